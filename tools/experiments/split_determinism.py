#!/usr/bin/env python3
"""Is the 512-triplet train step bit-reproducible run to run, with the column-split recurrence off / on / mixed?"""
import copy, os, sys
from pathlib import Path
root = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(root), str(root / "tests" / "golden")]
import numpy as np, torch, synth
import twotowermlretrieval_amd as tt

def run(flags, steps=3, B=512, conc=True):
    V, E, H = 500, 300, 256
    torch.manual_seed(5)
    m0 = tt.TwoTowerModel({"VOCAB_SIZE": V, "EMBED_DIM": E, "HIDDEN_DIM": H}, synth.make_table(4, V, E)).cuda().train()
    ms = [copy.deepcopy(m0) for _ in flags]
    os_ = [tt.FusedClipAdam(m.parameters(), lr=1e-3, max_norm=1.0) for m in ms]
    for step in range(steps):
        ids = [torch.from_numpy(synth.make_ids(60 + 3 * step + s, B, T, V)).cuda() for s, T in enumerate((9, 60, 70))]
        for m, o, f in zip(ms, os_, flags):
            os.environ["TT_GRU_SPLIT"] = str(f)
            tt.train_step(m, o, *ids, margin=0.5, concurrent_towers=conc)
            torch.cuda.synchronize()
        base = os_[0]
        for i, o in enumerate(os_[1:], 1):
            dp = (o.flat_params - base.flat_params).abs()
            dg = (o.flat_grads - base.flat_grads).abs()
            print(f"flags={flags} conc={conc} step {step}: run {i} vs 0: params differ at {int((dp > 0).sum())} (max {float(dp.max()):.3e}), "
                  f"grads differ at {int((dg > 0).sum())} (max {float(dg.max()):.3e}, rel {float(dg.max() / base.flat_grads.abs().max()):.2e})", flush=True)
        # which tensors
        if step == steps - 1:
            off = 0
            for (nm, p) in ms[0].named_parameters():
                if not p.requires_grad: continue
                k = p.numel()
                for i, o in enumerate(os_[1:], 1):
                    d = (o.flat_grads[off:off + k] - base.flat_grads[off:off + k]).abs()
                    if float(d.max()) > 0:
                        print(f"   {nm}: run {i} grads differ at {int((d > 0).sum())} of {k}", flush=True)
                off += k

for conc in (True, False):
    run((0, 0), conc=conc)
    run((1, 1), conc=conc)
    run((0, 1), conc=conc)
