#!/usr/bin/env python3
"""The eight-wave member of the split backward recurrence against the four-wave one (TT_GRU_SPLIT_BWD=4): same inputs, same
weights -- weight gradients must agree bit for bit (same accumulator chains), bias gradients to rounding (another summation
order); run-to-run reproducibility of the new kernel; time of forward + backward."""
import os, sys, time
from pathlib import Path
root = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(root), str(root / "tests" / "golden")]
import numpy as np, torch, synth
import twotowermlretrieval_amd as tt

V, E, H = 500, 300, 256
torch.manual_seed(5)
enc = tt.RNNEncoder(V, E, H, pretrained_embeddings=synth.make_table(4, V, E)).cuda().train()
names = ("dW_ih", "dW_hh", "db_ih", "db_hh")
for B, T in ((16, 9), (40, 30), (64, 70), (250, 33), (512, 70), (1024, 70), (1024, 20)):
    ids = torch.from_numpy(synth.make_ids(60 + B, B, T, V)).cuda()
    d_out = torch.from_numpy(np.random.RandomState(B).standard_normal((B, H)).astype(np.float32)).cuda()
    res = {}
    for mode in ("4", "8", "8 again"):
        os.environ["TT_GRU_SPLIT_BWD"] = "4" if mode == "4" else "1"
        for rep in range(3):
            enc.zero_grad()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            y = enc(ids)
            y.backward(d_out)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        res[mode] = ([p.grad.clone() for p in enc._flat_params()], dt)
    out = []
    for n, a, b, c in zip(names, res["4"][0], res["8"][0], res["8 again"][0]):
        scale = float(a.abs().max())
        out.append(f"{n}: differing {int((a != b).sum())}/{a.numel()} max {float((a - b).abs().max()) / max(scale, 1e-30):.1e} of max, rerun differing {int((b != c).sum())}")
    print(f"B={B} T={T}: 4-wave {res['4'][1] * 1e3:.3f} ms, 8-wave {res['8'][1] * 1e3:.3f} ms | " + " | ".join(out), flush=True)
