#!/usr/bin/env python3
"""Run-to-run reproducibility of one tower's forward + backward (same inputs, same weights), per gradient tensor."""
import os, sys
from pathlib import Path
root = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(root), str(root / "tests" / "golden")]
import numpy as np, torch, synth
import twotowermlretrieval_amd as tt

V, E, H = 500, 300, 256
torch.manual_seed(5)
enc = tt.RNNEncoder(V, E, H, pretrained_embeddings=synth.make_table(4, V, E)).cuda().train()
for B, T in ((64, 70), (256, 70), (512, 70), (1024, 70), (1024, 20)):
    ids = torch.from_numpy(synth.make_ids(60 + B, B, T, V)).cuda()
    d_out = torch.from_numpy(np.random.RandomState(B).standard_normal((B, H)).astype(np.float32)).cuda()
    runs = []
    for rep in range(4):
        enc.zero_grad()
        y = enc(ids)
        y.backward(d_out)
        torch.cuda.synchronize()
        runs.append((y.detach().clone(), [p.grad.clone() for p in enc._flat_params()]))
    msg = []
    for rep in range(1, 4):
        same_y = torch.equal(runs[rep][0], runs[0][0])
        diffs = [int((a != b).sum()) for a, b in zip(runs[rep][1], runs[0][1])]
        msg.append(f"rep{rep}: y_equal={same_y} grad_diff_counts(dW_ih,dW_hh,db_ih,db_hh)={diffs}")
    print(f"B={B} T={T} split={os.environ.get('TT_GRU_SPLIT','1')}: " + " | ".join(msg), flush=True)
