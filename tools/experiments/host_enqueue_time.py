#!/usr/bin/env python3
"""How long the HOST needs to enqueue one train step (the call returns without waiting for the GPU, apart from the one status
read) against the GPU's time for it: is the step GPU-bound or launch-bound?"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np, torch
import twotowermlretrieval_amd as tt
from bench import make_ids

dev = torch.device("cuda"); rs = np.random.RandomState(0)
V, E, H, B = 400003, 300, 256, 512
table = (rs.standard_normal((V, E)) * 0.3).astype(np.float32)
torch.manual_seed(0)
m = tt.TwoTowerModel({"VOCAB_SIZE": V, "EMBED_DIM": E, "HIDDEN_DIM": H}, table).to(dev)
q, _ = make_ids(rs, B, 6, 1, 30, V); p, _ = make_ids(rs, B, 70, 10, 250, V); n, _ = make_ids(rs, B, 70, 10, 250, V)
q, p, n = q.to(dev), p.to(dev), n.to(dev)
m.train()
opt = tt.FusedClipAdam(m.parameters(), lr=5e-5, max_norm=1.0)
import gc
gc.collect(); gc.freeze()
for checks in (True, False, True, False):
    for enc in (m.query_encoder, m.doc_encoder): enc.check_inputs = checks
    for _ in range(5): tt.train_step(m, opt, q, p, n, margin=0.5)
    torch.cuda.synchronize()
    enq, tot = [], []
    for _ in range(50):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        tt.train_step(m, opt, q, p, n, margin=0.5)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        enq.append((t1 - t0) * 1e3); tot.append((t2 - t0) * 1e3)
    # back to back (the host runs ahead as far as it can)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): tt.train_step(m, opt, q, p, n, margin=0.5)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"input checks {checks}: host time in the call median {sorted(enq)[25]:.3f} ms, call + wait {sorted(tot)[25]:.3f} ms; "
          f"50 back-to-back steps: host done after {(t1 - t0) / 50 * 1e3:.3f} ms per step, GPU after {(t2 - t0) / 50 * 1e3:.3f} ms per step")
