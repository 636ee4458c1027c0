#!/usr/bin/env python3
"""cProfile of the host side of the eager direct train step (where do the ~0.4 ms of enqueueing go?)."""
import sys, cProfile, pstats
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
for _ in range(10): tt.train_step(m, opt, q, p, n, margin=0.5, defer_check=True)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(300): tt.train_step(m, opt, q, p, n, margin=0.5, defer_check=True)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
