#!/usr/bin/env python3
"""Per-step wall times of the concurrent-tower train step (each step synchronised), to see stalls a mean would hide."""
import sys, json, time
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
for enc in (m.query_encoder, m.doc_encoder): enc.check_inputs = False
q, qt = make_ids(rs, B, 6, 1, 30, V); p, pt = make_ids(rs, B, 70, 10, 250, V); n, nt = make_ids(rs, B, 70, 10, 250, V)
q, p, n = q.to(dev), p.to(dev), n.to(dev)
m.train()
opt = tt.FusedClipAdam(m.parameters(), lr=5e-5, max_norm=1.0)
for _ in range(3): tt.train_step(m, opt, q, p, n, margin=0.5, concurrent_towers=True)
import gc
gc_log = []
def _cb(phase, info):
    if phase == 'start': gc_log.append([len(ts), info['generation'], time.perf_counter()])
    else: gc_log[-1][2] = round((time.perf_counter() - gc_log[-1][2]) * 1e3, 2)
gc.callbacks.append(_cb)
ts = []
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loss = tt.train_step(m, opt, q, p, n, margin=0.5, concurrent_towers=True)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("ms per step:", " ".join(f"{t:.2f}" for t in ts))
print("median %.3f max %.3f loss %s" % (sorted(ts)[len(ts) // 2], max(ts), float(loss)))
slow = [(i, round(t, 2)) for i, t in enumerate(ts) if t > 1.5 * sorted(ts)[len(ts) // 2]]
print("steps over 1.5 x median:", slow)
print('gc events (step, generation, ms):', [g for g in gc_log if g[2] > 0.5 or g[1] == 2])
print('allocator:', {k: v for k, v in torch.cuda.memory_stats().items() if k in ('num_alloc_retries', 'num_device_alloc', 'num_device_free', 'segment.all.allocated')})
