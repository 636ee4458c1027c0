#!/usr/bin/env python3
"""Serving-size batches over the 10M x 256 corpus: exact fp32 kernel (K4) vs the screened index (streaming form for B <= 64)."""
import sys, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import bench
import twotowermlretrieval_amd as tt
n = int(sys.argv[1]) if len(sys.argv) > 1 else bench.N_DOCS
dev = torch.device("cuda:0")
docs = bench.gen_rows(0, n, dev)
ix_s = tt.BruteForceIndex(docs, screen=True)
ix_e = tt.BruteForceIndex(docs)


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for B in (1, 8, 32, 33, 64, 65, 96, 128, 129, 192, 256, 257, 384, 512, 513, 768, 1024):
    q = bench.gen_queries(B, dev, seed=B)
    ve, ie = ix_e.search(q, 10)
    vs, is_ = ix_s.search(q, 10)
    same = bool(torch.equal(ve, vs) and torch.equal(ie, is_))
    te = timeit(lambda: ix_e.search(q, 10), iters=(5 if B > 64 else 20))
    ts = timeit(lambda: ix_s.search(q, 10))
    print(json.dumps(dict(B=B, docs=n, exact_ms=round(te, 4), screened_ms=round(ts, 4), identical=same,
                          screened_GBps=round(n * 512 / ts / 1e6, 1), flags=int(ix_s.fallback_flags.ne(0).sum().item()))), flush=True)
