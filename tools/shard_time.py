#!/usr/bin/env python3
"""Step time of one rank's share of the bench step for several shard sizes / k (single GPU, no collective)."""
import sys, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import bench
import twotowermlretrieval_amd as tt
dev = torch.device("cuda:0")
q = bench.gen_queries(1024, dev)
sizes = [int(a) for a in sys.argv[1:]] or [1_250_000, 2_500_000, 5_000_000, 10_000_000]
for n in sizes:
    docs = bench.gen_rows(0, n, dev)
    ix = tt.BruteForceIndex(docs, screen=True)
    for k in (10, 50):
        for _ in range(3):
            ix.search(q, k)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ix.search(q, k)
        e1.record()
        torch.cuda.synchronize()
        print(json.dumps(dict(docs=n, k=k, ms=round(e0.elapsed_time(e1) / 10, 4))), flush=True)
    # the sharded step as the product runs it (ShardedIndex.submit, world 1: per-shard top-50 -> exchange -> merge to
    # top-10, the tail launches and the exchange on the second stream)
    sx = tt.ShardedIndex(docs, 0, shard_k=50, screen=True)
    pend = []
    def step():
        pend.append(sx.submit(q, 10))
        return pend.pop(0).result() if len(pend) > 1 else None
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        step()
    while pend:
        pend.pop(0).result()
    e1.record()
    torch.cuda.synchronize()
    print(json.dumps(dict(docs=n, k="50->10 pipelined submit", ms=round(e0.elapsed_time(e1) / 20, 4))), flush=True)
    del sx
    del ix, docs
    torch.cuda.empty_cache()
