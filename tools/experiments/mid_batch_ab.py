#!/usr/bin/env python3
"""Screened search times at B = 33 .. 512 over 10M documents (the shared-tile form with one query group)."""
import sys, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
import bench
import twotowermlretrieval_amd as tt
dev = torch.device("cuda:0")
docs = bench.gen_rows(0, 10_000_000, dev)
ix = tt.BruteForceIndex(docs, screen=True)
out = {}
for B in (33, 64, 128, 256, 512):
    q = bench.gen_queries(B, dev, seed=B)
    for _ in range(3): ix.search(q, 10)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ix.search(q, 10)
    e1.record(); torch.cuda.synchronize()
    out[f"b{B}_ms"] = round(e0.elapsed_time(e1) / 20, 4)
print(json.dumps(out), flush=True)
