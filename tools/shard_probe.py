#!/usr/bin/env python3
"""One 8-GPU rank's share of the bench step on a single GPU (1.25M docs, B=1024, per-shard top-50), for kernel traces."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import bench
import twotowermlretrieval_amd as tt
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device("cuda:0")
docs = bench.gen_rows(0, n, dev)
q = bench.gen_queries(1024, dev)
ix = tt.BruteForceIndex(docs, screen=True)
for _ in range(6):
    ix.search(q, k)
torch.cuda.synchronize()
