#!/usr/bin/env python3
"""A few screened searches at B=1024 over 10M docs, for rocprofv3 kernel traces / PMC passes."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import bench
import twotowermlretrieval_amd as tt
n = int(sys.argv[1]) if len(sys.argv) > 1 else bench.N_DOCS
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device("cuda:0")
docs = bench.gen_rows(0, n, dev)
q = bench.gen_queries(B, dev)
ix = tt.BruteForceIndex(docs, screen=True)
for _ in range(4):
    ix.search(q, 10)
torch.cuda.synchronize()
print("flags", int(ix.fallback_flags.ne(0).sum().item()))
