#!/usr/bin/env python3
"""Exact top-10 over wide embeddings (d = 512: 2 KiB per document) with the 16-query-tile kernel.
GBps = ALGORITHMIC bytes (one pass over the corpus, N*d*4) per call; B > 16 makes ceil(B/16) query tiles, whose
concurrent waves share the stream through L2, so HBM traffic stays near one pass (the round-1 figure multiplied by
the tile count and exceeded the HBM peak)."""
import sys, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import twotowermlretrieval_amd as tt
dev = torch.device("cuda:0")
N, d = 2_000_000, 512
g = torch.Generator(device=dev).manual_seed(0)
D = torch.randn((N, d), device=dev, generator=g); D /= D.norm(dim=1, keepdim=True)
for B in (1, 16, 64, 256):
    q = torch.randn((B, d), device=dev, generator=g); q /= q.norm(dim=1, keepdim=True)
    for _ in range(2): tt.score_topk(q, D, 10)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): tt.score_topk(q, D, 10)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(json.dumps(dict(B=B, N=N, d=d, ms=round(ms, 3), GBps=round(N * d * 4 / ms / 1e6, 1), passes=-(-B // 16), qps=round(B / ms * 1e3))), flush=True)
