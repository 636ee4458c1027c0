#!/usr/bin/env python3
"""BASELINE configs[4], one GPU's share: 12.5M x 256 bf16 passages (6.4 GB) in pinned host DRAM streamed
through the GPU (double-buffered async copies overlapped with scoring), exact top-10; vs the same shard
resident in HBM."""
import argparse, json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import twotowermlretrieval_amd as tt

ap = argparse.ArgumentParser(); ap.add_argument("--n", type=int, default=12_500_000); ap.add_argument("--block", type=int, default=1 << 20)
a = ap.parse_args()
dev = torch.device("cuda")
host = torch.empty((a.n, 256), dtype=torch.bfloat16).pin_memory()
g = torch.Generator(device=dev).manual_seed(5)
for lo in range(0, a.n, 1_000_000):
    hi = min(a.n, lo + 1_000_000)
    x = torch.randn((hi - lo, 256), device=dev, generator=g); x /= x.norm(dim=1, keepdim=True)
    host[lo:hi].copy_(x.to(torch.bfloat16))
torch.cuda.synchronize()
t0 = time.perf_counter(); ix = tt.StreamedIndex(host, block_docs=a.block); torch.cuda.synchronize()
print(json.dumps(dict(what="build pass (norm scan)", s=round(time.perf_counter() - t0, 3), dmax=ix.dmax_norm)), flush=True)
for B in (32, 1024):
    q = torch.randn((B, 256), device=dev, generator=g); q /= q.norm(dim=1, keepdim=True)
    ix.search(q, 10); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): v, i = ix.search(q, 10)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(json.dumps(dict(what="streamed pass", B=B, N=a.n, block=a.block, ms=round(dt * 1e3, 2), pcie_GBps=round(a.n * 512 / dt / 1e9, 1),
                          qps=round(B / dt, 1))), flush=True)
    res = tt.BruteForceIndex(host[:2_000_000].to(dev).to(torch.float32), screen=True)   # resident check on a prefix
    rv, ri = res.search(q, 10)
    sv, si = tt.StreamedIndex(host[:2_000_000], block_docs=a.block).search(q, 10)
    torch.cuda.synchronize()
    print(json.dumps(dict(what="streamed == resident on a 2M prefix", B=B, equal=bool(torch.equal(ri, si) and torch.equal(rv, sv)))), flush=True)
    del res

t0 = time.perf_counter(); res = ix.resident(); torch.cuda.synchronize()
print(json.dumps(dict(what="widen the shard into HBM (one streamed pass)", s=round(time.perf_counter() - t0, 3),
                      hbm_GB=round(a.n * 256 * 6 / 1e9, 1))), flush=True)
for B in (32, 1024):
    q = torch.randn((B, 256), device=dev, generator=g); q /= q.norm(dim=1, keepdim=True)
    sv, si = ix.search(q, 10)
    rv, ri = res.search(q, 10); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): res.search(q, 10)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(json.dumps(dict(what="resident pass", B=B, N=a.n, ms=round(dt * 1e3, 3), qps=round(B / dt, 1),
                          equals_streamed=bool(torch.equal(ri, si) and torch.equal(rv, sv)))), flush=True)
