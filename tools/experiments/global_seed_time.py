#!/usr/bin/env python3
"""What a GLOBAL seed buys one rank of the 8-GPU step, measured on one GPU: the 10M corpus is cut into 8 row shards, the seed
phase runs on each (k_seed = 10), the element-wise maximum is what an all-reduce would hand every rank; then shard 0's step
(B = 1024, per-shard list 50 -> top 10) is timed with its own seed and with the global one."""
import sys, json, ctypes as C
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
import bench
import twotowermlretrieval_amd as tt
from twotowermlretrieval_amd import _lib
dev = torch.device("cuda:0")
L = _lib.lib()
q = bench.gen_queries(1024, dev)
st = lambda: torch.cuda.current_stream().cuda_stream
seeds, seeds2, tops = [], [], []
for r in range(8):
    lo, hi = tt.shard_bounds(10_000_000, r, 8)
    docs = bench.gen_rows(lo, hi, dev)
    ix = tt.BruteForceIndex(docs, idx_offset=lo, screen=True)
    got = []
    ix.search(q, 50, _seed_exchange=lambda s: got.append(s.clone()), _k_seed=10)
    seeds.append(got[0])
    got2 = []
    ix.search(q, 50, _seed_exchange=lambda s: got2.append(s.clone()), _k_seed=2)   # ceil(k / world) = 2
    seeds2.append(got2[0])
    for j in range(1, 11):   # the shard's 10 largest sample maxima per query, one seed call per rank in the list
        gj = []
        ix.search(q, 50, _seed_exchange=lambda s: gj.append(s.clone()), _k_seed=j)
        tops.append(gj[0])
    if r > 0:
        del ix, docs
    else:
        ix0, docs0 = ix, docs
gseed = torch.stack(seeds).max(0).values
gmin2 = torch.stack(seeds2).min(0).values   # every rank holds >= 2 documents at least this good: 16 >= 10 in the union
gunion = torch.stack(tops).topk(10, dim=0).values[9]   # the 10th best of the union of the shards' top-10 sample maxima
torch.cuda.synchronize()
print("own 10th (rank 0) mean %.4f, max over ranks of the 10th %.4f, min over ranks of the 2nd %.4f" % (float(seeds[0].mean()), float(gseed.mean()), float(gmin2.mean())))
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
own = timeit(lambda: ix0.search(q, 50))
two = timeit(lambda: ix0.search(q, 50, _seed_exchange=lambda s: None, _k_seed=10))
glob = timeit(lambda: ix0.search(q, 50, _seed_exchange=lambda s: s.copy_(torch.maximum(s, gseed)), _k_seed=10))
glob2 = timeit(lambda: ix0.search(q, 50, _seed_exchange=lambda s: s.copy_(torch.maximum(s, gmin2)), _k_seed=10))
globu = timeit(lambda: ix0.search(q, 50, _seed_exchange=lambda s: s.copy_(torch.maximum(s, gunion)), _k_seed=10))
vu, iu = ix0.search(q, 50, _seed_exchange=lambda s: s.copy_(torch.maximum(s, gunion)), _k_seed=10)
print(json.dumps(dict(union_10th_seed_mean=round(float(gunion.mean()), 4), union_seed_ms=round(globu, 4),
                      listed_per_query=round(float((iu >= 0).sum(1).float().mean()), 1))), flush=True)
ev, ei = tt.score_topk(q, docs0, 10, 0)
v2, i2 = ix0.search(q, 50, _seed_exchange=lambda s: s.copy_(torch.maximum(s, gmin2)), _k_seed=10)
print(json.dumps(dict(min_of_2nd_seed_ms=round(glob2, 4), listed_per_query=round(float((i2 >= 0).sum(1).float().mean()), 1))), flush=True)
v0, i0 = ix0.search(q, 50, _seed_exchange=lambda s: s.copy_(torch.maximum(s, gseed)), _k_seed=10)
print(json.dumps(dict(shard_rows=docs0.shape[0], own_seed_k50_ms=round(own, 4), two_phase_own_seed_k10_ms=round(two, 4),
                      global_seed_ms=round(glob, 4), listed_per_query=round(float((i0 >= 0).sum(1).float().mean()), 1))), flush=True)
