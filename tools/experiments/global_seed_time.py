#!/usr/bin/env python3
"""What the UNION seed (ShardedIndex._local_search: all-gather of every rank's 10 largest sample maxima per query, seed =
the 10th largest of the union) buys one rank of the 8-GPU step, EMULATED on one GPU: the 10M corpus is cut into 8 row
shards, each shard's seed list comes from the real entry point (tt_score_topk_screened_seed_list_f32), the eight lists
are stacked as the all-gather would leave them, and shard 0's step (B = 1024, per-shard list 50 -> top 10) is timed
with its own seed (world 1) and with the union seed (tt_seed_union_f32 over the 8 lists; the all-gather itself -- 40 KB per
rank -- cannot be timed on one GPU).  Also checks that the merged top-10 of the 8 union-seeded shard lists is the exact one."""
import sys, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
import bench
import twotowermlretrieval_amd as tt
from twotowermlretrieval_amd.index import _local_seed, seed_union
dev = torch.device("cuda:0")
q = bench.gen_queries(1024, dev)
lists, idx = [], []
for r in range(8):
    lo, hi = tt.shard_bounds(10_000_000, r, 8)
    ix = tt.BruteForceIndex(bench.gen_rows(lo, hi, dev), idx_offset=lo, screen=True)
    got = []
    ix.search(q, 50, _seed_union=lambda l: (got.append(l.clone()), _local_seed(l))[1], _k_seed=10)
    lists.append(got[0])
    idx.append(ix)
    if r > 0:
        ix.docs = ix._sdocs = ix.docs  # (kept: the exactness check below searches every shard)
stack = torch.stack(lists).contiguous()                 # [8, B, 10]: what the all-gather leaves on every rank
torch.cuda.synchronize()
own_seed, union_seed = _local_seed(lists[0]), seed_union(stack, 8)
print("seed means: shard 0's own 10th %.4f, union 10th %.4f" % (float(own_seed.mean()), float(union_seed.mean())), flush=True)

def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

def with_union(l):                                      # rank 0's view: its own fresh list in block 0 of the gathered buffer
    stack[0].copy_(l)
    return seed_union(stack, 8)

ix0 = idx[0]
rows = []
for rep in range(2):                                    # interleaved
    rows.append(dict(rep=rep, own_seed_ms=round(timeit(lambda: ix0.search(q, 50, _seed_union=_local_seed, _k_seed=10)), 4),
                     union_seed_ms=round(timeit(lambda: ix0.search(q, 50, _seed_union=with_union, _k_seed=10)), 4)))
    print(json.dumps(rows[-1]), flush=True)
vs, is_ = [], []
for r in range(8):
    stack_r = stack.clone()
    v, i = idx[r].search(q, 50, _seed_union=lambda l: seed_union(stack_r, 8), _k_seed=10)
    vs.append(v); is_.append(i)
mv, mi = tt.topk_merge(torch.cat(vs, 1), torch.cat(is_, 1), 10)
listed = float(torch.stack([(i >= 0).sum(1).float().mean() for i in is_]).mean())
full = torch.cat([ix.docs for ix in idx])
ev, ei = tt.score_topk(q, full, 10)
print(json.dumps(dict(shard_rows=idx[0].docs.shape[0], listed_per_query_and_shard=round(listed, 1),
                      merged_equals_exact=bool(torch.equal(mv, ev) and torch.equal(mi, ei)))), flush=True)
