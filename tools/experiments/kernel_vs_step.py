#!/usr/bin/env python3
"""The screen launch alone (HIP events inside the search) against the whole search (events around a run of searches)."""
import sys, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
import bench
import twotowermlretrieval_amd as tt
from twotowermlretrieval_amd import _lib
dev = torch.device("cuda:0")
q = bench.gen_queries(1024, dev)
docs = bench.gen_rows(0, 10_000_000, dev)
ix = tt.BruteForceIndex(docs, screen=True)
L = _lib.lib()
for rep in range(3):
    step = bench.time_search(ix, q, 10, iters=20, warm=5)
    pairs = [bench._event_pair(L) for _ in range(20)]
    for evs in pairs:
        ix.search(q, 10, _prof_events=evs)
    torch.cuda.synchronize()
    ks = [bench._pair_ms(L, evs) for evs in pairs]
    pairs2 = [bench._event_pair(L) for _ in range(20)]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for evs in pairs2:
        ix.search(q, 10, _prof_events=evs)
    e1.record(); torch.cuda.synchronize()
    ks2 = [bench._pair_ms(L, evs) for evs in pairs2]
    print(json.dumps(dict(step_ms=round(step, 4), step_with_events_ms=round(e0.elapsed_time(e1) / 20, 4),
                          kernel_ms_mean=round(sum(ks) / len(ks), 4), kernel_ms_min=round(min(ks), 4), kernel_ms_max=round(max(ks), 4))), flush=True)
