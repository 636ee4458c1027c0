#!/usr/bin/env python3
"""Why does a 0.3 ms host-to-device copy add its whole time to a 4 ms encode?  Same 16 batches as cu_mask_copy.py:
  a        ids resident
  b        copy(k) on a side stream right in front of encode(k), encode waits for it
  b_early  ALL copies issued first (own buffers, own events), then the encodes, each waiting for its copy's event
  b_nowait as b without the wait (a race: results meaningless, only the time counts)
  b_aheadD copy(k + D) issued in front of encode(k), encode(k) waits for copy(k)'s EVENT
  b_same_stream   the copy on the compute stream itself (no side stream)"""
import sys, json, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np, torch
import bench

dev = torch.device("cuda:0")
inp = bench.make_encoder_inputs(dev, with_index_batch=False)
model = inp["model"].eval()
V, rs = bench.ENC_V, np.random.RandomState(3)
n_docs, bs = 262_144, 16384
host = [bench.make_ids_bulk(rs, bs, 70, 10, 250, V).pin_memory() for _ in range(n_docs // bs)]
resident = [h.to(dev) for h in host]
bufs = [torch.empty_like(r) for r in resident]
side, cur = torch.cuda.Stream(device=dev), torch.cuda.current_stream(dev)


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return round(best * 1e3, 2)


def a():
    with torch.no_grad():
        for x in resident:
            model.encode_document(x)


def b(wait=True):
    with torch.no_grad():
        for k, h in enumerate(host):
            with torch.cuda.stream(side):
                bufs[k].copy_(h, non_blocking=True)
            if wait:
                cur.wait_stream(side)
            model.encode_document(bufs[k])


def b_early():
    evs = []
    with torch.no_grad():
        with torch.cuda.stream(side):
            for k, h in enumerate(host):
                bufs[k].copy_(h, non_blocking=True)
                e = torch.cuda.Event(); e.record(side); evs.append(e)
        for k in range(len(host)):
            cur.wait_event(evs[k])
            model.encode_document(bufs[k])


def b_ahead(d):
    evs = {}
    def issue(k):
        with torch.cuda.stream(side):
            bufs[k].copy_(host[k], non_blocking=True)
            e = torch.cuda.Event(); e.record(side); evs[k] = e
    with torch.no_grad():
        for k in range(min(d, len(host))):
            issue(k)
        for k in range(len(host)):
            if k + d < len(host):
                issue(k + d)
            cur.wait_event(evs[k])
            model.encode_document(bufs[k])


def b_sync():
    with torch.no_grad():
        for k, h in enumerate(host):
            bufs[k].copy_(h, non_blocking=True)
            model.encode_document(bufs[k])


def copies_only():
    with torch.cuda.stream(side):
        for k, h in enumerate(host):
            bufs[k].copy_(h, non_blocking=True)


for name, fn in (("a", a), ("copies_only", copies_only), ("b", b), ("b_early", b_early), ("b_ahead1", lambda: b_ahead(1)), ("b_ahead2", lambda: b_ahead(2)), ("b_ahead0_event", lambda: b_ahead(0)), ("b_nowait", lambda: b(False)), ("b_same_stream", b_sync), ("a again", a)):
    print(json.dumps({"what": name, "ms": timed(fn)}), flush=True)
