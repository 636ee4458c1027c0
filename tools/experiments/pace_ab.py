#!/usr/bin/env python3
"""Exact K4 (tt.score_topk) over 10M x 256 at several batch sizes with the chunk pacing on (default), off
(TT_SCORE_PACE=0) and with other lags (TT_SCORE_PACE_LAG), interleaved on one box:
python3 tools/experiments/pace_ab.py [B ...].  The child prints one JSON line per batch size (whole call, sample pass and
merges included)."""
import json, os, subprocess, sys
from pathlib import Path
root = Path(__file__).resolve().parent.parent.parent
if os.environ.get("_PACE_CHILD"):
    sys.path.insert(0, str(root))
    import torch
    import bench
    import twotowermlretrieval_amd as tt
    dev = torch.device("cuda:0")
    docs = bench.gen_rows(0, 10_000_000, dev)
    for B in [int(a) for a in sys.argv[1:]]:
        q = bench.gen_queries(B, dev)
        iters = 3 if B >= 512 else 10
        ref = None
        for _ in range(2):
            ref = tt.score_topk(q, docs, 10)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            tt.score_topk(q, docs, 10)
        e1.record()
        torch.cuda.synchronize()
        print(json.dumps(dict(pace=os.environ.get("TT_SCORE_PACE", "1"), lag=os.environ.get("TT_SCORE_PACE_LAG", "auto"),
                              g=os.environ.get("TT_SCORE_PACE_G", "auto"), B=B, ms=round(e0.elapsed_time(e1) / iters, 4),
                              idx_sum=int(ref[1].sum().item()))), flush=True)
    sys.exit(0)
sizes = sys.argv[1:] or ["64", "128", "256", "1024"]
for rep in range(2):
    for var in (dict(TT_SCORE_PACE="1"), dict(TT_SCORE_PACE="0"), dict(TT_SCORE_PACE_LAG="1"), dict(TT_SCORE_PACE_LAG="3")):
        env = dict(os.environ, _PACE_CHILD="1", **var)
        out = subprocess.run([sys.executable, __file__, *sizes], env=env, capture_output=True, text=True, timeout=900)
        for l in out.stdout.splitlines():
            if l.startswith("{"):
                print(l, flush=True)
        if out.returncode:
            print(out.stderr[-2000:], flush=True)
            sys.exit(1)
