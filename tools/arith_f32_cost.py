#!/usr/bin/env python3
"""What the reference's own arithmetic (RNNEncoder.arith = "f32", TT_ENC_F32) costs against the default fp16 hi/lo split: the bench's
towers at 512 rows (forward) and the 512-triplet train step, interleaved."""
import json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import bench
import twotowermlretrieval_amd as tt

dev = torch.device("cuda:0")
inp = bench.make_encoder_inputs(dev, with_index_batch=False)
m = inp["model"]
q, p, n = (inp[k].to(dev) for k in "qpn")


def t_of(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / iters * 1e3, 4)


res = {}
for arith in ("split16", "f32", "split16", "f32"):
    for enc in (m.query_encoder, m.doc_encoder):
        enc.arith = arith
    m.eval()
    with torch.no_grad():
        res.setdefault(f"{arith}_doc_b512_ms", []).append(t_of(lambda: m.encode_document(p)))
        res.setdefault(f"{arith}_query_b512_ms", []).append(t_of(lambda: m.encode_query(q)))
    m.train()
    opt = tt.FusedClipAdam(m.parameters(), lr=5e-5, max_norm=1.0)
    res.setdefault(f"{arith}_train_step_ms", []).append(t_of(lambda: tt.train_step(m, opt, q, p, n, margin=0.5, defer_check=True), 6, 2))
    opt.settle()
    del opt
print(json.dumps(res), flush=True)
