#!/usr/bin/env python3
"""Interleaved A/B of the projected table (tt_encoder_forward_projected_f32) against the call that projects its own tokens
(tt_encoder_forward_prepared_f32) on the bench's encoder shapes: one process, one box, arms alternated."""
import json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import bench


def t_of(fn, iters, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    dev = torch.device("cuda:0")
    inp = bench.make_encoder_inputs(dev)
    m = inp["model"].eval()
    big = inp["big"].to(dev)
    cases = [("query_b1", m.query_encoder, inp["q"][:1].to(dev), 50), ("query_b512", m.query_encoder, inp["q"].to(dev), 30),
             ("doc_b512", m.doc_encoder, inp["p"].to(dev), 20), ("doc_b8192", m.doc_encoder, big, 5),
             ("doc_b16384", m.doc_encoder, torch.cat([big, big], 0), 4), ("doc_b32768", m.doc_encoder, torch.cat([big] * 4, 0), 3)]
    with torch.no_grad():
        for name, enc, ids, iters in cases:
            res = {"case": name, "tokens": int((ids != 0).sum())}
            for rep in range(3):
                for arm, flag in (("projecting", False), ("projected", None)):
                    enc.projected_table = flag
                    res.setdefault(arm, []).append(round(t_of(lambda: enc(ids), iters), 4))
            res["speedup_best"] = round(min(res["projecting"]) / min(res["projected"]), 3)
            print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
