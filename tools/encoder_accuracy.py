#!/usr/bin/env python3
"""Error of the encoder against the CPU oracle (fp32 throughout) on a north-star-shaped batch, for the two arithmetic
paths of the PRODUCT library: RNNEncoder(arith="f32") (TT_ENC_F32: fp32 MFMA everywhere) and the default (fp16 hi/lo split on the
f16 pipes for the input projection, the recurrence, its backward and the weight-gradient products).
    python tools/encoder_accuracy.py            # one JSON line per path"""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests" / "golden")]


def child(arith):
    import numpy as np
    import torch
    import synth
    from oracle import oracle as o
    from twotowermlretrieval_amd.model import RNNEncoder
    o.build()
    V, E, H, B, T = 2000, 300, 256, 48, 120
    table = synth.make_table(3, V, E)
    sd = synth.make_encoder_state(4, E, H)
    enc = RNNEncoder(V, E, H, pretrained_embeddings=table, arith=arith)
    full = {"embedding.weight": torch.from_numpy(table)}
    full.update({k: torch.from_numpy(v) for k, v in sd.items()})
    enc.load_state_dict(full)
    enc = enc.cuda().train()
    ids = synth.make_ids(5, B, T, V, zero_inside=0.05)
    quads = synth.weight_quads(sd)
    y = enc(torch.from_numpy(ids).cuda())
    want = o.encoder_forward(ids, table, quads, H)
    d_out = np.random.RandomState(6).standard_normal((B, H)).astype(np.float32)
    y.backward(torch.from_numpy(d_out).cuda())
    og, _, _ = o.encoder_backward(ids, table, quads, H, d_out)
    grads = {n: p.grad.cpu().numpy() for n, p in enc.named_parameters() if p.requires_grad}
    rel = {}
    for (name, got), ref in zip(grads.items(), og[0]):
        rel[name] = float(np.abs(got - ref).max() / np.abs(ref).max())
    print(json.dumps({"path": "fp32 MFMA (arith='f32', TT_ENC_F32)" if arith == "f32" else "fp16 hi/lo split on f16 MFMA (default)",
                      "shape": dict(B=B, T=T, E=E, H=H), "out_max_abs_err": float(np.abs(y.detach().cpu().numpy() - want).max()),
                      "grad_max_err_over_max": rel}), flush=True)


if __name__ == "__main__":
    for arith in ("f32", "split16"):
        child(arith)
