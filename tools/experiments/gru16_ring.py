#!/usr/bin/env python3
"""Resident fragments vs ring depth of the one-workgroup forward recurrence (csrc/gru16.hip: TT_G16_R fragments of a wave's W_hh
slice stay in registers, TT_G16_NR streamed ones are in flight): document-tower inference calls of several batch sizes on variant
libraries (ab/libtt_<name>.so, tools/build_variant.py), each in its own process.  usage: gru16_ring.py [name ...] ("product")"""
import sys, json, subprocess, os
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, str(ROOT))
    import numpy as np, torch
    from twotowermlretrieval_amd import _lib
    name = sys.argv[2]
    if name != "product":
        _lib.LIB_PATH = ROOT / "ab" / f"libtt_{name}.so"
    import twotowermlretrieval_amd as tt
    from bench import make_ids
    dev = torch.device("cuda"); rs = np.random.RandomState(0)
    V, E, H = 400003, 300, 256
    table = (rs.standard_normal((V, E)) * 0.3).astype(np.float32)
    torch.manual_seed(0)
    m = tt.TwoTowerModel({"VOCAB_SIZE": V, "EMBED_DIM": E, "HIDDEN_DIM": H}, table).to(dev).eval()
    row = {"variant": name}
    ref = {}
    for B in (512, 2048, 8192, 16384):
        ids, _ = make_ids(rs, B, 70, 10, 250, V)
        ids = ids.to(dev)
        m.doc_encoder.one_workgroup = True
        with torch.no_grad():
            y = m.encode_document(ids)
            for _ in range(2):
                m.encode_document(ids)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                m.encode_document(ids)
            e1.record(); torch.cuda.synchronize()
        row[f"ms_b{B}"] = round(e0.elapsed_time(e1) / 8, 4)
        row[f"sum_b{B}"] = float(y.double().sum())
    print(json.dumps(row), flush=True)
    sys.exit(0)
names = sys.argv[1:] or ["product"]
for rep in range(2):
    for n in names:
        r = subprocess.run([sys.executable, __file__, "--child", n], capture_output=True, text=True)
        print(r.stdout.strip() or r.stderr[-400:], flush=True)
