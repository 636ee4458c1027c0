#!/usr/bin/env python3
"""Per-workgroup start/end clocks of the main screen launch (needs the instrumented build: patch -p0 twotowermlretrieval_amd/csrc/screen.hip < tools/experiments/screen_clocks.patch, rebuild; TT_SCREEN_CLOCKS=1)."""
import os, sys, ctypes as C
from pathlib import Path
os.environ["TT_SCREEN_CLOCKS"] = "1"
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np, torch
import bench
import twotowermlretrieval_amd as tt
dev = torch.device("cuda:0")
q = bench.gen_queries(1024, dev)
for n, k in ((1_250_000, 50), (10_000_000, 10)):
    docs = bench.gen_rows(0, n, dev)
    ix = tt.BruteForceIndex(docs, screen=True)
    for _ in range(3):
        ix.search(q, k)
    torch.cuda.synchronize()
    ptr = int(open("/tmp/tt_clk_ptr").read())
    a = np.ctypeslib.as_array((C.c_longlong * 512).from_address(ptr)).copy().reshape(256, 2)
    t0 = a[:, 0].min()
    st, en = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0   # wall_clock64: 100 MHz -> microseconds
    dur = en - st
    print(f"N={n} k={k}: start spread {st.max():.1f} us; end min/median/max {en.min():.1f}/{np.median(en):.1f}/{en.max():.1f} us; "
          f"duration min/median/max {dur.min():.1f}/{np.median(dur):.1f}/{dur.max():.1f} us")
    order = np.argsort(en)
    print("  slowest WGs (block, xcd, start, end):", [(int(b), int(b) % 8, round(float(st[b]), 1), round(float(en[b]), 1)) for b in order[-6:]])
    print("  fastest WGs:", [(int(b), int(b) % 8, round(float(st[b]), 1), round(float(en[b]), 1)) for b in order[:6]])
    byx = [float(np.median(dur[np.arange(256) % 8 == x])) for x in range(8)]
    print("  median duration by blockIdx%8:", [round(v, 1) for v in byx])
    del ix, docs
    torch.cuda.empty_cache()
