#!/usr/bin/env python3
"""Phase clocks of the one-workgroup forward recurrence (a -DTT_G16_DBG build: tools/build_variant.py g16dbg -DTT_G16_DBG), document
tower calls with the projected table at 4 096 (one tile per workgroup), 8 192 and 32 768 passages (two tiles)."""
import ctypes as C, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import torch
import bench
from twotowermlretrieval_amd import _lib

lib = C.CDLL(str(ROOT / "ab" / f"libtt_{sys.argv[1] if len(sys.argv) > 1 else 'g16dbg'}.so"))
for n, (res, args) in _lib.SIGNATURES.items():
    fn = getattr(lib, n)
    fn.restype, fn.argtypes = res, args
_lib.lib()
dev = torch.device("cuda:0")
inp = bench.make_encoder_inputs(dev)
enc = inp["model"].eval().doc_encoder
big = inp["big"].to(dev)
with torch.no_grad():
    enc(big)                      # (product library: caches built)
    _lib._lib = lib
    for name, ids in (("b4096", big[:4096].contiguous()), ("b8192", big), ("b32768", torch.cat([big] * 4, 0))):
        print("==", name, flush=True)
        sys.stderr.write(f"== {name}\n"); sys.stderr.flush()
        for _ in range(8):
            enc(ids)
        torch.cuda.synchronize()
