#!/usr/bin/env python3
"""Interleaved A/B of K4 (score_topk_kernel, the NT one-query-tile build) at B = 32 over 10M x 256 fp32 across macro variants of
the library (ab/libtt_<name>.so, tools/build_variant.py): HIP events around the main launch, one process, one box.
    python tools/experiments/k4_b32_ab.py product k4nst5 k4nolgkm ..."""
import ctypes as C, json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import torch
import bench
from twotowermlretrieval_amd import _lib


def load(name):
    if name == "product":
        return _lib.lib()
    lib = C.CDLL(str(ROOT / "ab" / f"libtt_{name}.so"))
    for n, (res, args) in _lib.SIGNATURES.items():
        fn = getattr(lib, n)
        fn.restype, fn.argtypes = res, args
    return lib


def main():
    names = sys.argv[1:] or ["product"]
    dev = torch.device("cuda:0")
    docs = bench.gen_rows(0, bench.N_DOCS, dev)
    q = bench.gen_queries(32, dev)
    libs = {n: load(n) for n in names}
    keep = _lib._lib
    ref = None
    out = {n: [] for n in names}
    for rep in range(4):
        for n in names:
            _lib._lib = libs[n]
            ms, _ = bench.kernel_only_ms(q, docs, 10, iters=10, warm=3)
            out[n].append(round(ms, 4))
            import twotowermlretrieval_amd as tt
            v, i = tt.score_topk(q, docs, 10)
            torch.cuda.synchronize()
            if ref is None:
                ref = (v.clone(), i.clone())
            assert torch.equal(v, ref[0]) and torch.equal(i, ref[1]), n
    _lib._lib = keep
    byts = bench.N_DOCS * 1024 + 32 * 1024 + 32 * 10 * 12
    for n in names:
        best = min(out[n])
        print(json.dumps({"variant": n, "kernel_ms": out[n], "best_ms": best, "hbm_frac_best": round(byts / best / 1e6 / 8000, 4)}), flush=True)


if __name__ == "__main__":
    main()
