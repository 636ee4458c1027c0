#!/usr/bin/env python3
"""Host front end of one 16 384-passage batch, stage by stage (the machine's cores, no GPU): gather / tt_tok_encode_units /
tt_tok_pad_i32 / the whole encode_batch, for 8 and 16 threads."""
import sys, time, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np, torch
import twotowermlretrieval_amd as tt
from twotowermlretrieval_amd import _lib, _pytext
V = 400003
words = ["the", ",", ".", "of", "and"] + [f"w{i}" for i in range(5, V - 1)]
tok = tt.PretrainedTokenizer(word2idx={w: i for i, w in enumerate(words)})
rs = np.random.RandomState(3)
n = 16384
lens_ = np.clip(rs.poisson(70, n), 10, 250)
z = rs.zipf(1.07, int(lens_.sum())) % (V - 1)
docs, p0 = [], 0
for L_ in lens_:
    docs.append(" ".join(map(words.__getitem__, z[p0:p0 + L_]))); p0 += L_
L = _lib.lib(); h = tok._native()
ptrs = np.empty(n, np.uint64); tlen = np.empty(n, np.int64); units = np.empty(n, np.uint8); off = np.empty(n + 1, np.int64)
lens = np.empty(n, np.int32); status = np.empty(n, np.int32)
n_ok, total, beyond = _pytext.gather(docs, ptrs.ctypes.data, tlen.ctypes.data, units.ctypes.data)
ragged = np.empty(total + 1, np.int64)
buf = torch.empty(n * 160, dtype=torch.int64, pin_memory=torch.cuda.is_available())
out32 = buf.view(torch.int32)
for nt in (8, 16):
    T = [0.0] * 5
    reps = 12
    for rep in range(reps + 2):
        t0 = time.perf_counter()
        tup = tuple(docs)
        _pytext.gather(tup, ptrs.ctypes.data, tlen.ctypes.data, units.ctypes.data)
        t1 = time.perf_counter()
        L.tt_tok_encode_units(h, ptrs.ctypes.data, tlen.ctypes.data, units.ctypes.data, n, off.ctypes.data, ragged.ctypes.data, lens.ctypes.data, status.ctypes.data, nt)
        t2 = time.perf_counter()
        width = int(lens.max())
        L.tt_tok_pad_i32(ragged.ctypes.data, off.ctypes.data, lens.ctypes.data, n, width, out32.data_ptr(), nt)
        t3 = time.perf_counter()
        tok.encode_batch(docs, n_threads=nt, out=buf, ids32=True)
        t4 = time.perf_counter()
        sl = docs[0:n]
        t5 = time.perf_counter()
        if rep >= 2:
            for k, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
                T[k] += d
    print(json.dumps({"threads": nt, "tokens": int(lens.sum()), "gather_ms": round(T[0] / reps * 1e3, 3), "encode_units_ms": round(T[1] / reps * 1e3, 3),
                      "pad_i32_ms": round(T[2] / reps * 1e3, 3), "encode_batch_ms": round(T[3] / reps * 1e3, 3), "slice_ms": round(T[4] / reps * 1e3, 3)}), flush=True)

# --- why is tt_tok_encode_units twice as slow here as in tok_harness (same tokens, same threads)?
import ctypes as C
print(open("/proc/self/smaps_rollup").read().split("AnonHugePages:")[1].split("\n")[0].strip(), "of anonymous huge pages in this process", flush=True)
blob = "\0".join(docs).encode("ascii")
base = C.cast(C.c_char_p(blob), C.c_void_p).value
offs = np.zeros(n, np.int64); offs[1:] = np.cumsum(tlen[:n - 1] + 1)
ptrs_c = (offs + base).astype(np.uint64)
for name, pp in (("scattered str objects", ptrs), ("one contiguous blob", ptrs_c), ("scattered str objects", ptrs)):
    ts = []
    for _ in range(10):
        t0 = time.perf_counter()
        L.tt_tok_encode_units(h, pp.ctypes.data, tlen.ctypes.data, units.ctypes.data, n, off.ctypes.data, ragged.ctypes.data, lens.ctypes.data, status.ctypes.data, 16)
        ts.append((time.perf_counter() - t0) * 1e3)
    print(json.dumps({"texts": name, "threads": 16, "encode_units_ms_min": round(min(ts), 3), "median": round(sorted(ts)[5], 3)}), flush=True)
for nt in (1, 4):
    t0 = time.perf_counter()
    L.tt_tok_encode_units(h, ptrs.ctypes.data, tlen.ctypes.data, units.ctypes.data, n, off.ctypes.data, ragged.ctypes.data, lens.ctypes.data, status.ctypes.data, nt)
    print(json.dumps({"threads": nt, "encode_units_ms": round((time.perf_counter() - t0) * 1e3, 3)}), flush=True)
