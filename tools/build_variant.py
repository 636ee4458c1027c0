#!/usr/bin/env python3
"""Build a macro variant of libtt.so for tools/ab_run.py:  python tools/build_variant.py NAME [-DX=Y ...]  ->  ab/libtt_NAME.so
(objects under ab/obj_NAME/; only sources that mention one of the macros, or all of them with no -D, are rebuilt with the flags;
the rest are taken from the package's own build directory)."""
import re, subprocess, sys
from pathlib import Path
root = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(root))
from twotowermlretrieval_amd import build as B

name, defs = sys.argv[1], sys.argv[2:]
B.build()
macros = [re.sub(r"^-D", "", d).split("=")[0] for d in defs]
hdr_hit = any(any(m in h.read_text() for m in macros) for h in B._headers())
objdir = root / "ab" / f"obj_{name}"
objdir.mkdir(parents=True, exist_ok=True)
flags = [f"--offload-arch={B.ARCH}", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]
objs, procs = [], []
for src in B.sources():
    if hdr_hit or not macros or any(m in src.read_text() for m in macros):
        obj = objdir / (src.stem + ".o")
        procs.append((src, subprocess.Popen([B._hipcc(), *flags, *defs, "-c", str(src), "-o", str(obj)])))
    else:
        obj = B.PKG / "build" / (src.stem + ".o")
    objs.append(obj)
for src, p in procs:
    if p.wait():
        sys.exit(f"hipcc failed on {src.name}")
out = root / "ab" / f"libtt_{name}.so"
subprocess.check_call([B._hipcc(), f"--offload-arch={B.ARCH}", "-shared", "-fPIC", "-o", str(out), *map(str, objs), "-ldl"])
print(out)
