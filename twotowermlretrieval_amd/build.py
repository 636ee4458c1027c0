"""Build libtt.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m twotowermlretrieval_amd.build [--force]

hipcc cross-compiles without a GPU; the .so sits next to this file so it
travels with the source tree (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "libtt.so"
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def sources():
    return sorted(CSRC.glob("*.hip"))


def _headers():
    return list(CSRC.glob("*.h")) + list((PKG.parent / "include").glob("*.h"))  # every source may include any of them


def _stale(obj: Path, src: Path) -> bool:
    return (not obj.exists()) or any(p.stat().st_mtime > obj.stat().st_mtime for p in [src, *_headers()])


def needs_build() -> bool:
    if not LIB.exists():
        return True
    objdir = PKG / "build"
    objs = [objdir / (src.stem + ".o") for src in sources()]
    return any(_stale(o, s) for o, s in zip(objs, sources())) or any(o.stat().st_mtime > LIB.stat().st_mtime for o in objs)


def build(force: bool = False, verbose: bool = False) -> Path:
    build_pytext(force)
    if not force and not needs_build():
        return LIB
    objdir = PKG / "build"
    objdir.mkdir(exist_ok=True)
    hipcc = _hipcc()
    flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
             "-Wall", "-Wno-unused-function"]
    objs = []
    procs = []
    for src in sources():
        obj = objdir / (src.stem + ".o")
        objs.append(obj)
        dep_newer = force or _stale(obj, src)
        if dep_newer:
            cmd = [hipcc, *flags, "-c", str(src), "-o", str(obj)]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src.name}:\n{out}")
        if verbose and out.strip():
            print(out)
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(LIB), *map(str, objs), "-ldl"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    return LIB


def build_pytext(force: bool = False) -> Path:
    """twotowermlretrieval_amd/_pytext.*.so from csrc/pytext.c (CPython API, host compiler): the str-pointer gather in front of
    tt_tok_encode_ptrs.  Not part of libtt.so; tokenizer.encode_batch falls back to its one-join form without it."""
    import sysconfig
    src = PKG / "csrc" / "pytext.c"
    out = PKG / ("_pytext" + sysconfig.get_config_var("EXT_SUFFIX"))
    if not force and out.exists() and out.stat().st_mtime > src.stat().st_mtime:
        return out
    cc = os.environ.get("CC") or shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        raise RuntimeError("no C compiler for csrc/pytext.c")
    r = subprocess.run([cc, "-O2", "-shared", "-fPIC", "-Wall", f"-I{sysconfig.get_paths()['include']}", str(src), "-o", str(out)],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode:
        raise RuntimeError(f"{cc} failed on pytext.c:\n{r.stdout}")
    return out


def build_variant(name: str, defs, force: bool = False) -> Path:
    """A macro variant of the library, NEVER loaded by the package itself: <repo>/ab/libtt_<name>.so, objects under
    ab/obj_<name>/.  Sources that mention one of the macros (all of them when a header does, or with no -D at all) are compiled
    with `defs`; the rest come from the product build.  `ab` = -DTT_AB, the comparison build (csrc/tt_common.h: A/B switches
    read from the environment at every call + the superseded kernels) that tests/ pin product kernels against; the mutation
    guard and tools/experiments build theirs the same way (tools/build_variant.py)."""
    import re
    build()
    root = PKG.parent
    out = root / "ab" / f"libtt_{name}.so"
    newest = max(p.stat().st_mtime for p in [*sources(), *_headers()])
    if not force and out.exists() and out.stat().st_mtime > newest:
        return out
    macros = [re.sub(r"^-D", "", d).split("=")[0] for d in defs]
    hdr_hit = any(any(m in h.read_text() for m in macros) for h in _headers())
    objdir = root / "ab" / f"obj_{name}"
    objdir.mkdir(parents=True, exist_ok=True)
    flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]
    objs, procs = [], []
    for src in sources():
        if hdr_hit or not macros or any(m in src.read_text() for m in macros):
            obj = objdir / (src.stem + ".o")
            procs.append((src, subprocess.Popen([_hipcc(), *flags, *defs, "-c", str(src), "-o", str(obj)],
                                                stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        else:
            obj = PKG / "build" / (src.stem + ".o")
        objs.append(obj)
    for src, p in procs:
        o, _ = p.communicate()
        if p.returncode:
            raise RuntimeError(f"hipcc failed on {src.name} ({' '.join(defs)}):\n{o}")
    r = subprocess.run([_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(out), *map(str, objs), "-ldl"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    return out


def build_ab(force: bool = False) -> Path:
    return build_variant("ab", ["-DTT_AB"], force)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    if "--ab" in sys.argv:
        print(build_ab(force="--force" in sys.argv))
