#!/usr/bin/env python3
"""A/B timing of several builds of libtt.so on ONE box (devices differ by up to ~10 % in wall time):
python tools/ab_run.py ab/libtt_A.so ab/libtt_B.so ... -- tools/shard_time.py [args]; runs each twice, interleaved."""
import shutil, subprocess, sys
from pathlib import Path
root = Path(__file__).resolve().parent.parent
sep = sys.argv.index("--")
libs, cmd = sys.argv[1:sep], sys.argv[sep + 1:]
target = root / "twotowermlretrieval_amd" / "libtt.so"
keep = target.read_bytes()
try:
    for rep in range(2):
        for lib in libs:
            shutil.copyfile(root / lib, target)
            out = subprocess.run([sys.executable, *cmd], cwd=root, capture_output=True, text=True, timeout=600)
            lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
            print(f"== {lib} (rep {rep}) rc={out.returncode}", flush=True)
            for l in lines:
                print(l, flush=True)
            if out.returncode:
                print(out.stderr[-2000:], flush=True)
                sys.exit(1)
finally:
    target.write_bytes(keep)
