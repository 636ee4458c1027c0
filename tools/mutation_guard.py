#!/usr/bin/env python3
"""Mutation guard for the fp16 hi/lo split: the parity tests must FAIL on a library whose `lo`
products are compiled out of any one f16-split kernel (csrc/tt_common.h: TT_MUTATE_DROP_LO bit mask).

    python tools/mutation_guard.py build        # here (no GPU): ab/libtt_mut{1,2,4,8}.so
    python tools/mutation_guard.py run          # on the GPU box: swaps each variant in, runs the encoder / training parity
                                                # tests, restores the product library; writes gpurun_out/r03_mutation_guard.json

Exit code 0 = every mutant was caught (its test run failed) AND the product library passes the same tests."""
import json, os, shutil, subprocess, sys
from pathlib import Path

root = Path(__file__).resolve().parent.parent
MASKS = {1: "K2 gru_seq16 (forward recurrence)", 2: "K7 gru_bwd16 (backward recurrence)",
         4: "K1 gemm_rows16 (input projection)", 8: "sgemm16 (weight / input gradients)"}
TESTS = ["tests/test_encoder_gpu.py", "tests/test_train_gpu.py", "tests/test_bench_size_gpu.py"]


def run_tests(tag):
    env = dict(os.environ, TT_TOL_REPORT=f"tol_{tag}")
    tests = [t for t in TESTS if (root / t).exists()]
    r = subprocess.run([sys.executable, "-m", "pytest", *tests, "-m", "gpu", "-q", "-x", "--no-header", "-p", "no:cacheprovider"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=1500)
    rep = root / "gpurun_out" / f"tol_{tag}.json"
    rows = json.loads(rep.read_text()) if rep.exists() else []
    worst = max(rows, key=lambda x: x["ratio"], default=None)
    tail = [l for l in r.stdout.splitlines() if l.strip()][-3:]
    return r.returncode, worst, tail


if sys.argv[1:] == ["build"]:
    for m in MASKS:
        subprocess.check_call([sys.executable, str(root / "tools" / "build_variant.py"), f"mut{m}", f"-DTT_MUTATE_DROP_LO={m}"])
    sys.exit(0)

target = root / "twotowermlretrieval_amd" / "libtt.so"
keep = target.read_bytes()
out = {"tests": TESTS, "mutants": {}}
ok = True
try:
    rc, worst, tail = run_tests("product")
    out["product"] = {"rc": rc, "worst": worst, "tail": tail}
    ok = ok and rc == 0
    print("product:", rc, worst, flush=True)
    for m, what in MASKS.items():
        shutil.copyfile(root / "ab" / f"libtt_mut{m}.so", target)
        rc, worst, tail = run_tests(f"mut{m}")
        out["mutants"][str(m)] = {"kernel": what, "rc": rc, "caught": rc != 0, "first_failure_worst": worst, "tail": tail}
        ok = ok and rc != 0
        print(f"mutant {m} ({what}): rc={rc} worst={worst}", flush=True)
finally:
    target.write_bytes(keep)
(root / "gpurun_out").mkdir(exist_ok=True)
(root / "gpurun_out" / "r03_mutation_guard.json").write_text(json.dumps(out, indent=1))
sys.exit(0 if ok else 1)
