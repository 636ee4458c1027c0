#!/usr/bin/env python3
"""Mutation guard for the fp16 hi/lo split: the parity tests must FAIL on a library whose `lo`
products are compiled out of any one f16-split kernel (csrc/tt_common.h: TT_MUTATE_DROP_LO bit mask).

    python tools/mutation_guard.py build        # here (no GPU): ab/libtt_mut{1,2,4,8,16}.so
    python tools/mutation_guard.py run          # on the GPU box: swaps each variant in, runs the encoder / training parity
                                                # tests WITHOUT -x, restores the product library; writes
                                                # gpurun_out/mutation_guard.json with every failing test of every mutant

Exit code 0 = every mutant was caught by a test of the kind it damages (MUST_FAIL below: a forward test for the forward
kernels, a GRADIENT test for the backward / weight-gradient kernels) AND the product library passes the same tests."""
import json, os, shutil, subprocess, sys
from pathlib import Path

root = Path(__file__).resolve().parent.parent
MASKS = {1: "K2 gru_seq16 / gru_seq16x4p (forward recurrence)", 2: "K7 gru_bwd16 / gru_bwd16x4p (backward recurrence)",
         4: "K1 gemm_rows16 (input projection)", 8: "sgemm16 (input gradients, tiled K1, tiled weight gradients)",
         16: "wgrad16 (dW_ih / dW_hh of the training step)"}
# a substring one of the mutant's failing tests must contain: the kernels that only the backward runs must be caught by a test that
# compares GRADIENTS (their names say so), not by a forward test that happens to share a code path
GRADIENT_TESTS = ("backward", "gradient", "autograd", "train_step_at_bench_size", "table_grad", "dropout_forward_and_backward")
MUST_FAIL = {1: ("test_",), 2: GRADIENT_TESTS, 4: ("test_",), 8: GRADIENT_TESTS, 16: GRADIENT_TESTS}
TESTS = ["tests/test_encoder_gpu.py", "tests/test_train_gpu.py", "tests/test_bench_size_gpu.py"]


def run_tests(tag):
    env = dict(os.environ, TT_TOL_REPORT=f"tol_{tag}")
    tests = [t for t in TESTS if (root / t).exists()]
    r = subprocess.run([sys.executable, "-m", "pytest", *tests, "-m", "gpu", "-q", "--no-header", "-p", "no:cacheprovider", "-rf",
                        "--tb=line"], cwd=root, env=env, capture_output=True, text=True, timeout=2400)
    rep = root / "gpurun_out" / f"tol_{tag}.json"
    rows = json.loads(rep.read_text()) if rep.exists() else []
    worst = max(rows, key=lambda x: x["ratio"], default=None)
    failed = sorted({l.split(" ")[1] for l in r.stdout.splitlines() if l.startswith("FAILED ")})
    tail = [l for l in r.stdout.splitlines() if l.strip()][-1:]
    print(tag, tail, flush=True)
    return r.returncode, worst, tail, failed


if sys.argv[1:] == ["build"]:
    for m in MASKS:
        subprocess.check_call([sys.executable, str(root / "tools" / "build_variant.py"), f"mut{m}", f"-DTT_MUTATE_DROP_LO={m}"])
    sys.exit(0)

target = root / "twotowermlretrieval_amd" / "libtt.so"
keep = target.read_bytes()
out = {"tests": TESTS, "mutants": {}}
ok = True
try:
    rc, worst, tail, failed = run_tests("product")
    out["product"] = {"rc": rc, "worst": worst, "tail": tail, "failed": failed}
    ok = ok and rc == 0
    print("product:", rc, worst, flush=True)
    for m, what in MASKS.items():
        shutil.copyfile(root / "ab" / f"libtt_mut{m}.so", target)
        rc, worst, tail, failed = run_tests(f"mut{m}")
        caught_by = [t for t in failed if any(sub in t.lower() for sub in MUST_FAIL[m])]
        out["mutants"][str(m)] = {"kernel": what, "rc": rc, "caught": bool(caught_by), "must_fail_one_of": list(MUST_FAIL[m]),
                                  "caught_by": caught_by, "failed": failed, "worst": worst, "tail": tail}
        ok = ok and bool(caught_by)
        print(f"mutant {m} ({what}): rc={rc} failed={len(failed)} caught_by={len(caught_by)} worst={worst}", flush=True)
finally:
    target.write_bytes(keep)
(root / "gpurun_out").mkdir(exist_ok=True)
(root / "gpurun_out" / "mutation_guard.json").write_text(json.dumps(out, indent=1))
sys.exit(0 if ok else 1)
