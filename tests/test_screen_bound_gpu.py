"""The screened search is exact only if the approximate score the REAL screen kernels compute obeys

    |s16 - s| <= eps_q = 1.10e-3 |q| Dmax + 1e-6 (|q| + Dmax)          (csrc/screen.hip: screen_eps)

for every (query, document) pair, where s is the defined exact score (fp32 FMA chain, oracle/tt_oracle.c:o_score_topk;
reference call site backend/evaluators.py:185-186).  No product entry point returns s16, so these tests read it through
the test-only export tt_debug_screen_s16 (include/tt_debug.h): the MAXONLY form of screen_stream_kernel /
screen_kernel<.,NSET> over a corpus whose 32-document tiles each hold 32 copies of ONE document, so the per-tile
maximum is that document's value.  Both arithmetic forms are observed: accumulators starting at +0 (sample pass)
and at -thr (main pass: the filter keeps a document iff t = fl(s16 - thr) >= +0 and reasons about v = t + thr).

Adversarial inputs: d parallel to q with every element just below an fp16 rounding midpoint (all 512 operand
roundings push the same way), all-positive vectors, corpora / queries scaled down to 1e-3 .. 1e-5 (fp16-subnormal
operands: a flush-to-zero in the MFMA would show here), elements near the top of the fp16 range, fp16-exact
operands with heavy cancellation (accumulation error alone), one dominant product next to 255 small ones
(truncation instead of rounding inside the MFMA's adder would show here).  The measured max |err| / eps_q per family
is written to gpurun_out/screen_bound.json and quoted in DESIGN.md.
"""
import ctypes as C
import json
from pathlib import Path

import numpy as np
import pytest
import torch

import synth
from conftest import ROOT

pytestmark = pytest.mark.gpu

D_FEAT = 256
FORMS = (0, 1, 2, 4)  # streaming form; shared-tile form with 16 / 32 / 64 queries per wave


def fma_chain(Q, D):
    """s[b, n] = fp32 FMA chain over the features in ascending order (the oracle's definition of the exact score).
    fl32(q*d + acc) is evaluated in float64 (the product of two fp32 is exact there) and rounded once more to fp32:
    the rare double-rounding cases differ from a true fma by one ulp of s, 1e-4 of the bound under test."""
    acc = np.zeros((Q.shape[0], D.shape[0]), dtype=np.float32)
    Qd, Dd = Q.astype(np.float64), D.astype(np.float64)
    for i in range(Q.shape[1]):
        acc = (np.outer(Qd[:, i], Dd[:, i]) + acc.astype(np.float64)).astype(np.float32)
    return acc


def eps_q(qn, dmax):
    """screen_eps in fp32, as the kernels evaluate it."""
    qn, dmax = np.float32(qn), np.float32(dmax)
    return np.float32(1.10e-3) * qn * dmax + np.float32(1e-6) * (qn + dmax)


class Observer:
    def __init__(self):
        from twotowermlretrieval_amd import _lib
        self.L = _lib.lib()
        self._lib = _lib
        self.L.tt_debug_screen_s16_workspace_bytes.restype = C.c_size_t
        self.L.tt_debug_screen_s16_workspace_bytes.argtypes = [C.c_int, C.c_int64, C.c_int]
        self.L.tt_debug_screen_s16.restype = C.c_int
        self.L.tt_debug_screen_s16.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_int,
                                               C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]

    def build(self, D):
        """fp16 shadow of the corpus with every document repeated 32 times (one document per tile), made by the
        product's own tt_index_build_f16; returns (D16 device tensor, Dmax, largest |element|)."""
        rep = torch.from_numpy(np.repeat(np.ascontiguousarray(D, dtype=np.float32), 32, axis=0)).cuda()
        d16 = torch.empty(rep.shape, dtype=torch.float16, device="cuda")
        stats = torch.zeros(2, dtype=torch.float32, device="cuda")
        self._lib.check(self.L.tt_index_build_f16(rep.data_ptr(), rep.shape[0], D_FEAT, d16.data_ptr(), stats.data_ptr(),
                                                  torch.cuda.current_stream().cuda_stream))
        dmax, amax = (float(x) for x in stats.tolist())
        return d16, dmax, amax

    def observe(self, Q, d16, dmax, form, thr=None):
        B, N = Q.shape[0], d16.shape[0]
        q = torch.from_numpy(np.ascontiguousarray(Q, dtype=np.float32)).cuda()
        out = torch.full((B, N // 32), float("nan"), dtype=torch.float32, device="cuda")
        ws = torch.empty(self.L.tt_debug_screen_s16_workspace_bytes(B, N, form), dtype=torch.uint8, device="cuda")
        t = torch.from_numpy(np.ascontiguousarray(thr, dtype=np.float32)).cuda() if thr is not None else None
        self._lib.check(self.L.tt_debug_screen_s16(q.data_ptr(), B, d16.data_ptr(), N, dmax,
                                                   t.data_ptr() if t is not None else None, form, out.data_ptr(),
                                                   ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        return out.cpu().numpy()


@pytest.fixture(scope="module")
def obs():
    return Observer()


RESULTS = {}


def run_family(obs, name, Q, D):
    """Observed s16 (C = +0) and v = t + thr (C = -thr, three thresholds) for every pair, every kernel form; returns
    the largest |error| / eps_q seen."""
    Q = np.ascontiguousarray(Q, dtype=np.float32)
    D = np.ascontiguousarray(D, dtype=np.float32)
    d16, dmax, amax = obs.build(D)
    assert amax < 6.0e4 and dmax < 6.0e4, "family outside the range the screen accepts"
    s = fma_chain(Q, D)
    qn = np.sqrt((Q.astype(np.float32) ** 2).sum(1, dtype=np.float32)).astype(np.float32)
    eps = eps_q(qn, dmax)[:, None]
    assert np.all(eps > 0)
    worst = 0.0
    thr_sets = {
        "floor": -(np.float32(1.01) * qn * np.float32(dmax)),            # no seed: lower bound of every score
        "quarter": np.float32(0.25) * qn * np.float32(dmax),             # a typical k-th score
        "median": np.median(s, axis=1).astype(np.float32),               # cancellation: t close to 0 for half the pairs
    }
    per_form = {}
    for form in FORMS:
        got = obs.observe(Q, d16, dmax, form)
        assert got.shape == s.shape and not np.isnan(got).any()
        r = float(np.max(np.abs(got.astype(np.float64) - s.astype(np.float64)) / eps))
        for tname, thr in thr_sets.items():
            t = obs.observe(Q, d16, dmax, form, thr=thr)
            v = (t + thr[:, None].astype(np.float32)).astype(np.float32)  # what the kernel stores: acc + thr_c in fp32
            r = max(r, float(np.max(np.abs(v.astype(np.float64) - s.astype(np.float64)) / eps)))
            # the filter's own statement: t >= +0  <=>  v >= thr
            assert np.array_equal(t >= 0, v >= thr[:, None])
        per_form[str(form)] = r
        worst = max(worst, r)
    RESULTS[name] = {"max_err_over_eps": worst, "per_form": per_form, "dmax": dmax,
                     "pairs": int(s.size), "forms": list(FORMS)}
    return worst


def below_midpoint(x16):
    """fp32 values just BELOW the rounding midpoint above each fp16 value: fp16() rounds every one of them down by
    (almost) half an ulp, the largest relative error 2^-11 an fp16 rounding can make."""
    x16 = x16.astype(np.float16)
    up = np.nextafter(x16, np.float16(np.inf)).astype(np.float32)
    mid = (x16.astype(np.float32) + up) * np.float32(0.5)
    return np.nextafter(mid, np.float32(-np.inf)).astype(np.float32)


def test_random_unit_vectors(obs):
    r = run_family(obs, "random_unit", synth.unit_rows(1, 64, D_FEAT), synth.unit_rows(2, 512, D_FEAT))
    assert r < 1.0


def test_parallel_vectors_on_rounding_midpoints(obs):
    """d = 2^j q with every element just below an fp16 midpoint: sum |q_i d_i| = |q||d| (Cauchy-Schwarz is tight) and
    all 512 operand roundings err in the same direction -- the worst case of the bound's leading term."""
    rs = np.random.RandomState(3)
    # mantissas 1024 (smallest: the relative error of rounding down is largest there) .. 1100, random exponents
    base = (1024 + rs.randint(0, 76, (64, D_FEAT))).astype(np.float32) * np.float32(2.0) ** rs.randint(-14, -8, (64, D_FEAT))
    Q = below_midpoint(base)
    Q *= rs.choice([-1.0, 1.0], Q.shape).astype(np.float32)
    D = np.concatenate([Q * np.float32(2.0 ** j) for j in (-3, 0, 2, 5)])[:512]
    r = run_family(obs, "parallel_midpoints", Q, D)
    assert r < 1.0
    assert r > 0.5, "this family is built to come close to the bound; a small ratio means it is not doing its job"


def test_all_positive_vectors(obs):
    rs = np.random.RandomState(4)
    Q = rs.uniform(0.0, 1.0, (64, D_FEAT)).astype(np.float32)
    D = rs.uniform(0.0, 1.0, (512, D_FEAT)).astype(np.float32)
    D[::2] = below_midpoint(D[::2])
    Q[::2] = below_midpoint(Q[::2])
    assert run_family(obs, "all_positive", Q, D) < 1.0


@pytest.mark.parametrize("scale", [1e-3, 1e-4, 1e-5, 3e-7])
def test_corpus_scaled_into_the_fp16_subnormal_range(obs, scale):
    """Elements of a unit vector are ~0.06; scaled by 1e-3 .. 1e-5 they are fp16 subnormals (< 6.1e-5) with 2^-25
    absolute rounding error each, which the bound's second term budgets.  If v_mfma_f32_16x16x32_f16 flushed subnormal
    operands the error would be the whole score and this test fails by orders of magnitude."""
    D = (synth.unit_rows(5, 512, D_FEAT) * np.float32(scale)).astype(np.float32)
    Q = synth.unit_rows(6, 64, D_FEAT)
    assert run_family(obs, f"corpus_x{scale:g}", Q, D) < 1.0


@pytest.mark.parametrize("scale", [1e-3, 1e-5])
def test_queries_scaled_into_the_fp16_subnormal_range(obs, scale):
    Q = (synth.unit_rows(7, 64, D_FEAT) * np.float32(scale)).astype(np.float32)
    D = synth.unit_rows(8, 512, D_FEAT)
    assert run_family(obs, f"queries_x{scale:g}", Q, D) < 1.0


def test_subnormal_times_large_operands(obs):
    """fp16-subnormal document elements against query elements in the thousands: each product is O(0.1), so a
    flushed operand would cost far more than the 2^-25 |q_i| per element the bound allows."""
    rs = np.random.RandomState(9)
    D = rs.uniform(-6e-5, 6e-5, (512, D_FEAT)).astype(np.float32)
    Q = rs.uniform(-4000.0, 4000.0, (64, D_FEAT)).astype(np.float32)
    assert run_family(obs, "subnormal_x_large", Q, D) < 1.0


def test_elements_near_the_top_of_the_fp16_range(obs):
    rs = np.random.RandomState(10)
    D = (synth.unit_rows(11, 512, D_FEAT) * np.float32(3000.0)).astype(np.float32)
    D[np.arange(512), rs.randint(0, D_FEAT, 512)] = rs.choice([-1.0, 1.0], 512) * rs.uniform(5.0e4, 5.9e4, 512)
    Q = (synth.unit_rows(12, 64, D_FEAT) * np.float32(2000.0)).astype(np.float32)
    Q[np.arange(64), rs.randint(0, D_FEAT, 64)] = rs.uniform(5.0e4, 5.9e4, 64)
    assert run_family(obs, "near_fp16_max", Q, D) < 1.0


def test_fp16_exact_operands_with_cancellation(obs):
    """Operands that fp16 holds exactly: the only error left is the MFMA's fp32 accumulation (budget 256 * 2^-24 of
    sum |terms| per side).  Large alternating terms cancel to a small score."""
    rs = np.random.RandomState(13)
    Q = (rs.randint(-2047, 2048, (64, D_FEAT)) / 2048.0).astype(np.float32)
    D = (rs.randint(-2047, 2048, (512, D_FEAT)) / 2048.0).astype(np.float32)
    assert np.array_equal(Q.astype(np.float16).astype(np.float32), Q)
    r = run_family(obs, "fp16_exact_cancellation", Q, D)
    assert r < 1.0
    # with exact operands only accumulation error is left: it must stay inside ITS share of the bound,
    # 2 * 257 * 2^-24 sum|q_i d_i| <= 3.07e-5 |q| Dmax, i.e. 3 % of eps
    assert r < 0.04, f"accumulation error alone takes {r:.3f} of eps: the MFMA adds worse than the fp32 chain the proof assumes"


def test_one_dominant_product_among_small_ones(obs):
    """One product ~2^20 (or ~2^10) next to 255 products ~0.5: an adder that TRUNCATES the small addends to the large
    one's exponent, instead of rounding each partial sum, loses up to an ulp of the large term per addend -- 255 ulps
    in all, more than the 256 * 2^-24 the proof budgets for the accumulation.  Operands are fp16-exact, so
    accumulation is the only error."""
    rs = np.random.RandomState(14)
    Q = rs.uniform(0.5, 1.0, (64, D_FEAT)).astype(np.float16).astype(np.float32)
    D = rs.uniform(0.5, 1.0, (512, D_FEAT)).astype(np.float16).astype(np.float32)
    D *= rs.choice([-1.0, 1.0], D.shape).astype(np.float32)
    qpos = rs.randint(0, D_FEAT, 64)
    Q[np.arange(64), qpos] *= np.float32(1024.0)
    dpos = np.where(np.arange(512) % 2 == 0, qpos[np.arange(512) % 64], rs.randint(0, D_FEAT, 512))
    D[np.arange(512), dpos] *= np.float32(1024.0)  # even documents: the two large elements meet for query n % 64
    assert np.array_equal(Q.astype(np.float16).astype(np.float32), Q) and np.array_equal(D.astype(np.float16).astype(np.float32), D)
    r = run_family(obs, "dominant_product", Q, D)
    assert r < 1.0
    assert r < 0.04, f"accumulation error alone takes {r:.3f} of eps"


def test_zz_write_report():
    """(runs last in this file) the measured ratios, for DESIGN.md."""
    out = ROOT / "gpurun_out"
    out.mkdir(exist_ok=True)
    (out / "screen_bound.json").write_text(json.dumps(RESULTS, indent=1, sort_keys=True))
    assert RESULTS and max(v["max_err_over_eps"] for v in RESULTS.values()) < 1.0
    print(json.dumps({k: round(v["max_err_over_eps"], 4) for k, v in RESULTS.items()}))
