"""GPU parity tests for K4 (fused score + top-k), K5 (merge) and the rank kernel, through the
C ABI (libtt.so).  Bar: BIT-EXACT scores and indices against oracle/tt_oracle.c (same
ascending-index fp32 FMA chain, same tie order), 1e-5 / near-tie-aware against the reference's
own torch outputs (tests/golden/g6_scoring.npz)."""
import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tt():
    import twotowermlretrieval_amd as m
    from twotowermlretrieval_amd import _lib
    _lib.lib()
    assert torch.cuda.is_available()
    return m


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def run(tt, Q, D, k, off=0):
    v, i = tt.score_topk(dev(Q), dev(D), k, idx_offset=off)
    torch.cuda.synchronize()
    return v.cpu().numpy(), i.cpu().numpy()


@pytest.mark.parametrize("k", [5, 10, 50])
def test_golden_reference_topk(tt, golden, k):
    g = golden("g6_scoring.npz")
    Q = synth.unit_rows(int(g["seed_q"]), 32, 256)
    D = synth.unit_rows(int(g["seed_d"]), 4096, 256)
    v, i = run(tt, Q, D, k)
    np.testing.assert_allclose(v, g[f"val_k{k}"], atol=1e-5, rtol=0)  # north_star: cosine within 1e-5
    safe = g["min_gap_top51"] > 1e-6
    assert np.array_equal(i[safe], g[f"idx_k{k}"][safe])
    for b in np.where(~safe)[0]:
        assert set(i[b]) == set(g[f"idx_k{k}"][b]) or np.abs(v[b] - g[f"val_k{k}"][b]).max() < 1e-6


@pytest.mark.parametrize("B,N,d,k", [
    (1, 1, 256, 1), (1, 31, 256, 5), (5, 32, 256, 10), (32, 33, 256, 10), (33, 1000, 256, 10),
    (70, 4103, 256, 16), (7, 2500, 256, 17), (40, 3000, 256, 50), (3, 5000, 256, 64),
    (9, 777, 128, 10), (64, 2049, 128, 50), (4, 600, 64, 7), (1, 100000, 256, 10),
    (2, 3, 256, 10), (1, 9, 64, 64), (11, 900, 32, 10), (40, 2000, 96, 5), (5, 1500, 192, 20),
])
def test_bit_exact_vs_oracle(tt, oracle, B, N, d, k):
    Q = synth.unit_rows(11 + B, B, d)
    D = synth.unit_rows(12 + N, N, d)
    v, i = run(tt, Q, D, k)
    ov, oi = oracle.score_topk(Q, D, k)
    assert np.array_equal(i, oi)
    assert np.array_equal(v, ov)  # bitwise: same FMA chain order
    if N < k:
        assert (i[:, N:] == -1).all() and np.isneginf(v[:, N:]).all()


@pytest.mark.parametrize("B,N,d,k", [(160, 1_500_000, 64, 10), (100, 400_000, 64, 50), (96, 3_000_000, 32, 10)])
def test_paced_chunks_and_shared_pool_draws_vs_oracle(tt, oracle, B, N, d, k):
    # three or more query tiles: the waves of a document chunk pace each other and (first and third shape: chunks long
    # enough for a pool) take the same pool blocks in the same order -- every tile must still be scored exactly once
    # for every query tile
    Q = synth.unit_rows(31 + B, B, d)
    D = synth.unit_rows(32 + B, N, d)
    v, i = run(tt, Q, D, k)
    ov, oi = oracle.score_topk(Q, D, k)
    assert np.array_equal(i, oi)
    assert np.array_equal(v, ov)


def test_non_unit_and_large_magnitudes(tt, oracle):
    rs = np.random.RandomState(5)
    Q = (rs.standard_normal((6, 256)) * 37.0).astype(np.float32)
    D = (rs.standard_normal((1500, 256)) * 1e-3).astype(np.float32)
    v, i = run(tt, Q, D, 10)
    ov, oi = oracle.score_topk(Q, D, 10)
    assert np.array_equal(i, oi) and np.array_equal(v, ov)


def test_exact_ties_index_ascending(tt, oracle):
    D = synth.unit_rows(708, 4096, 256).copy()
    D[99] = D[7]
    D[3000] = D[7]
    D[4095] = D[7]
    v, i = run(tt, D[7:8], D, 5)
    assert list(i[0, :4]) == [7, 99, 3000, 4095] and v[0, 0] == v[0, 3]
    # every document identical: top-k must be indices 0..k-1 for every query
    D2 = np.repeat(D[:1], 1000, axis=0)
    v, i = run(tt, D[:3], D2, 10)
    assert (i == np.arange(10)).all()
    ov, oi = oracle.score_topk(D[:3], D2, 10)
    assert np.array_equal(i, oi) and np.array_equal(v, ov)


def test_adversarial_ascending_scores_every_doc_inserts(tt, oracle):
    # document n = unit(q + noise shrinking with n): scores increase with the index, so the
    # running threshold is beaten by (almost) every new document.
    rs = np.random.RandomState(9)
    q = synth.unit_rows(1, 1, 256)
    N = 3000
    noise = rs.standard_normal((N, 256)).astype(np.float32)
    w = np.linspace(3.0, 0.0, N, dtype=np.float32)[:, None]
    D = q + w * noise / 16
    D /= np.linalg.norm(D, axis=1, keepdims=True)
    D = D.astype(np.float32)
    for k in (10, 50):
        v, i = run(tt, np.repeat(q, 4, 0), D, k)
        ov, oi = oracle.score_topk(np.repeat(q, 4, 0), D, k)
        assert np.array_equal(i, oi) and np.array_equal(v, ov)


def test_idx_offset_and_shard_merge_equals_full(tt, oracle):
    Q = synth.unit_rows(21, 37, 256)
    D = synth.unit_rows(22, 9001, 256)
    fv, fi = run(tt, Q, D, 10)
    parts = []
    for lo, hi in [(0, 2000), (2000, 2001), (2001, 7000), (7000, 9001)]:
        parts.append(tt.score_topk(dev(Q), dev(D[lo:hi]), 50 if hi - lo >= 1 else 50, idx_offset=lo))
    gv = torch.cat([p[0] for p in parts], 1)
    gi = torch.cat([p[1] for p in parts], 1)
    mv, mi = tt.topk_merge(gv, gi, 10)
    torch.cuda.synchronize()
    assert np.array_equal(mi.cpu().numpy(), fi) and np.array_equal(mv.cpu().numpy(), fv)
    ov, oi = oracle.topk_merge(gv.cpu().numpy(), gi.cpu().numpy(), 10)
    assert np.array_equal(mi.cpu().numpy(), oi) and np.array_equal(mv.cpu().numpy(), ov)


def test_merge_kernel_vs_oracle_random_with_padding(tt, oracle):
    rs = np.random.RandomState(3)
    B, M = 19, 777
    vals = rs.standard_normal((B, M)).astype(np.float32)
    vals[:, ::7] = vals[:, 1:2]  # many exact ties
    idx = rs.permutation(B * M).reshape(B, M).astype(np.int64)
    idx[:, ::5] = -1             # padding entries
    for k in (1, 10, 64):
        mv, mi = tt.topk_merge(dev(vals), dev(idx), k)
        ov, oi = oracle.topk_merge(vals, idx, k)
        assert np.array_equal(mi.cpu().numpy(), oi) and np.array_equal(mv.cpu().numpy(), ov)
    # fewer valid candidates than k
    idx2 = np.full((2, 9), -1, dtype=np.int64)
    idx2[:, :3] = [[5, 2, 9], [1, 0, 3]]
    mv, mi = tt.topk_merge(dev(vals[:2, :9]), dev(idx2), 5)
    ov, oi = oracle.topk_merge(vals[:2, :9], idx2, 5)
    assert np.array_equal(mi.cpu().numpy(), oi) and np.array_equal(mv.cpu().numpy(), ov)


def test_score_rank_vs_oracle_and_golden(tt, oracle, golden):
    g = golden("g7_batch_eval.npz")
    tgt = np.arange(len(g["q"]))
    r = tt.score_rank(dev(g["q"]), dev(g["d"]), dev(tgt)).cpu().numpy()
    assert np.array_equal(r, oracle.score_rank(g["q"], g["d"], tgt))
    assert abs(np.mean(1.0 / r) - float(g["mrr"])) < 1e-9
    Q = synth.unit_rows(31, 9, 256)
    D = synth.unit_rows(32, 5000, 256)
    D[77] = D[4000]
    tgt = np.array([0, 77, 4000, 4999, 5, 6, 7, 8, 9])
    r = tt.score_rank(dev(Q), dev(D), dev(tgt)).cpu().numpy()
    assert np.array_equal(r, oracle.score_rank(Q, D, tgt))


def test_single_query_1d_and_index_object(tt, oracle):
    Q = synth.unit_rows(41, 1, 256)
    D = synth.unit_rows(42, 2222, 256)
    ix = tt.BruteForceIndex(dev(D))
    v, i = ix.search(dev(Q[0]), k=10)  # [H] query like sim_scores.squeeze(0) at evaluators.py:185
    assert v.shape == (10,) and i.dtype == torch.int64
    ov, oi = oracle.score_topk(Q, D, 10)
    assert np.array_equal(i.cpu().numpy(), oi[0]) and np.array_equal(v.cpu().numpy(), ov[0])
    assert ix.ntotal == 2222


def test_baseline_config2_size_properties(tt):
    """BASELINE configs[1]: 1M x 256 fp32, B=1024, top-10 -- too big for the CPU oracle, so check
    size-independent properties: planted documents are found, returned scores equal an independent
    fp64 recomputation within 1e-5, rows are sorted, two-shard search + merge == unsharded search."""
    g = torch.Generator(device="cuda").manual_seed(1)
    N, B, d = 1_000_000, 1024, 256
    D = torch.randn(N, d, device="cuda", generator=g)
    D /= D.norm(dim=1, keepdim=True)
    Q = torch.randn(B, d, device="cuda", generator=g)
    Q /= Q.norm(dim=1, keepdim=True)
    plant = torch.randperm(N, device="cuda", generator=g)[:B]
    D[plant] = Q
    v, i = tt.score_topk(Q, D, 10)
    torch.cuda.synchronize()
    assert (i[:, 0] == plant).all()
    assert (v[:, 0] - 1).abs().max() < 1e-5
    assert (v[:, 1:] <= v[:, :-1]).all()
    ref = (D[i.reshape(-1)].double().view(B, 10, d) * Q.double()[:, None, :]).sum(-1)
    assert (ref - v.double()).abs().max() < 1e-5
    # no document outside the returned set beats the 10th score (checked on a query sample)
    s = Q[:16] @ D.t()
    s.scatter_(1, i[:16], -2.0)
    assert (s.max(dim=1).values <= v[:16, 9] + 1e-6).all()
    half = N // 2 + 13
    a = tt.score_topk(Q, D[:half], 50)
    b = tt.score_topk(Q, D[half:], 50, idx_offset=half)
    mv, mi = tt.topk_merge(torch.cat([a[0], b[0]], 1), torch.cat([a[1], b[1]], 1), 10)
    assert torch.equal(mi, i) and torch.equal(mv, v)


@pytest.mark.parametrize("d,B,N,k", [(512, 5, 3000, 10), (512, 40, 20001, 10), (320, 16, 4097, 5), (384, 17, 9000, 50),
                                     (448, 1, 700, 64), (512, 33, 300000, 10), (512, 3, 31, 8)])
def test_wide_embeddings_16_query_tiles_bit_exact(oracle, d, B, N, k):
    """256 < d <= 512 (score_topk16_kernel: 16-query tiles on v_mfma_f32_16x16x4_f32): still the oracle's fp32 FMA
    chain, bit for bit, with the same tie order; N = 300000 exercises the sample pass."""
    import twotowermlretrieval_amd as tt
    Q = synth.unit_rows(500 + d + B, B, d)
    D = synth.unit_rows(600 + N, N, d).copy()
    if N > 1000:
        D[N // 2] = D[7]                     # an exact tie: the lower index must come first
        Q[0] = D[7]
    k = min(k, N)
    v, i = tt.score_topk(torch.from_numpy(Q).cuda(), torch.from_numpy(D).cuda(), k, 5)
    torch.cuda.synchronize()
    ov, oi = oracle.score_topk(Q, D, k, idx_offset=5)
    assert np.array_equal(i.cpu().numpy(), oi) and np.array_equal(v.cpu().numpy(), ov)
    if N > 1000:
        assert list(oi[0][:2]) == [12, N // 2 + 5]


def test_a_wave_that_gives_up_its_pool_draw_is_redone_on_the_device(tt, oracle):
    """The one wait of the exact kernel that cannot be skipped without losing documents (a wave waiting for a chunk-mate's draw
    from the shared tile pool, B >= 96) is bounded; a wave whose budget runs out marks its lists (+inf, TT_TOPK_INVALID_INDEX)
    and tt_score_topk_f32 redoes the affected query tiles on the static split, on the device (round 4's marker was read by
    nobody: the merge would have ranked it first).  The comparison build forces the give-up for EVERY wave that did not draw
    itself (TT_DRAW_POLLS=-1): the redo flags are then raised and the results are still the product run's, bit for bit, and the
    oracle's."""
    import ctypes as C
    from conftest import ab_library
    from twotowermlretrieval_amd import _lib
    B, N, d, k = 200, 700_000, 128, 10
    L = _lib.lib()
    off = L.tt_score_topk_redo_flags_offset(B, N, d, k)
    assert off != C.c_size_t(-1).value, "this shape should draw from a shared pool"
    assert L.tt_score_topk_redo_flags_offset(32, N, d, k) == C.c_size_t(-1).value       # fewer than three query tiles: no pool
    Q, D = synth.unit_rows(71, B, d), synth.unit_rows(72, N, d)
    D[600_000] = Q[5]                                  # a document deep in the pool's part of the corpus is query 5's best
    qd, dd = dev(Q), dev(D)
    ws = torch.zeros(L.tt_score_topk_workspace_bytes(B, N, d, k), dtype=torch.uint8, device="cuda")
    ntile = (B + 31) // 32
    v0, i0 = tt.score_topk(qd, dd, k, 0, ws)
    torch.cuda.synchronize()
    assert int(ws[off:off + 4 * ntile].view(torch.int32).ne(0).sum()) == 0            # an ordinary run redoes nothing
    with ab_library(TT_DRAW_POLLS=-1):
        v1, i1 = tt.score_topk(qd, dd, k, 0, ws)
        torch.cuda.synchronize()
        redone = int(ws[off:off + 4 * ntile].view(torch.int32).ne(0).sum())
    assert redone > 0, "the forced give-up did not happen"
    assert torch.equal(i1, i0) and torch.equal(v1, v0)
    assert int(i1.max()) < N and int(i1[5, 0]) == 600_000 and bool(torch.isfinite(v1).all())
    for q in (0, 5, 199):
        ov, oi = oracle.score_topk(Q[q:q + 1], D, k)
        assert np.array_equal(i1[q].cpu().numpy(), oi[0]) and np.array_equal(v1[q].cpu().numpy(), ov[0]), q
