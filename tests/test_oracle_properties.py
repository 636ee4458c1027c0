"""Size-independent properties of the oracle itself (hypothesis): the checker has to be right before it checks
anything.  Runs on CPU in seconds."""
import numpy as np
from hypothesis import given, settings, strategies as st


def _rows(rs, n, d):
    x = rs.standard_normal((n, d)).astype(np.float32)
    return x / np.linalg.norm(x, axis=1, keepdims=True)


@settings(max_examples=40, deadline=None, derandomize=True)
@given(st.integers(1, 9), st.integers(1, 400), st.sampled_from([32, 64, 256]), st.integers(1, 20), st.integers(0, 2 ** 31 - 1))
def test_score_topk_is_the_sorted_prefix_of_the_fmaf_scores(oracle, B, N, d, k, seed):
    rs = np.random.RandomState(seed)
    Q, D = _rows(rs, B, d), _rows(rs, N, d)
    if N > 3:
        D[N - 1] = D[0]                                   # an exact tie
    v, i = oracle.score_topk(Q, D, k, idx_offset=3)
    s = np.zeros((B, N), dtype=np.float32)
    for x in range(d):                                    # the defined score: fp32 FMA chain, features ascending
        # (fp32 x fp32 is exact in the 64-bit mantissa of long double and so, bar astronomically rare double rounding,
        #  is the sum: this is fmaf)
        s = (Q[:, x:x + 1].astype(np.longdouble) * D[None, :, x].astype(np.longdouble) + s.astype(np.longdouble)).astype(np.float32)
    for b in range(B):
        order = sorted(range(N), key=lambda n: (-s[b, n], n))[:k]
        assert list(i[b, :len(order)]) == [n + 3 for n in order]
        assert np.array_equal(v[b, :len(order)], s[b, order])
        assert (i[b, len(order):] == -1).all() and np.isneginf(v[b, len(order):]).all()


@settings(max_examples=30, deadline=None, derandomize=True)
@given(st.integers(1, 6), st.integers(2, 300), st.integers(1, 12), st.integers(2, 5), st.integers(0, 2 ** 31 - 1))
def test_sharded_search_plus_merge_equals_unsharded(oracle, B, N, k, world, seed):
    """What ShardedIndex relies on: per-shard top-k' (k' >= k) + merge == global top-k, for any shard count."""
    rs = np.random.RandomState(seed)
    Q, D = _rows(rs, B, 64), _rows(rs, N, 64)
    kp = k + rs.randint(0, 5)
    parts_v, parts_i = [], []
    for r in range(world):
        base, rem = divmod(N, world)
        lo = r * base + min(r, rem)
        hi = lo + base + (1 if r < rem else 0)
        if hi > lo:
            v, i = oracle.score_topk(Q, D[lo:hi], kp, idx_offset=lo)
        else:
            v, i = np.full((B, kp), -np.inf, np.float32), np.full((B, kp), -1, np.int64)
        parts_v.append(v)
        parts_i.append(i)
    mv, mi = oracle.topk_merge(np.concatenate(parts_v, 1), np.concatenate(parts_i, 1), k)
    gv, gi = oracle.score_topk(Q, D, k)
    assert np.array_equal(mi, gi) and np.array_equal(mv, gv)


@settings(max_examples=30, deadline=None, derandomize=True)
@given(st.integers(1, 8), st.integers(1, 200), st.integers(0, 2 ** 31 - 1))
def test_score_rank_is_the_position_in_the_full_ordering(oracle, B, N, seed):
    rs = np.random.RandomState(seed)
    Q, D = _rows(rs, B, 32), _rows(rs, N, 32)
    tgt = rs.randint(0, N, B).astype(np.int64)
    rank = oracle.score_rank(Q, D, tgt)
    _, idx = oracle.score_topk(Q, D, min(N, 64))
    for b in range(B):
        pos = np.flatnonzero(idx[b] == tgt[b])
        if len(pos):
            assert rank[b] == pos[0] + 1
        else:
            assert rank[b] > 64
