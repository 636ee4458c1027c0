"""Pins the CPU oracle (oracle/tt_oracle.c) to outputs of the reference itself.

The fixtures in tests/golden/*.npz were produced by importing the reference's
backend/model.py etc. (tests/golden/gen_golden.py).  Tolerances: the oracle
restates the same fp32 math with its own summation order, so encoder outputs
agree to ~1e-6; gradients to ~1e-5 relative.
"""
import json

import numpy as np
import pytest

import synth
from conftest import GOLDEN, assert_grad_close

ATOL = 2e-6


def _enc(oracle, g, key_ids, dims, layers=1, bi=False, normalize=True):
    V, E, H, seed = [int(x) for x in dims[:4]]
    table = synth.make_table(seed, V, E)
    sd = synth.make_encoder_state(seed + 1, E, H, layers, bi)
    quads = synth.weight_quads(sd, layers, bi)
    return oracle.encoder_forward(g[key_ids], table, quads, H, layers, bi,
                                  sd.get("projection.weight"), sd.get("projection.bias"), normalize)


def test_g1_encoder_uni_small(oracle, golden):
    g = golden("g1_encoder_uni.npz")
    y = _enc(oracle, g, "small_ids", g["small_dims"])
    np.testing.assert_allclose(y, g["small_out"], atol=ATOL, rtol=0)
    np.testing.assert_allclose(np.linalg.norm(y, axis=1), 1.0, atol=1e-6)
    # quirk a2-1: rows that differ only beyond position count_nonzero give identical output
    assert np.array_equal(y[2], y[3])
    assert np.array_equal(g["small_out"][2], g["small_out"][3])


def test_g1_encoder_uni_northstar_shape(oracle, golden):
    g = golden("g1_encoder_uni.npz")
    y = _enc(oracle, g, "big_ids", g["big_dims"])
    np.testing.assert_allclose(y, g["big_out"], atol=ATOL, rtol=0)


def test_g2_encoder_bidirectional_stacked(oracle, golden):
    g = golden("g2_encoder_bi.npz")
    y = _enc(oracle, g, "ids", g["dims"], layers=2, bi=True)
    np.testing.assert_allclose(y, g["out"], atol=ATOL, rtol=0)


def test_g3_encoder_no_normalize(oracle, golden):
    g = golden("g3_encoder_nonorm.npz")
    y = _enc(oracle, g, "ids", g["dims"], normalize=False)
    np.testing.assert_allclose(y, g["out"], atol=ATOL, rtol=0)
    assert np.abs(np.linalg.norm(y, axis=1) - 1).max() > 1e-3


@pytest.mark.parametrize("tag,margin", [("uni", 0.5), ("uni", 0.2), ("bi", 0.5), ("bi", 0.2)])
def test_g4_triplet_loss_and_grads(oracle, golden, tag, margin):
    g = golden("g4_triplet.npz")
    V, E, H, seed, layers, bi = [int(x) for x in g[f"{tag}_dims"]]
    mt = f"{tag}_m{int(margin * 10)}"
    # loss + embedding gradients from the recorded embeddings
    loss, dq, dp, dn = oracle.triplet_loss(g[f"{mt}_emb_q"], g[f"{mt}_emb_p"], g[f"{mt}_emb_n"], margin)
    assert abs(loss - float(g[f"{mt}_loss"])) < 1e-6
    for nm, d in zip("qpn", (dq, dp, dn)):
        np.testing.assert_allclose(d, g[f"{mt}_demb_{nm}"], atol=2e-7, rtol=1e-4)
    assert g[f"{mt}_hinge"].any()
    # full chain: forward + backward through the encoder for each tower
    table = synth.make_table(seed, V, E)
    towers = {"query_encoder.": [(g[f"{tag}_q"], dq)], "doc_encoder.": [(g[f"{tag}_p"], dp), (g[f"{tag}_n"], dn)]}
    for ti, (prefix, calls) in enumerate(towers.items()):
        sd = synth.make_encoder_state(seed + 10 + ti, E, H, layers, bi, prefix=prefix)
        quads = synth.weight_quads(sd, layers, bi, prefix)
        pw, pb = sd.get(prefix + "projection.weight"), sd.get(prefix + "projection.bias")
        tot = None
        for ids, d_out in calls:
            emb = oracle.encoder_forward(ids, table, quads, H, layers, bi, pw, pb, True)
            grads, gpw, gpb = oracle.encoder_backward(ids, table, quads, H, d_out, layers, bi, pw, pb, True)
            flat = [x for quad in grads for x in quad] + ([gpw, gpb] if bi else [])
            tot = flat if tot is None else [a + b for a, b in zip(tot, flat)]
        names = []
        for layer in range(layers):
            for d in range(2 if bi else 1):
                sfx = f"_l{layer}" + ("_reverse" if d else "")
                names += [f"{prefix}rnn.{n}{sfx}" for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
        if bi:
            names += [prefix + "projection.weight", prefix + "projection.bias"]
        for name, got in zip(names, tot):
            want = g[f"{mt}_grad_{name}"]
            assert_grad_close(got, want, what=name, floor=1e-6)


def test_g5_clip_adam(oracle, golden):
    g = golden("g5_clip_adam.npz")
    n_t = 4
    p = np.concatenate([g[f"p0_{i}"].ravel() for i in range(n_t)]).astype(np.float32)
    m = np.zeros_like(p)
    v = np.zeros_like(p)
    for step in range(3):
        gr = np.concatenate([g[f"g{step}_{i}"].ravel() for i in range(n_t)]).astype(np.float32)
        tn = oracle.clip_adam_step(p, gr, m, v, step + 1, 5e-5, max_norm=1.0)
        assert abs(tn - float(g[f"norm_{step}"])) / float(g[f"norm_{step}"]) < 1e-6
        want = np.concatenate([g[f"p{step + 1}_{i}"].ravel() for i in range(n_t)])
        np.testing.assert_allclose(p, want, atol=1e-9, rtol=1e-6)


@pytest.mark.parametrize("k", [5, 10, 50])
def test_g6_scoring_topk(oracle, golden, k):
    g = golden("g6_scoring.npz")
    Q = synth.unit_rows(int(g["seed_q"]), 32, 256)
    D = synth.unit_rows(int(g["seed_d"]), 4096, 256)
    val, idx = oracle.score_topk(Q, D, k)
    # cosine scores within 1e-5 of the reference (north_star tolerance)
    np.testing.assert_allclose(val, g[f"val_k{k}"], atol=1e-5, rtol=0)
    # indices: bit-identical wherever the reference's own top-51 gap exceeds 2x the fp32
    # accumulation-order drift (3e-7); inside a near-tie only the SET may differ.
    safe = g["min_gap_top51"] > 1e-6
    assert safe.sum() >= 28
    assert np.array_equal(idx[safe], g[f"idx_k{k}"][safe])
    for b in np.where(~safe)[0]:
        assert set(idx[b]) == set(g[f"idx_k{k}"][b]) or np.abs(val[b] - g[f"val_k{k}"][b]).max() < 1e-6
    assert (np.diff(val, axis=1) <= 0).all()


def test_g6_single_query_call_site(oracle, golden):
    g = golden("g6_scoring.npz")
    Q = synth.unit_rows(int(g["seed_q"]), 32, 256)
    D = synth.unit_rows(int(g["seed_d"]), 4096, 256)
    val, idx = oracle.score_topk(Q[:1], D, 10)
    np.testing.assert_allclose(val[0], g["single_val"], atol=1e-5, rtol=0)
    if g["min_gap_top51"][0] > 1e-6:
        assert np.array_equal(idx[0], g["single_idx"])


def test_g6_exact_ties_defined_order(oracle, golden):
    g = golden("g6_scoring.npz")
    D = synth.unit_rows(int(g["seed_d"]), 4096, 256).copy()
    D[99] = D[7]
    D[3000] = D[7]
    val, idx = oracle.score_topk(D[7:8], D, 5)
    # the three bitwise-equal rows tie exactly; the oracle DEFINES index-ascending order,
    # the reference returns the same set in an unspecified order.
    assert list(idx[0, :3]) == [7, 99, 3000]
    assert val[0, 0] == val[0, 1] == val[0, 2]
    assert set(g["tie_idx_torch_unspecified"][0, :3]) == {7, 99, 3000}
    np.testing.assert_allclose(val[0], g["tie_val"][0], atol=1e-5, rtol=0)


def test_g7_batch_evaluator_rank_metrics(oracle, golden):
    g = golden("g7_batch_eval.npz")
    rank = oracle.score_rank(g["q"], g["d"], np.arange(len(g["q"])))
    assert abs(np.mean(1.0 / rank) - float(g["mrr"])) < 1e-9
    for k in (1, 5, 10):
        assert abs(np.mean(rank <= k) - float(g[f"recall{k}"])) < 1e-12
    # validation loss = mean over the 3 batches of triplet loss (evaluators.py:36-37,76)
    losses = [oracle.triplet_loss(g["q"][s:s + 16], g["d"][s:s + 16], g["n"][s:s + 16], 0.5, False)[0]
              for s in range(0, 48, 16)]
    assert abs(np.mean(losses) - float(g["val_loss"])) < 1e-6


def test_topk_merge_equals_topk_of_concatenation(oracle):
    Q = synth.unit_rows(1, 5, 64)
    D = synth.unit_rows(2, 3000, 64)
    full_v, full_i = oracle.score_topk(Q, D, 10)
    parts = [oracle.score_topk(Q, D[s:s + 700], 10, idx_offset=s) for s in range(0, 3000, 700)]
    mv, mi = oracle.topk_merge(np.concatenate([p[0] for p in parts], 1),
                               np.concatenate([p[1] for p in parts], 1), 10)
    assert np.array_equal(mi, full_i) and np.array_equal(mv, full_v)


def test_topk_fewer_docs_than_k(oracle):
    Q = synth.unit_rows(1, 2, 16)
    D = synth.unit_rows(2, 3, 16)
    v, i = oracle.score_topk(Q, D, 5)
    assert (i[:, 3:] == -1).all() and np.isneginf(v[:, 3:]).all() and (i[:, :3] >= 0).all()


def test_g10_error_cases(oracle):
    err = json.loads((GOLDEN / "g10_errors.json").read_text())
    assert err["all_zero_row"].startswith("RuntimeError")
    table = synth.make_table(101, 64, 16)
    quads = synth.weight_quads(synth.make_encoder_state(102, 16, 32))
    with pytest.raises(oracle.OracleError) as e:
        oracle.encoder_forward(np.array([[3, 4, 0], [0, 0, 0]]), table, quads, 32)
    assert e.value.code == oracle.O_ERR_ZERO_LENGTH
    assert err["index_out_of_range"].startswith("IndexError")
    with pytest.raises(oracle.OracleError) as e:
        oracle.encoder_forward(np.array([[1, 64, 2]]), table, quads, 32)
    assert e.value.code == oracle.O_ERR_BAD_INDEX


def test_dropout_mask_statistics_and_eval_identity(oracle):
    """The build defines the inter-layer dropout mask (torch's RNG stream cannot be matched): check it
    behaves like Bernoulli(1-p)/(1-p), differs per layer and per seed, and that p = 0 is the identity."""
    for p in (0.2, 0.5):
        m = oracle.dropout_mask(12345, 0, 400_000, p)
        keep = m != 0
        assert abs(keep.mean() - (1 - p)) < 4e-3
        assert np.allclose(m[keep], 1.0 / (1.0 - p))
        assert abs(m.mean() - 1.0) < 1e-2                       # expectation preserved
        assert abs(np.corrcoef(keep[:-1], keep[1:])[0, 1]) < 1e-2  # no serial correlation
    a, b, c = (oracle.dropout_mask(s, l, 10_000, 0.2) for s, l in ((1, 0), (1, 1), (2, 0)))
    assert (a != b).mean() > 0.2 and (a != c).mean() > 0.2
    V, E, H = 64, 16, 32
    table = synth.make_table(1, V, E)
    sd = synth.make_encoder_state(2, E, H, 2, True)
    quads = synth.weight_quads(sd, 2, True)
    ids = synth.make_ids(3, 6, 8, V)
    y0 = oracle.encoder_forward(ids, table, quads, H, 2, True, sd["projection.weight"], sd["projection.bias"])
    y1 = oracle.encoder_forward(ids, table, quads, H, 2, True, sd["projection.weight"], sd["projection.bias"],
                                dropout_p=0.2, dropout_seed=7)
    assert np.abs(y0 - y1).max() > 1e-3


@pytest.mark.parametrize("tag", ["uni", "bi"])
def test_g12_trainable_embedding_table_gradient(oracle, golden, tag):
    """Without GloVe vectors the reference trains nn.Embedding(padding_idx=0) (model.py:23-27): the oracle's table
    gradient (and, through the same call, every GRU weight gradient) against the reference's autograd."""
    g = golden("g12_table_grad.npz")
    V, E, H, seed, layers, bi = [int(x) for x in g[f"{tag}_dims"]]
    table = synth.make_table(seed, V, E)
    sd = synth.make_encoder_state(seed + 1, E, H, layers, bool(bi))
    quads = synth.weight_quads(sd, layers, bool(bi))
    pw, pb = sd.get("projection.weight"), sd.get("projection.bias")
    ids = g[f"{tag}_ids"]
    out = oracle.encoder_forward(ids, table, quads, H, layers, bool(bi), pw, pb, True)
    np.testing.assert_allclose(out, g[f"{tag}_out"], atol=2e-6, rtol=0)
    grads, gpw, gpb, gt = oracle.encoder_backward(ids, table, quads, H, g[f"{tag}_c"], layers, bool(bi), pw, pb, True,
                                                  table_grad=True)
    want = g[f"{tag}_grad_embedding.weight"]
    assert not gt[0].any() and not want[0].any()            # padding_idx: row 0 gets no gradient
    assert (ids == 0).any() and np.abs(want).max() > 0.1
    assert_grad_close(gt, want, what='embedding.weight')
    names = []
    for layer in range(layers):
        for d in range(2 if bi else 1):
            sfx = f"_l{layer}" + ("_reverse" if d else "")
            names += [f"rnn.{n}{sfx}" for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    for name, got in zip(names, [x for quad in grads for x in quad]):
        w = g[f"{tag}_grad_{name}"]
        assert_grad_close(got, w, what=name, floor=1e-6)


@pytest.mark.parametrize("cell", ["LSTM", "RNN"])
@pytest.mark.parametrize("tag", ["uni", "bi"])
def test_g13_lstm_and_vanilla_rnn_towers(oracle, golden, cell, tag):
    """RNN_TYPE = LSTM / RNN (getattr(nn, rnn_type.upper()), model.py:30; LSTM keeps h_n, :59-60): the oracle's
    forward and every parameter gradient against the reference's outputs and autograd."""
    g = golden("g13_lstm_rnn.npz")
    key = f"{cell}_{tag}"
    V, E, H, seed, layers, bi, gates = [int(x) for x in g[f"{key}_dims"]]
    table = synth.make_table(seed, V, E)
    sd = synth.make_encoder_state(seed + 1, E, H, layers, bool(bi), gates=gates)
    quads = synth.weight_quads(sd, layers, bool(bi))
    pw, pb = sd.get("projection.weight"), sd.get("projection.bias")
    ids = g[f"{key}_ids"]
    out = oracle.encoder_forward(ids, table, quads, H, layers, bool(bi), pw, pb, True, rnn_type=cell)
    np.testing.assert_allclose(out, g[f"{key}_out"], atol=2e-6, rtol=0)
    grads, gpw, gpb = oracle.encoder_backward(ids, table, quads, H, g[f"{key}_c"], layers, bool(bi), pw, pb, True,
                                              rnn_type=cell)
    names = []
    for layer in range(layers):
        for d in range(2 if bi else 1):
            sfx = f"_l{layer}" + ("_reverse" if d else "")
            names += [f"rnn.{n}{sfx}" for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    flat = [x for quad in grads for x in quad]
    if bi:
        names += ["projection.weight", "projection.bias"]
        flat += [gpw, gpb]
    for name, got in zip(names, flat):
        w = g[f"{key}_grad_{name}"]
        assert got.shape == w.shape, name
        assert_grad_close(got, w, what=name, floor=1e-6)
