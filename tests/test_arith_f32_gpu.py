"""The reference's own arithmetic as a PRODUCT option (include/tt.h TT_ENC_F32, RNNEncoder(arith="f32"), config key ARITH):
every matrix product of the encoder calls on the fp32-MFMA kernels -- plain fp32 multiply-adds, as nn.GRU computes them
(backend/model.py:31-37, :59-62) -- instead of three f16 MFMAs on fp16 hi/lo splits.  Held to HALF the split path's tolerances
(outputs 1e-6 absolute, gradients 1e-5 of the tensor's largest element; conftest.py has the split path's 2e-6 / 2e-5) against
the reference-generated fixtures and the oracle, through the product library (no comparison build, no environment)."""
import numpy as np
import pytest
import torch

import synth
from conftest import FWD_ATOL, GRAD_TOL, assert_fwd_close, assert_grad_close
from test_encoder_gpu import make_encoder, run
from test_train_gpu import build_two_tower, dev

pytestmark = pytest.mark.gpu
F32_ATOL, F32_GRAD = FWD_ATOL / 2, GRAD_TOL / 2


def test_g1_fixtures_on_the_fp32_kernels(golden, oracle):
    from twotowermlretrieval_amd import _lib
    assert _lib.lib() is _lib._lib and "ab" not in str(_lib.LIB_PATH.name)      # the product library
    g = golden("g1_encoder_uni.npz")
    for tag in ("small", "big"):
        V, E, H, seed = [int(x) for x in g[f"{tag}_dims"]]
        enc, table, sd = make_encoder(V, E, H, seed)
        split = run(enc, g[f"{tag}_ids"])
        enc.arith = "f32"
        y = run(enc, g[f"{tag}_ids"])
        assert_fwd_close(y, g[f"{tag}_out"], atol=F32_ATOL, what=f"_f32_{tag}")
        assert_fwd_close(split, g[f"{tag}_out"])
        if H in (128, 256):
            assert not np.array_equal(y, split), "TT_ENC_F32 did not change the arithmetic"
        assert enc.split_workgroups(64) == 0
        o = oracle.encoder_forward(g[f"{tag}_ids"], table, synth.weight_quads(sd), H)
        assert_fwd_close(y, o, atol=F32_ATOL, what=f"_f32_oracle_{tag}")


@pytest.mark.parametrize("tag,margin", [("uni", 0.5), ("bi", 0.2)])
def test_g4_training_loop_body_on_the_fp32_kernels(golden, tag, margin):
    """backend/main.py:249-254 on the reference's autograd fixtures with ARITH = f32: forward AND backward carry the bit."""
    from twotowermlretrieval_amd.model import triplet_loss_cosine
    g = golden("g4_triplet.npz")
    V, E, H, seed, layers, bi = [int(x) for x in g[f"{tag}_dims"]]
    m, _ = build_two_tower(V, E, H, seed, layers, bool(bi))
    for enc in (m.query_encoder, m.doc_encoder):
        enc.arith = "f32"
    mt = f"{tag}_m{int(margin * 10)}"
    q, p, n = (dev(g[f"{tag}_{k}"]) for k in "qpn")
    m.zero_grad()
    eq, ep, en = m.encode_query(q), m.encode_document(p), m.encode_document(n)
    assert_fwd_close(eq.detach().cpu().numpy(), g[f"{mt}_emb_q"], atol=F32_ATOL, what="_f32")
    assert_fwd_close(en.detach().cpu().numpy(), g[f"{mt}_emb_n"], atol=F32_ATOL, what="_f32")
    loss = triplet_loss_cosine((eq, ep, en), margin=margin)
    loss.backward()
    assert abs(loss.item() - float(g[f"{mt}_loss"])) < 1e-6
    for name, prm in m.named_parameters():
        if prm.requires_grad:
            assert_grad_close(prm.grad.cpu().numpy(), g[f"{mt}_grad_{name}"], tol=F32_GRAD, what=name, floor=1e-6)


def test_northstar_shape_batch_on_the_fp32_kernels_forward_backward_and_train_step(oracle):
    """E = 300, H = 256, 1 layer (the shape whose products otherwise ALL take the f16 split): forward + backward vs the oracle at
    the tighter tolerance; config key ARITH; the direct train step (both towers f32) equals the autograd path bit for bit."""
    import copy
    import twotowermlretrieval_amd as tt
    V, E, H, B, T = 500, 300, 256, 48, 60
    seed = 910
    enc, table, sd = make_encoder(V, E, H, seed)
    enc.arith = "f32"
    enc.train()
    ids = synth.make_ids(seed + 2, B, T, V, zero_inside=0.05)
    d_out = np.random.RandomState(seed + 3).standard_normal((B, H)).astype(np.float32)
    y = enc(dev(ids))
    y.backward(dev(d_out))
    quads = synth.weight_quads(sd)
    assert_fwd_close(y.detach().cpu().numpy(), oracle.encoder_forward(ids, table, quads, H), atol=F32_ATOL, what="_f32_ns")
    og, _, _ = oracle.encoder_backward(ids, table, quads, H, d_out, 1, False, None, None, True)
    for i, (got, want) in enumerate(zip([p.grad.cpu().numpy() for p in enc._flat_params()], [x for quad in og for x in quad])):
        assert_grad_close(got, want, tol=F32_GRAD, what=f"f32_{i}", floor=1e-6)
    # eval: the prepared / projected caches are the split kernels'; an f32 encoder uses neither and still matches
    enc.eval()
    ye = run(enc, ids)
    assert not enc._proj and not enc._prep
    assert_fwd_close(ye, oracle.encoder_forward(ids, table, quads, H), atol=F32_ATOL, what="_f32_eval")
    # the two-tower model built from a config, trained one step both ways
    torch.manual_seed(3)
    m0 = tt.TwoTowerModel({"VOCAB_SIZE": V, "EMBED_DIM": E, "HIDDEN_DIM": H, "ARITH": "f32"}, table).cuda().train()
    assert m0.query_encoder.arith == m0.doc_encoder.arith == "f32"
    tri = [dev(synth.make_ids(seed + 10 + k, 32, t, V)) for k, t in enumerate((6, 30, 28))]
    outs = []
    for direct in (True, False):
        m = copy.deepcopy(m0)
        opt = tt.FusedClipAdam(m.parameters(), lr=1e-3, max_norm=1.0)
        loss = tt.train_step(m, opt, *tri, margin=0.5, direct=direct)
        torch.cuda.synchronize()
        outs.append((float(loss.item()), opt.flat_params.detach().clone()))
    assert outs[0][0] == outs[1][0] and torch.equal(outs[0][1], outs[1][1])
    with pytest.raises(ValueError):
        tt.RNNEncoder(V, E, H, arith="bf16")
