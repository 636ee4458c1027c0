"""GPU parity tests for the training path: K6 triplet loss, K7 encoder backward (BPTT), K8 fused
clip+Adam -- against the reference's own autograd outputs (tests/golden/g4_triplet.npz,
g5_clip_adam.npz) and the CPU oracle.  Gradient tolerance: conftest.GRAD_TOL = 2e-5 of the tensor's largest
element (observed <= 3e-6: fp32 summation order); outputs conftest.FWD_ATOL = 2e-6.  Dropping the `lo` halves of the
fp16 hi/lo split fails these (tools/mutation_guard.py)."""
import numpy as np
import pytest
import torch

import synth
from conftest import assert_fwd_close, assert_grad_close

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def build_two_tower(V, E, H, seed, layers, bi):
    from twotowermlretrieval_amd.model import TwoTowerModel
    table = synth.make_table(seed, V, E)
    m = TwoTowerModel({"VOCAB_SIZE": V, "EMBED_DIM": E, "HIDDEN_DIM": H, "NUM_LAYERS": layers, "BIDIRECTIONAL": bi,
                       "DROPOUT": 0.0}, table)
    sd = {}
    for i, tower in enumerate(("query_encoder.", "doc_encoder.")):
        sd[tower + "embedding.weight"] = torch.from_numpy(table)
        sd.update({k: torch.from_numpy(v) for k, v in
                   synth.make_encoder_state(seed + 10 + i, E, H, layers, bi, prefix=tower).items()})
    m.load_state_dict(sd)
    return m.cuda().train(), table


@pytest.mark.parametrize("tag,margin", [("uni", 0.5), ("uni", 0.2), ("bi", 0.5), ("bi", 0.2)])
def test_g4_triplet_loss_values_and_embedding_grads(golden, tag, margin):
    from twotowermlretrieval_amd.model import triplet_loss_cosine
    g = golden("g4_triplet.npz")
    mt = f"{tag}_m{int(margin * 10)}"
    e = [dev(g[f"{mt}_emb_{n}"]).requires_grad_(True) for n in "qpn"]
    loss = triplet_loss_cosine(tuple(e), margin=margin)
    loss.backward()
    assert loss.dim() == 0 and abs(loss.item() - float(g[f"{mt}_loss"])) < 1e-6
    for t, n in zip(e, "qpn"):
        np.testing.assert_allclose(t.grad.cpu().numpy(), g[f"{mt}_demb_{n}"], atol=2e-7, rtol=1e-4)


@pytest.mark.parametrize("tag,margin", [("uni", 0.5), ("uni", 0.2), ("bi", 0.5), ("bi", 0.2)])
def test_g4_full_backward_matches_reference_autograd(golden, tag, margin):
    """The reference's own loop body (main.py:249-254) run on this package's model."""
    from twotowermlretrieval_amd.model import triplet_loss_cosine
    g = golden("g4_triplet.npz")
    V, E, H, seed, layers, bi = [int(x) for x in g[f"{tag}_dims"]]
    m, _ = build_two_tower(V, E, H, seed, layers, bool(bi))
    mt = f"{tag}_m{int(margin * 10)}"
    q, p, n = (dev(g[f"{tag}_{k}"]) for k in "qpn")
    m.zero_grad()
    eq, ep, en = m.encode_query(q), m.encode_document(p), m.encode_document(n)
    assert_fwd_close(eq.detach().cpu().numpy(), g[f"{mt}_emb_q"])
    assert_fwd_close(en.detach().cpu().numpy(), g[f"{mt}_emb_n"])
    loss = triplet_loss_cosine((eq, ep, en), margin=margin)
    loss.backward()
    assert abs(loss.item() - float(g[f"{mt}_loss"])) < 2e-6
    checked = 0
    for name, prm in m.named_parameters():
        if not prm.requires_grad:
            assert prm.grad is None
            continue
        want = g[f"{mt}_grad_{name}"]
        got = prm.grad.cpu().numpy()
        assert_grad_close(got, want, what=name, floor=1e-6)
        checked += 1
    assert checked == (8 if tag == "uni" else 36)  # 8 / 36 trainable tensors


@pytest.mark.parametrize("B,T,E,H,layers,bi", [(40, 30, 300, 256, 1, False), (19, 21, 200, 128, 2, True),
                                               (6, 64, 48, 64, 3, False),
                                               (300, 7, 20, 32, 1, True),   # B >= 256: the projection gradient's split-K path
                                               (2050, 4, 20, 32, 1, True)])  # B >= 2048: the projection head as GEMMs, then backward
def test_encoder_backward_vs_oracle(oracle, B, T, E, H, layers, bi):
    from twotowermlretrieval_amd.model import RNNEncoder
    V, seed = 300, 77 + B
    table = synth.make_table(seed, V, E)
    sd = synth.make_encoder_state(seed + 1, E, H, layers, bi)
    enc = RNNEncoder(V, E, H, pretrained_embeddings=table, num_layers=layers, bidirectional=bi)
    full = {"embedding.weight": torch.from_numpy(table)}
    full.update({k: torch.from_numpy(v) for k, v in sd.items()})
    enc.load_state_dict(full)
    enc = enc.cuda().train()
    ids = synth.make_ids(seed + 2, B, T, V, zero_inside=0.05)
    rs = np.random.RandomState(seed + 3)
    d_out = rs.standard_normal((B, H)).astype(np.float32)
    y = enc(dev(ids))
    y.backward(dev(d_out))
    quads = synth.weight_quads(sd, layers, bi)
    og, gpw, gpb = oracle.encoder_backward(ids, table, quads, H, d_out, layers, bi, sd.get("projection.weight"),
                                           sd.get("projection.bias"), True)
    flat_want = [x for quad in og for x in quad] + ([gpw, gpb] if bi else [])
    flat_got = [p.grad.cpu().numpy() for p in enc._flat_params()]
    for i, (got, want) in enumerate(zip(flat_got, flat_want)):
        assert_grad_close(got, want, what=str(i), floor=1e-6)


def test_g5_fused_clip_adam_matches_torch_sequence(golden):
    from twotowermlretrieval_amd.trainer import FusedClipAdam
    g = golden("g5_clip_adam.npz")
    params = [torch.nn.Parameter(dev(g[f"p0_{i}"])) for i in range(4)]
    opt = FusedClipAdam(params, lr=5e-5, max_norm=1.0)
    for step in range(3):
        opt.zero_grad()
        for i, prm in enumerate(params):
            prm.grad.copy_(dev(g[f"g{step}_{i}"]))
        tn = opt.step()
        assert abs(tn.item() - float(g[f"norm_{step}"])) / float(g[f"norm_{step}"]) < 1e-6
        for i, prm in enumerate(params):
            np.testing.assert_allclose(prm.detach().cpu().numpy(), g[f"p{step + 1}_{i}"], atol=1e-9, rtol=1e-6)


def test_fused_clip_adam_tolerates_grad_reset_and_matches_oracle(oracle):
    from twotowermlretrieval_amd.trainer import FusedClipAdam
    rs = np.random.RandomState(4)
    p0 = [rs.standard_normal(s).astype(np.float32) for s in [(33, 7), (5,), (64, 64)]]
    params = [torch.nn.Parameter(dev(a)) for a in p0]
    opt = FusedClipAdam(params, lr=1e-3, max_norm=0.5)
    flat = np.concatenate([a.ravel() for a in p0])
    m = np.zeros_like(flat)
    v = np.zeros_like(flat)
    for step in range(1, 5):
        gs = [rs.standard_normal(a.shape).astype(np.float32) * (10.0 if step % 2 else 0.01) for a in p0]
        for prm, gnp in zip(params, gs):
            prm.grad = dev(gnp) if step != 3 else None  # fresh tensors / None, like zero_grad(set_to_none=True)
        if step == 3:
            gs = [np.zeros_like(a) for a in p0]
        opt.step()
        oracle.clip_adam_step(flat, np.concatenate([x.ravel() for x in gs]), m, v, step, 1e-3, max_norm=0.5)
        got = np.concatenate([prm.detach().cpu().numpy().ravel() for prm in params])
        np.testing.assert_allclose(got, flat, atol=1e-8, rtol=2e-6)


def test_train_steps_reduce_loss_and_match_reference_loop_with_torch_optimizer():
    """Same model, two optimisers: the reference's torch calls (clip_grad_norm_ + Adam, main.py:257-259)
    and the fused K8 path must walk the same trajectory."""
    from twotowermlretrieval_amd.model import triplet_loss_cosine
    from twotowermlretrieval_amd.trainer import FusedClipAdam, train_step
    V, E, H = 200, 52, 64
    ma, _ = build_two_tower(V, E, H, 5, 1, False)
    mb, _ = build_two_tower(V, E, H, 5, 1, False)
    opt_t = torch.optim.Adam(ma.parameters(), lr=1e-3)
    opt_f = FusedClipAdam(mb.parameters(), lr=1e-3, max_norm=1.0)
    q, p, n = (dev(synth.make_ids(s, 32, t, V)) for s, t in ((1, 6), (2, 20), (3, 18)))
    la, lb = [], []
    for _ in range(6):
        opt_t.zero_grad()
        loss = triplet_loss_cosine((ma.encode_query(q), ma.encode_document(p), ma.encode_document(n)), margin=0.5)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(ma.parameters(), max_norm=1.0)
        opt_t.step()
        la.append(loss.item())
        lb.append(train_step(mb, opt_f, q, p, n, margin=0.5, concurrent_towers=(_ % 2 == 0)).item())
    assert la[-1] < la[0]
    np.testing.assert_allclose(la, lb, atol=2e-5)
    for (na, pa), (nb, pb) in zip(ma.named_parameters(), mb.named_parameters()):
        np.testing.assert_allclose(pa.detach().cpu().numpy(), pb.detach().cpu().numpy(), atol=2e-5, err_msg=na)


@pytest.mark.parametrize("layers,bi,p", [(2, True, 0.2), (3, False, 0.5)])
def test_inter_layer_dropout_forward_and_backward_vs_oracle(oracle, layers, bi, p):
    """config.json's model (NUM_LAYERS 2, BIDIRECTIONAL, DROPOUT 0.2) in TRAIN mode: the mask is the build's
    counter-based hash (seeded from torch's CPU generator), identical in the oracle."""
    from twotowermlretrieval_amd.model import RNNEncoder
    V, E, H, B, T, seed = 120, 40, 64, 21, 13, 91
    table = synth.make_table(seed, V, E)
    sd = synth.make_encoder_state(seed + 1, E, H, layers, bi)
    enc = RNNEncoder(V, E, H, pretrained_embeddings=table, num_layers=layers, bidirectional=bi, dropout=p)
    full = {"embedding.weight": torch.from_numpy(table)}
    full.update({k: torch.from_numpy(v) for k, v in sd.items()})
    enc.load_state_dict(full)
    enc = enc.cuda()
    ids = synth.make_ids(seed + 2, B, T, V, zero_inside=0.05)
    quads = synth.weight_quads(sd, layers, bi)
    pw, pb = sd.get("projection.weight"), sd.get("projection.bias")
    # eval mode: dropout is the identity
    enc.eval()
    with torch.no_grad():
        y_eval = enc(dev(ids)).cpu().numpy()
    assert_fwd_close(y_eval, oracle.encoder_forward(ids, table, quads, H, layers, bi, pw, pb))
    # train mode: same seed draw as the autograd node will make
    enc.train()
    torch.manual_seed(4242)
    mask_seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    torch.manual_seed(4242)
    y = enc(dev(ids))
    want = oracle.encoder_forward(ids, table, quads, H, layers, bi, pw, pb, True, p, mask_seed)
    assert_fwd_close(y.detach().cpu().numpy(), want)
    assert np.abs(want - y_eval).max() > 1e-3
    d_out = np.random.RandomState(5).standard_normal((B, H)).astype(np.float32)
    y.backward(dev(d_out))
    og, gpw, gpb = oracle.encoder_backward(ids, table, quads, H, d_out, layers, bi, pw, pb, True, p, mask_seed)
    flat_want = [x for quad in og for x in quad] + ([gpw, gpb] if bi else [])
    for i, (prm, w) in enumerate(zip(enc._flat_params(), flat_want)):
        assert_grad_close(prm.grad.cpu().numpy(), w, what=str(i), floor=1e-6)


@pytest.mark.parametrize("tag", ["uni", "bi"])
def test_g12_trainable_embedding_table_matches_reference_autograd(golden, tag):
    """RNNEncoder without pretrained vectors: the table is a trainable nn.Embedding(padding_idx=0) (model.py:23-27).
    Gradient of the table and of every GRU tensor vs the reference's autograd; then a fused clip+Adam step moves the
    table rows that occurred and leaves row 0 and unseen rows alone."""
    from twotowermlretrieval_amd.model import RNNEncoder
    from twotowermlretrieval_amd.trainer import FusedClipAdam
    g = golden("g12_table_grad.npz")
    V, E, H, seed, layers, bi = [int(x) for x in g[f"{tag}_dims"]]
    table = synth.make_table(seed, V, E)
    sd = synth.make_encoder_state(seed + 1, E, H, layers, bool(bi))
    enc = RNNEncoder(V, E, H, pretrained_embeddings=None, num_layers=layers, bidirectional=bool(bi))
    full = {"embedding.weight": torch.from_numpy(table)}
    full.update({k: torch.from_numpy(v) for k, v in sd.items()})
    enc.load_state_dict(full)
    enc = enc.cuda().train()
    assert enc.embedding.weight.requires_grad
    opt = FusedClipAdam(enc.parameters(), lr=1e-2, max_norm=1.0)   # (re-points .data / .grad at its flat buffers)
    ids = dev(g[f"{tag}_ids"])
    y = enc(ids)
    assert_fwd_close(y.detach().cpu().numpy(), g[f"{tag}_out"])
    (y * dev(g[f"{tag}_c"])).sum().backward()
    torch.cuda.synchronize()
    for name, prm in enc.named_parameters():
        want = g[f"{tag}_grad_{name}"]
        got = prm.grad.cpu().numpy()
        assert_grad_close(got, want, what=name, floor=1e-6)
    gt = enc.embedding.weight.grad.cpu().numpy()
    assert not gt[0].any()                                   # padding_idx
    before = enc.embedding.weight.detach().clone()
    opt.step()
    torch.cuda.synchronize()
    moved = (enc.embedding.weight.detach() - before).abs().amax(dim=1).cpu().numpy() > 0
    has_grad = np.abs(gt).max(axis=1) > 0
    assert np.array_equal(moved, has_grad) and has_grad.sum() >= 10 and not moved[0]
    seen = np.zeros(V, dtype=bool)
    seen[np.unique(g[f"{tag}_ids"])] = True
    assert not has_grad[~seen].any()                         # ids that never occur get no gradient


@pytest.mark.parametrize("layers,bi,drop", [(1, False, 0.0), (2, True, 0.0), (2, True, 0.3)])
def test_direct_train_step_equals_the_autograd_path_bit_for_bit(layers, bi, drop):
    """train_step's autograd-free shortcut (gradients written straight into the optimizer's flat buffer) against the same
    step through torch.autograd: same kernels on the same inputs -> identical parameters after three steps."""
    import copy
    import twotowermlretrieval_amd as tt
    V, E, H, B = 80, 20, 32, 24
    cfg = {"VOCAB_SIZE": V, "EMBED_DIM": E, "HIDDEN_DIM": H, "NUM_LAYERS": layers, "BIDIRECTIONAL": bi, "DROPOUT": drop}
    torch.manual_seed(3)
    m1 = tt.TwoTowerModel(cfg, synth.make_table(4, V, E)).cuda().train()
    m2 = copy.deepcopy(m1)
    o1 = tt.FusedClipAdam(m1.parameters(), lr=1e-2, max_norm=1.0)
    o2 = tt.FusedClipAdam(m2.parameters(), lr=1e-2, max_norm=1.0)
    losses = []
    for step in range(3):
        ids = [torch.from_numpy(synth.make_ids(60 + 3 * step + s, B, T, V)).cuda() for s, T in enumerate((6, 11, 9))]
        torch.manual_seed(100 + step)          # the dropout seeds are drawn from torch's CPU generator
        l1 = tt.train_step(m1, o1, *ids, margin=0.5, direct=True)
        torch.manual_seed(100 + step)
        l2 = tt.train_step(m2, o2, *ids, margin=0.5, direct=False)
        losses.append((float(l1.item()), float(l2.item())))
    torch.cuda.synchronize()
    assert all(a == b for a, b in losses), losses
    assert torch.equal(o1.flat_params, o2.flat_params)
    assert torch.equal(o1.exp_avg, o2.exp_avg) and torch.equal(o1.exp_avg_sq, o2.exp_avg_sq)


def test_direct_train_step_reports_bad_input_like_the_reference():
    import twotowermlretrieval_amd as tt
    V, E, H = 50, 20, 32
    m = tt.TwoTowerModel({"VOCAB_SIZE": V, "EMBED_DIM": E, "HIDDEN_DIM": H}, synth.make_table(1, V, E)).cuda().train()
    opt = tt.FusedClipAdam(m.parameters(), lr=1e-3)
    ids = [torch.from_numpy(synth.make_ids(7 + s, 4, T, V)).cuda() for s, T in enumerate((5, 6, 7))]
    ids[1][2] = 0                               # an all-padding row: "Length of all samples has to be greater than 0"
    with pytest.raises(RuntimeError):
        tt.train_step(m, opt, *ids, margin=0.5)
    torch.cuda.synchronize()


@pytest.mark.parametrize("Ba,Ta,Bb,Tb", [(5, 7, 3, 11), (512, 70, 512, 93), (1, 1, 1, 1), (4, 9, 4, 9), (0, 5, 3, 2)])
def test_concat_ids_kernel_is_torch_zeros_plus_two_slice_copies(Ba, Ta, Bb, Tb):
    """tt_concat_ids_i64 (the 2B-row document call of the train step: positives then negatives, padded with id 0 to the longer T)
    against the torch ops it replaced (backend/main.py:244-259 runs the two batches through the same tower)."""
    from twotowermlretrieval_amd.trainer import _concat_ids
    rs = np.random.RandomState(Ba * 131 + Tb)
    a = torch.from_numpy(rs.randint(0, 2 ** 40, size=(Ba, Ta)).astype(np.int64)).cuda()
    b = torch.from_numpy(rs.randint(0, 2 ** 40, size=(Bb, Tb)).astype(np.int64)).cuda()
    got = _concat_ids(a, b)
    torch.cuda.synchronize()
    T = max(Ta, Tb)
    want = torch.zeros((Ba + Bb, T), dtype=torch.int64, device="cuda")
    want[:Ba, :Ta] = a
    want[Ba:, :Tb] = b
    assert got.shape == want.shape and torch.equal(got, want)
    # a non-contiguous view (a collate_fn's slice) goes through .contiguous()
    if Ta > 2 and Ba > 0:
        got2 = _concat_ids(a[:, : Ta - 1], b)
        want2 = torch.zeros((Ba + Bb, max(Ta - 1, Tb)), dtype=torch.int64, device="cuda")
        want2[:Ba, : Ta - 1] = a[:, : Ta - 1]
        want2[Ba:, :Tb] = b
        assert torch.equal(got2, want2)


def test_a_bad_batch_raises_and_leaves_the_weights_untouched():
    """The towers' status words are folded into the gate words behind the gradients and the optimizer kernel is predicated on them
    on the device (tt_clip_adam_step_gated_f32; with a process group the gate is all-reduced with the gradients, so the ranks
    decide together: tests/test_multirank_gpu.py): an id out of range raises the reference's IndexError (nn.Embedding,
    backend/model.py:49), a zero-length row its RuntimeError (pack_padded_sequence, model.py:56), parameters, Adam moments and the
    step number stay what they were, and the next good batch trains as if nothing had happened."""
    import copy
    import twotowermlretrieval_amd as tt
    V, E, H, B = 300, 300, 256, 32
    torch.manual_seed(3)
    m = tt.TwoTowerModel({"VOCAB_SIZE": V, "EMBED_DIM": E, "HIDDEN_DIM": H}, synth.make_table(4, V, E)).cuda().train()
    opt = tt.FusedClipAdam(m.parameters(), lr=1e-3, max_norm=1.0)
    ids = [torch.from_numpy(synth.make_ids(70 + s, B, T, V)).cuda() for s, T in enumerate((7, 20, 25))]
    tt.train_step(m, opt, *ids, margin=0.5)          # one good step: moments are non-zero from here on
    torch.cuda.synchronize()
    ref = copy.deepcopy(m)
    ref_opt = tt.FusedClipAdam(ref.parameters(), lr=1e-3, max_norm=1.0)
    ref_opt.exp_avg.copy_(opt.exp_avg); ref_opt.exp_avg_sq.copy_(opt.exp_avg_sq); ref_opt.step_count = opt.step_count
    before = (opt.flat_params.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), opt.step_count)
    bad_id = [t.clone() for t in ids]
    bad_id[1][3, 0] = V + 5                            # a positive passage with an id out of range
    with pytest.raises(IndexError):
        tt.train_step(m, opt, *bad_id, margin=0.5)
    empty = [t.clone() for t in ids]
    empty[0][5, :] = 0                                 # a query of padding only
    with pytest.raises(RuntimeError):
        tt.train_step(m, opt, *empty, margin=0.5)
    torch.cuda.synchronize()
    assert torch.equal(opt.flat_params, before[0]) and torch.equal(opt.exp_avg, before[1]) and torch.equal(opt.exp_avg_sq, before[2])
    assert opt.step_count == before[3]
    l1 = tt.train_step(m, opt, *ids, margin=0.5)
    l2 = tt.train_step(ref, ref_opt, *ids, margin=0.5)
    torch.cuda.synchronize()
    assert float(l1) == float(l2) and torch.equal(opt.flat_params, ref_opt.flat_params)


def test_deferred_check_raises_one_call_late_and_skips_only_the_bad_step():
    """train_step(defer_check=True): the gate words of step i are copied to pinned memory behind the step and read inside call
    i + 1, after step i + 1 has been enqueued -- no host synchronisation inside a step.  The device already decided that the bad
    step is not applied; its exception comes out of the next call (or of optimizer.settle()), and the trajectory is the eager
    one without the bad batch, bit for bit."""
    import copy
    import twotowermlretrieval_amd as tt
    V, E, H, B = 300, 300, 256, 32
    torch.manual_seed(3)
    m = tt.TwoTowerModel({"VOCAB_SIZE": V, "EMBED_DIM": E, "HIDDEN_DIM": H}, synth.make_table(4, V, E)).cuda().train()
    ref = copy.deepcopy(m)
    opt = tt.FusedClipAdam(m.parameters(), lr=1e-3, max_norm=1.0)
    ref_opt = tt.FusedClipAdam(ref.parameters(), lr=1e-3, max_norm=1.0)
    good = [[torch.from_numpy(synth.make_ids(400 + 3 * i + s, B, T, V)).cuda() for s, T in enumerate((7, 20, 25))] for i in range(3)]
    bad = [t.clone() for t in good[0]]
    bad[2][1, 3] = -1
    tt.train_step(m, opt, *good[0], margin=0.5, defer_check=True)
    tt.train_step(m, opt, *bad, margin=0.5, defer_check=True)              # nothing raised yet
    with pytest.raises(IndexError):
        tt.train_step(m, opt, *good[1], margin=0.5, defer_check=True)      # the bad step's exception, behind this good step
    tt.train_step(m, opt, *good[2], margin=0.5, defer_check=True)
    assert opt.settle() is None                                             # the last step was fine
    for g in good:
        tt.train_step(ref, ref_opt, *g, margin=0.5)
    torch.cuda.synchronize()
    assert opt.step_count == 3 and torch.equal(opt.flat_params, ref_opt.flat_params) and torch.equal(opt.exp_avg_sq, ref_opt.exp_avg_sq)


def test_deferred_checks_of_consecutive_bad_steps_are_all_reported():
    """snapshot_gate registers THIS step's check before it settles the previous one (round 4 settled first: when that raised,
    the step enqueued in the same call was never snapshotted and a bad batch in it went unreported).  Two bad batches in a row:
    the first one's exception comes out of the call that enqueues the second, the second one's out of the next call -- and
    neither step was applied."""
    import copy
    import twotowermlretrieval_amd as tt
    V, E, H, B = 300, 300, 256, 32
    torch.manual_seed(3)
    m = tt.TwoTowerModel({"VOCAB_SIZE": V, "EMBED_DIM": E, "HIDDEN_DIM": H}, synth.make_table(4, V, E)).cuda().train()
    ref = copy.deepcopy(m)
    opt = tt.FusedClipAdam(m.parameters(), lr=1e-3, max_norm=1.0)
    ref_opt = tt.FusedClipAdam(ref.parameters(), lr=1e-3, max_norm=1.0)
    good = [[torch.from_numpy(synth.make_ids(500 + 3 * i + s, B, T, V)).cuda() for s, T in enumerate((7, 20, 25))] for i in range(2)]
    bad_id = [t.clone() for t in good[0]]; bad_id[1][2, 1] = V + 9
    empty = [t.clone() for t in good[0]]; empty[0][4, :] = 0
    tt.train_step(m, opt, *good[0], margin=0.5, defer_check=True)
    tt.train_step(m, opt, *bad_id, margin=0.5, defer_check=True)           # nothing raised yet
    with pytest.raises(IndexError):
        tt.train_step(m, opt, *empty, margin=0.5, defer_check=True)        # the first bad step's exception; this step IS registered
    with pytest.raises(RuntimeError, match="Length of all samples"):
        tt.train_step(m, opt, *good[1], margin=0.5, defer_check=True)      # the second bad step's, one call late as well
    assert opt.settle() is None
    for g in good:
        tt.train_step(ref, ref_opt, *g, margin=0.5)
    torch.cuda.synchronize()
    assert opt.step_count == 2 and torch.equal(opt.flat_params, ref_opt.flat_params)


def test_step_gate_folds_any_number_of_status_words():
    """tt_step_gate_f32 took at most eight status words; a watched model whose step follows more tower calls than that (gradient
    accumulation) failed with TT_ERR_BAD_SHAPE.  Twenty words, eight per launch."""
    import ctypes as C
    from twotowermlretrieval_amd import _lib
    from twotowermlretrieval_amd.trainer import _hip_step_gate
    words = [torch.tensor([w], dtype=torch.int32, device="cuda") for w in [0, 1, 2, 4, 3, 0, 0, 7, 1, 1, 0, 2, 0, 0, 4, 4, 0, 6, 0, 1]]
    gate = torch.full((_lib.TT_STEP_GATE_WORDS,), 9.0, device="cuda")
    _hip_step_gate(words, gate)
    torch.cuda.synchronize()
    vals = [int(w.item()) for w in words]
    assert gate.tolist() == [float(sum((v >> b) & 1 for v in vals)) for b in range(3)] + [0.0]
    _hip_step_gate(words[:3], gate)                                          # and a short list overwrites, it does not add
    assert gate.tolist() == [1.0, 1.0, 0.0, 0.0]
    arr = (C.c_void_p * 1)()
    _lib.check(_lib.lib().tt_step_gate_f32(arr, 0, gate.data_ptr(), torch.cuda.current_stream().cuda_stream))   # no words: zeros
    assert gate.tolist() == [0.0] * 4


def test_a_recurrence_time_out_redoes_the_step_on_the_one_workgroup_kernels():
    """Status bit 2 (a column-split recurrence gave up waiting for a partner workgroup: CUs held by other work) is transient and
    rank-local.  It reaches the optimizer like the data errors -- through the gate behind the gradients, so every rank sees it --
    but train_step does not raise: the step is redone with TT_ENC_ONE_WORKGROUP (ordinary relaunch, nothing to wait for).  The
    time-out cannot be provoked on an idle device, so it is injected: the first document-tower backward of the step ORs bit 2
    into its status word, exactly where gru_bwd16x4p_kernel would.  The redone step equals, bit for bit, a step of a model whose
    encoders run the one-workgroup recurrences from the start; the failed attempt left no trace (step number 1, not 2)."""
    import copy
    import twotowermlretrieval_amd as tt
    from twotowermlretrieval_amd import _lib
    V, E, H, B = 300, 300, 256, 48
    torch.manual_seed(3)
    m = tt.TwoTowerModel({"VOCAB_SIZE": V, "EMBED_DIM": E, "HIDDEN_DIM": H}, synth.make_table(4, V, E)).cuda().train()
    ref = copy.deepcopy(m)
    opt = tt.FusedClipAdam(m.parameters(), lr=1e-3, max_norm=1.0)
    ref_opt = tt.FusedClipAdam(ref.parameters(), lr=1e-3, max_norm=1.0)
    ids = [torch.from_numpy(synth.make_ids(70 + s, B, T, V)).cuda() for s, T in enumerate((7, 20, 25))]
    calls = []
    bwd0 = m.doc_encoder._run_backward

    def flaky_backward(ids_, ws, d_out, *a, status=None, opts=None, **k):
        calls.append(opts)
        out = bwd0(ids_, ws, d_out, *a, status=status, opts=opts, **k)
        if len(calls) == 1:
            status.bitwise_or_(4)          # on the backward's stream, behind its kernels
        return out
    m.doc_encoder._run_backward = flaky_backward
    for direct in (True, False):
        calls.clear()
        loss = tt.train_step(m, opt, *ids, margin=0.5, direct=direct)
        torch.cuda.synchronize()
        assert len(calls) == 2 and not (calls[0] & _lib.TT_ENC_ONE_WORKGROUP) and (calls[1] & _lib.TT_ENC_ONE_WORKGROUP)
        for enc in (ref.query_encoder, ref.doc_encoder):
            enc.one_workgroup = True
        ref_loss = tt.train_step(ref, ref_opt, *ids, margin=0.5, direct=direct)
        torch.cuda.synchronize()
        assert float(loss) == float(ref_loss) and torch.equal(opt.flat_params, ref_opt.flat_params)
        assert opt.step_count == ref_opt.step_count == (1 if direct else 2)
        assert not m.doc_encoder.one_workgroup and not m.query_encoder.one_workgroup   # (the option was the step's, not the model's)
    # a hand-written autograd loop gets the exception itself (nothing can redo somebody else's loop)
    calls.clear()
    opt.watch(m)
    opt.zero_grad()
    loss = tt.triplet_loss_cosine((m.encode_query(ids[0]), m.encode_document(ids[1]), m.encode_document(ids[2])), margin=0.5)
    loss.backward()
    with pytest.raises(tt.model.SplitRecurrenceTimeout):
        opt.step()
    assert opt.step_count == 2


def test_graphed_train_step_replays_the_eager_step_bit_for_bit():
    """GraphedTrainStep: the whole direct step (two towers on two streams, loss, backwards, gate, clip + Adam with the step
    number on the device) captured once and replayed.  Three steps with batches of different widths: parameters, moments, loss
    and step number equal, bit for bit, those of the eager train_step on the same ids padded to the captured widths; a bad
    batch raises the reference's exception out of the replay and leaves everything untouched; the next batch trains."""
    import copy
    import twotowermlretrieval_amd as tt
    V, E, H, B = 300, 300, 256, 64
    torch.manual_seed(9)
    m = tt.TwoTowerModel({"VOCAB_SIZE": V, "EMBED_DIM": E, "HIDDEN_DIM": H}, synth.make_table(4, V, E)).cuda().train()
    ref = copy.deepcopy(m)
    opt = tt.FusedClipAdam(m.parameters(), lr=1e-3, max_norm=1.0)
    ref_opt = tt.FusedClipAdam(ref.parameters(), lr=1e-3, max_norm=1.0)
    before = opt.flat_params.clone()
    step = tt.GraphedTrainStep(m, opt, batch=B, q_width=16, doc_width=48, margin=0.5)
    torch.cuda.synchronize()
    assert torch.equal(opt.flat_params, before) and opt.step_count == 0     # warm-up and capture ran on all-padding ids: gate closed

    def padded(t, w):
        out = torch.zeros((t.shape[0], w), dtype=torch.int64, device=t.device)
        out[:, : t.shape[1]] = t
        return out
    for i, widths in enumerate(((7, 20, 25), (16, 48, 31), (3, 9, 48))):
        ids = [torch.from_numpy(synth.make_ids(200 + 3 * i + s, B, T, V)).cuda() for s, T in enumerate(widths)]
        loss = step(*ids)
        ref_loss = tt.train_step(ref, ref_opt, padded(ids[0], 16), padded(ids[1], 48), padded(ids[2], 48), margin=0.5)
        torch.cuda.synchronize()
        assert float(loss) == float(ref_loss)
        assert torch.equal(opt.flat_params, ref_opt.flat_params) and torch.equal(opt.exp_avg_sq, ref_opt.exp_avg_sq)
        assert opt.step_count == ref_opt.step_count == i + 1
    # the prepared-weight cache of the eval path follows the replayed updates
    m.eval(); ref.eval()
    with torch.no_grad():
        assert torch.equal(m.encode_query(ids[0]), ref.encode_query(ids[0]))
    m.train(); ref.train()
    snap = (opt.flat_params.clone(), opt.exp_avg.clone(), opt.step_count)
    bad = [t.clone() for t in ids]
    bad[1][3, 0] = V + 5
    with pytest.raises(IndexError):
        step(*bad)
    torch.cuda.synchronize()
    assert torch.equal(opt.flat_params, snap[0]) and torch.equal(opt.exp_avg, snap[1]) and opt.step_count == snap[2]
    with pytest.raises(ValueError):
        step(ids[0], padded(ids[1], 64), ids[2])                           # wider than the capture
    loss = step(*ids)
    ref_loss = tt.train_step(ref, ref_opt, padded(ids[0], 16), padded(ids[1], 48), padded(ids[2], 48), margin=0.5)
    torch.cuda.synchronize()
    assert float(loss) == float(ref_loss) and torch.equal(opt.flat_params, ref_opt.flat_params)
    # defer_check: the gate words of step i are looked at inside call i + 1, after step i + 1 has been enqueued -- the bad
    # batch's exception comes out one call late, its step was not applied, the step behind it was
    lazy = tt.GraphedTrainStep(m, opt, batch=B, q_width=16, doc_width=48, margin=0.5, defer_check=True)
    n0 = opt.step_count
    lazy(*ids)
    lazy(*bad)                                  # no exception yet
    with pytest.raises(IndexError):
        lazy(*ids)                              # ... here, behind this (good, applied) step
    lazy.flush()                                # the last step was fine
    torch.cuda.synchronize()
    assert opt.step_count == n0 + 2
    for _ in range(2):
        ref_loss = tt.train_step(ref, ref_opt, padded(ids[0], 16), padded(ids[1], 48), padded(ids[2], 48), margin=0.5)
    torch.cuda.synchronize()
    assert torch.equal(opt.flat_params, ref_opt.flat_params)


def test_graphed_train_step_with_inter_layer_dropout_replays_the_eager_step():
    """The reference's default model (backend/config.json:13-17: 2 layers, bidirectional, dropout 0.2) in a HIP graph: the dropout
    seeds are device words the captured kernels read when they run (TT_ENC_SEED_ON_DEVICE), drawn per call from torch's CPU
    generator in the eager step's order.  With the generator in the same state a replay computes the eager step's bits --
    loss, parameters, moments -- for three different batches (different masks every step), and a replay with another generator
    state computes something else (the mask is not frozen into the graph)."""
    import copy
    import twotowermlretrieval_amd as tt
    V, E, H, B = 300, 200, 256, 64
    torch.manual_seed(19)
    cfg = {"VOCAB_SIZE": V, "EMBED_DIM": E, "HIDDEN_DIM": H, "NUM_LAYERS": 2, "BIDIRECTIONAL": True, "DROPOUT": 0.2}
    m = tt.TwoTowerModel(cfg, synth.make_table(4, V, E)).cuda().train()
    ref = copy.deepcopy(m)
    opt = tt.FusedClipAdam(m.parameters(), lr=1e-3, max_norm=1.0)
    ref_opt = tt.FusedClipAdam(ref.parameters(), lr=1e-3, max_norm=1.0)
    step = tt.GraphedTrainStep(m, opt, batch=B, q_width=16, doc_width=48, margin=0.5)

    def padded(t, w):
        out = torch.zeros((t.shape[0], w), dtype=torch.int64, device=t.device)
        out[:, : t.shape[1]] = t
        return out
    losses = []
    for i, widths in enumerate(((7, 20, 25), (16, 48, 31), (3, 9, 48))):
        ids = [torch.from_numpy(synth.make_ids(700 + 3 * i + s, B, T, V)).cuda() for s, T in enumerate(widths)]
        torch.manual_seed(500 + i)
        loss = step(*ids)
        torch.manual_seed(500 + i)
        ref_loss = tt.train_step(ref, ref_opt, padded(ids[0], 16), padded(ids[1], 48), padded(ids[2], 48), margin=0.5)
        torch.cuda.synchronize()
        assert float(loss) == float(ref_loss)
        assert torch.equal(opt.flat_params, ref_opt.flat_params) and torch.equal(opt.exp_avg_sq, ref_opt.exp_avg_sq)
        assert opt.step_count == ref_opt.step_count == i + 1
        losses.append(float(loss))
    # the same batch twice with different generator states: different masks, different losses
    torch.manual_seed(1)
    a = float(step(*ids))
    torch.manual_seed(2)
    b = float(step(*ids))
    assert a != b


def test_trainer_with_graphs_buckets_the_widths_and_equals_the_eager_trainer():
    """DataParallelTrainer(graphs=True): one HIP graph per (batch, query width, document width) bucket of 32 columns, least
    recently used evicted; every step equals, bit for bit, the eager trainer's step on ids padded to the bucket's widths."""
    import copy
    import twotowermlretrieval_amd as tt
    V, E, H, B = 300, 300, 256, 32
    torch.manual_seed(13)
    m = tt.TwoTowerModel({"VOCAB_SIZE": V, "EMBED_DIM": E, "HIDDEN_DIM": H}, synth.make_table(4, V, E)).cuda()
    ref = copy.deepcopy(m)
    tr = tt.DataParallelTrainer(m, lr=1e-3, margin=0.5, graphs=True, max_graphs=2)
    tr_ref = tt.DataParallelTrainer(ref, lr=1e-3, margin=0.5)

    def padded(t, w):
        out = torch.zeros((t.shape[0], w), dtype=torch.int64, device=t.device)
        out[:, : t.shape[1]] = t
        return out
    seen = []
    for i, widths in enumerate(((7, 20, 25), (9, 40, 33), (5, 30, 64), (7, 70, 20), (6, 31, 12))):
        ids = [torch.from_numpy(synth.make_ids(300 + 3 * i + s, B, T, V)).cuda() for s, T in enumerate(widths)]
        wq, wd = 32, -(-max(widths[1:]) // 32) * 32
        loss = tr.step(*ids)
        ref_loss = tr_ref.step(padded(ids[0], wq), padded(ids[1], wd), padded(ids[2], wd))
        torch.cuda.synchronize()
        assert float(loss) == float(ref_loss) and torch.equal(tr.optimizer.flat_params, tr_ref.optimizer.flat_params), i
        seen.append((B, wq, wd))
        assert list(tr._graphs)[-1] == (B, wq, wd) and len(tr._graphs) <= 2
    assert len(set(seen)) == 3 and tr.optimizer.step_count == 5


def test_weight_gradient_kernel_against_the_tiled_one():
    """wgrad16 (256-row output tiles, K-major LDS images, one K slab per workgroup) and the tiled f16-split GEMM it replaced
    (the comparison build with TT_WGRAD_TILED=1) compute dW_ih / dW_hh from the same dGi / dGh with different slab partitions: they
    agree to the gradient tolerance (both are fp32-grade; the oracle comparisons elsewhere pin each of them), everything else is
    the same bits."""
    from conftest import ab_library
    V, E, H, B, T = 500, 300, 256, 512, 40
    torch.manual_seed(11)
    import twotowermlretrieval_amd as tt
    enc = tt.RNNEncoder(V, E, H, pretrained_embeddings=synth.make_table(4, V, E)).cuda().train()
    ids = torch.from_numpy(synth.make_ids(91, B, T, V)).cuda()
    d_out = torch.from_numpy(np.random.RandomState(5).standard_normal((B, H)).astype(np.float32)).cuda()

    def grads():
        enc.zero_grad()
        y = enc(ids)
        y.backward(d_out)
        torch.cuda.synchronize()
        return {n: p.grad.clone() for n, p in enc.named_parameters() if p.grad is not None}
    res = {"0": grads()}                      # the product library
    with ab_library(TT_WGRAD_TILED=1):
        res["1"] = grads()
    with ab_library():                        # the comparison build with every switch at its default = the product's bits
        same = grads()
    assert all(torch.equal(res["0"][n], same[n]) for n in same)
    differ = 0
    for n in res["0"]:
        a, b = res["0"][n], res["1"][n]
        if "weight" in n:
            assert_grad_close(a.cpu().numpy(), b.cpu().numpy(), what=n)
            differ += int((a != b).sum())
        else:
            assert torch.equal(a, b), n
    assert differ > 0   # (the switch really selects another kernel)
