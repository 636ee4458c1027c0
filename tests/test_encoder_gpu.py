"""GPU parity tests for the encoder tower (K1-K3) through the drop-in Python surface
(twotowermlretrieval_amd.model) and the C ABI.  Tolerance: conftest.FWD_ATOL = 2e-6 absolute on the (unit-norm)
outputs (observed <= 4e-7: summation order and the v_exp/v_rcp based sigmoid/tanh; north_star's cosine tolerance is
1e-5), gradients conftest.GRAD_TOL = 2e-5 of the tensor's largest element (observed <= 3e-6).  A build without the `lo`
halves of the fp16 hi/lo split fails both (tools/mutation_guard.py)."""
import json

import numpy as np
import pytest
import torch

import synth
from conftest import FWD_ATOL, GOLDEN, assert_fwd_close, assert_grad_close

pytestmark = pytest.mark.gpu
ATOL = FWD_ATOL


def make_encoder(V, E, H, seed, layers=1, bi=False, normalize=True):
    from twotowermlretrieval_amd.model import RNNEncoder
    table = synth.make_table(seed, V, E)
    sd = synth.make_encoder_state(seed + 1, E, H, layers, bi)
    enc = RNNEncoder(V, E, H, pretrained_embeddings=table, num_layers=layers, bidirectional=bi,
                     normalize_output=normalize)
    full = {"embedding.weight": torch.from_numpy(table)}
    full.update({k: torch.from_numpy(v) for k, v in sd.items()})
    enc.load_state_dict(full)  # exact reference key names, strict
    return enc.cuda().eval(), table, sd


def run(enc, ids):
    with torch.no_grad():
        y = enc(torch.from_numpy(ids).cuda())
    torch.cuda.synchronize()
    return y.cpu().numpy()


def test_g1_small_all_quirks(golden, oracle):
    g = golden("g1_encoder_uni.npz")
    V, E, H, seed = [int(x) for x in g["small_dims"]]
    enc, table, sd = make_encoder(V, E, H, seed)
    y = run(enc, g["small_ids"])
    assert_fwd_close(y, g["small_out"])
    np.testing.assert_allclose(np.linalg.norm(y, axis=1), 1.0, atol=1e-6)
    assert np.array_equal(y[2], y[3])  # interior-zero quirk: tokens beyond count_nonzero are dropped
    o = oracle.encoder_forward(g["small_ids"], table, synth.weight_quads(sd), H)
    assert_fwd_close(y, o)


def test_g1_northstar_shape(golden):
    g = golden("g1_encoder_uni.npz")
    V, E, H, seed = [int(x) for x in g["big_dims"]]
    enc, _, _ = make_encoder(V, E, H, seed)
    assert_fwd_close(run(enc, g["big_ids"]), g["big_out"])


def test_g2_two_layer_bidirectional_projection(golden):
    g = golden("g2_encoder_bi.npz")
    V, E, H, seed = [int(x) for x in g["dims"]]
    enc, _, _ = make_encoder(V, E, H, seed, layers=2, bi=True)
    assert_fwd_close(run(enc, g["ids"]), g["out"])


def test_g3_normalize_off(golden):
    g = golden("g3_encoder_nonorm.npz")
    V, E, H, seed = [int(x) for x in g["dims"]]
    enc, _, _ = make_encoder(V, E, H, seed, normalize=False)
    assert_fwd_close(run(enc, g["ids"]), g["out"])


@pytest.mark.parametrize("B,T,E,H,layers,bi", [
    (70, 40, 300, 256, 1, False),   # north-star tower, ragged batch, > 4 recurrence blocks
    (33, 17, 200, 128, 2, True),    # config.json family (E=200, stacked, bidirectional)
    (5, 250, 300, 256, 1, False),   # long passages
    (130, 9, 52, 64, 3, False),
    (16, 12, 100, 512, 1, True),    # widest supported hidden size
    (41, 30, 200, 256, 2, True),    # config.json's own shape: token-stationary K1 with 13 k-steps, layer 2 on the tiled GEMM
    (67, 21, 300, 128, 1, False),   # 6 column chunks over 4 waves: two waves run a second pass
    (9, 70, 256, 256, 1, False),    # 16 k-steps: no refill-only ring turns
    (2100, 5, 20, 32, 1, True),     # B >= 2048, bidirectional: the projection head as two accumulating GEMMs
])
def test_random_batches_vs_oracle(oracle, B, T, E, H, layers, bi):
    V, seed = 500, 31 + B
    enc, table, sd = make_encoder(V, E, H, seed, layers, bi)
    ids = synth.make_ids(seed + 5, B, T, V, zero_inside=0.05)
    y = run(enc, ids)
    o = oracle.encoder_forward(ids, table, synth.weight_quads(sd, layers, bi), H, layers, bi,
                               sd.get("projection.weight"), sd.get("projection.bias"), True)
    assert_fwd_close(y, o)


def test_state_dict_keys_match_reference_layout():
    from twotowermlretrieval_amd.model import TwoTowerModel
    m = TwoTowerModel({"VOCAB_SIZE": 50, "EMBED_DIM": 20, "HIDDEN_DIM": 32, "NUM_LAYERS": 2, "BIDIRECTIONAL": True},
                      synth.make_table(1, 50, 20))
    keys = set(m.state_dict().keys())
    want = set()
    for tower in ("query_encoder.", "doc_encoder."):
        want.add(tower + "embedding.weight")
        for layer in range(2):
            for sfx in ("", "_reverse"):
                for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                    want.add(f"{tower}rnn.{n}_l{layer}{sfx}")
        want |= {tower + "projection.weight", tower + "projection.bias"}
    assert keys == want
    assert m.query_encoder.embedding.embedding_dim == 20          # read by backend/main.py:104
    assert not m.query_encoder.embedding.weight.requires_grad     # frozen GloVe table (model.py:25-27)
    assert sum(p.numel() for p in m.parameters() if p.requires_grad) == 2 * (
        3 * 32 * (20 + 32 + 2) + 3 * 32 * (20 + 32 + 2) + 2 * 3 * 32 * (64 + 32 + 2) + 32 * 64 + 32)


def test_two_tower_towers_are_independent(oracle):
    from twotowermlretrieval_amd.model import TwoTowerModel
    V, E, H = 80, 24, 64
    table = synth.make_table(9, V, E)
    m = TwoTowerModel({"VOCAB_SIZE": V, "EMBED_DIM": E, "HIDDEN_DIM": H}, table)
    sd = {}
    quads = {}
    for i, tower in enumerate(("query_encoder.", "doc_encoder.")):
        sd[tower + "embedding.weight"] = torch.from_numpy(table)
        st = synth.make_encoder_state(20 + i, E, H, prefix=tower)
        quads[tower] = synth.weight_quads(st, prefix=tower)
        sd.update({k: torch.from_numpy(v) for k, v in st.items()})
    m.load_state_dict(sd)
    m = m.cuda().eval()
    m.device = torch.device("cuda")  # callers set this attribute (backend/main.py:196)
    q = synth.make_ids(1, 6, 5, V)
    d = synth.make_ids(2, 6, 13, V)
    with torch.no_grad():
        eq, ed = m(torch.from_numpy(q).cuda(), torch.from_numpy(d).cuda())
    assert_fwd_close(eq.cpu().numpy(), oracle.encoder_forward(q, table, quads["query_encoder."], H))
    assert_fwd_close(ed.cpu().numpy(), oracle.encoder_forward(d, table, quads["doc_encoder."], H))


def test_error_behaviour_matches_reference():
    err = json.loads((GOLDEN / "g10_errors.json").read_text())
    enc, _, _ = make_encoder(64, 16, 32, 101)
    with pytest.raises(RuntimeError, match="Length of all samples has to be greater than 0"):
        enc(torch.tensor([[3, 4, 0], [0, 0, 0]]).cuda())
    assert err["all_zero_row"].startswith("RuntimeError: Length of all samples")
    with pytest.raises(IndexError):
        enc(torch.tensor([[1, 64, 2]]).cuda())
    with pytest.raises((RuntimeError, ValueError), match="Cannot pack empty tensors"):
        enc(torch.zeros((1, 0), dtype=torch.long).cuda())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        enc(torch.tensor([[1, 2]]))
    enc.check_inputs = False  # async mode: no host sync, bad rows give finite garbage
    y = enc(torch.tensor([[3, 4, 0], [0, 0, 0]]).cuda())
    assert torch.isfinite(y).all()


def test_unknown_tower_type_fails_like_the_reference():
    """The reference does getattr(nn, rnn_type.upper()) (model.py:30): anything but GRU / LSTM / RNN is an AttributeError."""
    from twotowermlretrieval_amd.model import RNNEncoder
    with pytest.raises(AttributeError):
        RNNEncoder(10, 8, 32, rnn_type="transformer")


@pytest.mark.parametrize("cell", ["LSTM", "RNN"])
@pytest.mark.parametrize("tag", ["uni", "bi"])
def test_g13_lstm_and_vanilla_rnn_towers_forward_and_backward(golden, oracle, cell, tag):
    """RNN_TYPE = LSTM / RNN through the HIP kernels: forward output and every parameter gradient against the reference's
    own outputs and autograd (tests/golden/g13_lstm_rnn.npz), state_dict keys and shapes as nn.LSTM / nn.RNN."""
    from twotowermlretrieval_amd.model import RNNEncoder
    g = golden("g13_lstm_rnn.npz")
    key = f"{cell}_{tag}"
    V, E, H, seed, layers, bi, gates = [int(x) for x in g[f"{key}_dims"]]
    table = synth.make_table(seed, V, E)
    sd = synth.make_encoder_state(seed + 1, E, H, layers, bool(bi), gates=gates)
    enc = RNNEncoder(V, E, H, pretrained_embeddings=table, rnn_type=cell.lower(), num_layers=layers, bidirectional=bool(bi))
    full = {"embedding.weight": torch.from_numpy(table)}
    full.update({k: torch.from_numpy(v) for k, v in sd.items()})
    enc.load_state_dict(full)                                  # strict: exact nn.LSTM / nn.RNN key names and shapes
    enc = enc.cuda().train()
    ids = torch.from_numpy(g[f"{key}_ids"]).cuda()
    y = enc(ids)
    assert_fwd_close(y.detach().cpu().numpy(), g[f"{key}_out"])
    (y * torch.from_numpy(g[f"{key}_c"]).cuda()).sum().backward()
    torch.cuda.synchronize()
    for name, prm in enc.named_parameters():
        if not prm.requires_grad:
            continue
        want = g[f"{key}_grad_{name}"]
        assert_grad_close(prm.grad.cpu().numpy(), want, what=name, floor=1e-6)
    with torch.no_grad():                                      # eval path (no stash) gives the same output
        assert_fwd_close(enc.eval()(ids).cpu().numpy(), g[f"{key}_out"])


@pytest.mark.parametrize("cell,B,T,E,H,layers,bi", [("LSTM", 37, 40, 300, 256, 1, False), ("RNN", 50, 33, 200, 128, 2, True),
                                                    ("LSTM", 9, 17, 52, 96, 3, True), ("RNN", 20, 60, 300, 512, 1, False)])
def test_lstm_rnn_towers_vs_oracle_on_larger_shapes(oracle, cell, B, T, E, H, layers, bi):
    from twotowermlretrieval_amd.model import RNNEncoder
    gates = 4 if cell == "LSTM" else 1
    V = 300
    table = synth.make_table(3, V, E)
    sd = synth.make_encoder_state(4, E, H, layers, bi, gates=gates)
    enc = RNNEncoder(V, E, H, pretrained_embeddings=table, rnn_type=cell, num_layers=layers, bidirectional=bi)
    full = {"embedding.weight": torch.from_numpy(table)}
    full.update({k: torch.from_numpy(v) for k, v in sd.items()})
    enc.load_state_dict(full)
    enc = enc.cuda().train()
    ids = synth.make_ids(5, B, T, V, zero_inside=0.05)
    quads = synth.weight_quads(sd, layers, bi)
    pw, pb = sd.get("projection.weight"), sd.get("projection.bias")
    y = enc(torch.from_numpy(ids).cuda())
    want = oracle.encoder_forward(ids, table, quads, H, layers, bi, pw, pb, True, rnn_type=cell)
    assert_fwd_close(y.detach().cpu().numpy(), want)
    d_out = np.random.RandomState(6).standard_normal((B, H)).astype(np.float32)
    y.backward(torch.from_numpy(d_out).cuda())
    og, gpw, gpb = oracle.encoder_backward(ids, table, quads, H, d_out, layers, bi, pw, pb, True, rnn_type=cell)
    flat = [x for quad in og for x in quad] + ([gpw, gpb] if bi else [])
    got = [p.grad.cpu().numpy() for n, p in enc.named_parameters() if p.requires_grad]
    assert len(flat) == len(got)
    for a, b in zip(got, flat):
        assert_grad_close(a, b, floor=1e-6)


def test_encoder_forward_replays_from_a_hip_graph():
    import twotowermlretrieval_amd as tt
    """The forward issues hipMemsetAsync (status block, zero rows) besides its kernels; captured in a HIP graph and
    replayed with new ids in the static input buffer, every replay must equal the eager call bit for bit
    (check_inputs off: the status read is a host synchronisation, which a capture cannot contain)."""
    V, E, H = 500, 52, 64
    torch.manual_seed(0)
    m = tt.TwoTowerModel({"VOCAB_SIZE": V, "EMBED_DIM": E, "HIDDEN_DIM": H, "NUM_LAYERS": 2, "BIDIRECTIONAL": True},
                         synth.make_table(9, V, E)).cuda().eval()
    enc = m.query_encoder
    enc.check_inputs = False
    ids = [torch.from_numpy(synth.make_ids(70 + s, 6, 11, V)).cuda() for s in range(4)]
    static = ids[0].clone()
    with torch.no_grad():
        want = [enc(x).clone() for x in ids]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            enc(static)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = enc(static)
        for rep in range(2):
            for x, w in zip(ids, want):
                static.copy_(x)
                g.replay()
                torch.cuda.synchronize()
                assert torch.equal(out, w), rep


@pytest.mark.parametrize("w_scale,x_scale", [(1.0e-3, 1.0), (4.0, 1.0), (1.0, 1.0e-3), (1.0, 20.0), (5.0e-5, 2.0e3)])
def test_f16_split_scaling_holds_for_extreme_weights_and_embeddings(oracle, w_scale, x_scale):
    """The f16-pipe kernels (K1, K2, K7) scale their operands by powers of two before the fp16 hi/lo split: W_ih / W_hh from
    their largest element, dGh per row and step, dGi from its largest element.  Weights three orders of magnitude smaller or
    larger than the default init, tiny and large embedding vectors: outputs and gradients must still match the fp32
    oracle at the usual tolerances (H = 256: the f16 recurrence; E = 300).  (Weights 30x the default init make the
    recurrence chaotic -- saturated gates flip on last-bit differences of ANY summation order -- and are not a test of
    anything; tiny weights are: the output normalisation divides by a ~1e-2 norm, which is what exposed the absolute
    error of the exp-based tanh and led to its small-argument polynomial.)"""
    from twotowermlretrieval_amd.model import RNNEncoder
    V, E, H, B, T = 200, 300, 256, 21, 37
    table = (synth.make_table(31, V, E) * np.float32(x_scale)).astype(np.float32)
    sd = {k: (v * np.float32(w_scale)).astype(np.float32) for k, v in synth.make_encoder_state(32, E, H).items()}
    enc = RNNEncoder(V, E, H, pretrained_embeddings=table)
    full = {"embedding.weight": torch.from_numpy(table)}
    full.update({k: torch.from_numpy(v) for k, v in sd.items()})
    enc.load_state_dict(full)
    enc = enc.cuda().train()
    ids = synth.make_ids(33, B, T, V, zero_inside=0.05)
    quads = synth.weight_quads(sd)
    y = enc(torch.from_numpy(ids).cuda())
    want = oracle.encoder_forward(ids, table, quads, H)
    assert_fwd_close(y.detach().cpu().numpy(), want)
    d_out = np.random.RandomState(34).standard_normal((B, H)).astype(np.float32)
    y.backward(torch.from_numpy(d_out).cuda())
    og, _, _ = oracle.encoder_backward(ids, table, quads, H, d_out)
    got = [p.grad.cpu().numpy() for n, p in enc.named_parameters() if p.requires_grad]
    for a, b in zip(got, og[0]):
        assert_grad_close(a, b)


@pytest.mark.parametrize("cell", ["LSTM", "RNN"])
def test_lstm_rnn_inter_layer_dropout_and_trainable_table_vs_oracle(oracle, cell):
    """Train-mode inter-layer dropout (the package's counter-based mask) and the trainable embedding table through the
    LSTM / RNN kernels: forward and every gradient, the table's included, against the oracle with the same mask."""
    from twotowermlretrieval_amd.model import RNNEncoder, _EncoderFn
    gates = 4 if cell == "LSTM" else 1
    V, E, H, B, T, layers, bi, p = 90, 24, 64, 11, 14, 2, True, 0.3
    table = synth.make_table(41, V, E)
    sd = synth.make_encoder_state(42, E, H, layers, bi, gates=gates)
    enc = RNNEncoder(V, E, H, pretrained_embeddings=None, rnn_type=cell, num_layers=layers, dropout=p, bidirectional=bi)
    full = {"embedding.weight": torch.from_numpy(table)}
    full.update({k: torch.from_numpy(v) for k, v in sd.items()})
    enc.load_state_dict(full)
    enc = enc.cuda().train()
    ids = synth.make_ids(43, B, T, V, zero_inside=0.1)
    torch.manual_seed(1234)
    seed = int(torch.randint(0, 2 ** 62, (1,)).item())      # what _EncoderFn.forward will draw
    torch.manual_seed(1234)
    y = enc(torch.from_numpy(ids).cuda())
    quads = synth.weight_quads(sd, layers, bi)
    pw, pb = sd["projection.weight"], sd["projection.bias"]
    want = oracle.encoder_forward(ids, table, quads, H, layers, bi, pw, pb, True, p, seed, rnn_type=cell)
    assert_fwd_close(y.detach().cpu().numpy(), want)
    d_out = np.random.RandomState(44).standard_normal((B, H)).astype(np.float32)
    y.backward(torch.from_numpy(d_out).cuda())
    og, gpw, gpb, gt = oracle.encoder_backward(ids, table, quads, H, d_out, layers, bi, pw, pb, True, p, seed,
                                               table_grad=True, rnn_type=cell)
    named = dict(enc.named_parameters())
    assert_grad_close(named["embedding.weight"].grad.cpu().numpy(), gt, what="embedding.weight")
    flat = [x for quad in og for x in quad] + [gpw, gpb]
    got = [prm.grad.cpu().numpy() for n, prm in named.items() if n != "embedding.weight"]
    assert len(flat) == len(got)
    for a, b in zip(got, flat):
        assert_grad_close(a, b)


def test_prepared_weights_are_bit_identical_and_follow_weight_changes(oracle):
    """Inference keeps the weights in kernel form (tt_encoder_prepare_f32 / tt_encoder_forward_prepared_f32): same bits as the
    per-call derivation, and every way the weights can change is noticed (in-place op, load_state_dict, optimizer step)."""
    import twotowermlretrieval_amd as tt
    for layers, bi, E, H in ((1, False, 300, 256), (2, True, 52, 64)):
        V, seed, B, T = 400, 77, 23, 19
        enc, table, sd = make_encoder(V, E, H, seed, layers, bi)
        ids = torch.from_numpy(synth.make_ids(seed + 1, B, T, V, zero_inside=0.05)).cuda()
        with torch.no_grad():
            enc.cache_prepared = False
            plain = enc(ids).clone()
            enc.cache_prepared = True
            first = enc(ids).clone()          # prepares
            blob = enc._prep[ids.device][1]
            again = enc(ids).clone()          # reuses
            assert enc._prep[ids.device][1] is blob
        assert torch.equal(plain, first) and torch.equal(plain, again)
        # 1. in-place change of one weight
        with torch.no_grad():
            enc.rnn.weight_hh_l0.mul_(0.5)
            changed = enc(ids).clone()
            enc.cache_prepared = False
            want = enc(ids).clone()
            enc.cache_prepared = True
        assert enc._prep[ids.device][1] is not blob
        assert torch.equal(changed, want) and not torch.equal(changed, plain)
        # 2. an optimizer step writes the parameters through raw pointers
        enc.train()
        opt = tt.FusedClipAdam(enc.parameters(), lr=1e-2, max_norm=1.0)
        enc(ids).square().sum().backward()
        opt.step()
        enc.eval()
        with torch.no_grad():
            after = enc(ids).clone()
            enc.cache_prepared = False
            want2 = enc(ids).clone()
            enc.cache_prepared = True
        assert torch.equal(after, want2) and not torch.equal(after, changed)
        # 3. writes the version counters do NOT see: through `.data`, and through the optimizer's flat buffer
        with torch.no_grad():
            enc.rnn.weight_ih_l0.data.mul_(0.5)
            enc.invalidate_prepared()          # (documented: `.data` writes need it)
            after_data = enc(ids).clone()
            opt.flat_params.mul_(1.25)
            opt.mark_params_changed()
            after_flat = enc(ids).clone()
            enc.cache_prepared = False
            want3 = enc(ids).clone()
            enc.cache_prepared = True
        assert not torch.equal(after_data, after) and torch.equal(after_flat, want3) and not torch.equal(after_flat, after_data)
        after = after_flat
        # 4. against the oracle with the final weights
        sd2 = {k: v.detach().cpu().numpy() for k, v in enc.state_dict().items() if k != "embedding.weight"}
        o = oracle.encoder_forward(ids.cpu().numpy(), table, synth.weight_quads(sd2, layers, bi), H, layers, bi,
                                   sd2.get("projection.weight"), sd2.get("projection.bias"), True)
        assert_fwd_close(after.cpu().numpy(), o)


def test_prepared_weights_inside_a_hip_graph_and_deepcopy():
    import copy
    V, E, H = 300, 300, 256
    enc, _, _ = make_encoder(V, E, H, 5)
    ids = torch.from_numpy(synth.make_ids(6, 8, 12, V)).cuda()
    with torch.no_grad():
        want = enc(ids).clone()               # cache filled outside the capture
        enc.check_inputs = False
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            enc(ids)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = enc(ids)
        g.replay(); g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, want)
        twin = copy.deepcopy(enc)             # the cache (and the module-level lock) must not break copying
        assert torch.equal(twin(ids), want)


@pytest.mark.parametrize("B,T", [(8, 12), (40, 300)])  # the one-workgroup prep (<= 8192 ids) and the four-kernel prep
def test_status_word_is_written_not_accumulated(B, T):
    """include/tt.h: `status` is WRITTEN on the stream.  The Python host hands the call an uninitialised word, so a stale
    value must never survive: poison it, run clean ids (-> 0), a zero-length row (-> exactly 1), an id >= V (-> exactly 2)."""
    import ctypes as C
    import twotowermlretrieval_amd as tt
    from twotowermlretrieval_amd import _lib
    from twotowermlretrieval_amd.model import _ptr_array, _stream
    V, E, H = 50, 16, 32
    torch.manual_seed(3)
    enc = tt.RNNEncoder(V, E, H).cuda().eval()
    L = _lib.lib()
    rs = np.random.RandomState(B)
    base = rs.randint(1, V, size=(B, T)).astype(np.int64)
    quads = [w.detach().contiguous() for quad in enc.rnn.quads() for w in quad]
    wptr = _ptr_array(quads)
    table = enc.embedding.weight.detach()
    need = L.tt_encoder_workspace_bytes(B, T, E, H, 1, 0, enc._cell, 0, 0)
    ws = torch.empty(max(need, 256), dtype=torch.uint8, device="cuda")
    out = torch.empty((B, H), dtype=torch.float32, device="cuda")

    def run(ids_np):
        ids = torch.from_numpy(ids_np).cuda()
        status = torch.full((1,), 0x7FFFFFFF, dtype=torch.int32, device="cuda")
        with torch.cuda.device(0):
            _lib.check(L.tt_encoder_forward_f32(ids.data_ptr(), B, T, table.data_ptr(), V, E, H, 1, 0, enc._cell, wptr, None,
                                                None, 1, 0, 0.0, 0, out.data_ptr(), ws.data_ptr(), ws.numel(),
                                                status.data_ptr(), None, _stream(ids.device)))
        torch.cuda.synchronize()
        return int(status.item())

    assert run(base) == 0
    empty_row = base.copy()
    empty_row[B // 2] = 0
    assert run(empty_row) == 1
    bad_id = base.copy()
    bad_id[1, T - 1] = V
    assert run(bad_id) == 2
    # and through the module: the reference's exceptions, from a word nobody zeroed
    with pytest.raises(RuntimeError):
        enc(torch.from_numpy(empty_row).cuda())
    with pytest.raises(IndexError):
        enc(torch.from_numpy(bad_id).cuda())
    assert torch.isfinite(enc(torch.from_numpy(base).cuda())).all()


def _with_split(mod, flag, fn, bwd=None):
    """Run fn with the recurrences of the encoders inside `mod` column-split (gru16x4.hip; True), on the one-workgroup kernels
    (False: TT_ENC_ONE_WORKGROUP, a per-call option of the product library), or -- "4" -- on the four-wave members the
    eight-wave ones replaced, which only the comparison build contains (conftest.ab_library).  bwd: the reverse-time recurrence
    alone (default: as flag)."""
    import contextlib
    from conftest import ab_library
    import twotowermlretrieval_amd as tt
    bwd = flag if bwd is None else bwd
    encs = [e for e in mod.modules() if isinstance(e, tt.RNNEncoder)]
    old = [(e.one_workgroup, e.one_workgroup_bwd) for e in encs]
    for e in encs:
        e.one_workgroup, e.one_workgroup_bwd = (flag is False), (bwd is False)
    four = {k: 4 for k, f in (("TT_GRU_SPLIT", flag), ("TT_GRU_SPLIT_BWD", bwd)) if f == "4"}
    try:
        with (ab_library(**four) if four else contextlib.nullcontext()):
            return fn()
    finally:
        for e, (o, ob) in zip(encs, old):
            e.one_workgroup, e.one_workgroup_bwd = o, ob


@pytest.mark.parametrize("B,T,layers,bi", [(70, 40, 1, False), (5, 250, 1, False), (1024, 33, 1, False), (1, 7, 1, False),
                                          (300, 21, 2, True), (512, 12, 1, True), (17, 9, 3, False),
                                          (1024, 20, 1, True), (700, 15, 2, True)])   # (the last two: one launch per direction)
def test_column_split_recurrence_vs_the_one_cu_kernels(B, T, layers, bi):
    """gru_seq16x4p_kernel (a row group's gate columns on four CUs, hidden state handed over through tagged granules every
    step) against gru_seq16_kernel (TT_ENC_ONE_WORKGROUP): same products in the same order -> the SAME BITS, in eval mode and in
    train mode, stash included: with the split forward and the one-CU backward every gradient is bit-identical to the all-one-CU
    run.  gru_bwd16x4p_kernel splits the REDUCTION (four partial chains per column, summed in member order): its gradients agree
    with the one-CU backward to the tests' gradient tolerance and are the same bits on every run.  Ragged lengths, one-row
    batches, 64 teams (every CU taken), both directions in one launch, stacked layers; bidirectional batches whose two directions
    do not fit the device together run one launch per direction (gru16x4_launches == 2: the reference's default model shape at
    512 triplets)."""
    V, E, H, seed = 400, 300, 256, 900 + B
    enc, table, sd = make_encoder(V, E, H, seed, layers, bi)
    ids = torch.from_numpy(synth.make_ids(seed + 5, B, T, V, zero_inside=0.05)).cuda()
    enc.cache_prepared = False
    with torch.no_grad():
        one = _with_split(enc, False, lambda: enc(ids).clone())
        four = _with_split(enc, True, lambda: enc(ids).clone())
        enc.cache_prepared = True      # (the prepared-weights entry point takes the option too)
        one_p = _with_split(enc, False, lambda: enc(ids).clone())
        enc.cache_prepared = False
    torch.cuda.synchronize()
    assert torch.equal(one, four) and torch.equal(one, one_p)
    assert torch.isfinite(four).all() and float(four.norm(dim=1).min()) > 0.99
    enc.train()
    d_out = torch.from_numpy(np.random.RandomState(seed).standard_normal((B, H)).astype(np.float32)).cuda()

    def grads(flag, bwd):
        enc.zero_grad()
        y = _with_split(enc, flag, lambda: enc(ids), bwd)
        _with_split(enc, flag, lambda: y.backward(d_out), bwd)
        torch.cuda.synchronize()
        return [p.grad.clone() for p in enc._flat_params()], y.detach().clone()
    g1, y1 = grads(False, False)
    g4f, y4f = grads(True, False)      # split forward, one-CU backward: the stash is the same bits
    assert torch.equal(y1, y4f) and all(torch.equal(a, b) for a, b in zip(g1, g4f))
    g4, y4 = grads(True, True)         # both split
    g4b, _ = grads(True, True)
    assert torch.equal(y1, y4) and all(torch.equal(a, b) for a, b in zip(g4, g4b))
    for a, b in zip(g4, g1):
        assert_grad_close(a.cpu().numpy(), b.cpu().numpy())


@pytest.mark.parametrize("B,T,layers,bi", [(70, 40, 1, False), (1, 7, 1, False), (33, 250, 1, False), (2000, 25, 1, False),
                                          (300, 21, 2, True), (17, 9, 3, False)])
def test_two_row_tiles_per_workgroup_vs_one(B, T, layers, bi):
    """gru_seq16_kernel<256, 2> (big batches: a workgroup takes 32 rows, every streamed W_hh fragment multiplies both tiles' h
    fragments) against the one-tile form: a row's products and their order are the same -> the SAME BITS, eval and train, stash
    included (the backward pass of both runs is the same kernel: every gradient the same bits).  The product library takes two
    tiles from two rounds of one-tile workgroups up (B >= 8 192 on 256 CUs; four rounds until round 5: tests/test_bench_size_gpu.py runs that size
    against the oracle); here the comparison build forces either form (TT_GRU16_RT = 3 / 1) on small shapes: a half-empty
    second tile, a one-row batch, both directions in one launch, stacked layers, ragged lengths."""
    from conftest import ab_library
    V, E, H, seed = 400, 300, 256, 2100 + B
    enc, table, sd = make_encoder(V, E, H, seed, layers, bi)
    ids = torch.from_numpy(synth.make_ids(seed + 5, B, T, V, zero_inside=0.05)).cuda()
    enc.cache_prepared = False
    enc.one_workgroup = enc.one_workgroup_bwd = True      # (below 1 025 rows the column-split kernels would take the call)

    def run(mode, train):
        enc.train(train)
        with ab_library(TT_GRU16_RT=mode):
            if not train:
                with torch.no_grad():
                    y = enc(ids).clone()
                torch.cuda.synchronize()
                return y, []
            enc.zero_grad()
            y = enc(ids)
            y.backward(d_out)
            torch.cuda.synchronize()
            return y.detach().clone(), [p.grad.clone() for p in enc._flat_params()]
    d_out = torch.from_numpy(np.random.RandomState(seed).standard_normal((B, H)).astype(np.float32)).cuda()
    y1, _ = run(1, False)
    y2, _ = run(3, False)
    assert torch.equal(y1, y2) and torch.isfinite(y2).all() and float(y2.norm(dim=1).min()) > 0.99
    with torch.no_grad():
        enc.eval()
        assert torch.equal(enc(ids), y1)                  # the product library (one tile at these sizes)
    t1, g1 = run(1, True)
    t2, g2 = run(3, True)
    assert torch.equal(t1, t2) and len(g1) == len(g2) > 0 and all(torch.equal(a, b) for a, b in zip(g1, g2))


@pytest.mark.parametrize("B,T,layers,bi", [(70, 40, 1, False), (1024, 33, 1, False), (300, 21, 2, True), (1, 7, 1, False)])
def test_eight_wave_members_vs_four_wave_members(B, T, layers, bi):
    """The split recurrences run a member as EIGHT waves (two per SIMD; gru_seq16x4p / gru_bwd16x4p_kernel) -- the four-wave
    members they replaced are compiled into the comparison build only (-DTT_AB, TT_GRU_SPLIT=4 / TT_GRU_SPLIT_BWD=4) and stay the
    reference here.  Forward: every column's accumulator chain is handed from one wave of a pair to the other, same products in
    the same order: outputs and stash are the same bits.  Backward: the pair splits the destination members, every partial is
    still one chain: dW_ih / dW_hh are the same bits; the bias sums accumulate per half (another order): gradient tolerance."""
    V, E, H, seed = 400, 300, 256, 1300 + B
    enc, table, sd = make_encoder(V, E, H, seed, layers, bi)
    ids = torch.from_numpy(synth.make_ids(seed + 5, B, T, V, zero_inside=0.05)).cuda()
    enc.cache_prepared = False
    enc.train()
    d_out = torch.from_numpy(np.random.RandomState(seed).standard_normal((B, H)).astype(np.float32)).cuda()

    def grads(fwd, bwd):
        enc.zero_grad()
        y = _with_split(enc, fwd, lambda: enc(ids), bwd)
        _with_split(enc, fwd, lambda: y.backward(d_out), bwd)
        torch.cuda.synchronize()
        return {n: p.grad.clone() for n, p in enc.named_parameters() if p.grad is not None}, y.detach().clone()
    g44, y44 = grads("4", "4")
    g84, y84 = grads(True, "4")    # eight-wave forward (product library), four-wave backward: the stash is the same bits
    assert torch.equal(y44, y84) and all(torch.equal(g44[n], g84[n]) for n in g44)
    g88, y88 = grads(True, True)
    assert torch.equal(y44, y88)
    for n in g44:
        if "weight" in n:
            assert torch.equal(g44[n], g88[n]), n
        else:
            assert_grad_close(g88[n].cpu().numpy(), g44[n].cpu().numpy(), what=n)


@pytest.mark.parametrize("B,T", [(1, 5), (33, 12), (512, 70), (700, 64)])
def test_k1_tail_split_writes_the_same_bits(B, T):
    """gemm_rows16's tail split (the token blocks beyond the last full round of resident workgroups are each given to three
    workgroups that take a third of every wave's passes) against the unsplit launch (the comparison build with TT_ROWS_SPLIT=0):
    same arithmetic per column, so the tower's output and the training stash must be the same bits -- for a serving-size batch
    (one round, every block split), a batch just over one round (512 x 70: ~560 token blocks on 512 slots) and one whose tail is
    too large to split."""
    import contextlib
    from conftest import ab_library
    V, E, H, seed = 400, 300, 256, 1700 + B
    enc, table, sd = make_encoder(V, E, H, seed, 1, False)
    ids = torch.from_numpy(synth.make_ids(seed + 5, B, T, V, zero_inside=0.05)).cuda()
    enc.cache_prepared = False
    outs = {}
    for flag in ("1", "0"):
        with (contextlib.nullcontext() if flag == "1" else ab_library(TT_ROWS_SPLIT=0)):   # "1": the product library
            with torch.no_grad():
                y = enc(ids).clone()
            enc.train()
            enc.zero_grad()
            yt = enc(ids)
            yt.backward(torch.ones_like(yt))
            torch.cuda.synchronize()
            outs[flag] = (y, yt.detach().clone(), [p.grad.clone() for p in enc._flat_params()])
            enc.eval()
    a, b = outs["1"], outs["0"]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and all(torch.equal(x, y) for x, y in zip(a[2], b[2]))


def test_train_step_keeps_the_split_recurrences_within_the_cus():
    """The train step launches the query tower and the 2B-row document tower on two streams.  Both column-split would be 32 + 64
    teams = 384 one-per-CU workgroups for 256 CUs: members of some teams would wait for a CU while their partners already sweep
    for them, and progress would rest on dispatch order.  trainer._towers_in_flight makes co-residency a matter of construction:
    the direct step ORDERS the two towers' forward recurrence launches with an event inside the calls (tt_enc_sync_t: the query
    tower's first; the document tower's call goes out in two halves around the query tower's, TT_ENC_PHASE_BEGIN / _FINISH, so
    that its projection is not delayed) and gives the smaller tower the one-workgroup BACKWARD recurrence; the autograd path and
    graph capture give the smaller tower the one-workgroup recurrences in both directions.  Checked here on what the tower calls
    really got.  Three steps: with the split forward alone
    the parameters are the all-one-workgroup run's bit for bit; with the backwards split too the run repeats itself bit for bit
    and its last gradient agrees with the one-workgroup run's to the gradient tolerance; direct and autograd paths agree."""
    import copy
    import twotowermlretrieval_amd as tt
    from twotowermlretrieval_amd import _lib
    V, E, H, B = 500, 300, 256, 512
    torch.manual_seed(5)
    m0 = tt.TwoTowerModel({"VOCAB_SIZE": V, "EMBED_DIM": E, "HIDDEN_DIM": H}, synth.make_table(4, V, E)).cuda().train()
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    runs = {}
    cases = {"one": (False, False, True, True), "fwd": (True, False, True, True), "both": (True, True, True, True),
             "both2": (True, True, True, True), "both_checked": (True, True, True, False), "both_autograd": (True, True, False, False)}
    for name, (flag, bwd, direct, defer) in cases.items():
        m = copy.deepcopy(m0)
        o = tt.FusedClipAdam(m.parameters(), lr=1e-5, max_norm=1.0)
        seen, begun = [], []
        for enc in (m.query_encoder, m.doc_encoder):
            fwd0 = enc._run_forward

            def spy(x, train, *a, _f=fwd0, _e=enc, **k):
                sy = k.get("sync")
                if k.get("phase", 0) == _lib.TT_ENC_PHASE_BEGIN:
                    begun.append(_e)          # (the first half of a two-phase call: prep + projection, no recurrence yet)
                else:
                    seen.append((_e, x.shape[0], _e._opts(), _e._opts_bwd(), None if sy is None else
                                 (bool(sy.wait_before_recurrence), bool(sy.record_after_recurrence))))
                return _f(x, train, *a, **k)
            enc._run_forward = spy
        losses = []
        for step in range(3):
            ids = [torch.from_numpy(synth.make_ids(60 + 3 * step + s, B, T, V)).cuda() for s, T in enumerate((9, 60, 70))]
            losses.append(float(_with_split(m, flag, lambda: tt.train_step(m, o, *ids, margin=0.5, direct=direct, defer_check=defer),
                                            bwd).item()))
            torch.cuda.synchronize()
        assert o.settle() is None
        runs[name] = (losses, o.flat_params.clone(), o.flat_grads.clone())
        # what was in flight together never asked for more CUs than the device has -- or its recurrences were ordered
        for i in range(0, len(seen), 2):
            pair = seen[i:i + 2]
            want = sum(_lib.lib().tt_encoder_split_workgroups(b, H, 0, 0) for e, b, of, ob, sy in pair if not (of and ob))
            ordered = sorted(sy for *_, sy in pair if sy is not None) == [(False, True), (True, False)]   # one records, one waits
            assert want <= cus or ordered, (name, pair)
            if ordered:   # the recording call (the query tower) was issued first
                assert pair[0][0] is m.query_encoder and pair[0][4] == (False, True), (name, pair)
        if cus < 384 and name in ("both", "both_checked"):
            assert all(sy is not None and not of for *_, of, ob, sy in seen), seen          # both forwards split, ordered by events
            assert all(ob for e, _, of, ob, sy in seen if e is m.query_encoder), seen       # the smaller tower's backward: one workgroup
            assert begun == [m.doc_encoder] * 3                                             # the document tower's call in two halves
        if cus < 384 and name == "both_autograd":
            q_calls = [t for t in seen if t[0] is m.query_encoder]
            assert all(of and ob and sy is None for *_, of, ob, sy in q_calls), seen         # the smaller tower: one workgroup
    assert runs["one"][0] == runs["fwd"][0] and torch.equal(runs["one"][1], runs["fwd"][1])
    assert runs["both"][0] == runs["both2"][0] and torch.equal(runs["both"][1], runs["both2"][1])
    np.testing.assert_allclose(runs["both"][0], runs["one"][0], atol=2e-6)
    assert_grad_close(runs["both"][2].cpu().numpy(), runs["one"][2].cpu().numpy())
    for other in ("both_checked", "both_autograd"):   # the plans differ in kernels, not in numbers beyond the gradient tolerance
        np.testing.assert_allclose(runs[other][0], runs["both"][0], atol=2e-6)
        assert_grad_close(runs[other][2].cpu().numpy(), runs["both"][2].cpu().numpy())
