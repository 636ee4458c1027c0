"""CPU tests of the host logic: tokenizer front end vs the reference's recorded behaviour, shard
arithmetic, and the multi-process (world_size 2, gloo) host logic of the sharded search and of the data-parallel
optimizer -- the package's PRIVATE helpers index._exchange_and_merge and trainer._FlatClipAdam, driven with the
oracle's search / merge / step on CPU tensors (the public classes take no such hooks: they run the HIP kernels and
refuse CPU tensors; nothing in the package imports the oracle)."""
import json
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import synth
from conftest import GOLDEN, ROOT


# ------------------------------------------------------------------ tokenizer (g8)
def test_tokenizer_matches_reference_cases():
    from twotowermlretrieval_amd.tokenizer import PretrainedTokenizer
    doc = json.loads((GOLDEN / "g8_tokenizer.json").read_text())
    tok = PretrainedTokenizer(word2idx=doc["vocab"])
    assert tok.unk_token_id == doc["unk_id"] and tok.vocab_size() == doc["vocab_size"]
    for case in doc["cases"]:
        assert tok.encode(case["text"]) == case["ids"], case["text"]
    assert tok.encode("") == [] and tok.encode(None) == [tok.unk_token_id]  # str(None) -> "none" -> <UNK>
    assert tok.decode([0, 3]) == "the of" and tok.get_word_index("zzz") == -1 and tok.contains_word("the")


def test_tokenizer_batch_padding_is_pad_sequence():
    from twotowermlretrieval_amd.tokenizer import PretrainedTokenizer
    doc = json.loads((GOLDEN / "g8_tokenizer.json").read_text())
    tok = PretrainedTokenizer(word2idx=doc["vocab"])
    texts = ["what is machine learning?", "", "w5 w6"]
    batch = tok.encode_batch(texts)
    want = torch.nn.utils.rnn.pad_sequence([torch.tensor(tok.encode(t), dtype=torch.long) for t in texts],
                                           batch_first=True, padding_value=0)
    assert batch.dtype == torch.int64 and torch.equal(batch, want)


def test_shard_bounds_cover_exactly():
    from twotowermlretrieval_amd import shard_bounds
    for n, w in [(10_000_000, 8), (10, 3), (7, 8), (1, 1)]:
        spans = [shard_bounds(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


# ------------------------------------------------------------------ gloo, world_size 2
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _sharded_worker(rank, world, port, out):
    sys.path[:0] = [str(ROOT), str(GOLDEN)]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as o
    from twotowermlretrieval_amd.index import _exchange_and_merge, shard_bounds

    def local_search(q, docs, k, off):
        v, i = o.score_topk(q.numpy(), docs.numpy(), k, idx_offset=off)
        return torch.from_numpy(v), torch.from_numpy(i)

    def merge(v, i, k):
        mv, mi = o.topk_merge(v.numpy(), i.numpy(), k)
        return torch.from_numpy(mv), torch.from_numpy(mi)

    D = torch.from_numpy(synth.unit_rows(5, 3001, 64).copy())
    D[2500] = D[17]  # an exact tie across the two shards: lower index must win
    Q = torch.from_numpy(synth.unit_rows(6, 9, 64))
    Q[0] = D[17]
    lo, hi = shard_bounds(D.shape[0], rank, world)
    lv, li = local_search(Q, D[lo:hi], 20, lo)          # per-shard top-20 with global indices
    v, i = _exchange_and_merge(lv, li, 10, merge)
    out[rank] = (v.numpy(), i.numpy())
    dist.destroy_process_group()


def test_sharded_index_two_ranks_equals_single(oracle):
    world, port = 2, _free_port()
    mgr = mp.get_context("spawn").Manager()
    out = mgr.dict()
    mp.spawn(_sharded_worker, args=(world, port, out), nprocs=world, join=True)
    D = synth.unit_rows(5, 3001, 64).copy()
    D[2500] = D[17]
    Q = synth.unit_rows(6, 9, 64).copy()
    Q[0] = D[17]
    fv, fi = oracle.score_topk(Q, D, 10)
    for r in range(world):  # identical on every rank and identical to the unsharded search
        assert np.array_equal(out[r][1], fi) and np.array_equal(out[r][0], fv)
    assert list(fi[0, :2]) == [17, 2500]


def _dp_worker(rank, world, port, out):
    sys.path[:0] = [str(ROOT), str(GOLDEN)]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as o
    from twotowermlretrieval_amd.trainer import _FlatClipAdam

    def step_fn(p, g, m, v, step, lr, betas, eps, max_norm, grad_scale, total_norm, scratch):
        gn = g.numpy()
        gn *= np.float32(grad_scale)  # the kernel applies 1/world before the norm
        total_norm[0] = o.clip_adam_step(p.numpy(), gn, m.numpy(), v.numpy(), step, lr, betas, eps, max_norm)

    rs = np.random.RandomState(3)
    shapes = [(12, 5), (7,), (4, 4)]
    params = [torch.nn.Parameter(torch.from_numpy(rs.standard_normal(s).astype(np.float32))) for s in shapes]
    opt = _FlatClipAdam(params, step_fn, lambda g: dist.all_reduce(g, op=dist.ReduceOp.SUM), world, lr=1e-2,
                        betas=(0.9, 0.999), eps=1e-8, max_norm=1.0, scratch_bytes=64)
    norms = []
    for step in range(3):
        opt.zero_grad()
        gr = np.random.RandomState(100 + 10 * step + rank)  # each rank: its own batch shard's gradient
        for prm in params:
            prm.grad.add_(torch.from_numpy(gr.standard_normal(tuple(prm.shape)).astype(np.float32)))
        if step == 1:
            opt.reduce_prefix(60)       # the first parameter's gradients go out early (the query tower's, in the train step);
            assert opt._reduced_upto == 60   # step() then reduces the rest of the bucket, gate words included
        norms.append(float(opt.step()[0]))
        assert opt._reduced_upto == 0
    out[rank] = (np.concatenate([p.detach().numpy().ravel() for p in params]), norms)
    dist.destroy_process_group()


def test_dp_allreduce_then_clip_then_adam_two_ranks(oracle):
    world, port = 2, _free_port()
    mgr = mp.get_context("spawn").Manager()
    out = mgr.dict()
    mp.spawn(_dp_worker, args=(world, port, out), nprocs=world, join=True)
    # single-process restatement: average the two ranks' gradients, THEN clip, then Adam
    rs = np.random.RandomState(3)
    shapes = [(12, 5), (7,), (4, 4)]
    p = np.concatenate([rs.standard_normal(s).astype(np.float32).ravel() for s in shapes])
    m, v = np.zeros_like(p), np.zeros_like(p)
    norms = []
    for step in range(3):
        gs = []
        for rank in range(world):
            gr = np.random.RandomState(100 + 10 * step + rank)
            gs.append(np.concatenate([gr.standard_normal(s).astype(np.float32).ravel() for s in shapes]))
        g = ((gs[0] + gs[1]) * np.float32(0.5)).astype(np.float32)
        norms.append(oracle.clip_adam_step(p, g, m, v, step + 1, 1e-2, max_norm=1.0))
    for r in range(world):
        np.testing.assert_allclose(out[r][0], p, atol=1e-7)
        np.testing.assert_allclose(out[r][1], norms, rtol=1e-6)
    assert np.array_equal(out[0][0], out[1][0])  # replicas stay bit-identical


def _dp_bad_rank_worker(rank, world, port, out):
    sys.path[:0] = [str(ROOT), str(GOLDEN)]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as o
    from twotowermlretrieval_amd.model import SplitRecurrenceTimeout
    from twotowermlretrieval_amd.trainer import _FlatClipAdam

    def step_fn(p, g, m, v, step, lr, betas, eps, max_norm, grad_scale, total_norm, scratch):
        gn = g.numpy()
        gn *= np.float32(grad_scale)
        total_norm[0] = o.clip_adam_step(p.numpy(), gn, m.numpy(), v.numpy(), step, lr, betas, eps, max_norm)

    rs = np.random.RandomState(3)
    params = [torch.nn.Parameter(torch.from_numpy(rs.standard_normal(s).astype(np.float32))) for s in [(12, 5), (7,)]]
    opt = _FlatClipAdam(params, step_fn, lambda g: dist.all_reduce(g, op=dist.ReduceOp.SUM), world, lr=1e-2,
                        betas=(0.9, 0.999), eps=1e-8, max_norm=1.0, scratch_bytes=64)
    log = []
    # per step: the status words this rank's two tower calls produced (bit 0 zero-length row, 1 id out of range, 2 time-out)
    plan = [((0, 0), (0, 0)), ((0, 0), (2, 0)), ((1, 0), (0, 0)), ((0, 4), (0, 0)), ((4, 0), (0, 3)), ((0, 0), (0, 0))]
    for step, per_rank in enumerate(plan):
        opt.zero_grad()
        gr = np.random.RandomState(100 + 10 * step + rank)
        for prm in params:
            prm.grad.add_(torch.from_numpy(gr.standard_normal(tuple(prm.shape)).astype(np.float32)))
        opt._pending_status.extend(torch.tensor([w], dtype=torch.int32) for w in per_rank[rank])
        before = opt.flat_params.clone()
        try:
            opt.step()
            log.append(("ok", opt.step_count))
            assert not torch.equal(before, opt.flat_params)
        except (IndexError, SplitRecurrenceTimeout, RuntimeError) as e:
            log.append((type(e).__name__, opt.step_count))
            assert torch.equal(before, opt.flat_params) and not opt._pending_status
    out[rank] = (opt.flat_params.detach().numpy().copy(), log)
    dist.destroy_process_group()


def test_a_failed_step_is_a_collective_decision_two_ranks(oracle):
    """The status words of a rank's encoder calls ride behind the gradients in the ONE all-reduced bucket
    (trainer._FlatClipAdam): whichever rank had the bad batch, BOTH ranks raise the same exception in the same step (the
    reference's: IndexError for an id out of range, RuntimeError for a zero-length row -- backend/main.py:244-259 is
    single-process and simply stops), neither applies the step, and the replicas stay bit-identical.  Data errors win over
    the transient time-out bit; a time-out alone raises SplitRecurrenceTimeout (train_step redoes the step on it)."""
    world, port = 2, _free_port()
    mgr = mp.get_context("spawn").Manager()
    out = mgr.dict()
    mp.spawn(_dp_bad_rank_worker, args=(world, port, out), nprocs=world, join=True)
    want = [("ok", 1), ("IndexError", 1), ("RuntimeError", 1), ("SplitRecurrenceTimeout", 1), ("IndexError", 1), ("ok", 2)]
    assert out[0][1] == want and out[1][1] == want
    assert np.array_equal(out[0][0], out[1][0])


def test_load_config_reads_json(tmp_path):
    from twotowermlretrieval_amd.query_inferencer import load_config
    (tmp_path / "config.json").write_text('{"HIDDEN_DIM": 256, "RNN_TYPE": "GRU"}')
    assert load_config(str(tmp_path / "config.json")) == {"HIDDEN_DIM": 256, "RNN_TYPE": "GRU"}


def test_seed_plan_of_a_sharded_search():
    """How many seeds a rank lists and which rank of the union is taken (index.seed_plan): k per rank while world * k fits the
    union kernel's 512-value tile, fewer per rank on wider jobs (the k-th of the union still has k distinct documents above
    it), every shard for itself when even that does not fit, when the ranks did not agree to exchange, or beyond the screen's k."""
    from twotowermlretrieval_amd.index import seed_plan
    assert seed_plan(8, 10) == (10, 10) and seed_plan(1, 10) == (10, 10) and seed_plan(8, 64) == (64, 64)
    assert seed_plan(16, 64) == (32, 64)          # 16 x 64 = 1024 > 512: 32 per rank, the 64th of 512
    assert seed_plan(512, 64) == (1, 64) and seed_plan(512, 1) == (1, 1)
    assert seed_plan(1024, 10) is None            # not even one value per rank fits
    assert seed_plan(8, 65) is None and seed_plan(8, 10, exchange=False) is None
    for world in (1, 2, 3, 7, 8, 16, 48, 100, 512):
        for k in (1, 5, 10, 50, 64):
            pl = seed_plan(world, k)
            if pl is not None:
                ks, kth = pl
                assert 1 <= ks <= k and kth == k and world * ks <= 512 and world * ks >= k


def test_committed_pmc_traffic_json_follows_from_the_committed_counter_summaries(tmp_path):
    """bench.py's `roofline.traffic`, `mfma_busy` and `clock_MHz` are read from profiles/pmc_traffic.json; that file must be what
    tools/pmc_traffic_update.py derives from the rocprofv3 summaries its `_source` / `_sq_source` name (evidence chain: counters
    -> summary -> json -> bench line), with the guide's gfx950 correction (FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024)."""
    import json, shutil, subprocess, sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    have = json.loads((root / "profiles" / "pmc_traffic.json").read_text())
    src = root / have["_source"].split(" : ")[0]
    sq = root / have["_sq_source"].split(" : ")[0]
    assert src.exists() and sq.exists(), (src, sq)
    (tmp_path / "profiles").mkdir()
    for f in (src, sq):
        shutil.copy(f, tmp_path / "profiles" / f.name)
    subprocess.run([sys.executable, str(root / "tools" / "pmc_traffic_update.py"), f"profiles/{src.name}", f"profiles/{sq.name}"],
                   cwd=tmp_path, check=True, capture_output=True)
    again = json.loads((tmp_path / "profiles" / "pmc_traffic.json").read_text())
    for k in ("b32", "b1024", "screen_b1024", "stream_b32", "screen_b1024_mfma_busy", "screen_b1024_clock_MHz"):
        assert again[k] == have[k], k
    # 5.12 GB fp16 shadow read about once; the busy share is the exact MFMA cycle count over SIMD-cycles
    assert 1.0 <= have["screen_b1024"] / 5.12e9 < 1.2 and 0.5 < have["screen_b1024_mfma_busy"] < 1.0
