#!/usr/bin/env python3
"""Headline benchmark: queries/sec of exact top-10 over a 10M x 256-d fp32 corpus (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W        (N > 1 without a launcher: starts N rank processes itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the hot path over one batch of B=1024 synthetic queries: every rank scores
the batch against its contiguous row shard of the SAME 10M-document corpus, resident in HBM, with
the screened exact search (fp16-MFMA filter over an fp16 shadow copy with a rigorous error bound,
exact fp32 rescoring of the survivors: bit-identical to the plain fp32 kernel and to the CPU
oracle), and for N > 1 the per-shard top-50 lists are all-gathered over RCCL and merged to the
global top-10 on every rank (strong scaling: the corpus is fixed, the shard shrinks with N).
Rank 0 prints ONE JSON line.

Legs EVERY rank executes after the timed loop (collectives; reported under `roofline.legs`): `streamed_bf16` (BASELINE configs[4]:
each GPU's 12.5M-row bf16 shard streamed from pinned host DRAM through a ShardedIndex, GB/s per GPU against PCIe, resident variant
beside it) and `dp_train` (BASELINE configs[2]: data-parallel triplet training, 512 triplets per GPU, all-reduce share).
Extra legs on rank 0 (untimed w.r.t. `value`): `roofline` (the step's dominant kernel,
screen_kernel<false>, timed alone with HIP events recorded around its launch) and, nested under `roofline.legs` so that
the driver's record carries them: `mfma_exact_f32` (the plain fp32-MFMA kernel on the same batch), `hbm_screen` (a
serving-size batch, B=32: the streaming form of the screen, bound by HBM streaming of the fp16 shadow corpus -- the
regime north_star's ">= 70 % of HBM" is graded in), `hbm_exact_f32` (the fp32 kernel at B=32), `encoder` / `train`
(SURVEY 8d's secondary metrics: tower tokens/s, index-build tokens/s, training triplets/s on synthetic
MS-MARCO-shaped batches, each with its fp32-MFMA fraction), `encoder_corpus` (the same search over a corpus of
document-tower outputs with query-tower outputs as queries: queries/s, what the filter let through, fallback count);
and at N=1 `cpu_baseline` (the reference's torch CPU idioms: `value` = the bench batch on a 1M-document sample scaled
x10; `cpu_baseline.legs` holds BASELINE.md section 2's rows run at their own sizes, incl. B=64 over all 10M documents,
the 512-passage doc-tower forward and the 512-triplet train step, and B=1 over all 10M documents).  `serve_b1` (N = 1): one query
string end to end through QueryInferencer + HybridSearcher over the resident corpus, p50 / p99.
"""
from __future__ import annotations

import argparse
import gc
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

N_DOCS = 10_000_000
DIM = 256
BATCH = 1024
TOPK = 10
SHARD_K = 50
GEN_BLOCK = 1_000_000  # corpus is generated in seeded 1M-row blocks -> identical for every world size
HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E peak
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32 MFMA peak
MFMA_F16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16/f16 MFMA peak


def gen_rows(lo: int, hi: int, device) -> torch.Tensor:
    """Rows [lo,hi) of the synthetic corpus: randn, L2-normalised, seed = 1000 + block."""
    out = torch.empty((hi - lo, DIM), dtype=torch.float32, device=device)
    b0, b1 = lo // GEN_BLOCK, (hi - 1) // GEN_BLOCK
    for blk in range(b0, b1 + 1):
        g = torch.Generator(device=device).manual_seed(1000 + blk)
        x = torch.randn((GEN_BLOCK, DIM), dtype=torch.float32, device=device, generator=g)
        x /= x.norm(dim=1, keepdim=True).clamp_min(1e-12)
        s, e = max(lo, blk * GEN_BLOCK), min(hi, (blk + 1) * GEN_BLOCK)
        out[s - lo:e - lo] = x[s - blk * GEN_BLOCK:e - blk * GEN_BLOCK]
        del x
    return out


def gen_queries(n: int, device, seed: int = 7) -> torch.Tensor:
    g = torch.Generator(device=device).manual_seed(seed)
    q = torch.randn((n, DIM), dtype=torch.float32, device=device, generator=g)
    return q / q.norm(dim=1, keepdim=True)


def _event_pair(L):
    from twotowermlretrieval_amd import _lib
    evs = (C.c_void_p * 2)()
    for i in range(2):
        e = C.c_void_p()
        _lib.check(L.tt_event_create(C.byref(e)))
        evs[i] = e.value
    return evs


def _pair_ms(L, evs):
    """Elapsed time between the two events of a pair (synchronises on the second), then frees them."""
    from twotowermlretrieval_amd import _lib
    ms = C.c_float()
    _lib.check(L.tt_event_elapsed_ms(evs[0], evs[1], C.byref(ms)))
    for i in range(2):
        L.tt_event_destroy(evs[i])
    return ms.value


def kernel_only_ms(q, docs, k, iters=5, warm=2):
    """(main_ms, bracket_ms): average duration of the streaming score+top-k launch alone -- HIP events
    recorded on the launch stream right before/after that launch (prof_events) -- and of the whole
    partials call (sample pass + threshold select + main pass) bracketed on the same stream."""
    from twotowermlretrieval_amd import _lib
    L = _lib.lib()
    B, d = q.shape
    N = docs.shape[0]
    ws = torch.empty(L.tt_score_topk_workspace_bytes(B, N, d, k), dtype=torch.uint8, device=q.device)
    pv, pi, pm = C.c_void_p(), C.c_void_p(), C.c_int()
    st = torch.cuda.current_stream().cuda_stream
    pairs = [_event_pair(L) for _ in range(iters)]

    def call(prof=None):
        _lib.check(L.tt_score_topk_partials_f32(q.data_ptr(), B, d, docs.data_ptr(), N, k, 0, ws.data_ptr(),
                                                ws.numel(), C.byref(pv), C.byref(pi), C.byref(pm), prof, st))
    for _ in range(warm):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for evs in pairs:  # back to back, no host synchronisation in between
        call(evs)
    e1.record()
    torch.cuda.synchronize()
    main = sum(_pair_ms(L, evs) for evs in pairs)
    off = L.tt_score_topk_pace_timeouts_offset(B, N, d, k)
    kernel_only_ms.pace_timeouts = int(ws[off:off + 4].view(torch.int32).item()) if off != C.c_size_t(-1).value else None
    return main / iters, e0.elapsed_time(e1) / iters


def screen_kernel_ms(index, q, k, iters=20, warm=5, k_seed=0):
    """Average duration of screen_kernel<false> alone: HIP events recorded on the launch stream right
    before and after that launch inside tt_score_topk_screened_f32 (prof_events).  k_seed: the sharded step's form of the
    search (list length k, thresholds seeded for the final k_seed: ShardedIndex._local_search)."""
    from twotowermlretrieval_amd import _lib
    L = _lib.lib()
    pairs = [_event_pair(L) for _ in range(iters)]
    from twotowermlretrieval_amd.index import _local_seed
    kw = dict(_seed_union=_local_seed, _k_seed=k_seed) if 0 < k_seed < k else {}
    for _ in range(warm):
        index.search(q, k, **kw)
    torch.cuda.synchronize()
    for evs in pairs:  # back to back, no host synchronisation in between
        index.search(q, k, _prof_events=evs, **kw)
    torch.cuda.synchronize()
    return sum(_pair_ms(L, evs) for evs in pairs) / iters


def time_search(index, q, k, iters=10, warm=2):
    """Average duration of a whole index.search call (all its kernels), back to back on the current stream."""
    for _ in range(warm):
        index.search(q, k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        index.search(q, k)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def pmc_traffic(name: str):
    """HBM bytes per launch from committed rocprofv3 PMC passes (profiles/pmc_traffic.json), or None."""
    f = ROOT / "profiles" / "pmc_traffic.json"
    if f.exists():
        try:
            return json.loads(f.read_text()).get(name)
        except Exception:  # noqa: BLE001
            return None
    return None


def cpu_baseline(q_gpu, docs_gpu, enc_inputs=None):
    """The reference's torch CPU idioms (oracle/torch_ref.py) on this box's host cores: BASELINE.md section 2's rows at
    their own sizes.  `value` is the bench batch (B=1024) on the first 1M documents scaled x10 -- the full 10M x 1024 score
    matrix would be 41 GB, which BASELINE.md excludes; every other row is RUN at the size it names."""
    from oracle import torch_ref
    n_s = 1_000_000
    cores = min(len(os.sched_getaffinity(0)), 64)
    torch.set_num_threads(cores)
    qs = q_gpu.cpu()
    dfull = docs_gpu.cpu()                      # 10.24 GB of host memory
    ds = dfull[:n_s]
    t = torch_ref.time_scoring_idiom(qs, ds, TOPK, warmup=1, reps=5)
    scale = N_DOCS / n_s
    out = {"value": round(BATCH / (t * scale), 2), "unit": "queries/s", "cores": cores, "kind": "port",
           "sample": f"B={BATCH} queries x first {n_s} of the {N_DOCS} docs, torch.matmul+torch.topk on CPU "
                     f"(reference idiom evaluators.py:185-186), median of 5 = {t:.3f} s, time scaled x{scale:.0f} "
                     f"(the full pass would materialise a 41 GB score matrix: BASELINE.md section 2 skips it)"}
    legs = {"scoring_b1024_n1m": {"value": round(BATCH / t, 1), "unit": "queries/s",
                                  "sample": f"B={BATCH} x N={n_s}, as run (no scaling), median of 5 = {t:.3f} s"}}
    q64 = qs[:64].contiguous()
    t64 = torch_ref.time_scoring_idiom(q64, ds, TOPK, warmup=1, reps=5)
    legs["scoring_b64_n1m"] = {"value": round(64 / t64, 1), "unit": "queries/s",
                               "sample": f"B=64 x N={n_s}, as run, median of 5 = {t64:.3f} s"}
    t64f = torch_ref.time_scoring_idiom(q64, dfull, TOPK, warmup=1, reps=3)
    legs["scoring_b64_n10m"] = {"value": round(64 / t64f, 2), "unit": "queries/s",
                                "sample": f"B=64 x N={dfull.shape[0]} (2.6 GB score matrix), as run, median of 3 = {t64f:.3f} s"}
    q1 = qs[:1].contiguous()
    t1 = torch_ref.time_scoring_idiom(q1, dfull, TOPK, warmup=1, reps=5)
    legs["scoring_b1_n10m"] = {"value": round(1 / t1, 2), "unit": "queries/s",
                               "sample": f"B=1 x N={dfull.shape[0]} (the reference's serving batch, frontend/main.py:152), as run, "
                                         f"median of 5 = {t1 * 1e3:.1f} ms per query"}
    del dfull, ds
    if enc_inputs is not None:
        table, q_ids, p_ids, n_ids = enc_inputs
        nb = q_ids.shape[0]  # the whole 512-row batches of the GPU legs
        qt, dt = torch_ref.TorchTower(table, ENC_H, seed=0), torch_ref.TorchTower(table, ENC_H, seed=1)
        qc, pc, nc = q_ids.cpu(), p_ids.cpu(), n_ids.cpu()
        tf = torch_ref.time_tower_forward(dt, pc, warmup=1, reps=5)
        ptok = int((pc != 0).sum())
        legs["doc_tower_forward"] = {"value": round(ptok / tf), "unit": "tokens/s",
                                     "sample": f"{nb} passages ({ptok} tokens), nn.Embedding + nn.GRU + F.normalize "
                                               f"(model.py:48-75), median of 5 = {tf:.3f} s"}
        tt_ = torch_ref.time_train_step(qt, dt, qc, pc, nc, margin=0.5, lr=5e-5, warmup=1, reps=3)
        legs["train_step"] = {"value": round(nb / tt_, 1), "unit": "triplets/s",
                              "sample": f"{nb} triplets per step (main.py:244-259: 3 forwards, loss, backward, "
                                        f"clip_grad_norm_, Adam), median of 3 = {tt_:.3f} s"}
    out["legs"] = legs
    return out


# ---- secondary metrics: synthetic MS-MARCO-shaped token batches, north-star model shape ----
ENC_V, ENC_E, ENC_H = 400_003, 300, 256
FLOP_PER_TOKEN_FWD = 2.0 * 3 * ENC_H * (ENC_E + ENC_H)             # input projection + recurrence
FLOP_PER_TOKEN_REC = 2.0 * 3 * ENC_H * ENC_H                       # the recurrence alone (inference with the projected table)
FLOP_PER_TOKEN_TRAIN = FLOP_PER_TOKEN_FWD + 2.0 * 3 * ENC_H * (ENC_E + 2 * ENC_H)  # + dW_ih, dW_hh, dh (table frozen)


def make_ids(rs, B, mean, lo, hi, V):
    """ids ~ Zipf(1.07) over [0,V) (id 0, "the", also occurs inside sentences), lengths ~ clip(Poisson(mean), lo, hi),
    first token non-zero, right-padded with 0 to the batch maximum."""
    import numpy as np
    L = np.clip(rs.poisson(mean, B), lo, hi)
    T = int(L.max())
    ids = np.zeros((B, T), dtype=np.int64)
    for b in range(B):
        z = rs.zipf(1.07, L[b]) % V
        z[0] = max(z[0], 1)
        ids[b, :L[b]] = z
    return torch.from_numpy(ids), int((ids != 0).sum())


def make_ids_bulk(rs, B, mean, lo, hi, V):
    """make_ids's distribution drawn in one vectorised call per batch (millions of passages: the per-row loop is ~25 us a row)."""
    import numpy as np
    L = np.clip(rs.poisson(mean, B), lo, hi).astype(np.int64)
    T = int(L.max())
    z = (rs.zipf(1.07, int(L.sum())) % V).astype(np.int64)
    start = np.concatenate([[0], np.cumsum(L)[:-1]])
    z[start] = np.maximum(z[start], 1)                      # first token of every row non-zero
    ids = np.zeros((B, T), dtype=np.int64)
    rows = np.repeat(np.arange(B), L)
    cols = np.arange(int(L.sum())) - np.repeat(start, L)
    ids[rows, cols] = z
    return torch.from_numpy(ids)


def encoder_corpus_leg(dev, model, n_docs=2_097_152, seed=11):
    """The screened search on embeddings the ENCODER produces (north_star: "synthetic MS-MARCO-shaped queries/passages"):
    corpus = document-tower outputs for n_docs synthetic Zipf passages (model.py:71-74: unit rows, but anisotropic --
    nothing like the isotropic randn rows of the headline corpus), queries = query-tower outputs.  Reports queries/s at
    B = 1024 and B = 32, what the fp16 filter let through (pooled candidates / survivors per query), the exact-kernel
    fallback count, and checks the results bit for bit against the plain fp32 kernel (K4)."""
    import numpy as np
    import twotowermlretrieval_amd as tt
    rs = np.random.RandomState(seed)
    model.eval()
    D = torch.empty((n_docs, ENC_H), dtype=torch.float32, device=dev)
    t0 = time.perf_counter()
    with torch.no_grad():
        for lo in range(0, n_docs, 8192):
            n = min(8192, n_docs - lo)
            D[lo:lo + n] = model.encode_document(make_ids_bulk(rs, n, 70, 10, 250, ENC_V).to(dev))
        q = model.encode_query(make_ids_bulk(rs, BATCH, 6, 1, 30, ENC_V).to(dev))
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t0
    ix = tt.BruteForceIndex(D, screen=True)
    assert ix.docs16 is not None
    ix.keep_stats = True
    # how far from isotropic: mean pairwise cosine of a sample, spread of one query's scores over the corpus
    samp = D[torch.randint(0, n_docs, (2048,), device=dev)]
    mean_cos = float(((samp @ samp.t()).sum() - 2048) / (2048 * 2047))
    sc = q[:64] @ D[:262144].t()
    out = {"docs": n_docs, "corpus": "doc-tower outputs (1-layer GRU, random init, Zipf(1.07) passages ~Poisson(70))",
           "queries": "query-tower outputs (~Poisson(6) tokens)", "mean_pairwise_cos": round(mean_cos, 4),
           "score_std_per_query": round(float(sc.std(dim=1).mean()), 5), "index_build_s_incl_host_id_generation": round(t_build, 2)}
    for B in (BATCH, 32):
        qb = q[:B].contiguous()
        t = time_search(ix, qb, TOPK)
        v, i = ix.search(qb, TOPK)
        st = ix.search_stats().to(torch.float32)
        flags = int(ix.fallback_flags.ne(0).sum().item())
        ev, ei = tt.score_topk(qb, D, TOPK)
        ms_k = screen_kernel_ms(ix, qb, TOPK, iters=10, warm=2)
        leg = {"queries_per_s": round(B / t * 1e3, 1), "search_ms": round(t, 4), "screen_kernel_ms": round(ms_k, 4),
               "pooled_per_query_mean": round(float(st[:, 0].mean()), 1), "pooled_per_query_max": int(st[:, 0].max()),
               "survivors_per_query_mean": round(float(st[:, 1].mean()), 1), "survivors_per_query_max": int(st[:, 1].max()),
               "exact_fallback_tiles": flags, "identical_to_exact_f32": bool(torch.equal(v, ev) and torch.equal(i, ei))}
        if B == 32:
            byts = n_docs * DIM * 2 + 32 * DIM * 4
            leg.update({"bound": "hbm", "achieved": round(byts / ms_k / 1e6, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": round(byts / ms_k / 1e6 / HBM_PEAK_GBPS, 4)})
        else:
            fl = 2.0 * B * n_docs * DIM
            leg.update({"bound": "mfma", "achieved": round(fl / ms_k / 1e9, 2), "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(fl / ms_k / 1e9 / MFMA_F16_PEAK_TFLOPS, 4)})
        out[f"b{B}"] = leg
    del ix, D
    torch.cuda.empty_cache()
    return out


def make_clustered_corpus(n_docs, n_queries, dev, seed=21, n_centres=20_000, dup_groups=1000, dup=64, noise=0.08, common=0.6):
    """A corpus the screen can lose on (MS MARCO has near-duplicate passages and a trained tower clusters them,
    backend/model.py:71-74): unit rows = one of n_centres centres + small noise, the centres themselves sharing a common
    direction (pairwise cosine of two random rows ~ `common`, of two rows of one cluster ~ 1 - noise^2); on top, dup_groups
    groups of `dup` EXACT duplicates (more tied documents than any k) scattered over the corpus.  Queries = noisy copies of
    centres, the first dup_groups of them copies of the duplicated rows (so the tie groups ARE the top of their lists).
    Returns (D [n_docs,256], Q [n_queries,256], first row index of every duplicate group's members [dup_groups, dup])."""
    g = torch.Generator(device=dev).manual_seed(seed)
    u = torch.randn(DIM, device=dev, generator=g)
    u /= u.norm()
    cen = torch.randn((n_centres, DIM), device=dev, generator=g)
    cen /= cen.norm(dim=1, keepdim=True)
    cen = (common ** 0.5) * u + ((1.0 - common) ** 0.5) * cen
    cen /= cen.norm(dim=1, keepdim=True)
    D = torch.empty((n_docs, DIM), dtype=torch.float32, device=dev)
    for lo in range(0, n_docs, GEN_BLOCK):
        n = min(GEN_BLOCK, n_docs - lo)
        z = torch.randint(0, n_centres, (n,), device=dev, generator=g)
        x = cen[z] + noise * torch.randn((n, DIM), device=dev, generator=g) / (DIM ** 0.5)
        D[lo:lo + n] = x / x.norm(dim=1, keepdim=True)
    dup_groups = min(dup_groups, n_docs // (2 * dup))
    members = torch.randperm(n_docs, device=dev, generator=g)[:dup_groups * dup].view(dup_groups, dup)
    D[members.reshape(-1)] = D[members[:, 0]].repeat_interleave(dup, dim=0)
    zq = torch.randint(0, n_centres, (n_queries,), device=dev, generator=g)
    Q = cen[zq] + noise * torch.randn((n_queries, DIM), device=dev, generator=g) / (DIM ** 0.5)
    nq_dup = min(dup_groups, n_queries // 4)      # a quarter of the queries at most: copies of duplicated rows
    if nq_dup:
        Q[:nq_dup] = D[members[:nq_dup, 0]] + 0.01 * torch.randn((nq_dup, DIM), device=dev, generator=g) / (DIM ** 0.5)
    Q /= Q.norm(dim=1, keepdim=True)
    return D, Q.contiguous(), members


def clustered_corpus_leg(dev, n_docs=4_194_304):
    """Queries/s, what the fp16 filter let through and how many 32-query tiles fell back to the exact kernel on the
    clustered corpus (make_clustered_corpus), next to the rate when EVERY tile falls back (the plain fp32 kernel on the same
    batch): the worst case of the screened search is that number, not a wrong result -- checked bit for bit against it here."""
    import twotowermlretrieval_amd as tt
    D, Q, members = make_clustered_corpus(n_docs, BATCH, dev)
    samp = D[torch.randint(0, n_docs, (2048,), device=dev)]
    mean_cos = float(((samp @ samp.t()).sum() - 2048) / (2048 * 2047))
    ix = tt.BruteForceIndex(D, screen=True)
    assert ix.docs16 is not None
    ix.keep_stats = True
    out = {"docs": n_docs, "corpus": "20 000 clustered centres + noise, 1 000 groups of 64 exact duplicates; queries = noisy "
                                     "copies of centres, a quarter of them of duplicated rows",
           "mean_pairwise_cos": round(mean_cos, 4), "duplicate_groups": int(members.shape[0]), "duplicates_per_group": int(members.shape[1])}
    for B in (BATCH, 32):
        qb = Q[:B].contiguous()
        t = time_search(ix, qb, TOPK)
        v, i = ix.search(qb, TOPK)
        st = ix.search_stats().to(torch.float32)
        flags = int(ix.fallback_flags.ne(0).sum().item())
        ev, ei = tt.score_topk(qb, D, TOPK)
        ws = torch.empty(max(_lib_ws_bytes(B, n_docs), 16), dtype=torch.uint8, device=dev)
        t_exact = _time_gpu(lambda: tt.score_topk(qb, D, TOPK, 0, ws), 3, 1) * 1e3
        out[f"b{B}"] = {"queries_per_s": round(B / t * 1e3, 1), "search_ms": round(t, 4),
                        "pooled_per_query_mean": round(float(st[:, 0].mean()), 1), "pooled_per_query_max": int(st[:, 0].max()),
                        "survivors_per_query_mean": round(float(st[:, 1].mean()), 1), "survivors_per_query_max": int(st[:, 1].max()),
                        "exact_fallback_tiles": flags, "tiles": (B + 31) // 32,
                        "queries_per_s_if_every_tile_falls_back": round(B / t_exact * 1e3, 1),
                        "identical_to_exact_f32": bool(torch.equal(v, ev) and torch.equal(i, ei))}
    del ix, D
    torch.cuda.empty_cache()
    return out


def _lib_ws_bytes(B, N):
    from twotowermlretrieval_amd import _lib
    return _lib.lib().tt_score_topk_workspace_bytes(B, N, DIM, TOPK)


def index_build_from_strings_leg(dev, model, gpu_docs_per_s, n_docs=524_288, seed=3):
    """Index build from STRINGS (backend/main.py:125-138 over tokenizer.py:41-43): synthetic Zipf passages as text ->
    native tokeniser on several producer threads -> pinned batches -> document tower (evaluators.embed_corpus), against the
    GPU-only rate of the `index_build_b8192` leg (ids already on the device)."""
    import numpy as np
    import twotowermlretrieval_amd as tt
    from twotowermlretrieval_amd.evaluators import embed_corpus
    words = ["the", ",", ".", "of", "and"] + [f"w{i}" for i in range(5, ENC_V - 1)]
    tok = tt.PretrainedTokenizer(word2idx={w: i for i, w in enumerate(words)})
    assert tok.vocab_size() == ENC_V
    rs = np.random.RandomState(seed)
    lens = np.clip(rs.poisson(70, n_docs), 10, 250)
    z = rs.zipf(1.07, int(lens.sum())) % (ENC_V - 1)
    docs, p0 = [], 0
    for L_ in lens:
        docs.append(" ".join(map(words.__getitem__, z[p0:p0 + L_])))
        p0 += L_
    n_tok = int(lens.sum())
    model.eval()
    embed_corpus(model, tok, docs, dev)                  # warm-up = one whole pass: vocabulary table, pinned ring, and the caching
    torch.cuda.synchronize()                             # allocator's workspace blocks for every batch width (a 10M-100M build runs warm)
    stats = {}
    _settle_gc()    # (a generation-2 collection inside a 75 ms timed region -- 262 k fresh strings invite one -- is a third of it)
    t0 = time.perf_counter()
    emb = embed_corpus(model, tok, docs, dev, stats=stats)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stage = torch.empty(16384 * 160, dtype=torch.int64, pin_memory=True)   # (a reused pinned block, as embed_corpus's ring)
    tok.encode_batch(docs[:16384], out=stage)
    t1 = time.perf_counter()
    for i in range(0, 65536, 16384):
        tok.encode_batch(docs[i:i + 16384], out=stage)
    t_host = (time.perf_counter() - t1) / 65536 * n_docs
    # the same build with typographic / accented characters in 5 % of the passages (MS MARCO has them): tokenised natively from
    # CPython's code units with the interpreter's own Unicode tables (tt_tok_encode_units) -- no Python fallback, no cliff
    docs5 = [d.replace(" ", " \u2019s caf\u00e9 ", 2) if i % 20 == 0 else d for i, d in enumerate(docs)]
    embed_corpus(model, tok, docs5, dev)     # (a whole pass again: other batch widths, other workspace blocks; Unicode tables)
    torch.cuda.synchronize()
    _settle_gc()
    t2 = time.perf_counter()
    embed_corpus(model, tok, docs5, dev)
    torch.cuda.synchronize()
    dt5 = time.perf_counter() - t2
    del docs5
    # the GPU's own rate on THESE passages at the build's batch size (ids resident: 16 consecutive batches of 32 768)
    bs = stats.get("batch_size", 32768)
    resident = [tok.encode_batch(docs[i:i + bs]).to(dev) for i in range(0, n_docs, bs)]
    with torch.no_grad():
        t_gpu = _time_gpu(lambda: [model.encode_document(x) for x in resident], 2, 1)
    gpu_same = n_docs / t_gpu
    return {"docs": n_docs, "tokens": n_tok, "s": round(dt, 3), "docs_per_s": round(n_docs / dt), "tokens_per_s": round(n_tok / dt),
            "host_front_end_alone_tokens_per_s_one_producer": round(n_tok / t_host),
            "producers": stats.get("producers"), "threads_per_producer": stats.get("threads_per_producer"),
            "host_cores": stats.get("host_cores"), "batch_size": bs,
            "gpu_only_docs_per_s": round(gpu_same), "frac_of_gpu_only_rate": round(n_docs / dt / gpu_same, 3),
            "gpu_only_docs_per_s_b8192_leg": gpu_docs_per_s, "docs_per_s_with_5pct_non_ascii_passages": round(n_docs / dt5),
            "rows": list(emb.shape)}


def _settle_gc():
    """A full collection now, and everything alive moved to the permanent generation: CPython's generation-2 pass walks every
    tracked object of the process (~40 ms with torch and numpy imported, once per ~100 train steps: tools/experiments/
    step_times.py) and would land inside some of the encoder / train legs' timed loops and not others.  The collector stays on.
    NOT used around the headline loop (10 - 20 search calls allocate too little to trigger a collection, and a full collection
    in front of them made the following calls 1 - 4 % slower, A/B on one box)."""
    if os.environ.get("TT_BENCH_NO_GC_FREEZE") == "1":  # (A/B switch)
        return
    gc.collect()
    gc.freeze()


def _time_gpu(fn, iters, warm):
    _settle_gc()   # BEFORE the warm-up: the first calls after a full collection pay ~3 ms once (A/B: TT_BENCH_NO_GC_FREEZE)
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def make_encoder_inputs(dev, with_index_batch=True):
    """The synthetic inputs of the encoder / train legs, also used by tests/test_bench_size_gpu.py to check
    these very launches against the oracle: embedding table randn * 0.3 [V,E] (CPU tensor), the two-tower model with
    torch.manual_seed(0) default init on `dev`, Zipf(1.07) id batches q / p / n (B=512) and the index-build batch
    (B=8192), each with its count of non-zero tokens."""
    import numpy as np
    import twotowermlretrieval_amd as tt
    rs = np.random.RandomState(0)
    g = torch.Generator(device=dev).manual_seed(5)
    table = (torch.randn((ENC_V, ENC_E), dtype=torch.float32, device=dev, generator=g) * 0.3).cpu()
    torch.manual_seed(0)
    m = tt.TwoTowerModel({"VOCAB_SIZE": ENC_V, "EMBED_DIM": ENC_E, "HIDDEN_DIM": ENC_H}, table.numpy()).to(dev)
    B = 512
    q, qt = make_ids(rs, B, 6, 1, 30, ENC_V)
    p, pt = make_ids(rs, B, 70, 10, 250, ENC_V)
    n, nt = make_ids(rs, B, 70, 10, 250, ENC_V)
    big, bt = make_ids(rs, 8192, 70, 10, 250, ENC_V) if with_index_batch else (None, 0)
    return {"table": table, "model": m, "B": B, "q": q, "p": p, "n": n, "big": big, "qt": qt, "pt": pt, "nt": nt, "bt": bt}


def graph_legs(dev):
    """The HIP-graph legs, run in a process of their own (graph_legs_child): the whole direct train step as ONE graph launch
    (trainer.GraphedTrainStep: ids copied into static buffers padded to the next multiple of 32 columns; same kernels, same failure
    semantics -- the gate words are read after every replay), with and without the deferred read, for the north-star model at 512
    triplets and for the reference's default model (config.json: 2 layers, bidirectional, dropout -- seeds as device words) at 64."""
    import numpy as np
    import twotowermlretrieval_amd as tt
    inp = make_encoder_inputs(dev, with_index_batch=False)
    m, B = inp["model"], inp["B"]
    q, p, n = inp["q"], inp["p"], inp["n"]
    qd, pd, nd = q.to(dev), p.to(dev), n.to(dev)
    tok = inp["qt"] + inp["pt"] + inp["nt"]
    r32 = lambda x: (int(x) + 31) // 32 * 32  # noqa: E731
    out = {}
    m.train()
    opt = tt.FusedClipAdam(m.parameters(), lr=5e-5, max_norm=1.0)
    try:
        gstep = tt.GraphedTrainStep(m, opt, batch=B, q_width=r32(q.shape[1]), doc_width=r32(max(p.shape[1], n.shape[1])), margin=0.5)
        t_g = _time_gpu(lambda: gstep(qd, pd, nd), 5, 2)
        out["graphed"] = {"ms_per_step": round(t_g * 1e3, 3), "triplets_per_s": round(B / t_g),
                          "q_width": gstep.q_width, "doc_width": gstep.doc_width,
                          "frac_f16_mfma_3x": round(3.0 * tok * FLOP_PER_TOKEN_TRAIN / t_g / 1e12 / MFMA_F16_PEAK_TFLOPS, 4)}
        del gstep
        lazy = tt.GraphedTrainStep(m, opt, batch=B, q_width=r32(q.shape[1]), doc_width=r32(max(p.shape[1], n.shape[1])), margin=0.5,
                                   defer_check=True)     # step i's gate words read inside call i + 1
        t_l = _time_gpu(lambda: lazy(qd, pd, nd), 5, 2)
        lazy.flush()
        out["graphed"]["deferred_check_ms_per_step"] = round(t_l * 1e3, 3)
        out["graphed"]["deferred_check_triplets_per_s"] = round(B / t_l)
        del lazy
    except Exception as e:  # noqa: BLE001 -- the eager leg is the record; say why the graph leg is missing
        out["graphed"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    del opt, m
    try:
        rs = np.random.RandomState(7)
        tab1 = torch.from_numpy((rs.standard_normal((ENC_V, 200)) * 0.3).astype(np.float32))
        torch.manual_seed(1)
        m1 = tt.TwoTowerModel({"VOCAB_SIZE": ENC_V, "EMBED_DIM": 200, "HIDDEN_DIM": ENC_H, "NUM_LAYERS": 2, "BIDIRECTIONAL": True,
                               "DROPOUT": 0.2}, tab1).to(dev)
        m1.train()
        opt1 = tt.FusedClipAdam(m1.parameters(), lr=5e-5, max_norm=1.0)
        q64, p64, n64 = qd[:64].contiguous(), pd[:64].contiguous(), nd[:64].contiguous()
        g64 = tt.GraphedTrainStep(m1, opt1, batch=64, q_width=r32(q64.shape[1]), doc_width=r32(max(p64.shape[1], n64.shape[1])),
                                  margin=0.5, defer_check=True)    # (dropout seeds as device words: include/tt.h TT_ENC_SEED_ON_DEVICE)
        tg = _time_gpu(lambda: g64(q64, p64, n64), 8, 2)
        g64.flush()
        out["config_json_model"] = {"graphed_deferred_ms_per_step_64_triplets": round(tg * 1e3, 3),
                                    "graphed_deferred_triplets_per_s_64": round(64 / tg)}
    except Exception as e:  # noqa: BLE001
        out["config_json_model"] = {"graphed_error": f"{type(e).__name__}: {e}"[:300]}
    return out


def graph_legs_child(timeout_s: int = 600):
    """graph_legs() in a child process (`bench.py --graph-legs-child`, one JSON line on stdout).  Stream capture is the one part
    of the bench that has taken a process down before -- a ROCm runtime fault inside hipStreamEndCapture for some stream patterns,
    DESIGN 4 -- and a fault there must cost the line these keys, not the line.  A started child is never exec'd over: it is an
    ordinary subprocess of this (GPU-initialised) process."""
    import subprocess
    # (a profiler around this process stays with this process: its preloaded library would open a second trace for the child)
    env = {k: v for k, v in os.environ.items()
           if k not in ("LD_PRELOAD", "HSA_TOOLS_LIB", "HSA_TOOLS_REPORT_LOAD_FAILURE") and not k.startswith(("ROCP", "ROCPROF"))}
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--graph-legs-child"], stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, text=True, timeout=timeout_s, env=env)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode == 0 and lines:
            return json.loads(lines[-1])
        why = f"child exit code {r.returncode}: {r.stderr.strip()[-240:]}"
    except Exception as e:  # noqa: BLE001  (time-out, no interpreter, unparsable output)
        why = f"{type(e).__name__}: {e}"[:300]
    return {"graphed": {"error": why}, "config_json_model": {"graphed_error": why}}


def encoder_legs(dev):
    """Tower forward (B=512), index build (B=8192) and the train step (512 triplets) on this rank's GPU.
    Returns (encoder dict, train dict, inputs for the CPU legs)."""
    import twotowermlretrieval_amd as tt
    inp = make_encoder_inputs(dev)
    table, m, B = inp["table"], inp["model"], inp["B"]
    q, p, n, big = inp["q"], inp["p"], inp["n"], inp["big"]
    qt, pt, nt, bt = inp["qt"], inp["pt"], inp["nt"], inp["bt"]
    qd, pd, nd, bigd = q.to(dev), p.to(dev), n.to(dev), big.to(dev)
    m.eval()
    with torch.no_grad():
        t_doc = _time_gpu(lambda: m.encode_document(pd), 10, 3)
        t_q = _time_gpu(lambda: m.encode_query(qd), 10, 3)
        t_big = _time_gpu(lambda: m.encode_document(bigd), 3, 1)

    projected = dev in m.doc_encoder._proj and dev in m.query_encoder._proj
    fpt = FLOP_PER_TOKEN_REC if projected else FLOP_PER_TOKEN_FWD

    def leg(tokens, t, extra):
        tf = tokens * fpt / t / 1e12
        return {"tokens_per_s": round(tokens / t), "ms": round(t * 1e3, 3), "tokens": tokens,
                "TFLOPs": round(tf, 2), "frac_f32_mfma": round(tf / MFMA_F32_PEAK_TFLOPS, 4),
                "frac_f16_mfma_3x": round(3.0 * tf / MFMA_F16_PEAK_TFLOPS, 4), **extra}
    enc = {"model": f"1-layer GRU, V={ENC_V}, E={ENC_E}, H={ENC_H}, fp32 (model.py:48-75)",
           "arith": "fp32-grade: every product is three f16 MFMAs on fp16 hi/lo splits of the fp32 operands (DESIGN 4); "
                    "TFLOPs counts each fp32 multiply-add once, so frac_f32_mfma may exceed 1; frac_f16_mfma_3x = 3 x "
                    "TFLOPs / f16 peak is the share of the matrix pipes actually used",
           "input_projection": ("projected table: table W_ih^T + b_ih of every vocabulary row computed once per weight version "
                                f"({ENC_V * 3 * ENC_H * 4 / 1e9:.2f} GB per tower), gathered by the recurrence kernels -- the flops "
                                "counted are the recurrence's only (2*3H*H per token)") if projected else "a GEMM over the batch's tokens in every call",
           "doc_tower_b512": leg(pt, t_doc, {"batch": B, "T": int(p.shape[1])}),
           "query_tower_b512": leg(qt, t_q, {"batch": B, "T": int(q.shape[1]), "queries_per_s": round(B / t_q)}),
           "index_build_b8192": leg(bt, t_big, {"batch": 8192, "T": int(big.shape[1]), "docs_per_s": round(8192 / t_big)})}
    # one serving query (frontend/main.py:152: batch 1) and the index build's own batch size
    with torch.no_grad():
        q1 = qd[:1].contiguous()
        t_q1 = _time_gpu(lambda: m.encode_query(q1), 50, 5)
        big4 = torch.cat([bigd] * 4, 0)
        t_b4 = _time_gpu(lambda: m.encode_document(big4), 3, 1)
        del big4
    enc["query_tower_b1"] = {"ms": round(t_q1 * 1e3, 4), "tokens": int((q1 != 0).sum())}
    enc["index_build_b32768"] = leg(4 * bt, t_b4, {"batch": 32768, "T": int(big.shape[1]), "docs_per_s": round(32768 / t_b4)})
    m.train()
    opt = tt.FusedClipAdam(m.parameters(), lr=5e-5, max_norm=1.0)
    t_tr = _time_gpu(lambda: tt.train_step(m, opt, qd, pd, nd, margin=0.5), 5, 2)
    tok = qt + pt + nt
    tf = tok * FLOP_PER_TOKEN_TRAIN / t_tr / 1e12
    train = {"triplets_per_s": round(B / t_tr), "ms_per_step": round(t_tr * 1e3, 3), "triplets": B, "tokens": tok,
             "step": "3 tower forwards + triplet loss + backward + clip_grad_norm_(1.0) + Adam (main.py:244-259), default "
                     "input checking",
             "TFLOPs": round(tf, 2), "frac_f32_mfma": round(tf / MFMA_F32_PEAK_TFLOPS, 4),
             "frac_f16_mfma_3x": round(3.0 * tf / MFMA_F16_PEAK_TFLOPS, 4)}
    # the eager step with the gate read one call late (no host synchronisation inside a step)
    t_d = _time_gpu(lambda: tt.train_step(m, opt, qd, pd, nd, margin=0.5, defer_check=True), 5, 2)
    opt.settle()
    train["deferred_check_ms_per_step"] = round(t_d * 1e3, 3)
    train["deferred_check_triplets_per_s"] = round(B / t_d)
    # the same step as ONE HIP graph launch, and the reference's default model's: measured in a CHILD process (graph_legs_child)
    graphs = graph_legs_child()
    train["graphed"] = graphs.get("graphed", {"error": "no result"})
    del opt
    # the reference's DEFAULT model (backend/config.json:13-17: E = 200, 2 layers, bidirectional, dropout 0.2 -- BASELINE
    # configs[0]'s model) on the same triplets' shapes, at 512 triplets and at config.json's own BATCH_SIZE of 64
    try:
        import numpy as np
        rs = np.random.RandomState(7)
        tab1 = torch.from_numpy((rs.standard_normal((ENC_V, 200)) * 0.3).astype(np.float32))
        torch.manual_seed(1)
        m1 = tt.TwoTowerModel({"VOCAB_SIZE": ENC_V, "EMBED_DIM": 200, "HIDDEN_DIM": ENC_H, "NUM_LAYERS": 2, "BIDIRECTIONAL": True,
                               "DROPOUT": 0.2}, tab1).to(dev)
        m1.train()
        opt1 = tt.FusedClipAdam(m1.parameters(), lr=5e-5, max_norm=1.0)
        t512 = _time_gpu(lambda: tt.train_step(m1, opt1, qd, pd, nd, margin=0.5), 4, 2)
        t64 = _time_gpu(lambda: tt.train_step(m1, opt1, qd[:64], pd[:64], nd[:64], margin=0.5), 8, 2)
        train["config_json_model"] = {"model": "E=200, H=256, 2 layers, bidirectional, dropout 0.2 (backend/config.json:13-17)",
                                      "ms_per_step_512_triplets": round(t512 * 1e3, 3), "triplets_per_s_512": round(512 / t512),
                                      "ms_per_step_64_triplets": round(t64 * 1e3, 3), "triplets_per_s_64": round(64 / t64)}
        train["config_json_model"].update(graphs.get("config_json_model", {"graphed_error": "no result"}))
        del opt1, m1, tab1
    except Exception as e:  # noqa: BLE001
        train["config_json_model"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    torch.cuda.empty_cache()
    return enc, train, (table, q, p, n), m


# ---- legs EVERY rank executes (their constructors and steps are collectives): same order on every rank ------------------------
PCIE_PEAK_GBPS = 64.0     # PCIe Gen5 x16 per direction (MI355X_MICROARCH.md: 63 GB/s spec)
STREAM_DOCS_PER_GPU = 12_500_000   # BASELINE configs[4]: 100M x 256 bf16 over 8 GPUs


def _max_over_ranks(x: float, dev, world: int) -> float:
    if world == 1:
        return x
    t = torch.tensor([x], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def _fence(world: int):
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()


def streamed_leg(dev, rank, world, n_shard=STREAM_DOCS_PER_GPU, iters=3):
    """BASELINE configs[4] as stated: every GPU owns n_shard = 12.5M rows of a bf16 corpus (N x 12.5M rows in all: at N = 8 the
    100M-passage corpus) that stay in PINNED host memory; one pass = every rank streams its shard through its GPU in 1M-row
    blocks (copy stream: hipMemcpyAsync, compute stream: widen + screened search of the block + running top-k), then the usual
    per-shard top-50 all-gather + merge to the global top-10 (ShardedIndex over a StreamedIndex).  Weak scaling: the shard per
    GPU is fixed, the corpus grows with N.  PCIe binds (6.4 GB per pass and GPU), so the figure is GB/s per GPU against the
    64 GB/s of a Gen5 x16 link; beside it the same shard widened once into HBM (it is 19.2 GB of 288) and searched resident."""
    import twotowermlretrieval_amd as tt
    lo = rank * n_shard
    host = torch.empty((n_shard, DIM), dtype=torch.bfloat16, pin_memory=True)
    for b0 in range(0, n_shard, GEN_BLOCK):
        b1 = min(n_shard, b0 + GEN_BLOCK)
        g = torch.Generator(device=dev).manual_seed(5000 + (lo + b0) // GEN_BLOCK)
        x = torch.randn((b1 - b0, DIM), dtype=torch.float32, device=dev, generator=g)
        x /= x.norm(dim=1, keepdim=True).clamp_min(1e-12)
        host[b0:b1].copy_(x.to(torch.bfloat16))
        del x
    torch.cuda.synchronize()
    q = gen_queries(BATCH, dev, seed=8)
    six = tt.StreamedIndex(host, block_docs=1 << 20, device=dev, idx_offset=lo)
    sh = tt.ShardedIndex(six, lo, shard_k=SHARD_K)

    def timed(fn, n, warm=1):
        for _ in range(warm):
            out = fn()
        _fence(world)
        t0 = time.perf_counter()
        for _ in range(n):
            out = fn()
        _fence(world)
        return _max_over_ranks((time.perf_counter() - t0) / n, dev, world), out

    t_s, (sv, si) = timed(lambda: sh.search(q, TOPK), iters)
    q32 = q[:32].contiguous()
    t_s32, _ = timed(lambda: sh.search(q32, TOPK), iters)
    # PCIe binds, not the kernels: a pass costs the same for four times the queries (the knob a streamed deployment has)
    q4k = gen_queries(4 * BATCH, dev, seed=9)
    t_s4k, _ = timed(lambda: sh.search(q4k, TOPK), 2)
    del q4k
    res = six.resident()                                     # widened once into HBM (fp32 rows + fp16 shadow)
    rsh = tt.ShardedIndex(res.docs, lo, shard_k=SHARD_K, screen=True)
    del res
    t_r, (rv, ri) = timed(lambda: rsh.search(q, TOPK), 10, warm=2)
    same = bool(torch.equal(sv, rv) and torch.equal(si, ri))
    byts = n_shard * DIM * 2
    out = {"workload": f"BASELINE configs[4]: exact top-{TOPK} of B={BATCH} queries over {world} x {n_shard} x {DIM} bf16 passages, "
                       f"each GPU's shard in pinned host DRAM streamed in 1M-row blocks (async HIP copies on a copy stream "
                       f"under the block searches), per-shard top-{SHARD_K} + all-gather + merge",
           "docs_per_gpu": n_shard, "docs_total": world * n_shard, "scaling": "weak", "bound": "pcie",
           "ms_per_pass": round(t_s * 1e3, 2), "queries_per_s": round(BATCH / t_s, 1),
           "achieved": round(byts / t_s / 1e9, 2), "peak": PCIE_PEAK_GBPS, "unit": "GB/s per GPU", "frac": round(byts / t_s / 1e9 / PCIE_PEAK_GBPS, 4),
           "b32_ms_per_pass": round(t_s32 * 1e3, 2), "b32_queries_per_s": round(32 / t_s32, 1),
           "b4096_ms_per_pass": round(t_s4k * 1e3, 2), "b4096_queries_per_s": round(4 * BATCH / t_s4k, 1),
           "resident": {"what": f"the same shards widened once into HBM (fp32 rows + fp16 shadow = {n_shard * DIM * 6 / 1e9:.1f} GB per GPU), same exchange",
                        "ms_per_pass": round(t_r * 1e3, 3), "queries_per_s": round(BATCH / t_r, 1)},
           "identical_to_resident": same, "collective": sh.collective if world > 1 else None}
    del sh, rsh, six, host
    torch.cuda.empty_cache()
    return out


def dp_train_leg(dev, rank, world, steps=5, warm=2):
    """BASELINE configs[2]: triplet training data-parallel over the ranks (backend/main.py:244-259 per rank; SURVEY 8e: ONE
    summing all-reduce of the flat gradient bucket, x 1/W, then clip, then Adam).  512 triplets per GPU (4096 at N = 8) of one
    seeded global batch, north-star model (GloVe-300-shaped frozen table, 1-layer GRU, H = 256).  Every rank constructs the
    trainer and runs the same steps; the time is the slowest rank's.  The all-reduce's share is event-timed on the bucket."""
    import numpy as np
    import twotowermlretrieval_amd as tt
    per = 512
    g = torch.Generator(device=dev).manual_seed(5)
    table = (torch.randn((ENC_V, ENC_E), dtype=torch.float32, device=dev, generator=g) * 0.3).cpu()
    torch.manual_seed(0)
    m = tt.TwoTowerModel({"VOCAB_SIZE": ENC_V, "EMBED_DIM": ENC_E, "HIDDEN_DIM": ENC_H}, table.numpy()).to(dev)
    del table
    rs = np.random.RandomState(0)
    G = per * world
    ids = [make_ids_bulk(rs, G, mean, lo_, hi_, ENC_V) for mean, lo_, hi_ in ((6, 1, 30), (70, 10, 250), (70, 10, 250))]
    mine = [x[rank * per:(rank + 1) * per] for x in ids]
    mine = [x[:, :max(int((x != 0).sum(1).max()), 1)].contiguous().to(dev) for x in mine]   # (the shard's own widest row)
    tok_global = int(sum((x != 0).sum() for x in ids))
    tr = tt.trainer.DataParallelTrainer(m, lr=5e-5, margin=0.5)
    tr.broadcast_parameters()
    opt = tr.optimizer

    def timed(n, w):
        for _ in range(w):
            tr.step(*mine)
        tr.flush()
        _fence(world)
        t0 = time.perf_counter()
        for _ in range(n):
            tr.step(*mine)
        tr.flush()
        _fence(world)
        return _max_over_ranks((time.perf_counter() - t0) / n, dev, world)

    _settle_gc()
    t = timed(steps, warm)
    opt.reduce_timing = []
    timed(steps, 0)
    torch.cuda.synchronize()
    red_ms = sum(e0.elapsed_time(e1) for e0, e1, _ in opt.reduce_timing) / steps if opt.reduce_timing else 0.0
    red_bytes = sum(b for _, _, b in opt.reduce_timing) // steps if opt.reduce_timing else 0
    n_red = len(opt.reduce_timing) // steps
    opt.reduce_timing = None
    tr.defer_check = True
    t_d = timed(steps, warm)
    out = {"workload": f"BASELINE configs[2]: {G} triplets per step = {per} per GPU x {world} (4096 at N = 8), GloVe-300-shaped frozen "
                       f"table V={ENC_V}, GRU H={ENC_H}; per rank: 3 tower forwards + triplet loss + backward, then ONE flat "
                       f"all-reduce (query tower's prefix early), x 1/W, clip_grad_norm_(1.0), Adam (main.py:244-259)",
           "scaling": "weak", "triplets_per_s": round(G / t), "ms_per_step": round(t * 1e3, 3), "triplets_per_step": G,
           "tokens_per_step": tok_global,
           "allreduce_ms_per_step_rank0": round(red_ms, 4), "allreduce_share_of_step": round(red_ms / (t * 1e3), 4),
           "allreduce_calls_per_step": n_red, "allreduce_bytes_per_step": int(red_bytes),
           "deferred_check_ms_per_step": round(t_d * 1e3, 3), "deferred_check_triplets_per_s": round(G / t_d),
           "collective": opt._coll.via if world > 1 else None}
    del tr, opt, m
    torch.cuda.empty_cache()
    return out


def collective_legs(dev, rank, world, a):
    legs = {}
    if not a.no_streamed:
        legs["streamed_bf16"] = streamed_leg(dev, rank, world, n_shard=a.stream_docs)
    if not a.no_secondary:
        legs["dp_train"] = dp_train_leg(dev, rank, world)
    return legs


def resident_100m_leg(dev, n=100_000_000):
    """BASELINE configs[4]'s WHOLE corpus (100M x 256) resident on ONE MI355X: fp32 rows + fp16 shadow = 153.6 GB of the 288 GB
    (what the sharded / streamed forms are the alternative to).  Screened exact top-10 at the bench batch and a serving batch;
    the serving batch is compared bit for bit with the plain fp32 kernel, planted documents must come back first."""
    import twotowermlretrieval_amd as tt
    D = gen_rows(0, n, dev)
    Q = gen_queries(BATCH, dev, seed=12)
    planted = torch.tensor([0, 77, n // 2 + 1, n - 1], device=dev)
    D[planted] = Q[:4]
    ix = tt.BruteForceIndex(D, screen=True)
    out = {"docs": n, "hbm_GB": round(n * DIM * 6 / 1e9, 1), "what": "exact top-10 over 100M x 256 passages resident on one GPU (fp32 rows + fp16 shadow)"}
    for B in (BATCH, 32):
        q = Q[:B].contiguous()
        t = time_search(ix, q, TOPK, iters=4, warm=2)
        v, i = ix.search(q, TOPK)
        ok = i[:4, 0].tolist() == planted.tolist() and bool((v[:, 1:] <= v[:, :-1]).all())
        leg = {"search_ms": round(t, 3), "queries_per_s": round(B / t * 1e3, 1), "exact_fallback_tiles": int(ix.fallback_flags.ne(0).sum().item()),
               "planted_first_and_sorted": ok}
        if B == 32:
            ev, ei = tt.score_topk(q, D, TOPK)
            leg["identical_to_exact_f32"] = bool(torch.equal(v, ev) and torch.equal(i, ei))
            leg["fp16_stream_GBps"] = round(n * DIM * 2 / t / 1e6, 1)
        out[f"b{B}"] = leg
    del ix, D
    torch.cuda.empty_cache()
    return out


class _SynthPassages:
    """documents.pkl's list for a synthetic corpus too large to hold as Python strings: passage i is a window (start and length
    derived from i) of ONE pre-generated stream of Zipf(1.07) words, ~Poisson(70) words long -- the same text every time it is
    asked for, a few microseconds to produce (the serving leg times the reference's handler, not a text generator)."""

    def __init__(self, n, words, seed=4):
        import numpy as np
        rs = np.random.RandomState(seed)
        self.n = int(n)
        self.stream = [words[j] for j in rs.zipf(1.07, 1 << 21) % len(words)]
        self.lens = np.clip(rs.poisson(70, 1 << 16), 10, 250)

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        i = int(i)
        if not 0 <= i < self.n:
            raise IndexError(i)
        L = int(self.lens[i & 0xFFFF])
        start = (i * 7919) % (len(self.stream) - 256)
        return " ".join(self.stream[start:start + L])


def serve_b1_leg(dev, index, n_queries=200, warm=20):
    """The reference's /search handler with alpha != 0 (frontend/main.py:150-198) for ONE query, host clock, result on the host:
    string -> tokenise -> query tower (QueryInferencer.get_query_embedding: a numpy vector, as the reference's) -> exact top-50
    over the resident 10M-passage corpus (in place of Chroma's HNSW top-50) -> TF-IDF of the query against the 50 candidates'
    texts (sklearn, CPU, as the reference) -> alpha blend -> top-10.  GloVe-size vocabulary, north-star tower, artifacts written
    and read back through the reference's formats.  The only number the reference publishes is for this path: 0.32 s per query
    on its author's machine, corpus size unknown -- context, not a like-for-like baseline."""
    import tempfile
    import numpy as np
    import twotowermlretrieval_amd as tt
    from sklearn.feature_extraction.text import TfidfVectorizer
    from twotowermlretrieval_amd.evaluators import save_inference_artifacts
    words = ["the", ",", ".", "of", "and"] + [f"w{i}" for i in range(5, ENC_V - 1)]
    tok = tt.PretrainedTokenizer(word2idx={w: i for i, w in enumerate(words)})
    cfg = {"HIDDEN_DIM": ENC_H, "NUM_LAYERS": 1, "BIDIRECTIONAL": False, "BATCH_SIZE": 64}
    table = (np.random.RandomState(1).standard_normal((tok.vocab_size(), ENC_E)) * 0.3).astype(np.float32)
    torch.manual_seed(0)
    m = tt.TwoTowerModel({**cfg, "VOCAB_SIZE": tok.vocab_size(), "EMBED_DIM": ENC_E}, table).to(dev)
    with tempfile.TemporaryDirectory() as d:
        save_inference_artifacts(d, m, cfg, tok, ["w5 w6"], dev, tfidf=False)
        del m, table
        inf = tt.QueryInferencer(d, device=dev)
    docs = _SynthPassages(index.ntotal, words)
    vec = TfidfVectorizer(stop_words="english", max_features=20000).fit([docs[i] for i in range(0, 20000)])
    hs = tt.HybridSearcher(inf, docs, tfidf_vectorizer=vec, doc_tfidf_matrix=None, n_candidates=50, index=index)
    rs = np.random.RandomState(0)
    queries = [" ".join(words[i] for i in (rs.zipf(1.07, rs.randint(3, 12)) % (ENC_V - 6)) + 5) for _ in range(n_queries + warm)]
    lat, enc, srch = [], [], []
    for i, qs in enumerate(queries):
        t0 = time.perf_counter()
        res = hs.search(qs, alpha=0.5, n_results=10)
        t1 = time.perf_counter()
        qv = torch.from_numpy(inf.get_query_embedding(qs)).to(dev)            # the stages again, separately
        t2 = time.perf_counter()
        index.search(qv, 50)[1].tolist()
        t3 = time.perf_counter()
        assert len(res) == 10
        if i >= warm:
            lat.append(t1 - t0); enc.append(t2 - t1); srch.append(t3 - t2)
    lat, enc, srch = (np.array(x) * 1e3 for x in (lat, enc, srch))
    del hs, inf
    torch.cuda.empty_cache()
    return {"what": "one query string -> top-10 of the hybrid rerank over the resident corpus (frontend/main.py:150-198 with the exact "
                    "GPU top-50 in place of Chroma), host clock, result on the host",
            "docs": index.ntotal, "queries": n_queries, "p50_ms": round(float(np.median(lat)), 3), "p99_ms": round(float(np.percentile(lat, 99)), 3),
            "mean_ms": round(float(lat.mean()), 3), "queries_per_s_one_caller": round(1e3 / float(lat.mean()), 1),
            "stage_p50_ms": {"tokenise_query_tower_to_numpy": round(float(np.median(enc)), 3),
                             "exact_top50_to_host": round(float(np.median(srch)), 3),
                             "tfidf_blend_on_cpu": round(float(np.median(lat) - np.median(enc) - np.median(srch)), 3)},
            "reference_published_s_per_query": 0.32,
            "reference_published_note": "README.md:9 screenshot: author's machine, corpus size unknown, Chroma HNSW top-50 (approximate)"}


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` from a bare shell: this parent makes NO GPU call; it starts N fresh rank processes
    (never an exec of itself), relays rank 0's JSON line and fails if any rank fails."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve()), *sys.argv[1:]], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    rc = 0
    try:
        out0, _ = procs[0].communicate()
        for pr in procs:
            rc = rc or pr.wait()
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    line = [l for l in (out0 or "").splitlines() if l.startswith("{")]
    if rc == 0 and line:
        print(line[-1], flush=True)
        return 0
    sys.stderr.write(f"bench.py: a rank failed (rc={rc}); rank 0 stdout was:\n{out0}\n")
    return rc or 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the encoder / train / encoder-corpus legs")
    ap.add_argument("--no-encoder-corpus", action="store_true", help="skip the encoder-produced-corpus leg (~20 s)")
    ap.add_argument("--no-streamed", action="store_true", help="skip the configs[4] leg (6.4 GB of pinned host memory per rank)")
    ap.add_argument("--stream-docs", type=int, default=STREAM_DOCS_PER_GPU, help="rows per GPU of the configs[4] leg (rehearsals)")
    ap.add_argument("--graph-legs-child", action="store_true", help="internal: graph_legs() on cuda:0, one JSON line")
    a = ap.parse_args()

    if a.graph_legs_child:
        torch.cuda.set_device(0)
        print(json.dumps(graph_legs(torch.device("cuda", 0))), flush=True)
        return

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(launch_ranks(a.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        sys.exit(f"--gpus {a.gpus} != WORLD_SIZE {world}")
    # TT_BENCH_BACKEND=gloo rehearses the N > 1 code path on a box with fewer GPUs than ranks (ranks then share
    # devices; the numbers mean nothing).  The default, and what the driver runs, is nccl = RCCL, one rank per GPU.
    backend = os.environ.get("TT_BENCH_BACKEND", "nccl")
    local = local if backend == "nccl" else local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import twotowermlretrieval_amd as tt
    lo, hi = tt.shard_bounds(N_DOCS, rank, world)
    docs = gen_rows(lo, hi, dev)
    q = gen_queries(BATCH, dev)
    if world > 1:
        index = tt.ShardedIndex(docs, lo, shard_k=SHARD_K, screen=True)
        local_index = index._index
    else:
        index = local_index = tt.BruteForceIndex(docs, screen=True)
    assert local_index.docs16 is not None, "fp16 shadow copy was not built"
    if world > 1:
        # pipelined steps: step i's all-gather + merge run on the index's second stream while step i+1's local
        # search runs on this one; a step's result is collected when the next one has been enqueued
        pending = []

        def step():
            pending.append(index.submit(q, TOPK))
            return pending.pop(0).result() if len(pending) > 1 else None

        def drain():
            out = None
            while pending:
                out = pending.pop(0).result()
            return out
    else:
        step = lambda: index.search(q, TOPK)  # noqa: E731
        drain = lambda: None  # noqa: E731

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        out = step()
    out = drain() or out
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
    out = drain() or out   # the last step's exchange + merge are inside the timed region
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    vals, idx = out
    assert vals.shape == (BATCH, TOPK) and bool((vals[:, 1:] <= vals[:, :-1]).all()) and int(idx.min()) >= 0

    flags = int(local_index.fallback_flags.ne(0).sum().item())
    # the legs that are collectives (configs[4] streamed shards, configs[2] data-parallel training): all ranks, same order
    shared_legs = collective_legs(dev, rank, world, a)
    if rank == 0:
        n_shard = hi - lo
        kp = TOPK if world == 1 else SHARD_K
        flops = 2.0 * BATCH * n_shard * DIM
        ms_s = screen_kernel_ms(local_index, q, kp, k_seed=TOPK if world > 1 else 0)
        roof = {"bound": "mfma", "kernel": "screen_kernel<false> (f16 MFMA 16x16x32, fp32 accumulate)",
                "achieved": round(flops / ms_s / 1e9, 2), "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(flops / ms_s / 1e9 / MFMA_F16_PEAK_TFLOPS, 4), "traffic": pmc_traffic("screen_b1024") if world == 1 else None,
                "traffic_source": "static: committed rocprofv3 PMC passes, not this run -- " + str(pmc_traffic("_source"))[:60].split(" : ")[0],
                # what the chip sustains on this kernel (committed SQ / GRBM pass over the same launch, profiles/pmc_traffic.json
                # `_sq_source`): the share of cycles the matrix pipes are busy, and the clock it holds meanwhile -- the datasheet
                # peak assumes 2400 MHz, so frac ~= mfma_busy x clock_MHz / 2400 (under the profiler the clock reads 2-3 % lower)
                "mfma_busy": pmc_traffic("screen_b1024_mfma_busy") if world == 1 else None,
                "clock_MHz": pmc_traffic("screen_b1024_clock_MHz") if world == 1 else None,
                "kernel_ms": round(ms_s, 4), "batch": BATCH, "docs_per_gpu": n_shard,
                "hbm_GBps_same_launch": round((-(-BATCH // 512) * n_shard * DIM * 2) / ms_s / 1e6, 1),
                "exact_fallback_tiles": flags}
        ms, ms_br = kernel_only_ms(q, docs, kp, iters=3, warm=1)
        roof_f32 = {"bound": "mfma", "kernel": "score_topk_kernel<8,*,false> (fp32 MFMA 32x32x2), same batch",
                    "achieved": round(flops / ms / 1e9, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(flops / ms / 1e9 / MFMA_F32_PEAK_TFLOPS, 4), "traffic": pmc_traffic("b1024") if world == 1 else None,
                    "kernel_ms": round(ms, 4), "with_sample_pass_ms": round(ms_br, 4), "batch": BATCH,
                    "docs_per_gpu": n_shard, "qps": round(BATCH / ms_br * 1e3, 1),
                    "pace_timeouts_last_launch": getattr(kernel_only_ms, "pace_timeouts", None)}
        qb = q[:32].contiguous()
        ms32, ms32_br = kernel_only_ms(qb, docs, TOPK)
        byts = n_shard * DIM * 4 + 32 * DIM * 4 + 32 * TOPK * 12
        roof_hbm_f32 = {"bound": "hbm", "kernel": "score_topk_kernel<8,64,false> (fp32 MFMA 32x32x2, B=32)",
                        "achieved": round(byts / ms32 / 1e6, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": round(byts / ms32 / 1e6 / HBM_PEAK_GBPS, 4),
                        "traffic": pmc_traffic("b32") if world == 1 else None,
                        "kernel_ms": round(ms32, 4), "with_sample_pass_ms": round(ms32_br, 4), "batch": 32,
                        "docs_per_gpu": n_shard, "qps": round(32 / ms32_br * 1e3, 1)}
        # the serving-size batch through the index: streaming form of the screen (fp16 shadow corpus, N x 512 B)
        ms32s = screen_kernel_ms(local_index, qb, TOPK)
        t32 = time_search(local_index, qb, TOPK)
        byts_s = n_shard * DIM * 2 + 32 * DIM * 4
        roof_hbm = {"bound": "hbm", "kernel": "screen_stream_kernel<false> (f16 MFMA 16x16x32, B=32, fp16 shadow corpus)",
                    "achieved": round(byts_s / ms32s / 1e6, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(byts_s / ms32s / 1e6 / HBM_PEAK_GBPS, 4),
                    "traffic": pmc_traffic("stream_b32") if world == 1 else None,
                    "kernel_ms": round(ms32s, 4), "search_ms": round(t32, 4), "batch": 32,
                    "docs_per_gpu": n_shard, "qps": round(32 / t32 * 1e3, 1)}
        line = {
            "metric": "queries/sec top-k over 10M x 256-d docs", "value": round(BATCH * a.steps / dt, 2),
            "unit": "queries/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32 results (bit-identical to the fp32 FMA chain); f16-MFMA screen + fp32 rescoring",
            "data": "synthetic",
            "config": {"workload": f"exact cosine top-{TOPK} of B={BATCH} queries over {N_DOCS} x {DIM} fp32 unit-norm "
                                   f"passages resident in HBM (BASELINE configs[3]; configs[1] batch), row-sharded "
                                   f"over {world} GPU(s), screened path (f16-MFMA filter + exact fp32 rescoring, "
                                   f"bit-identical to the fp32 kernel)"
                                   + (f", union seed (all-gather of every rank's {TOPK} largest sample maxima per query), "
                                      f"per-shard lists of up to {SHARD_K} (the shard's documents above the global "
                                      f"threshold) + RCCL all-gather + merge" if world > 1 else ""),
                       "n_docs": N_DOCS, "dim": DIM, "batch": BATCH, "k": TOPK, "parallelism": f"rowshard{world}",
                       "collective": index.collective if world > 1 else None},
            "roofline": roof,
        }
        # every other leg rides INSIDE `roofline` (the driver's record keeps `roofline` and `cpu_baseline` whole):
        #   hbm_screen      north_star's ">= 70 % of HBM" regime: the streaming screen at B = 32 over the fp16 shadow corpus
        #   hbm_exact_f32   the plain fp32 kernel at B = 32 (N x 1024 bytes)
        #   mfma_exact_f32  the plain fp32 kernel on the bench batch (the strict-precision reading of the step)
        #   encoder_corpus  the same index over embeddings the encoder produces (anisotropic), with the filter's statistics
        #   clustered_corpus  ... over clustered rows with groups of exact duplicates: fallback count and the all-fallback rate
        #   index_build_from_strings  text -> tokeniser threads -> document tower, as a fraction of the GPU-only rate
        #   encoder, train  SURVEY 8d's secondary metrics (tower / index-build tokens/s, training triplets/s)
        roof["legs"] = {"hbm_screen": roof_hbm, "hbm_exact_f32": roof_hbm_f32, "mfma_exact_f32": roof_f32, **shared_legs}
        if world == 1 and not a.no_secondary:
            try:
                roof["legs"]["serve_b1"] = serve_b1_leg(dev, local_index)
            except Exception as e:  # noqa: BLE001 -- (sklearn missing, ...): say so in the line
                roof["legs"]["serve_b1"] = {"error": f"{type(e).__name__}: {e}"[:300]}
            torch.cuda.empty_cache()
            free_b, total_b = torch.cuda.mem_get_info(dev)
            if not a.no_encoder_corpus and free_b > 170e9:   # (153.6 GB + workspaces; a smaller card skips the leg)
                roof["legs"]["resident_100m"] = resident_100m_leg(dev)
        enc_inputs = None
        # N = 1 only: these legs build optimizers and trainers on rank 0 alone, and with a process group up every such object is
        # a COLLECTIVE construction (FusedClipAdam(group=None) means the default group, as DDP's does): rank 0 would wait for
        # peers that are already at the closing barrier.  (They are single-GPU figures anyway.)
        if not a.no_secondary and world == 1:
            del index, local_index
            torch.cuda.empty_cache()
            roof["legs"]["encoder"], roof["legs"]["train"], enc_inputs, model = encoder_legs(dev)
            if not a.no_encoder_corpus:
                roof["legs"]["encoder_corpus"] = encoder_corpus_leg(dev, model)
                roof["legs"]["index_build_from_strings"] = index_build_from_strings_leg(
                    dev, model, roof["legs"]["encoder"]["index_build_b8192"]["docs_per_s"])
            del model
            torch.cuda.empty_cache()
            if not a.no_encoder_corpus:
                roof["legs"]["clustered_corpus"] = clustered_corpus_leg(dev)
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(q, docs, enc_inputs)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
