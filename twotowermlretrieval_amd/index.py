"""Exact brute-force cosine/dot top-k over a resident document-embedding matrix.

Drop-in for the reference's scoring idiom

    sim = torch.matmul(query_emb, doc_embeddings.t()); torch.topk(sim, k)
    (backend/evaluators.py:185-186, :269-272; backend/trainer.py:62-65)

with the same return convention as torch.topk: (values [B,k] f32 descending,
indices [B,k] int64).  The [B,N] score matrix is never materialised.  Ties are
DEFINED (score desc, index asc); torch.topk leaves them unspecified.

Everything here runs through libtt.so (hand-written HIP, gfx950).  There is no
PyTorch/CPU fallback: inputs must be CUDA(ROCm) tensors.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch

from . import _lib

__all__ = ["GraphedSearch", "score_topk", "topk_merge", "score_rank", "score_all", "BruteForceIndex", "ShardedIndex", "PendingSearch", "StreamedIndex",
           "shard_bounds", "seed_union"]


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def _need_cuda(*ts: torch.Tensor) -> None:
    for t in ts:
        if not t.is_cuda:
            raise RuntimeError("twotowermlretrieval_amd: this path runs only on an AMD GPU via libtt.so; "
                               f"got a {t.device} tensor (no CPU fallback exists)")


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise TypeError(f"expected float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def score_topk(q: torch.Tensor, docs: torch.Tensor, k: int, idx_offset: int = 0,
               workspace: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """topk(q @ docs.T, k) fused.  q [B,d] or [d]; docs [N,d] (row i <-> document i)."""
    squeeze = q.dim() == 1
    if squeeze:
        q = q.unsqueeze(0)
    _need_cuda(q, docs)
    q, docs = _f32c(q), _f32c(docs)
    B, d = q.shape
    N = docs.shape[0]
    if docs.dim() != 2 or docs.shape[1] != d:
        raise ValueError(f"shape mismatch: q {tuple(q.shape)} vs docs {tuple(docs.shape)}")
    L = _lib.lib()
    vals = torch.empty((B, k), dtype=torch.float32, device=q.device)
    idx = torch.empty((B, k), dtype=torch.int64, device=q.device)
    need = L.tt_score_topk_workspace_bytes(B, N, d, k)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(max(need, 16), dtype=torch.uint8, device=q.device)
    with torch.cuda.device(q.device):
        _lib.check(L.tt_score_topk_f32(q.data_ptr(), B, d, docs.data_ptr(), N, k, idx_offset, vals.data_ptr(),
                                       idx.data_ptr(), workspace.data_ptr(), workspace.numel(), _stream(q)))
    return (vals[0], idx[0]) if squeeze else (vals, idx)


def topk_merge(vals: torch.Tensor, idx: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Top-k of [B,M] unordered candidates (idx < 0 = padding); (score desc, index asc)."""
    _need_cuda(vals, idx)
    vals = _f32c(vals)
    idx = idx.contiguous()
    if idx.dtype != torch.int64 or vals.shape != idx.shape or vals.dim() != 2:
        raise ValueError("topk_merge wants vals f32 [B,M] and idx int64 [B,M]")
    B, M = vals.shape
    ov = torch.empty((B, k), dtype=torch.float32, device=vals.device)
    oi = torch.empty((B, k), dtype=torch.int64, device=vals.device)
    with torch.cuda.device(vals.device):
        _lib.check(_lib.lib().tt_topk_merge(vals.data_ptr(), idx.data_ptr(), B, M, k, ov.data_ptr(), oi.data_ptr(),
                                            _stream(vals)))
    return ov, oi


def score_rank(q: torch.Tensor, docs: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """1-based rank of docs[target[b]] for query b (BatchEvaluator's sort + nonzero, evaluators.py:58-65)."""
    _need_cuda(q, docs, target)
    q, docs = _f32c(q), _f32c(docs)
    target = target.to(torch.int64).contiguous()
    B, d = q.shape
    if ((target < 0) | (target >= docs.shape[0])).any():
        raise IndexError("score_rank: target out of range")
    rank = torch.empty(B, dtype=torch.int64, device=q.device)
    with torch.cuda.device(q.device):
        _lib.check(_lib.lib().tt_score_rank_f32(q.data_ptr(), B, d, docs.data_ptr(), docs.shape[0], target.data_ptr(),
                                                rank.data_ptr(), _stream(q)))
    return rank


def score_all(q: torch.Tensor, docs: torch.Tensor) -> torch.Tensor:
    """q @ docs.T as a [B,N] matrix (same fp32 FMA chain as score_topk's scores): for callers that blend the dense score
    of EVERY document with another signal (backend/simple_hybrid.py:53-56).  Small corpora only: it materialises B*N."""
    squeeze = q.dim() == 1
    if squeeze:
        q = q.unsqueeze(0)
    _need_cuda(q, docs)
    q, docs = _f32c(q), _f32c(docs)
    if docs.dim() != 2 or docs.shape[1] != q.shape[1]:
        raise ValueError(f"shape mismatch: q {tuple(q.shape)} vs docs {tuple(docs.shape)}")
    out = torch.empty((q.shape[0], docs.shape[0]), dtype=torch.float32, device=q.device)
    with torch.cuda.device(q.device):
        _lib.check(_lib.lib().tt_score_all_f32(q.data_ptr(), q.shape[0], q.shape[1], docs.data_ptr(), docs.shape[0],
                                               out.data_ptr(), _stream(q)))
    return out[0] if squeeze else out


SCREEN_MIN_BATCH = 1     # the screened path wins at every batch size once the corpus is large enough to sample:
                         # B <= 64 streaming form (half the bytes of the fp32 kernel), above it the shared-tile form
SCREEN_PADDED_MIN_BATCH = 33  # d < 256 (zero-padded screen copy): only where the exact kernel is MFMA-bound
SCREEN_MIN_DOCS = 65536  # below this there is no sample pass to seed thresholds and the exact kernel is faster


class BruteForceIndex:
    """A [N,d] fp32 document matrix resident in HBM (document_embeddings.npy layout,
    backend/main.py:125-138: row i <-> documents[i]) with exact top-k search.

    screen=True additionally keeps an fp16 shadow copy (N*d*2 bytes) so that large query batches
    run the screened path (fp16 MFMA filter + exact fp32 rescoring, tt_score_topk_screened_f32):
    same bit-exact result, an order of magnitude more queries/s than the fp32-MFMA-bound kernel.
    """

    def __init__(self, doc_embeddings: torch.Tensor, idx_offset: int = 0, screen: bool = False):
        _need_cuda(doc_embeddings)
        self.docs = _f32c(doc_embeddings)
        self.idx_offset = int(idx_offset)
        self.docs16: Optional[torch.Tensor] = None
        self.dmax_norm = float("nan")
        self.fallback_flags = torch.zeros(1, dtype=torch.int32, device=self.docs.device)  # per 32-query tile
        self.keep_stats = False   # True: the most recent screened search's workspace is kept for search_stats()
        self._last_ws = None
        N, d = self.docs.shape
        self._sdocs = self.docs  # what the screened path scores against: [N,256] fp32
        if screen and N > 0 and (d == 256 or (d < 256 and d % 4 == 0)):
            L = _lib.lib()
            if d < 256:
                # narrower embeddings (HIDDEN_DIM 64, 128, ...): zero-padded to the screen kernels' 256 features.
                # Padding adds fmaf(0, 0, acc) terms to the fp32 chain, which leave every score bit-identical; the
                # padded copy costs N KiB and is used for batches above 32 queries (MFMA-bound), smaller batches
                # stream the original rows through the exact kernel, which already moves only N*d*4 bytes.
                self._sdocs = torch.zeros((N, 256), dtype=torch.float32, device=self.docs.device)
                self._sdocs[:, :d] = self.docs
            self.docs16 = torch.empty((N, 256), dtype=torch.float16, device=self.docs.device)
            stats = torch.zeros(2, dtype=torch.float32, device=self.docs.device)
            with torch.cuda.device(self.docs.device):
                _lib.check(L.tt_index_build_f16(self._sdocs.data_ptr(), N, 256, self.docs16.data_ptr(), stats.data_ptr(),
                                                _stream(self.docs)))
            dmax, amax = (float(x) for x in stats.tolist())  # one sync, at index-build time
            if dmax == dmax and amax < 6.0e4 and dmax < 6.0e4:
                self.dmax_norm = dmax
            else:
                self.docs16 = None  # outside the fp16 range: exact kernel only

    @classmethod
    def _from_buffers(cls, docs32: torch.Tensor, docs16: Optional[torch.Tensor], dmax_norm: float, idx_offset: int):
        """An index over caller-managed device buffers (StreamedIndex's per-block view)."""
        self = cls.__new__(cls)
        self.docs, self.docs16, self.dmax_norm, self.idx_offset = docs32, docs16, float(dmax_norm), int(idx_offset)
        self._sdocs = docs32
        self.fallback_flags = torch.zeros(1, dtype=torch.int32, device=docs32.device)
        self.keep_stats, self._last_ws = False, None
        return self

    @property
    def ntotal(self) -> int:
        return self.docs.shape[0]

    @property
    def device(self) -> torch.device:
        return self.docs.device

    def search_stats(self) -> Optional[torch.Tensor]:
        """int32 [B,2] = (pooled candidates, survivors rescored exactly) per query of the most recent screened search made
        with keep_stats = True, or None.  Diagnostic: what the fp16 filter let through on this corpus."""
        if self._last_ws is None:
            return None
        ws, B, k = self._last_ws
        with torch.cuda.device(self.docs.device):
            off = _lib.lib().tt_score_topk_screened_stats_offset(B, self.docs.shape[0], 256, k)
        return ws[off:off + 8 * B].view(torch.int32).view(B, 2).clone()

    def search(self, q: torch.Tensor, k: int = 10, _prof_events=None, out=None, _seed_union=None,
               _k_seed: int = 0, _k_list: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
        """out: optional (vals f32 [B,k], idx int64 [B,k]) contiguous device tensors to write into (2-D q only).
        _seed_union (ShardedIndex): a callable that turns this shard's seed list [B, _k_seed] f32 (its _k_seed largest
        sample maxima per query, tt_score_topk_screened_seed_list_f32) into the seed thresholds [B] f32 -- the _k_seed-th
        largest of the UNION of the ranks' lists (one all-gather + tt_seed_union_f32) -- on the current stream; the screen
        then runs with that global seed and `out` holds this shard's documents above it."""
        B = 1 if q.dim() == 1 else q.shape[0]
        N, d = self.docs.shape
        L = _lib.lib()
        _need_cuda(q)
        if q.device != self.docs.device:
            raise ValueError(f"queries on {q.device} but the index lives on {self.docs.device}")
        if q.shape[-1] != d:
            raise ValueError(f"shape mismatch: q {tuple(q.shape)} vs docs {tuple(self.docs.shape)}")
        if (self.docs16 is not None and B >= (SCREEN_MIN_BATCH if d == 256 else SCREEN_PADDED_MIN_BATCH)
                and N >= SCREEN_MIN_DOCS and k <= 64):
            _need_cuda(q)
            if q.dim() == 1:  # single query (QueryInferencer / hybrid rerank): same path, squeezed result
                vals, idx = self.search(q.unsqueeze(0), k, _prof_events)
                return vals[0], idx[0]
            q = _f32c(q)
            if d < 256:
                qp = torch.zeros((B, 256), dtype=torch.float32, device=q.device)
                qp[:, :d] = q
                q, d = qp, 256
            if out is not None:
                vals, idx = out
            else:
                vals = torch.empty((B, k), dtype=torch.float32, device=q.device)
                idx = torch.empty((B, k), dtype=torch.int64, device=q.device)
            with torch.cuda.device(self.docs.device):  # workspace sizing depends on the device's CU count
                # per-call workspace and flags (cached allocator blocks): safe for concurrent callers and streams
                need = L.tt_score_topk_screened_workspace_bytes(B, N, d, k)
                ws_s = torch.empty(need, dtype=torch.uint8, device=self.docs.device)
                flags = torch.empty((B + 31) // 32, dtype=torch.int32, device=self.docs.device)
                if self.keep_stats:
                    self._last_ws = (ws_s, B, k)
                if _seed_union is not None:
                    ks = min(_k_list or _k_seed or k, k)  # entries per seed list (the caller ranks the union)
                    lst = torch.empty((B, ks), dtype=torch.float32, device=self.docs.device)
                    _lib.check(L.tt_score_topk_screened_seed_list_f32(q.data_ptr(), B, d, self.docs16.data_ptr(), N, k, ks,
                                                                      self.dmax_norm, flags.data_ptr(), lst.data_ptr(),
                                                                      ws_s.data_ptr(), ws_s.numel(), _stream(q)))
                    seed = _seed_union(lst)
                    if seed.shape != (B,) or seed.dtype != torch.float32 or not seed.is_contiguous():
                        raise ValueError("_seed_union must return a contiguous float32 [B] tensor")
                    _lib.check(L.tt_score_topk_screened_seeded_f32(q.data_ptr(), B, d, self._sdocs.data_ptr(),
                                                                   self.docs16.data_ptr(), N, k, self.dmax_norm,
                                                                   self.idx_offset, vals.data_ptr(), idx.data_ptr(),
                                                                   flags.data_ptr(), seed.data_ptr(), ws_s.data_ptr(),
                                                                   ws_s.numel(), _prof_events, _stream(q)))
                    self.fallback_flags = flags
                    return vals, idx
                _lib.check(L.tt_score_topk_screened_f32(q.data_ptr(), B, d, self._sdocs.data_ptr(), self.docs16.data_ptr(),
                                                        N, k, self.dmax_norm, self.idx_offset, vals.data_ptr(),
                                                        idx.data_ptr(), flags.data_ptr(), ws_s.data_ptr(), ws_s.numel(),
                                                        _prof_events, _stream(q)))
            self.fallback_flags = flags  # of the most recent search (per 32-query tile; non-zero = exact kernel took over)
            return vals, idx
        with torch.cuda.device(self.docs.device):
            need = L.tt_score_topk_workspace_bytes(B, N, d, k)
        ws = torch.empty(max(need, 16), dtype=torch.uint8, device=self.docs.device)
        v, i = score_topk(q, self.docs, k, self.idx_offset, ws)
        if out is not None:
            out[0].copy_(v)
            out[1].copy_(i)
            return out
        return v, i


class GraphedSearch:
    """One search of a fixed shape (B, k) captured in a HIP graph and replayed: the nine launches of a
    screened search (flag reset, sample pass, threshold select, screen, finish, predicated exact kernels)
    become one graph launch, which matters at serving sizes where the whole search is < 1 ms.
    Everything in the library is asynchronous on the caller's stream with caller-owned memory, so plain
    stream capture works; queries are copied into a static buffer, results are returned in static buffers
    (valid until the next call)."""

    def __init__(self, index: "BruteForceIndex", batch: int, k: int = 10):
        self.index, self.B, self.k = index, int(batch), int(k)
        dev = index.docs.device
        d = index.docs.shape[1]
        self.q = torch.zeros((self.B, d), dtype=torch.float32, device=dev)
        self.vals = torch.empty((self.B, self.k), dtype=torch.float32, device=dev)
        self.idx = torch.empty((self.B, self.k), dtype=torch.int64, device=dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):  # warm-up outside capture: workspaces get allocated, kernels loaded
            for _ in range(2):
                index.search(self.q, self.k, out=(self.vals, self.idx))
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            index.search(self.q, self.k, out=(self.vals, self.idx))

    def __call__(self, q: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        if tuple(q.shape) != tuple(self.q.shape):
            raise ValueError(f"GraphedSearch was captured for queries of shape {tuple(self.q.shape)}, got {tuple(q.shape)}")
        self.q.copy_(q)
        self.graph.replay()
        return self.vals, self.idx


def shard_bounds(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous row shard of rank r: [r*N/W, (r+1)*N/W) with the remainder spread over the
    first ranks."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _exchange_and_merge(vals: torch.Tensor, idx: torch.Tensor, k: int, merge: Callable, group=None):
    """Host logic of the sharded search on ordinary tensors: pack this rank's [B,kp] lists into one byte block,
    ONE all_gather_into_tensor, unpack to [B, world*kp] candidates, merge.  Private: the CPU tests drive it over
    gloo with the oracle's search and merge; the product path is ShardedIndex below, which exchanges in place."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    B, kp = vals.shape
    packed = torch.cat([vals.contiguous().view(torch.uint8).reshape(-1), idx.contiguous().view(torch.uint8).reshape(-1)])
    out = torch.empty(world * packed.numel(), dtype=torch.uint8, device=packed.device)
    dist.all_gather_into_tensor(out, packed, group=group)
    out = out.view(world, -1)
    nv = B * kp * 4
    gv = out[:, :nv].contiguous().view(torch.float32).view(world, B, kp)
    gi = out[:, nv:].contiguous().view(torch.int64).view(world, B, kp)
    gv = gv.permute(1, 0, 2).reshape(B, world * kp).contiguous()
    gi = gi.permute(1, 0, 2).reshape(B, world * kp).contiguous()
    return merge(gv, gi, k)


SEED_UNION_MAX = 512  # values per query tt_seed_union_f32 ranks (one LDS tile)


def seed_union(lists: torch.Tensor, world: int, kth: Optional[int] = None) -> torch.Tensor:
    """lists [world, B, ks] f32 (every rank's ks largest sample maxima per query) -> seed [B]: the kth-th (default ks-th)
    largest of each query's world * ks values (tt_seed_union_f32), on the current stream."""
    world_, B, ks = lists.shape
    assert world_ == world and lists.is_contiguous() and lists.dtype == torch.float32
    seed = torch.empty(B, dtype=torch.float32, device=lists.device)
    with torch.cuda.device(lists.device):
        _lib.check(_lib.lib().tt_seed_union_f32(lists.data_ptr(), world, B, ks, ks if kth is None else int(kth), seed.data_ptr(),
                                                _stream(lists)))
    return seed


def seed_plan(world: int, k: int, exchange: bool = True):
    """(list length per rank, rank taken from the union) of a sharded search's seed exchange, or None when the union does not
    apply: world * list length is bounded by tt_seed_union_f32's tile (SEED_UNION_MAX values per query); a job too wide for k
    values per rank lists fewer -- the k-th of the union is still reached by k distinct documents -- and one too wide even for
    that (or a k beyond the screen's 64) seeds every shard by itself."""
    if not exchange or k > 64 or world < 1:
        return None
    ks = min(k, SEED_UNION_MAX // world)
    if ks < 1 or world * ks < k:
        return None
    return ks, k


def _local_seed(lst: torch.Tensor) -> torch.Tensor:
    """No exchange: the union over one shard is that shard's own ks-th largest sample maximum."""
    return seed_union(lst.unsqueeze(0), 1)


class _Slot:
    """One set of exchange buffers of a ShardedIndex (three per shape: two that submit() alternates between -- a step's
    all-gather + merge overlaps the next step's search -- and one of search()'s own): send block [vals f32 [B,kp] |
    idx int64 [B,kp]], receive buffer of `world` such blocks, outputs."""

    def __init__(self, B: int, kp: int, k: int, world: int, dev):
        self.nv = (B * kp * 4 + 7) // 8 * 8           # idx block 8-byte aligned
        self.stride = self.nv + B * kp * 8
        self.send = torch.empty(self.stride, dtype=torch.uint8, device=dev)
        self.recv = torch.empty(world * self.stride, dtype=torch.uint8, device=dev)
        self.send_v = self.send[:B * kp * 4].view(torch.float32).view(B, kp)
        self.send_i = self.send[self.nv:].view(torch.int64).view(B, kp)
        self.out_v = torch.empty((B, k), dtype=torch.float32, device=dev)
        self.out_i = torch.empty((B, k), dtype=torch.int64, device=dev)
        self.sampled = torch.cuda.Event()   # caller's stream: the seed list is written
        self.seeded = torch.cuda.Event()    # exchange stream: the ranks' seed lists have been gathered
        self.searched = torch.cuda.Event()  # caller's stream: the send block is written
        self.merged = torch.cuda.Event()    # exchange stream: outputs are valid, send / receive buffers free

    def tensors(self):
        return (self.send, self.recv, self.out_v, self.out_i)


class PendingSearch:
    """Result of ShardedIndex.submit(): result() makes the CURRENT stream wait for the exchange + merge (no host
    synchronisation) and returns (values, indices) views valid until two more submits of the same shape on the same
    index (search() has a slot of its own and never overwrites them)."""

    def __init__(self, slot: _Slot, index: "ShardedIndex"):
        self._slot, self._index = slot, index

    def result(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """(A COLLECTIVE when this step's exchange has not been issued yet -- it normally goes out inside the NEXT submit():
        every rank must then call result() at the same point of its sequence of submit / search / result calls.)"""
        self._index._flush(self._slot)
        torch.cuda.current_stream(self._slot.out_v.device).wait_event(self._slot.merged)
        return self._slot.out_v, self._slot.out_i


class ShardedIndex:
    """Row-sharded corpus: every rank holds rows [lo,hi) and the SAME queries; search =
    local top-k' (global indices) -> one all-gather (RCCL over xGMI) -> merge on every rank.  Result is identical on
    all ranks and identical to the single-GPU search for any world size (the global top-k is a subset of the union of
    the shard top-k's).

    No glue kernels: the local search writes its [B,k'] lists straight into this rank's send block, ONE all-gather
    (tt_allgather_topk on torch.distributed's RCCL communicator; torch.distributed's own call for other backends),
    and the merge kernel reads the receive buffer in place (tt_topk_merge_shards).

    submit() pipelines consecutive searches on two streams, and keeps the collectives off the search's critical path.  One
    steady-state step (caller's stream C, exchange stream X; AG = all-gather on the one communicator):

        C: sample(i+1) ──sampled──┐                      ┌──seeded── union + screen(i+1) + rescoring ──searched(i+1)──> ...
        X:                        └─ seedAG(i+1) ────────┴─ listAG(i) ─ merge(i) ──merged(i)──> result(i)

    screen(i+1) waits for seedAG(i+1) only (k floats per query and rank: 40 KB per rank at B = 1024, k = 10); step i's list
    exchange (410-614 KB per rank) and merge are ISSUED BEHIND it and run under screen(i+1).  (Round 3 issued listAG(i) + merge(i)
    at the end of submit(i), in front of seedAG(i+1) on the same stream: the screen then waited for the whole previous exchange.)
    Every rank issues the same collectives in the same order -- seedAG(1), seedAG(2), listAG(1), seedAG(3), listAG(2), ... -- as
    long as all ranks make the same sequence of submit / result / search calls, which a sharded search requires anyway; a
    result() or search() that finds the last step's exchange still unissued issues it first, at the same point on every rank.

    Whether the seed exchange happens at all is agreed ONCE, in the constructor (a collective): every rank has an fp16 shadow
    copy and at least SCREEN_MIN_DOCS rows, or no rank exchanges seeds -- a rank-local decision (shard sizes straddling the
    minimum, one shard outside the fp16 range) would leave some ranks in an all-gather the others never enter."""

    def __init__(self, local_docs, row_offset: int, group=None, shard_k: int = 50, screen: bool = False,
                 comm=None, block_docs: int = 1 << 20, device=None):
        """local_docs: this rank's rows [row_offset, row_offset + n) of the corpus, as
          * a device fp32 [n,d] tensor: resident shard (BruteForceIndex; BASELINE configs[3]), or
          * a CPU bfloat16 [n,d] tensor / a StreamedIndex: the shard stays in pinned host DRAM and every search streams it
            through the GPU in blocks of block_docs rows (BASELINE configs[4]: 100M x 256 bf16 over 8 GPUs = 12.5M rows,
            6.4 GB per rank, PCIe-bound).  The streamed shard's list is the exact top-k' of its rows (bf16 -> fp32 is exact),
            so the exchange and the merge are the resident path's, unchanged.
        A job may mix the two kinds (a rank that has the HBM keeps its shard resident).  The seed exchange is a property of the
        whole job: with ANY streamed shard no rank exchanges seeds.  (A streamed shard searches block by block, each block
        seeded by its own sample pass, which is valid on its own; a union seed would need the sample maxima of the whole
        shard before the first block is screened, i.e. one more pass over PCIe, which is what binds.)"""
        from .collective import Collective
        self.group = group
        self.row_offset = int(row_offset)
        self.shard_k = int(shard_k)
        if isinstance(local_docs, StreamedIndex):
            if local_docs.idx_offset != self.row_offset:
                raise ValueError(f"the StreamedIndex numbers its rows from {local_docs.idx_offset}, the shard starts at {row_offset}")
            self._index = local_docs
        elif not local_docs.is_cuda and local_docs.dtype == torch.bfloat16:
            self._index = StreamedIndex(local_docs, block_docs=block_docs, device=device, idx_offset=row_offset, screen=True)
        else:
            self._index = BruteForceIndex(local_docs, idx_offset=row_offset, screen=screen)
        self.streamed = isinstance(self._index, StreamedIndex)
        self._dev = self._index.device
        self._coll = Collective(group, self._dev, comm=comm)
        self._slots = {}
        self._n_submitted = 0
        self._xs: Optional[torch.cuda.Stream] = None
        self._deferred = None  # (slot, B, kp, k) of the last submit(): its list exchange goes out behind the next seed gather
        if self.streamed:
            mine = 0
        else:
            N, d = self._index.docs.shape
            mine = int(self._index.docs16 is not None and d == 256 and N >= SCREEN_MIN_DOCS)
        if self._coll.world > 1:
            dev = self._dev
            send = torch.tensor([mine], dtype=torch.int64, device=dev)
            recv = torch.empty(self._coll.world, dtype=torch.int64, device=dev)
            self._coll.all_gather_blocks(send.view(torch.uint8), recv.view(torch.uint8))
            mine = int(recv.min().item())
        self._seed_exchange = bool(mine)  # the same on every rank

    def _seed_plan(self, k: int):
        return seed_plan(self._coll.world, k, self._seed_exchange)

    def _local_search(self, q: torch.Tensor, kp: int, k: int, sl: "_Slot", comm_stream=None) -> None:
        """This shard's list for the exchange: up to kp = max(k, shard_k) entries, best first.  The screen is seeded for
        the FINAL k with the UNION seed: every rank lists its k largest sample maxima per query, ONE small all-gather
        (k floats per query and rank: 40 KB per rank at B = 1024), and seed[q] = the k-th largest of the union -- k distinct
        documents of the whole corpus reach it, so it bounds the global k-th score from below as well as the unsharded
        search's own sample would (a shard's own k-th sample maximum is much weaker, and on a small shard the candidates
        that get through, not the matrix pipes, set the screen's pace: 1.25M rows, emulated 8 shards, 0.640 -> 0.582 ms).
        The shard then owes the exchange only its documents above that global threshold: entries beyond them are padding
        (-inf / -1), which the merge ignores; the merged top-k is exact (two real-kernel ranks vs the oracle,
        tests/test_multirank_gpu.py).
        comm_stream: the stream the seed all-gather is issued on (submit(): the index's exchange stream; the previous step's
        deferred list exchange is issued right behind it); None = the caller's."""
        coll, world = self._coll, self._coll.world
        plan = self._seed_plan(k)
        B = q.shape[0]
        padded = (not self.streamed) and self._index.docs.shape[1] != 256
        screens = plan is not None and B >= (SCREEN_MIN_BATCH if not padded else SCREEN_PADDED_MIN_BATCH)  # (BruteForceIndex.search's rule)
        if not screens or (world == 1 and kp == k):
            # no seed exchange on any rank (agreed in the constructor; B and k are the same everywhere): the shard's own search
            if comm_stream is not None:
                self._flush()
            self._index.search(q, kp, out=(sl.send_v, sl.send_i))
            return
        ks, kth = plan

        def union(lst: torch.Tensor) -> torch.Tensor:
            if world == 1:
                return seed_union(lst.unsqueeze(0), 1, kth)
            recv = torch.empty((world,) + tuple(lst.shape), dtype=torch.float32, device=lst.device)
            send_b, recv_b = lst.view(-1).view(torch.uint8), recv.view(-1).view(torch.uint8)
            if comm_stream is None:
                coll.all_gather_blocks(send_b, recv_b)
            else:
                cur = torch.cuda.current_stream(lst.device)
                sl.sampled.record(cur)
                with torch.cuda.stream(comm_stream):
                    comm_stream.wait_event(sl.sampled)
                    lst.record_stream(comm_stream)
                    recv.record_stream(comm_stream)
                    coll.all_gather_blocks(send_b, recv_b)
                    sl.seeded.record(comm_stream)
                self._flush()                # the PREVIOUS step's list exchange + merge: behind this step's seed gather
                cur.wait_event(sl.seeded)    # (the event, not the stream: the screen does not wait for that exchange)
            return seed_union(recv, world, kth)

        self._index.search(q, kp, out=(sl.send_v, sl.send_i), _seed_union=union, _k_seed=k, _k_list=ks)

    @property
    def collective(self) -> str:
        """Which transport the exchange uses (reported by bench.py)."""
        return self._coll.via

    @classmethod
    def from_global(cls, docs: torch.Tensor, group=None, **kw) -> "ShardedIndex":
        import torch.distributed as dist
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        lo, hi = shard_bounds(docs.shape[0], rank, world)
        return cls(docs[lo:hi], lo, group=group, **kw)

    @classmethod
    def from_host_bf16(cls, host_docs: torch.Tensor, group=None, **kw) -> "ShardedIndex":
        """BASELINE configs[4]: `host_docs` is the WHOLE bf16 corpus [N,d] in host memory (an np.memmap-backed tensor will do:
        only this rank's rows are touched); rank r pins and streams rows [r*N/W, (r+1)*N/W) only."""
        import torch.distributed as dist
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        lo, hi = shard_bounds(host_docs.shape[0], rank, world)
        return cls(host_docs[lo:hi], lo, group=group, **kw)

    @classmethod
    def from_documents(cls, model, tokenizer, documents, device, group=None, **kw) -> "ShardedIndex":
        """Index build across ranks (third row: embarrassingly parallel over documents, no collective): every rank
        holds the same document list (documents.pkl order, backend/main.py:134-136), embeds ONLY its contiguous shard
        [lo, hi) with the document tower and keeps those rows; row i of the global index is documents[i] on every rank."""
        import torch.distributed as dist
        from .evaluators import embed_corpus
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        lo, hi = shard_bounds(len(documents), rank, world)
        model.eval()
        with torch.no_grad():
            local = embed_corpus(model, tokenizer, documents[lo:hi], device)
        return cls(local, lo, group=group, **kw)

    _MAX_SLOT_SHAPES = 8

    def _slot(self, B: int, kp: int, k: int, which: int) -> _Slot:
        """which: 0 / 1 = the two slots submit() alternates between, 2 = search()'s own (a search() between a submit() and
        its .result() must not overwrite that PendingSearch's outputs).  Slot sets are kept per (B, kp, k) shape; when
        more than _MAX_SLOT_SHAPES shapes have been seen the oldest set is retired, after its queued exchanges have
        completed (its buffers are used on the exchange stream, which the caching allocator does not know about)."""
        key = (B, kp, k, self._coll.world)
        if key not in self._slots:
            while len(self._slots) >= self._MAX_SLOT_SHAPES:
                old = self._slots.pop(next(iter(self._slots)))
                for sl in old:
                    sl.merged.synchronize()
            self._slots[key] = [_Slot(B, kp, k, self._coll.world, self._dev) for _ in range(3)]
        return self._slots[key][which]

    def _flush(self, only: Optional[_Slot] = None) -> None:
        """Issue the deferred list exchange + merge of the last submit() on the exchange stream (only: just if it is that
        slot's)."""
        d = self._deferred
        if d is None or (only is not None and d[0] is not only):
            return
        self._deferred = None
        sl, B, kp, k = d
        with torch.cuda.stream(self._xs):
            self._xs.wait_event(sl.searched)
            for t in sl.tensors():           # allocated on the caller's stream, used on the exchange stream
                t.record_stream(self._xs)
            self._exchange_merge(sl, B, kp, k)
            sl.merged.record(self._xs)

    def _exchange_merge(self, sl: _Slot, B: int, kp: int, k: int) -> None:
        """all-gather + in-place merge of one slot on the CURRENT stream."""
        self._coll.all_gather_blocks(sl.send, sl.recv)
        with torch.cuda.device(sl.recv.device):
            _lib.check(_lib.lib().tt_topk_merge_shards(sl.recv.data_ptr(), self._coll.world, sl.stride, sl.nv, B, kp, k,
                                                       sl.out_v.data_ptr(), sl.out_i.data_ptr(), _stream(sl.recv)))

    def search(self, q: torch.Tensor, k: int = 10) -> Tuple[torch.Tensor, torch.Tensor]:
        """One search, everything on the caller's stream; fresh result tensors."""
        if q.dim() == 1:  # a single query vector: same path, squeezed result
            v, i = self.search(q.unsqueeze(0), k)
            return v[0], i[0]
        kp = max(k, self.shard_k)
        sl = self._slot(q.shape[0], kp, k, 2)
        cur = torch.cuda.current_stream(sl.send.device)
        cur.wait_event(sl.merged)  # (a search() on another stream may still own the slot)
        self._flush()              # (a submitted step's exchange goes out first: one issue order on every rank)
        self._local_search(q, kp, k, sl)
        self._exchange_merge(sl, q.shape[0], kp, k)
        sl.merged.record(cur)
        return sl.out_v.clone(), sl.out_i.clone()

    def submit(self, q: torch.Tensor, k: int = 10) -> PendingSearch:
        """Pipelined search of a [B,d] batch: the local search is enqueued on the caller's stream now; its list exchange and
        merge go out on this index's second stream inside the NEXT submit() (behind that step's seed gather), or when
        .result() / search() asks for them.  Call .result() when the answer is needed."""
        if q.dim() != 2:
            raise ValueError("submit wants a [B,d] batch")
        kp = max(k, self.shard_k)
        B = q.shape[0]
        sl = self._slot(B, kp, k, self._n_submitted % 2)
        self._n_submitted += 1
        dev = sl.send.device
        if self._xs is None:
            self._xs = torch.cuda.Stream(device=dev)
        cur = torch.cuda.current_stream(dev)
        self._flush(sl)                      # (the slot's own previous step, if nobody collected it: issue before reuse)
        cur.wait_event(sl.merged)            # the slot's previous exchange has read its send block
        self._local_search(q, kp, k, sl, comm_stream=self._xs)
        self._flush()                        # (a local search without a seed exchange has not issued the previous step's yet)
        sl.searched.record(cur)
        self._deferred = (sl, B, kp, k)
        return PendingSearch(sl, self)


class StreamedIndex:
    """BASELINE configs[4]: a corpus kept as bf16 rows in (pinned) host DRAM and streamed through the GPU.

    search() walks the corpus in blocks: a copy stream moves block i+1 host -> device (hipMemcpyAsync from
    pinned memory) while the compute stream widens block i to fp32 (+ fp16 shadow), searches it with the
    resident-corpus kernels (idx_offset = block start) and folds the block's top-k into the running top-k
    with the merge kernel.  Two staging buffers; events order the two streams.  The result is the exact
    top-k over the bf16 corpus (bf16 -> fp32 is exact), with the usual (score desc, index asc) order.
    PCIe-bound by design: ~55-60 GB/s on Gen5 x16, i.e. ~110 ms per pass over a 6.4 GB shard.
    """

    def __init__(self, host_docs: torch.Tensor, block_docs: int = 1 << 20, device=None, idx_offset: int = 0,
                 screen: bool = True):
        if host_docs.is_cuda or host_docs.dtype != torch.bfloat16 or host_docs.dim() != 2:
            raise ValueError("StreamedIndex wants a CPU bfloat16 [N,d] tensor (pinned for full PCIe speed)")
        self.host = host_docs if (host_docs.shape[0] == 0 or host_docs.is_pinned()) else host_docs.pin_memory()
        self.N, self.d = self.host.shape
        self.block = int(min(block_docs, max(self.N, 1)))
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.idx_offset = int(idx_offset)
        dev = self.device
        self._stage = [torch.empty((self.block, self.d), dtype=torch.bfloat16, device=dev) for _ in range(2)]
        self._d32 = [torch.empty((self.block, self.d), dtype=torch.float32, device=dev) for _ in range(2)]
        self._d16 = [torch.empty((self.block, self.d), dtype=torch.float16, device=dev) for _ in range(2)] \
            if (screen and self.d == 256) else [None, None]
        self._copy = torch.cuda.Stream(device=dev)
        self._ready = [torch.cuda.Event() for _ in range(2)]
        self._free = [torch.cuda.Event() for _ in range(2)]
        # one streaming pass at build time: the corpus-wide largest row norm bounds the screen's error term
        stats = torch.zeros(2, dtype=torch.float32, device=dev)
        self._walk(lambda s, lo, n: None, stats=stats)
        dmax, amax = (float(x) for x in stats.tolist())
        self.dmax_norm = dmax
        if not (dmax == dmax and amax < 6.0e4 and dmax < 6.0e4):
            self._d16 = [None, None]

    def _walk(self, visit, stats=None):
        L = _lib.lib()
        cur = torch.cuda.current_stream(self.device)
        for s in range(2):
            self._free[s].record(cur)
        nblk = (self.N + self.block - 1) // self.block
        for i in range(nblk):
            s = i % 2
            lo = i * self.block
            n = min(self.block, self.N - lo)
            with torch.cuda.stream(self._copy):
                self._copy.wait_event(self._free[s])          # the staging buffer has been consumed
                self._stage[s][:n].copy_(self.host[lo:lo + n], non_blocking=True)
                self._ready[s].record(self._copy)
            cur.wait_event(self._ready[s])
            with torch.cuda.device(self.device):
                _lib.check(L.tt_index_build_from_bf16(
                    self._stage[s].data_ptr(), n, self.d, self._d32[s].data_ptr(),
                    self._d16[s].data_ptr() if self._d16[s] is not None else None,
                    stats.data_ptr() if stats is not None else None, 0, cur.cuda_stream))
            self._free[s].record(cur)
            visit(s, lo, n)

    def resident(self) -> "BruteForceIndex":
        """The same corpus widened once into HBM (N*d*4 bytes fp32 + N*d*2 bytes fp16 shadow): configs[4]'s shard
        of 12.5M passages is 19.2 GB of a 288 GB GPU, after which it is searched at the resident rates."""
        d32 = torch.empty((self.N, self.d), dtype=torch.float32, device=self.device)
        d16 = torch.empty((self.N, self.d), dtype=torch.float16, device=self.device) if self._d16[0] is not None else None

        def visit(s, lo, n):
            d32[lo:lo + n].copy_(self._d32[s][:n])
            if d16 is not None:
                d16[lo:lo + n].copy_(self._d16[s][:n])

        self._walk(visit)
        return BruteForceIndex._from_buffers(d32, d16, self.dmax_norm, self.idx_offset)

    @property
    def ntotal(self) -> int:
        return self.N

    def search(self, q: torch.Tensor, k: int = 10, out=None) -> Tuple[torch.Tensor, torch.Tensor]:
        """out: optional (vals f32 [B,k], idx int64 [B,k]) device tensors to write the result into (ShardedIndex's send block)."""
        _need_cuda(q)
        if q.dim() == 1:  # a single query vector: squeezed result, like BruteForceIndex
            v, i = self.search(q.unsqueeze(0), k)
            return v[0], i[0]
        if q.device != self.device:
            raise ValueError(f"queries on {q.device} but the index streams through {self.device}")
        if q.shape[1] != self.d:
            raise ValueError(f"shape mismatch: q {tuple(q.shape)} vs docs {(self.N, self.d)}")
        q = _f32c(q)
        run = [None, None]

        def visit(s, lo, n):
            blk = BruteForceIndex._from_buffers(self._d32[s][:n], self._d16[s][:n] if self._d16[s] is not None else None,
                                                self.dmax_norm, self.idx_offset + lo)
            v, i = blk.search(q, k)
            if run[0] is None:
                run[0], run[1] = v, i
            else:
                run[0], run[1] = topk_merge(torch.cat([run[0], v], 1), torch.cat([run[1], i], 1), k)

        self._walk(visit)
        if run[0] is None:
            run = [torch.full((q.shape[0], k), float("-inf"), device=q.device),
                   torch.full((q.shape[0], k), -1, dtype=torch.int64, device=q.device)]
        if out is not None:
            out[0].copy_(run[0])
            out[1].copy_(run[1])
            return out
        return run[0], run[1]
