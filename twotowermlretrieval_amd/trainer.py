"""Triplet-loss training step, data-parallel over RCCL.

Mirrors the live train step of the reference (backend/main.py:244-259):

    optimizer.zero_grad()
    q, p, n = model.encode_query(queries), model.encode_document(pos), model.encode_document(neg)
    loss = triplet_loss_cosine((q, p, n), margin)          # model.py:109-114
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)
    optimizer.step()                                       # Adam(lr), main.py:222

`FusedClipAdam` replaces the last two lines with ONE kernel pair over a single flat fp32 buffer
(tt_clip_adam_step_f32) and, when a process group is given, ONE summing all-reduce of that buffer
(RCCL over xGMI for the nccl backend) followed by the 1/world scale inside the kernel -- so every rank
clips the SAME averaged gradient, exactly as the reference would on the global batch.
The reference's own pair of torch calls also keeps working on this package's model (its gradients
are ordinary tensors); the fused optimizer is the MI355X-native path.
"""
from __future__ import annotations

from typing import Callable, Iterable, Tuple

import torch

from . import _lib
from .model import TwoTowerModel, deferred_input_checks, triplet_loss_cosine

__all__ = ["FusedClipAdam", "train_step", "DataParallelTrainer"]


def _hip_clip_adam(flat_p, flat_g, m, v, step, lr, betas, eps, max_norm, grad_scale, total_norm, scratch):
    L = _lib.lib()
    with torch.cuda.device(flat_p.device):
        _lib.check(L.tt_clip_adam_step_f32(flat_p.data_ptr(), flat_g.data_ptr(), m.data_ptr(), v.data_ptr(),
                                           flat_p.numel(), step, lr, betas[0], betas[1], eps, max_norm, grad_scale,
                                           total_norm.data_ptr(), scratch.data_ptr(),
                                           torch.cuda.current_stream(flat_p.device).cuda_stream))


class _FlatClipAdam:
    """Host logic of the fused optimizer on ordinary tensors: all trainable parameters are re-pointed at views of one
    contiguous fp32 buffer (`flat_params`) and their .grad at views of `flat_grads`, so the optimizer and the
    data-parallel all-reduce touch two pointers; step() = summing all-reduce -> step_fn(..., grad_scale = 1/world).
    Private: the CPU tests drive it over gloo with the oracle's step; the product class is FusedClipAdam."""

    def __init__(self, params: Iterable[torch.nn.Parameter], step_fn: Callable, all_reduce: Callable, world: int,
                 lr: float, betas: Tuple[float, float], eps: float, max_norm: float, scratch_bytes: int):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev = self.params[0].device
        self.lr, self.betas, self.eps, self.max_norm = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(max_norm)
        self._step_fn, self._all_reduce, self.world = step_fn, all_reduce, int(world)
        n = sum(p.numel() for p in self.params)
        self.flat_params = torch.empty(n, dtype=torch.float32, device=dev)
        self.flat_grads = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.total_norm = torch.zeros(1, dtype=torch.float32, device=dev)
        self._scratch = torch.empty(scratch_bytes, dtype=torch.uint8, device=dev)
        self._views = []
        off = 0
        with torch.no_grad():
            for p in self.params:
                k = p.numel()
                self.flat_params[off:off + k].copy_(p.detach().reshape(-1))
                p.data = self.flat_params[off:off + k].view_as(p)
                gv = self.flat_grads[off:off + k].view_as(p)
                p.grad = gv
                self._views.append(gv)
                off += k
        self.step_count = 0

    def zero_grad(self) -> None:
        self.flat_grads.zero_()
        for p, gv in zip(self.params, self._views):
            p.grad = gv

    def _collect(self) -> None:
        """Tolerate callers that reset .grad to None / fresh tensors (model.zero_grad(set_to_none=True))."""
        for p, gv in zip(self.params, self._views):
            if p.grad is None:
                gv.zero_()
            elif p.grad.data_ptr() != gv.data_ptr():
                gv.copy_(p.grad)
            p.grad = gv

    def step(self) -> torch.Tensor:
        """Returns the pre-clip global gradient norm (device tensor, no sync)."""
        self._collect()
        if self.world > 1:
            self._all_reduce(self.flat_grads)  # one 3.4 MB bucket
        self.step_count += 1
        self._step_fn(self.flat_params, self.flat_grads, self.exp_avg, self.exp_avg_sq, self.step_count, self.lr,
                      self.betas, self.eps, self.max_norm, 1.0 / self.world, self.total_norm, self._scratch)
        self.mark_params_changed()
        return self.total_norm

    def mark_params_changed(self) -> None:
        """Call after ANY write to `flat_params` that did not go through the parameters themselves (this class's own step,
        a broadcast into the flat buffer, a checkpoint copied into it): the parameters are views whose version counters are
        separate from the flat buffer's, and RNNEncoder keys its cache of kernel-form weights on those counters -- without
        the bump an eval forward would keep serving the weights of before the write."""
        torch.autograd.graph.increment_version(self.params)


class FusedClipAdam(_FlatClipAdam):
    """clip_grad_norm_(max_norm) + Adam(lr, betas, eps, weight_decay=0) over one flat buffer in ONE kernel pair
    (tt_clip_adam_step_f32); with a process group, ONE summing all-reduce of that buffer first (tt_allreduce_grads on
    torch.distributed's RCCL communicator; torch.distributed's own call for other backends)."""

    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-4, betas: Tuple[float, float] = (0.9, 0.999),
                 eps: float = 1e-8, max_norm: float = 1.0, group=None, comm=None):
        from .collective import Collective
        params = [p for p in params if p.requires_grad]
        if params and params[0].device.type != "cuda":
            raise RuntimeError("FusedClipAdam runs only on an AMD GPU via libtt.so (no CPU fallback)")
        self.group = group
        self._coll = Collective(group, params[0].device if params else None, comm=comm)
        super().__init__(params, _hip_clip_adam, self._coll.all_reduce_sum, self._coll.world, lr, betas, eps, max_norm,
                         _lib.lib().tt_clip_adam_scratch_bytes())


_TOWER_STREAMS = {}


def _tower_streams(device):
    key = (device.type, device.index)
    if key not in _TOWER_STREAMS:
        _TOWER_STREAMS[key] = [torch.cuda.Stream(device=device) for _ in range(3)]
    return _TOWER_STREAMS[key]


def _concat_ids(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """[a; b] padded to the longer T with id 0, on the CURRENT stream: one launch (tt_concat_ids_i64) instead of torch's zero
    fill + two slice copies -- the document tower's whole chain waits for it."""
    from . import _lib
    T = max(a.shape[1], b.shape[1])
    a, b = a.contiguous(), b.contiguous()
    out = torch.empty((a.shape[0] + b.shape[0], T), dtype=torch.int64, device=a.device)
    with torch.cuda.device(a.device):
        _lib.check(_lib.lib().tt_concat_ids_i64(a.data_ptr(), a.shape[0], a.shape[1], b.data_ptr(), b.shape[0], b.shape[1],
                                                out.data_ptr(), T, torch.cuda.current_stream(a.device).cuda_stream))
    return out


def _train_step_direct(model: TwoTowerModel, optimizer, queries, pos_docs, neg_docs, margin: float):
    """The same step without the autograd engine: tower forwards (train mode), the fused loss + gradient kernel, tower backwards
    written STRAIGHT into the optimizer's flat gradient buffer, optimizer step.  Every parameter receives its gradient exactly
    once (query tower once; positives and negatives as one 2B-row document-tower call), so nothing has to be zeroed or
    accumulated: what the autograd path adds per step -- grad_output plumbing around the loss, four accumulate-adds per
    tower, the 3.4 MB zero fill -- is ~0.1 ms of small launches on the document tower's critical path.  Returns None when the
    shortcut does not apply (another optimizer, a trainable embedding table, parameters without the optimizer's gradient
    views): the caller then takes the autograd path, which computes the same numbers."""
    from . import _lib
    from .model import _raise_status
    if not isinstance(optimizer, _FlatClipAdam) or not torch.is_grad_enabled():
        return None
    # the document tower (2B rows of ~70 tokens) is the step's critical path: its launches go out FIRST, the query tower's
    # ~15 small launches then overlap it instead of delaying it by the ~0.1 ms the host needs to issue them
    encs = (model.doc_encoder, model.query_encoder)
    views = {id(p): gv for p, gv in zip(optimizer.params, optimizer._views)}
    into = []
    for enc in encs:
        ps = enc._flat_params()
        if enc.embedding.weight.requires_grad or not enc.training or any(id(p) not in views for p in ps):
            return None
        into.append([views[id(p)] for p in ps])
    if len({id(p) for enc in encs for p in enc._flat_params()}) != len(optimizer.params):
        return None  # (the optimizer holds parameters no tower call would write)
    dev = queries.device
    cur = torch.cuda.current_stream(dev)
    B = queries.shape[0]
    streams = _tower_streams(dev)[:2]
    if pos_docs.dtype != torch.int64 or neg_docs.dtype != torch.int64:
        return None
    streams[0].wait_stream(cur)
    with torch.cuda.stream(streams[0]):  # (on the document tower's own stream: no hop from the caller's)
        pos_docs.record_stream(streams[0])
        neg_docs.record_stream(streams[0])
        both = _concat_ids(pos_docs, neg_docs)
    ids_of = (both, queries)
    # dropout seeds from torch's CPU generator in the order the autograd path draws them (query tower, then document tower)
    seeds = {id(enc): (int(torch.randint(0, 2 ** 62, (1,)).item()) if enc.dropout > 0.0 else 0)
             for enc in (model.query_encoder, model.doc_encoder)}
    fw = []
    for enc, ids, s in zip(encs, ids_of, streams):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            ids.record_stream(s)
            p_drop = enc.dropout
            seed = seeds[id(enc)]
            check, enc.check_inputs = enc.check_inputs, False
            try:
                out, ws, status = enc._run_forward(ids, train=True, dropout_p=p_drop, dropout_seed=seed)
            finally:
                enc.check_inputs = check
            fw.append((out, ws, status, p_drop, seed))
    # The document tower's stream carries the step's critical path from here on: the loss, the document backward and the
    # optimizer are enqueued on IT (a hop to the caller's stream and back cost ~20 us each way on that path: event wait +
    # launch); the query tower's stream joins for the loss and again before the optimizer.
    s_doc, s_qry = streams
    pn, q = fw[0][0], fw[1][0]
    p, n = pn[:B], pn[B:]
    H = q.shape[1]
    s_doc.wait_stream(s_qry)
    with torch.cuda.stream(s_doc):
        loss = torch.empty((), dtype=torch.float32, device=dev)
        dq = torch.empty_like(q)
        dpn = torch.empty_like(pn)
        rows = torch.empty(B, dtype=torch.float32, device=dev)
        for t in (q, loss, dq, dpn, rows):
            t.record_stream(s_doc)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().tt_triplet_loss_f32(q.data_ptr(), p.data_ptr(), n.data_ptr(), B, H, float(margin), loss.data_ptr(),
                                                      dq.data_ptr(), dpn[:B].data_ptr(), dpn[B:].data_ptr(), rows.data_ptr(),
                                                      s_doc.cuda_stream))
    s_qry.wait_stream(s_doc)
    for enc, ids, s, f, d_out, grads in zip(encs, ids_of, streams, fw, (dpn, dq), into):
        with torch.cuda.stream(s):
            d_out.record_stream(s)
            enc._run_backward(ids.contiguous(), f[1], d_out, f[3], f[4], into=grads)
    s_doc.wait_stream(s_qry)
    for p_, gv in zip(optimizer.params, optimizer._views):
        p_.grad = gv
    with torch.cuda.stream(s_doc):
        # The towers' status words (zero-length rows / ids out of range raise as in the reference; a column-split recurrence
        # that gave up) are read HERE, behind the backward kernels: the read is the one host synchronisation of the step, and
        # the optimizer step is not enqueued for a bad batch -- the weights stay untouched (the gradient buffer holds garbage,
        # which the next step overwrites).
        for enc, f in zip(encs, fw):
            if enc.check_inputs:
                _raise_status(int(f[2].item()))
        for t in (optimizer.flat_params, optimizer.flat_grads, optimizer.exp_avg, optimizer.exp_avg_sq, optimizer.total_norm,
                  optimizer._scratch):
            t.record_stream(s_doc)
        optimizer.step()
    cur.wait_stream(s_doc)
    loss.record_stream(cur)
    return loss


def train_step(model: TwoTowerModel, optimizer: FusedClipAdam, queries: torch.Tensor, pos_docs: torch.Tensor,
               neg_docs: torch.Tensor, margin: float = 0.2, concurrent_towers: bool = True, direct: bool = True) -> torch.Tensor:
    """One step of backend/main.py:244-259 on this rank's (equal-sized) share of the global batch.
    Returns the local loss as a 0-d device tensor (no .item(): the reference's per-step sync is dropped).

    concurrent_towers: the encoder calls are independent and each recurrence kernel only occupies ceil(B/16)
    CUs, so the query tower and the document tower (positives and negatives in one 2B-row call) are issued
    on separate HIP streams (autograd replays each call's backward on the stream its forward ran on).
    direct: skip the autograd engine when the step has the standard shape (_train_step_direct: same kernels, same numbers)."""
    if direct and concurrent_towers and queries.is_cuda and neg_docs.shape[0] == pos_docs.shape[0] == queries.shape[0]:
        loss = _train_step_direct(model, optimizer, queries, pos_docs, neg_docs, margin)
        if loss is not None:
            return loss
    optimizer.zero_grad()
    if concurrent_towers and queries.is_cuda:
        cur = torch.cuda.current_stream(queries.device)
        B = pos_docs.shape[0]
        if neg_docs.shape[0] == B:
            # positives and negatives go through the SAME tower: one call over 2B rows (rows are independent),
            # so the recurrence kernels fill twice the CUs and the weight-gradient GEMMs run once
            both = _concat_ids(pos_docs.long(), neg_docs.long())
            calls = ((model.encode_query, queries), (model.encode_document, both))
        else:
            calls = ((model.encode_query, queries), (model.encode_document, pos_docs), (model.encode_document, neg_docs))
        outs = []
        # input checking stays on (zero-length rows / out-of-range ids raise as in the reference), but the status
        # words of the towers are read ONCE, after all of them have been enqueued: a per-call read would make the
        # host wait for the query tower before it could launch the document tower
        with deferred_input_checks(model.query_encoder, model.doc_encoder):
            for s, (fn, ids) in zip(_tower_streams(queries.device), calls):
                s.wait_stream(cur)
                with torch.cuda.stream(s):
                    ids.record_stream(s)
                    outs.append(fn(ids))
            for s, o in zip(_tower_streams(queries.device), outs):
                cur.wait_stream(s)
                o.record_stream(cur)
        if len(outs) == 2:
            q, p, n = outs[0], outs[1][:B], outs[1][B:]
        else:
            q, p, n = outs
    else:
        q = model.encode_query(queries)
        p = model.encode_document(pos_docs)
        n = model.encode_document(neg_docs)
    loss = triplet_loss_cosine((q, p, n), margin=margin)
    loss.backward()
    optimizer.step()
    return loss.detach()


class DataParallelTrainer:
    """Replicated model, per-rank batch shard, gradient all-reduce inside FusedClipAdam.step()."""

    def __init__(self, model: TwoTowerModel, lr: float = 1e-4, margin: float = 0.2, max_norm: float = 1.0, group=None):
        self.model = model
        self.margin = margin
        self.optimizer = FusedClipAdam(model.parameters(), lr=lr, max_norm=max_norm, group=group)

    def broadcast_parameters(self, src: int = 0) -> None:
        import torch.distributed as dist
        if dist.is_initialized() and dist.get_world_size(self.optimizer.group) > 1:
            dist.broadcast(self.optimizer.flat_params, src=src, group=self.optimizer.group)
        self.optimizer.mark_params_changed()  # (the broadcast wrote the flat buffer, not the parameter tensors)

    def step(self, queries, pos_docs, neg_docs) -> torch.Tensor:
        self.model.train()
        return train_step(self.model, self.optimizer, queries, pos_docs, neg_docs, self.margin)
