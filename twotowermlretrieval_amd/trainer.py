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

from typing import Callable, Iterable, Optional, Tuple

import torch

from . import _lib
from .model import TwoTowerModel, deferred_input_checks, triplet_loss_cosine

__all__ = ["FusedClipAdam", "train_step", "DataParallelTrainer", "GraphedTrainStep"]


def _hip_clip_adam(flat_p, flat_g, m, v, step, lr, betas, eps, max_norm, grad_scale, total_norm, scratch):
    L = _lib.lib()
    with torch.cuda.device(flat_p.device):
        _lib.check(L.tt_clip_adam_step_f32(flat_p.data_ptr(), flat_g.data_ptr(), m.data_ptr(), v.data_ptr(),
                                           flat_p.numel(), step, lr, betas[0], betas[1], eps, max_norm, grad_scale,
                                           total_norm.data_ptr(), scratch.data_ptr(),
                                           torch.cuda.current_stream(flat_p.device).cuda_stream))


def _hip_clip_adam_gated(flat_p, flat_g, m, v, step_dev, lr, betas, eps, max_norm, grad_scale, total_norm, gate, scratch):
    L = _lib.lib()
    with torch.cuda.device(flat_p.device):
        _lib.check(L.tt_clip_adam_step_gated_f32(flat_p.data_ptr(), flat_g.data_ptr(), m.data_ptr(), v.data_ptr(),
                                                 flat_p.numel(), step_dev.data_ptr(), lr, betas[0], betas[1], eps, max_norm,
                                                 grad_scale, total_norm.data_ptr(),
                                                 gate.data_ptr() if gate is not None else None, scratch.data_ptr(),
                                                 torch.cuda.current_stream(flat_p.device).cuda_stream))


def _hip_step_gate(status_words, gate):
    """gate[b] = how many of the status words have bit b set (one launch, no host read)."""
    import ctypes as C
    cur = torch.cuda.current_stream(gate.device)
    arr = (C.c_void_p * len(status_words))()
    for i, st in enumerate(status_words):
        st.record_stream(cur)  # (written on a tower's stream, read here)
        arr[i] = st.data_ptr()
    with torch.cuda.device(gate.device):
        _lib.check(_lib.lib().tt_step_gate_f32(arr, len(status_words), gate.data_ptr(), cur.cuda_stream))


def _torch_step_gate(status_words, gate):
    """The same on ordinary tensors (the CPU rehearsals of the host logic)."""
    bits = torch.stack([st.reshape(()).to(torch.int64) for st in status_words])
    for b in range(3):
        gate[b] = float(((bits >> b) & 1).sum())
    gate[3] = 0.0


class _FlatClipAdam:
    """Host logic of the fused optimizer on ordinary tensors: all trainable parameters are re-pointed at views of one
    contiguous fp32 buffer (`flat_params`) and their .grad at views of `flat_grads`, so the optimizer and the
    data-parallel all-reduce touch two pointers; step() = summing all-reduce -> step_fn(..., grad_scale = 1/world).

    A FAILED STEP IS A COLLECTIVE DECISION.  The gradient bucket has TT_STEP_GATE_WORDS extra floats behind the gradients (the
    `gate`): before the all-reduce they hold, per status bit, how many of this rank's encoder calls raised it (zero-length row,
    id out of range, column-split recurrence timed out: the status words the watched encoders handed over -- `watch`); the ONE
    all-reduce sums them with the gradients, so afterwards every rank holds the same counts, the device applies the step on
    all ranks or on none, and every rank raises the same exception.  The reference's step is single-process (a bad batch
    raises and the run stops, backend/main.py:244-259); a rank-local raise in front of the all-reduce would leave the peers
    waiting in it forever.

    Private: the CPU tests drive it over gloo with the oracle's step; the product class is FusedClipAdam."""

    GATE = _lib.TT_STEP_GATE_WORDS

    def __init__(self, params: Iterable[torch.nn.Parameter], step_fn: Callable, all_reduce: Callable, world: int,
                 lr: float, betas: Tuple[float, float], eps: float, max_norm: float, scratch_bytes: int,
                 gate_fn: Callable = _torch_step_gate, gated_step_fn: Optional[Callable] = None):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev = self.params[0].device
        self.lr, self.betas, self.eps, self.max_norm = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(max_norm)
        self._step_fn, self._all_reduce, self.world = step_fn, all_reduce, int(world)
        self._gate_fn, self._gated_step_fn = gate_fn, gated_step_fn
        n = sum(p.numel() for p in self.params)
        self.flat_params = torch.empty(n, dtype=torch.float32, device=dev)
        self._bucket = torch.zeros(n + self.GATE, dtype=torch.float32, device=dev)  # what the all-reduce sees
        self.flat_grads = self._bucket[:n]
        self.gate = self._bucket[n:]
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.total_norm = torch.zeros(1, dtype=torch.float32, device=dev)
        self._scratch = torch.empty(scratch_bytes, dtype=torch.uint8, device=dev)
        self._step_dev = torch.zeros(1, dtype=torch.int64, device=dev)  # steps applied (the gated kernel counts on the device)
        self._step_host = 0
        self._pending_status: list = []  # status words of the watched encoders' training calls since the last step()
        self._gate_host, self._gate_ev, self._gate_n, self._deferred = None, None, 0, None   # (snapshot_gate / settle)
        self._reduced_upto = 0           # gradients [0, _reduced_upto) of this step have been all-reduced already (reduce_prefix)
        self.reduce_timing: Optional[list] = None   # a list: every all-reduce appends (event before, event after, bytes) -- bench.py
        self.check = True                # step() reads the reduced gate (one host synchronisation) and raises
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

    # ---- step number -----------------------------------------------------------
    @property
    def step_count(self) -> int:
        """Steps applied so far (reads the device counter of the gated kernel: a host synchronisation)."""
        return int(self._step_dev.item()) if self._gated_step_fn is not None else self._step_host

    @step_count.setter
    def step_count(self, value: int) -> None:
        self._step_host = int(value)
        self._step_dev.fill_(int(value))

    # ---- status words of the encoders -------------------------------------------
    def watch(self, *modules) -> None:
        """The RNNEncoders inside `modules` hand the status words of their TRAINING calls to this optimizer instead of raising
        at the call: step() reduces them over the ranks with the gradients and raises -- on every rank -- what the reference
        would have raised.  DataParallelTrainer and train_step do this for their model; a hand-written loop around
        FusedClipAdam(group=...) must, or a bad batch on one rank leaves the others waiting in the all-reduce.
        What a watched encoder hands over is EVERY grad-enabled forward since the last step(): all of them (any number:
        gradient accumulation) decide the next step() together.  Forwards that belong to no step -- validation -- must run under
        torch.no_grad() (as the reference's evaluators do, backend/evaluators.py:31), or their bad batch vetoes the next training
        step instead of raising where it happened; unwatch() ends the hand-over."""
        from .model import RNNEncoder
        for mod in modules:
            for enc in mod.modules():
                if isinstance(enc, RNNEncoder):
                    enc._status_sink = self._pending_status

    def unwatch(self, *modules) -> None:
        from .model import RNNEncoder
        for mod in modules:
            for enc in mod.modules():
                if isinstance(enc, RNNEncoder) and enc._status_sink is self._pending_status:
                    enc._status_sink = None

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

    def step(self, check: Optional[bool] = None) -> torch.Tensor:
        """Returns the pre-clip global gradient norm (device tensor).  With status words pending or a process group, the step
        is gated (class docstring) and -- unless check is False -- the reduced gate is read back (the step's one host
        synchronisation): IndexError / RuntimeError as the reference raises them, model.SplitRecurrenceTimeout for a time-out,
        on every rank alike; parameters, moments and step number are then untouched."""
        self._collect()
        return self._reduce_apply(self._fold_status(), check)

    def _fold_status(self) -> bool:
        """The pending status words -> the gate words behind the gradients (one launch, no host read).  Returns whether the step
        is gated (status words were pending, or there is a process group)."""
        status = list(self._pending_status)
        self._pending_status.clear()  # (in place: the watched encoders hold this list)
        gated = bool(status) or self.world > 1
        if status:
            self._gate_fn(status, self.gate)
        elif gated:
            self.gate.zero_()
        return gated

    def reduce_prefix(self, upto: int) -> None:
        """Data-parallel only: all-reduce the bucket's first `upto` gradients NOW, on the current stream -- the part of the
        bucket whose gradients are already complete (the query tower's, while the document tower's weight-gradient kernels still
        run: the exchange of half the bucket hides under them).  _reduce_apply then reduces the rest, gate words included: the
        failure decision still rides in the LAST all-reduce of the step.  Every rank must make the same calls in the same order
        (a function of the model's structure only)."""
        if self.world > 1 and 0 < upto <= self.flat_grads.numel() and self._reduced_upto == 0:
            self._timed_all_reduce(self._bucket[:upto])
            self._reduced_upto = int(upto)

    def _timed_all_reduce(self, part: torch.Tensor) -> None:
        """The all-reduce of one part of the bucket on the current stream; with `reduce_timing` set, bracketed by two timing
        events on that stream (what the collective costs THIS rank, its wait for the slowest peer included)."""
        if self.reduce_timing is None or not part.is_cuda:
            self._all_reduce(part)
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        self._all_reduce(part)
        e1.record()
        self.reduce_timing.append((e0, e1, part.numel() * 4))

    def _reduce_apply(self, gated: bool, check: Optional[bool] = None) -> torch.Tensor:
        if self.world > 1:
            # the bucket: 3.4 MB of gradients + the gate words (minus what reduce_prefix has sent already)
            self._timed_all_reduce(self._bucket[self._reduced_upto:] if self._reduced_upto else self._bucket)
            self._reduced_upto = 0
        args = (self.lr, self.betas, self.eps, self.max_norm, 1.0 / self.world, self.total_norm)
        words = None
        if self._gated_step_fn is not None:
            self._gated_step_fn(self.flat_params, self.flat_grads, self.exp_avg, self.exp_avg_sq, self._step_dev, *args,
                                self.gate if gated else None, self._scratch)
        else:  # ordinary tensors: the gate is looked at on the host
            words = self.gate.tolist() if gated else None
            if not words or not any(words):
                self._step_host += 1
                self._step_fn(self.flat_params, self.flat_grads, self.exp_avg, self.exp_avg_sq, self._step_host, *args,
                              self._scratch)
        self.mark_params_changed()
        if gated and (self.check if check is None else check):
            self.raise_for_gate(self.gate.tolist() if words is None else words)
        return self.total_norm

    @staticmethod
    def raise_for_gate(words) -> None:
        from .model import _raise_status
        _raise_status(sum(1 << b for b in range(3) if words[b] != 0))

    # ---- the gate read, one step late (train_step(defer_check=True), GraphedTrainStep(defer_check=True)) --------------------
    def snapshot_gate(self, redo: Optional[Callable] = None) -> None:
        """Copy the (reduced) gate words of the step just enqueued to pinned host memory, asynchronously on the current stream, and
        make them THE pending check of this optimizer; the previous pending check -- whose step has long finished -- is settled
        afterwards (it raises here, one call late).  redo: called instead of raising when the words say "recurrence time-out".
        Order matters: THIS step's copy is enqueued and registered first, so neither an exception out of the previous check nor
        its redo (which runs a whole step: new gate words in the same device buffer) can lose it -- the copy is already in the
        stream in front of anything the redo enqueues, and it stays pending for the next call's settle().  A redone step is
        applied BEHIND the step enqueued in this call (it was skipped on the device when its turn came; the weights it now starts
        from include this call's update): the order of two updates changes, none is lost."""
        if self._gate_host is None:
            self._gate_host = [torch.zeros(self.GATE, dtype=torch.float32).pin_memory() if self.gate.is_cuda
                               else torch.zeros(self.GATE, dtype=torch.float32) for _ in range(2)]
            self._gate_ev = [torch.cuda.Event() if self.gate.is_cuda else None for _ in range(2)]
        slot = self._gate_n & 1          # (the other slot holds the previous pending check, if there is one)
        self._gate_n += 1
        self._gate_host[slot].copy_(self.gate, non_blocking=True)
        if self._gate_ev[slot] is not None:
            self._gate_ev[slot].record(torch.cuda.current_stream(self.gate.device))
        previous, self._deferred = self._deferred, (slot, redo)
        self._settle_one(previous)

    def settle(self):
        """Look at the pending check, if any: raises what its step would have raised (IndexError / RuntimeError), or returns
        redo()'s result after a recurrence time-out, or None."""
        pending, self._deferred = self._deferred, None
        return self._settle_one(pending)

    def _settle_one(self, pending):
        from .model import SplitRecurrenceTimeout
        if pending is None:
            return None
        slot, redo = pending
        if self._gate_ev[slot] is not None:
            self._gate_ev[slot].synchronize()
        try:
            self.raise_for_gate(self._gate_host[slot].tolist())
        except SplitRecurrenceTimeout:
            if redo is None:
                raise
            return redo()
        return None

    def mark_params_changed(self) -> None:
        """Call after ANY write to `flat_params` that did not go through the parameters themselves (this class's own step,
        a broadcast into the flat buffer, a checkpoint copied into it): the parameters are views whose version counters are
        separate from the flat buffer's, and RNNEncoder keys its cache of kernel-form weights on those counters -- without
        the bump an eval forward would keep serving the weights of before the write."""
        torch.autograd.graph.increment_version(self.params)


class FusedClipAdam(_FlatClipAdam):
    """clip_grad_norm_(max_norm) + Adam(lr, betas, eps, weight_decay=0) over one flat buffer in ONE kernel pair
    (tt_clip_adam_step_gated_f32: step number and bias corrections on the device, the step predicated on the reduced gate);
    with a process group, ONE summing all-reduce of the bucket first (tt_allreduce_grads on torch.distributed's RCCL
    communicator; torch.distributed's own call for other backends)."""

    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-4, betas: Tuple[float, float] = (0.9, 0.999),
                 eps: float = 1e-8, max_norm: float = 1.0, group=None, comm=None):
        from .collective import Collective
        params = [p for p in params if p.requires_grad]
        if params and params[0].device.type != "cuda":
            raise RuntimeError("FusedClipAdam runs only on an AMD GPU via libtt.so (no CPU fallback)")
        self.group = group
        self._coll = Collective(group, params[0].device if params else None, comm=comm)
        super().__init__(params, _hip_clip_adam, self._coll.all_reduce_sum, self._coll.world, lr, betas, eps, max_norm,
                         _lib.lib().tt_clip_adam_scratch_bytes(), gate_fn=_hip_step_gate, gated_step_fn=_hip_clip_adam_gated)


_TOWER_STREAMS = {}


def _tower_streams(device):
    key = (device.type, device.index)
    if key not in _TOWER_STREAMS:
        # [0] document tower, [1] query tower, [2] a third tower call of the autograd path.  (A high-priority query stream was
        # tried: its split recurrence still only gets whole CUs when the document tower's input projection ends -- that kernel's
        # workgroups are resident for its whole duration -- 161 us against 148 without, profiles/r04_l_train_timeline.txt.)
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


_ORDER_EVENTS = {}
_TWO_PHASE = True   # (tools/experiments: False issues the query tower first instead of splitting the document tower's call)
# How the two towers' FORWARD recurrences share the CUs when both column-split would not fit (tools/experiments/fwd_order_ab.py):
#   "query_first"  the query tower's recurrence first, the document tower's waits for it (event inside the calls); default
#   "doc_first"    the document tower's first, the query tower's waits (it has slack at the END of the step, none here: the
#                  loss needs its output)
#   "query_one_wg" no ordering: the query tower on the one-workgroup recurrence beside the document tower's split one
_FWD_ORDER = "query_first"


def _order_events(device):
    """Two hipEvents per device (tt_event_create: plain HIP events the C calls record / wait on, include/tt.h tt_enc_sync_t)."""
    import ctypes as C
    key = (device.type, device.index)
    if key not in _ORDER_EVENTS:
        evs = []
        with torch.cuda.device(device):
            for _ in range(2):
                e = C.c_void_p()
                _lib.check(_lib.lib().tt_event_create(C.byref(e)))
                evs.append(e.value)
        _ORDER_EVENTS[key] = tuple(evs)
    return _ORDER_EVENTS[key]


class _towers_in_flight:
    """For the duration of one train step: (1) the model's encoders hand their status words to `optimizer` (watch), so that
    the step's failure is decided in optimizer.step(), by all ranks together; (2) the towers that are in flight at the same time
    never have more column-split workgroups in flight than the device has CUs.  A split launch needs every member of every
    team resident at once (one workgroup per CU), which ONE launch guarantees against the CU count and two launches on two
    streams do not (query tower 128 + document tower 256 on 256 CUs: round 3 rested on in-order dispatch).  Two ways out, both
    co-residency BY CONSTRUCTION:
      * ordered (the direct step), FORWARD: the towers' recurrence launches are ordered by an event inside the calls
        (tt_enc_sync_t): the query tower's recurrence first, the document tower's behind it, everything else of the two towers
        still overlaps; the document tower's call goes out in two halves around the query tower's (TT_ENC_PHASE_BEGIN / _FINISH),
        so that the record is issued before the wait without the document tower's projection waiting for the host.  BACKWARD: the smaller tower runs the
        one-workgroup recurrence (its workgroups wait for nobody).  Ordering the backwards too was measured and dropped: the query
        tower's split recurrence behind the document tower's lands under the document tower's weight-gradient kernels, whose
        one-per-CU workgroups keep it off the CUs until they end (graph replay 1.146 -> 1.207 ms), and in front of it it would
        add its ~100 us to the critical path;
      * one_workgroup (the autograd path, whose calls cannot carry events): the smaller towers run the one-workgroup
        recurrences (TT_ENC_ONE_WORKGROUP: same bits, no hand-off) until the sum fits.  Measured (profiles/
        r04_j_train_timeline.txt): the query tower then holds 32 CUs for 280 + 390 us and 32 of the document tower's split
        workgroups wait for them -- correct (nobody waits for a workgroup that cannot be scheduled), ~60 us per step slower.
    (csrc/gru16x4.hip, include/tt.h)"""

    def __init__(self, model: TwoTowerModel, optimizer, rows: dict, force_one_workgroup: bool = False):
        self.model, self.optimizer, self.rows, self.force = model, optimizer, rows, force_one_workgroup
        self.needs_order = False

    def __enter__(self):
        encs = [self.model.query_encoder, self.model.doc_encoder]
        self._saved = [(e, e.one_workgroup, e.one_workgroup_bwd, e._status_sink) for e in encs]
        if isinstance(self.optimizer, _FlatClipAdam):
            self.optimizer.watch(*encs)
        self._need = {}
        if self.force:
            for e in encs:
                e.one_workgroup, e.one_workgroup_bwd = True, None
        elif self.rows and encs[0].embedding.weight.is_cuda:
            L = _lib.lib()
            cus = torch.cuda.get_device_properties(encs[0].embedding.weight.device).multi_processor_count
            self._need = {e: (0 if (e.one_workgroup and e.one_workgroup_bwd is not False) else e.split_workgroups(B))
                          for e, B in self.rows.items()}
            self.needs_order = sum(self._need.values()) > cus
            if self.needs_order and any(e.num_layers > 1 for e in encs):
                # stacked layers: the ordering event is recorded behind a call's LAST recurrence launch, i.e. behind ALL the layers
                # of the smaller tower -- the document tower's first layer would wait for the whole query tower (207 us of a
                # 5.5 ms step at the reference's default shape, profiles/r04_z_config1_train_timeline.txt).  The smaller tower
                # on the one-workgroup recurrences instead: its few short sequences cost it ~90 us off the critical path.
                self.use_one_workgroup()
            if self.needs_order:   # the backward: the smaller towers on the one-workgroup recurrence until the sum fits
                left = dict(self._need)
                for e in sorted(left, key=lambda e: (left[e], self.rows.get(e, 0))):
                    if sum(left.values()) <= cus:
                        break
                    if left[e]:
                        e.one_workgroup_bwd, left[e] = True, 0
        return self

    def use_one_workgroup(self) -> None:
        """The plan for calls that cannot carry events: the smaller towers on the one-workgroup recurrences until the sum fits."""
        if not self.needs_order:
            return
        cus = torch.cuda.get_device_properties(self.model.query_encoder.embedding.weight.device).multi_processor_count
        need = self._need
        for e in sorted(need, key=lambda e: (need[e], self.rows.get(e, 0))):  # smallest first (fewest rows among equals)
            if sum(need.values()) <= cus:
                break
            if need[e]:
                e.one_workgroup, e.one_workgroup_bwd, need[e] = True, None, 0
        self.needs_order = False

    def __exit__(self, exc_type, exc, tb):
        for e, one, one_bwd, sink in self._saved:
            e.one_workgroup, e.one_workgroup_bwd, e._status_sink = one, one_bwd, sink
        if exc_type is not None and isinstance(self.optimizer, _FlatClipAdam):
            self.optimizer._pending_status.clear()  # (a step that died before optimizer.step(): its words are nobody's)
            self.optimizer._reduced_upto = 0
        return False


def _train_step_direct(model: TwoTowerModel, optimizer, queries, pos_docs, neg_docs, margin: float, phase: str = "all",
                       join_on_caller: bool = False, both: Optional[torch.Tensor] = None, plan: Optional[_towers_in_flight] = None,
                       seed_words: Optional[dict] = None):
    """The same step without the autograd engine: tower forwards (train mode), the fused loss + gradient kernel, tower backwards
    written STRAIGHT into the optimizer's flat gradient buffer, optimizer step.  Every parameter receives its gradient exactly
    once (query tower once; positives and negatives as one 2B-row document-tower call), so nothing has to be zeroed or
    accumulated: what the autograd path adds per step -- grad_output plumbing around the loss, four accumulate-adds per
    tower, the 3.4 MB zero fill -- is ~0.1 ms of small launches on the document tower's critical path.  Returns None when the
    shortcut does not apply (another optimizer, a trainable embedding table, parameters without the optimizer's gradient
    views): the caller then takes the autograd path, which computes the same numbers.
    phase: "all" = through optimizer.step(); "fold" = up to the gate words behind the gradients (optimizer._fold_status), the
    caller reduces and applies (GraphedTrainStep with a process group: the all-reduce stays outside the captured graph).
    join_on_caller: the towers' streams meet on the CALLER's stream (loss and optimizer run there) instead of on the document
    tower's.  Eagerly that costs two hops per join on the critical path (~20 us each); under stream capture it is free (the hops
    are graph edges) and it is the only form this ROCm can capture: a side stream that waits for ANOTHER side stream and then
    goes on makes hipStreamEndCapture crash inside the runtime (tools/experiments/graph_pattern_probe.py: fork / join through
    the origin stream is fine, side-to-side joins segfault, whatever the kernels).
    both: the 2B-row document batch [pos_docs; neg_docs] already concatenated (GraphedTrainStep stages it outside the graph).
    plan: the step's _towers_in_flight; when the two towers' split recurrences do not fit the device together their recurrence
    launches are ordered with events (query tower's first in the forward, last in the backward).
    seed_words: {id(encoder): device int64[1]} -- the towers' dropout seeds as device words the kernels read when they run
    (TT_ENC_SEED_ON_DEVICE: GraphedTrainStep writes a fresh seed there before every replay); default: drawn here, passed by value."""
    if not isinstance(optimizer, _FlatClipAdam) or not torch.is_grad_enabled():
        return None
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
    if both is None and (pos_docs.dtype != torch.int64 or neg_docs.dtype != torch.int64):
        return None
    s_doc, s_qry = streams
    if join_on_caller and plan is not None:
        # a capture: no ordering events (an event edge between the two tower streams is a side-to-side edge, which this ROCm
        # cannot capture; with the query tower moved onto the capturing stream it can, but replays slower: 1.19-1.21 ms against
        # 1.16 with the smaller tower on the one-workgroup recurrences, profiles/r04_m_graph_probe.log)
        plan.use_one_workgroup()
    s_main = cur if join_on_caller else s_doc     # where the towers meet: loss, optimizer
    ordered = plan is not None and plan.needs_order
    sync_f = sync_b = {}
    fwd_opts = {}
    two_phase = _TWO_PHASE
    if ordered:
        ev_f, ev_b = _order_events(dev)
        q_enc, d_enc = model.query_encoder, model.doc_encoder
        if _FWD_ORDER == "doc_first":
            sync_f = {id(d_enc): _lib.EncSync(None, ev_f), id(q_enc): _lib.EncSync(ev_f, None)}   # document records, query waits
            two_phase = None
        elif _FWD_ORDER == "query_one_wg":
            fwd_opts[id(q_enc)] = q_enc._opts() | _lib.TT_ENC_ONE_WORKGROUP
            two_phase = None
        else:
            sync_f = {id(q_enc): _lib.EncSync(None, ev_f), id(d_enc): _lib.EncSync(ev_f, None)}   # query records, document waits
        # (backward: no events -- the plan gave the smaller tower the one-workgroup recurrence, _towers_in_flight)

    def join():   # s_main waits for both towers
        for t in (s_qry, s_doc):
            if t is not s_main:
                s_main.wait_stream(t)

    def fork():   # both towers wait for s_main
        for t in (s_qry, s_doc):
            if t is not s_main:
                t.wait_stream(s_main)
    try:
        s_doc.wait_stream(cur)
        if both is None:
            with torch.cuda.stream(s_doc):  # (on the document tower's own stream: no hop from the caller's)
                pos_docs.record_stream(s_doc)
                neg_docs.record_stream(s_doc)
                both = _concat_ids(pos_docs, neg_docs)
        ids_of = (both, queries)
        # dropout seeds from torch's CPU generator in the order the autograd path draws them (query tower, then document tower)
        seeds = seed_words if seed_words is not None else _draw_dropout_seeds(model)
        # Issue order: the document tower (2B rows of ~70 tokens: the step's critical path) FIRST, the query tower's ~15 small
        # launches then overlap it instead of delaying it by the ~0.1 ms the host needs to issue them.  Ordered recurrences: the
        # call that RECORDS the ordering event must be issued before the WAIT for it (include/tt.h), so the document tower's call
        # goes out in two halves around the query tower's: BEGIN (prep, weight conversion, input projection: 130 us of GPU
        # work), the query tower (records behind its recurrence), FINISH (waits, then recurrence + head).
        fw = [None, None]
        begun = None
        for k, half in (((0, 0), (1, 0)) if (not ordered or two_phase is None) else
                        (((0, _lib.TT_ENC_PHASE_BEGIN), (1, 0), (0, _lib.TT_ENC_PHASE_FINISH)) if two_phase else ((1, 0), (0, 0)))):
            enc, ids, s = encs[k], ids_of[k], streams[k]
            if s is not cur and half != _lib.TT_ENC_PHASE_FINISH:
                s.wait_stream(cur)
            with torch.cuda.stream(s):
                ids.record_stream(s)
                p_drop = enc.dropout
                seed = seeds[id(enc)]
                # (the encoders are watched -- _towers_in_flight -- so the status word goes to the optimizer instead of a read here)
                res = enc._run_forward(ids, train=True, dropout_p=p_drop, dropout_seed=seed, sync=sync_f.get(id(enc)) if half != _lib.TT_ENC_PHASE_BEGIN else None,
                                       phase=half, resume=begun if half == _lib.TT_ENC_PHASE_FINISH else None, opts=fwd_opts.get(id(enc)))
                if half == _lib.TT_ENC_PHASE_BEGIN:
                    begun = res
                    continue
                out, ws, status = res
                fw[k] = (out, ws, status, p_drop, seed, enc._opts_bwd())
        # The document tower's stream carries the step's critical path from here on: the loss, the document backward and the
        # optimizer are enqueued on IT (a hop to the caller's stream and back cost ~20 us each way on that path: event wait +
        # launch); the query tower's stream joins for the loss and again before the optimizer.
        pn, q = fw[0][0], fw[1][0]
        p, n = pn[:B], pn[B:]
        H = q.shape[1]
        join()
        with torch.cuda.stream(s_main):
            loss = torch.empty((), dtype=torch.float32, device=dev)
            dq = torch.empty_like(q)
            dpn = torch.empty_like(pn)
            rows = torch.empty(B, dtype=torch.float32, device=dev)
            for t in (q, pn, loss, dq, dpn, rows):
                t.record_stream(s_main)
            with torch.cuda.device(dev):
                _lib.check(_lib.lib().tt_triplet_loss_f32(q.data_ptr(), p.data_ptr(), n.data_ptr(), B, H, float(margin), loss.data_ptr(),
                                                          dq.data_ptr(), dpn[:B].data_ptr(), dpn[B:].data_ptr(), rows.data_ptr(),
                                                          s_main.cuda_stream))
        fork()
        for enc, ids, s, f, d_out, grads in zip(encs, ids_of, streams, fw, (dpn, dq), into):
            with torch.cuda.stream(s):
                d_out.record_stream(s)
                # (a time-out of the split backward recurrence ORs bit 2 into the forward's word, which the optimizer reads)
                enc._run_backward(ids.contiguous(), f[1], d_out, f[3], f[4], into=grads, status=f[2], opts=f[5],
                                  sync=sync_b.get(id(enc)))
                if enc is model.query_encoder and optimizer.world > 1 and not join_on_caller:
                    # data-parallel: the query tower's gradients are complete long before the document tower's (its backward is
                    # a tenth of the tokens) -- when they are the bucket's prefix their all-reduce goes out HERE, on the query
                    # tower's stream, and runs under the document tower's weight-gradient kernels; the rest of the bucket (with
                    # the gate words) follows in optimizer.step()
                    base = optimizer.flat_grads.data_ptr()
                    spans = sorted(((g.data_ptr() - base) // 4, g.numel()) for g in grads)
                    if spans[0][0] == 0 and all(a + n == b for (a, n), (b, _) in zip(spans, spans[1:])):
                        optimizer.reduce_prefix(spans[-1][0] + spans[-1][1])
        join()
        for p_, gv in zip(optimizer.params, optimizer._views):
            p_.grad = gv
        with torch.cuda.stream(s_main):
            # The optimizer is enqueued unconditionally: the towers' status words (zero-length rows / ids out of range raise as
            # in the reference; a column-split recurrence that gave up) are folded into the gate behind the gradients, reduced
            # over the ranks with them, and the device applies the step on every rank or on none.  The read of the reduced gate
            # inside step() is the one host synchronisation of the step; for a bad batch the weights stay untouched (the
            # gradient buffer holds garbage, which the next step overwrites).
            for t in (optimizer.flat_params, optimizer._bucket, optimizer.exp_avg, optimizer.exp_avg_sq, optimizer.total_norm,
                      optimizer._scratch, optimizer._step_dev):
                t.record_stream(s_main)
            if phase == "all":
                optimizer.step()
            else:
                optimizer._fold_status()
    finally:
        # (also when step() raised: the caller's stream joins the towers' streams, nothing of this step is left running
        #  behind the caller's back)
        for t in (s_qry, s_doc):
            if t is not cur:
                cur.wait_stream(t)
    loss.record_stream(cur)
    return loss


def _draw_dropout_seeds(model: TwoTowerModel) -> dict:
    """{id(encoder): seed} from torch's CPU generator in the order the autograd path draws them (query tower, then document
    tower); 0 for a tower without inter-layer dropout (nothing is drawn for it)."""
    return {id(enc): (int(torch.randint(0, 2 ** 62, (1,)).item()) if enc.dropout > 0.0 else 0)
            for enc in (model.query_encoder, model.doc_encoder)}


def _train_step_once(model, optimizer, queries, pos_docs, neg_docs, margin, concurrent_towers, direct, plan=None):
    if direct and concurrent_towers and queries.is_cuda and neg_docs.shape[0] == pos_docs.shape[0] == queries.shape[0]:
        loss = _train_step_direct(model, optimizer, queries, pos_docs, neg_docs, margin, plan=plan)
        if loss is not None:
            return loss
    if plan is not None:
        plan.use_one_workgroup()   # (autograd calls cannot carry the ordering events)
    optimizer.zero_grad()
    watched = isinstance(optimizer, _FlatClipAdam)
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

        def launch():
            for s, (fn, ids) in zip(_tower_streams(queries.device), calls):
                s.wait_stream(cur)
                with torch.cuda.stream(s):
                    ids.record_stream(s)
                    outs.append(fn(ids))
            for s, o in zip(_tower_streams(queries.device), outs):
                cur.wait_stream(s)
                o.record_stream(cur)
        if watched:
            launch()  # (the status words go to the optimizer, which reads them -- reduced over the ranks -- in step())
        else:
            # another optimizer: input checking stays on (zero-length rows / out-of-range ids raise as in the reference), but the
            # status words of the towers are read ONCE, after all of them have been enqueued: a per-call read would make the
            # host wait for the query tower before it could launch the document tower
            with deferred_input_checks(model.query_encoder, model.doc_encoder):
                launch()
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


def train_step(model: TwoTowerModel, optimizer: FusedClipAdam, queries: torch.Tensor, pos_docs: torch.Tensor,
               neg_docs: torch.Tensor, margin: float = 0.2, concurrent_towers: bool = True, direct: bool = True,
               defer_check: bool = False) -> torch.Tensor:
    """One step of backend/main.py:244-259 on this rank's (equal-sized) share of the global batch.
    Returns the local loss as a 0-d device tensor (no .item(): the reference's per-step sync is dropped).

    concurrent_towers: the encoder calls are independent, so the query tower and the document tower (positives and negatives in
    one 2B-row call) are issued on separate HIP streams (autograd replays each call's backward on the stream its forward ran on).
    direct: skip the autograd engine when the step has the standard shape (_train_step_direct: same kernels, same numbers).

    Failure is collective (FusedClipAdam): a zero-length row or an id out of range on ANY rank raises the reference's
    exception on EVERY rank, inside this call, with parameters, moments and step number untouched everywhere.  A time-out of a
    column-split recurrence (transient: CUs held by other work) does not raise: all ranks see it in the same reduced gate and
    redo the step on the one-workgroup kernels (ordinary relaunch, same bits as the split forward).

    defer_check (FusedClipAdam only): the step's one host synchronisation -- the read of the reduced gate words -- moves one call
    back: the words are copied to pinned memory behind the step and looked at inside the NEXT call (or optimizer.settle()), after
    that step has been enqueued, so the host runs ahead of the GPU and the ~0.1 ms the GPU idles per step while the host
    enqueues the next 40 launches is gone.  A bad batch's exception then comes out one call late; its step was not applied
    (the device decided that), the step behind it is an ordinary step on the same weights."""
    from .model import SplitRecurrenceTimeout
    if defer_check and isinstance(optimizer, _FlatClipAdam) and queries.is_cuda:
        keep, optimizer.check = optimizer.check, False
        try:
            loss = train_step(model, optimizer, queries, pos_docs, neg_docs, margin, concurrent_towers, direct)
        finally:
            optimizer.check = keep
        if keep:
            optimizer.snapshot_gate(redo=lambda: train_step(model, optimizer, queries, pos_docs, neg_docs, margin, concurrent_towers, direct))
        return loss
    encs = (model.query_encoder, model.doc_encoder)
    n_doc = pos_docs.shape[0] + neg_docs.shape[0] if (concurrent_towers and neg_docs.shape[0] == pos_docs.shape[0]) else max(
        pos_docs.shape[0], neg_docs.shape[0])
    rows = {model.query_encoder: queries.shape[0], model.doc_encoder: n_doc} if concurrent_towers else {}
    rng = torch.get_rng_state() if any(e.dropout > 0.0 and e.training for e in encs) else None
    try:
        with _towers_in_flight(model, optimizer, rows if queries.is_cuda else {}) as plan:
            return _train_step_once(model, optimizer, queries, pos_docs, neg_docs, margin, concurrent_towers, direct, plan)
    except SplitRecurrenceTimeout:
        if rng is not None:
            torch.set_rng_state(rng)  # (the redone step draws the dropout seeds the failed attempt drew)
        with _towers_in_flight(model, optimizer, {}, force_one_workgroup=True) as plan:
            return _train_step_once(model, optimizer, queries, pos_docs, neg_docs, margin, concurrent_towers, direct, plan)


class DataParallelTrainer:
    """Replicated model, per-rank batch shard, gradient all-reduce (with the step's failure gate) inside FusedClipAdam.step().

    graphs=True: steps are replayed from HIP graphs (GraphedTrainStep), one per (batch, query width, document width) bucket --
    widths rounded up to `width_step` columns, the `max_graphs` most recently used buckets kept (a graph owns its workspaces:
    ~0.3 GB at 512 triplets x 128 columns).  A batch that does not fit the rules of GraphedTrainStep (unchecked inputs,
    non-int64 ids, a trainable table) takes the eager train_step; results are the eager step's on ids padded to the bucket's widths.
    defer_check: see GraphedTrainStep (exceptions one call late; call flush() after the last step)."""

    def __init__(self, model: TwoTowerModel, lr: float = 1e-4, margin: float = 0.2, max_norm: float = 1.0, group=None,
                 graphs: bool = False, width_step: int = 32, max_graphs: int = 4, defer_check: bool = False):
        self.model = model
        self.margin = margin
        self.optimizer = FusedClipAdam(model.parameters(), lr=lr, max_norm=max_norm, group=group)
        self.optimizer.watch(model)  # (a hand-written loop over self.model / self.optimizer fails collectively too)
        self.graphs, self.width_step, self.max_graphs, self.defer_check = bool(graphs), int(width_step), int(max_graphs), bool(defer_check)
        self._graphs: "dict" = {}

    def broadcast_parameters(self, src: int = 0) -> None:
        import torch.distributed as dist
        if dist.is_initialized() and dist.get_world_size(self.optimizer.group) > 1:
            dist.broadcast(self.optimizer.flat_params, src=src, group=self.optimizer.group)
        self.optimizer.mark_params_changed()  # (the broadcast wrote the flat buffer, not the parameter tensors)

    def _graph_for(self, queries, pos_docs, neg_docs) -> Optional["GraphedTrainStep"]:
        encs = (self.model.query_encoder, self.model.doc_encoder)
        if not (queries.is_cuda and queries.dtype == pos_docs.dtype == neg_docs.dtype == torch.int64
                and queries.shape[0] == pos_docs.shape[0] == neg_docs.shape[0]
                and all(e.check_inputs and not e.embedding.weight.requires_grad for e in encs)):
            return None
        up = lambda x: max(self.width_step, -(-int(x) // self.width_step) * self.width_step)  # noqa: E731
        key = (queries.shape[0], up(queries.shape[1]), up(max(pos_docs.shape[1], neg_docs.shape[1])))
        g = self._graphs.pop(key, None)
        if g is None:
            while len(self._graphs) >= self.max_graphs:           # least recently used first
                self._graphs.pop(next(iter(self._graphs)))
            g = GraphedTrainStep(self.model, self.optimizer, key[0], key[1], key[2], self.margin, defer_check=self.defer_check)
        self._graphs[key] = g                                      # (re-inserted: most recently used last)
        return g

    def flush(self):
        """defer_check: settle the last step's pending check."""
        return self.optimizer.settle()

    def step(self, queries, pos_docs, neg_docs) -> torch.Tensor:
        self.model.train()
        if self.graphs:
            g = self._graph_for(queries, pos_docs, neg_docs)
            if g is not None:
                return g(queries, pos_docs, neg_docs)
        return train_step(self.model, self.optimizer, queries, pos_docs, neg_docs, self.margin, defer_check=self.defer_check)


def _stage_ids(out: torch.Tensor, a: torch.Tensor, b: Optional[torch.Tensor] = None) -> None:
    """out [Ba + Bb, T] <- rows of a, then rows of b, zero-padded to out's width: ONE launch (tt_concat_ids_i64) on the current
    stream, into an existing buffer (GraphedTrainStep's static inputs)."""
    a = a.contiguous()
    b = b.contiguous() if b is not None else None
    with torch.cuda.device(out.device):
        _lib.check(_lib.lib().tt_concat_ids_i64(a.data_ptr(), a.shape[0], a.shape[1], b.data_ptr() if b is not None else None,
                                                b.shape[0] if b is not None else 0, b.shape[1] if b is not None else 0,
                                                out.data_ptr(), out.shape[1], torch.cuda.current_stream(out.device).cuda_stream))


class GraphedTrainStep:
    """The direct train step of ONE batch shape captured in a HIP graph and replayed: the ~40 launches of a step (two towers on
    two streams, loss, two backwards, gate, clip + Adam) become one graph launch, so the host is out of the step's way -- what
    is left per step is two staging launches (the batch's ids into static buffers, zero-padded), one hipGraphLaunch and the read
    of the gate words (the eager step leaves the GPU idle for ~0.1 ms per step while the host enqueues the next step's launches
    behind its one synchronisation).  Possible because nothing in the captured launches changes from step to step: the step number
    and the bias corrections live on the device (tt_clip_adam_step_gated_f32), the failure decision is a device predicate (the
    gate), and every library call is capture-clean (no allocation, no synchronisation, kernels instead of memsets: include/tt.h).

        step = GraphedTrainStep(model, optimizer, batch=512, q_width=32, doc_width=128, margin=0.5)
        loss = step(queries, pos_docs, neg_docs)      # ids [batch, <= width]

    Padding columns of id 0 change nothing (lengths are counts of non-zero ids, backend/model.py:52): results are bit-identical
    to train_step on ids padded to the same widths (the launch geometry -- slab partitions of the weight-gradient products --
    follows the padded width, so against the UNPADDED eager step the last bits of a gradient sum may differ).
    Failure semantics are train_step's: the reduced gate is read after the replay (the step's one synchronisation) and the
    reference's exceptions are raised, on every rank, with the weights untouched; a recurrence time-out redoes the step eagerly
    on the one-workgroup kernels.  defer_check=True moves that read one step back: step i's gate words are copied to pinned host
    memory behind its replay and looked at inside call i + 1, AFTER step i + 1 has been enqueued -- the GPU never waits for the
    host, and a bad batch's exception comes out of the NEXT call (or of flush()); its step was not applied, the following one
    (already enqueued) is an ordinary step on the same weights.  With a process group the graph ends at the gate words and the
    all-reduce + clip + Adam are issued eagerly behind it (RCCL inside a capture is not rehearsable here).  Needs: the
    optimizer's own parameters on a GPU, GloVe-frozen tables, input checking on (default).  Inter-layer dropout (the reference's
    default model has it) is captured with its seeds as device words (TT_ENC_SEED_ON_DEVICE, include/tt.h): every call draws them
    from torch's CPU generator in the eager step's order and copies them in front of the replay, so a replay computes what the
    eager step would have computed with the generator in the same state."""

    def __init__(self, model: TwoTowerModel, optimizer: FusedClipAdam, batch: int, q_width: int, doc_width: int, margin: float = 0.2,
                 defer_check: bool = False):
        encs = (model.query_encoder, model.doc_encoder)
        if not isinstance(optimizer, _FlatClipAdam) or optimizer._gated_step_fn is None:
            raise TypeError("GraphedTrainStep needs a FusedClipAdam")
        if not all(e.check_inputs for e in encs):
            raise ValueError("GraphedTrainStep: input checking must be on (the gate is what keeps a bad batch from being applied)")
        self.model, self.optimizer, self.margin = model, optimizer, float(margin)
        self.B, self.q_width, self.doc_width = int(batch), int(q_width), int(doc_width)
        self.defer_check = bool(defer_check)
        dev = optimizer.flat_params.device
        self.q = torch.zeros((self.B, self.q_width), dtype=torch.int64, device=dev)
        self.both = torch.zeros((2 * self.B, self.doc_width), dtype=torch.int64, device=dev)
        # inter-layer dropout: the seeds are device words the captured kernels read when they run (TT_ENC_SEED_ON_DEVICE); every
        # call draws them from torch's CPU generator exactly as the eager step does and copies them here in front of the replay
        self._seed_words = torch.zeros(2, dtype=torch.int64, device=dev) if any(e.dropout > 0.0 for e in encs) else None
        self._seed_host = torch.zeros((8, 2), dtype=torch.int64).pin_memory() if self._seed_words is not None else None
        self._seed_events, self._seed_step = [None] * 8, 0
        self._phase = "all" if optimizer.world == 1 else "fold"
        # (the optimizer's hyper-parameters are arguments of the captured launches: changing them means capturing again)
        self._hyper = (optimizer.lr, optimizer.betas, optimizer.eps, optimizer.max_norm)
        model.train()
        keep_check, optimizer.check = optimizer.check, False        # (no host read inside a capture)
        try:
            # the static buffers hold id 0 only: every row is a zero-length row, the gate closes and neither the warm-up run nor
            # anything else before the first real batch touches the weights
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(2):                                   # allocator pools, kernel attributes, lazy handles
                    if self._run() is None:
                        raise ValueError("GraphedTrainStep: the model / optimizer pair does not take the direct step "
                                         "(trainer._train_step_direct)")
                    if self._phase == "fold":
                        optimizer.gate.zero_()
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.loss = self._run()
        finally:
            optimizer.check = keep_check
            optimizer._pending_status.clear()

    def _run(self):
        rows = {self.model.query_encoder: self.B, self.model.doc_encoder: 2 * self.B}
        words = None
        if self._seed_words is not None:
            words = {id(self.model.query_encoder): self._seed_words[0:1], id(self.model.doc_encoder): self._seed_words[1:2]}
        with _towers_in_flight(self.model, self.optimizer, rows) as plan:
            return _train_step_direct(self.model, self.optimizer, self.q, None, None, self.margin, phase=self._phase,
                                      join_on_caller=True, both=self.both, plan=plan, seed_words=words)

    def flush(self):
        """defer_check: settle the optimizer's pending check (raises that step's exception, if any)."""
        return self.optimizer.settle()

    def __call__(self, queries: torch.Tensor, pos_docs: torch.Tensor, neg_docs: torch.Tensor) -> torch.Tensor:
        """Returns the step's loss (a static 0-d device tensor, valid until the next call)."""
        for src, width, what in ((queries, self.q_width, "queries"), (pos_docs, self.doc_width, "pos_docs"), (neg_docs, self.doc_width, "neg_docs")):
            if src.dim() != 2 or src.shape[0] != self.B or src.shape[1] > width or src.dtype != torch.int64:
                raise ValueError(f"GraphedTrainStep was captured for int64 {what} [{self.B}, <= {width}], got {src.dtype} {tuple(src.shape)}")
        opt = self.optimizer
        if self._phase == "all" and (opt.lr, opt.betas, opt.eps, opt.max_norm) != self._hyper:
            raise ValueError("GraphedTrainStep: the optimizer's lr / betas / eps / max_norm changed since the capture (they are "
                             "arguments of the captured launches); build a new GraphedTrainStep")
        _stage_ids(self.q, queries)
        _stage_ids(self.both, pos_docs, neg_docs)
        if self._seed_words is not None:
            drawn = _draw_dropout_seeds(self.model)
            k = self._seed_step % self._seed_host.shape[0]
            self._seed_step += 1
            if self._seed_events[k] is not None:
                self._seed_events[k].synchronize()   # (this pinned pair was a copy's source eight calls ago: long done)
            self._seed_host[k, 0], self._seed_host[k, 1] = drawn[id(self.model.query_encoder)], drawn[id(self.model.doc_encoder)]
            self._seed_words.copy_(self._seed_host[k], non_blocking=True)
            self._seed_events[k] = torch.cuda.Event()
            self._seed_events[k].record()
        self.graph.replay()
        if self._phase == "fold":
            opt._reduce_apply(True, check=False)
        opt.mark_params_changed()
        if not opt.check:
            return self.loss
        # (the eager redo takes the one-workgroup retry itself if it must)
        opt.snapshot_gate(redo=lambda: train_step(self.model, opt, queries, pos_docs, neg_docs, self.margin))
        if not self.defer_check:
            redone = opt.settle()            # the step's one host synchronisation
            if redone is not None:
                return redone
        return self.loss
