"""Drop-in mirror of the reference's backend/model.py public surface, computed by libtt.so.

    RNNEncoder(vocab_size, embed_dim, hidden_dim, pretrained_embeddings=None, rnn_type='GRU',
               num_layers=1, dropout=0.0, bidirectional=False, normalize_output=True)   model.py:11-46
    RNNEncoder.forward(LongTensor[B,T]) -> FloatTensor[B,H]                              model.py:48-75
    TwoTowerModel(config, pretrained_embeddings).encode_query / encode_document / forward model.py:78-106
    triplet_loss_cosine((q, p, n), margin=0.2) -> 0-d tensor                             model.py:109-114

Same constructor arguments, attribute names (`query_encoder.embedding.embedding_dim` is read by
backend/main.py:104) and state_dict keys ({query,doc}_encoder.{embedding.weight,
rnn.weight_ih_l0[_reverse], ..., projection.weight/bias}), so a model.pth written by either side
loads in the other.  The nn.Module objects here only OWN the parameters; every forward/backward
runs in hand-written HIP kernels (csrc/encoder*.hip, gru16.hip, train.hip).  There is no PyTorch fallback:
CPU tensors raise.  RNN_TYPE GRU, LSTM and RNN (tanh) are the three the reference's getattr(nn, ...) accepts.
"""
from __future__ import annotations

import ctypes as C
import math
import threading
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import _lib

__all__ = ["RNNEncoder", "TwoTowerModel", "triplet_loss_cosine", "SplitRecurrenceTimeout"]


class SplitRecurrenceTimeout(RuntimeError):
    """Status bit 2: a column-split GRU recurrence (csrc/gru16x4.hip) gave up waiting for a partner workgroup -- a transient
    condition of the device (CUs held by other work), not of the data.  The call's outputs are invalid; the same call with
    `one_workgroup` recurrences (TT_ENC_ONE_WORKGROUP) cannot time out and returns the same bits.  RNNEncoder's inference
    forward and trainer.train_step redo the call that way themselves; this exception reaches the caller of a hand-written
    autograd loop only."""


def _stream(dev) -> int:
    return torch.cuda.current_stream(dev).cuda_stream


_PREP_LOCK = threading.Lock()  # module level: nn.Module instances must stay deep-copyable / picklable

_CELLS = {"GRU": (0, 3), "LSTM": (1, 4), "RNN": (2, 1)}  # RNN_TYPE -> (C-ABI rnn_type, gate rows / H)


class _GRUParams(nn.Module):
    """Parameter container with torch.nn.GRU's / nn.LSTM's / nn.RNN's names, shapes (gates * H rows) and default init
    (U(-1/sqrt(H), 1/sqrt(H)))."""

    def __init__(self, input_size: int, hidden_size: int, num_layers: int, bidirectional: bool, gates: int = 3):
        super().__init__()
        self.input_size, self.hidden_size = input_size, hidden_size
        self.num_layers, self.bidirectional = num_layers, bidirectional
        k = 1.0 / math.sqrt(hidden_size)
        ndir = 2 if bidirectional else 1
        for layer in range(num_layers):
            I = input_size if layer == 0 else ndir * hidden_size
            for d in range(ndir):
                sfx = f"_l{layer}" + ("_reverse" if d else "")
                for name, shape in (("weight_ih", (gates * hidden_size, I)), ("weight_hh", (gates * hidden_size, hidden_size)),
                                    ("bias_ih", (gates * hidden_size,)), ("bias_hh", (gates * hidden_size,))):
                    self.register_parameter(name + sfx, nn.Parameter(torch.empty(shape).uniform_(-k, k)))

    def quads(self):
        """[(W_ih, W_hh, b_ih, b_hh)] ordered (layer, dir): the order of the C ABI's `weights` array."""
        out = []
        for layer in range(self.num_layers):
            for d in range(2 if self.bidirectional else 1):
                sfx = f"_l{layer}" + ("_reverse" if d else "")
                out.append(tuple(getattr(self, n + sfx) for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")))
        return out


class _LinearParams(nn.Module):
    """Parameter container with nn.Linear's names and default init."""

    def __init__(self, in_features: int, out_features: int):
        super().__init__()
        k = 1.0 / math.sqrt(in_features)
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features).uniform_(-k, k))
        self.bias = nn.Parameter(torch.empty(out_features).uniform_(-k, k))


def _ptr_array(tensors):
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


class deferred_input_checks:
    """Context manager: the encoders' per-call status reads (one D2H sync each) are collected instead and checked
    ONCE on exit, so independent tower calls can be enqueued back to back on several streams.  The same exceptions
    are raised, only later: after all the towers of a step have been launched, before anything consumes them."""

    def __init__(self, *encoders: "RNNEncoder"):
        self.encoders = encoders

    def __enter__(self):
        self._box = []
        for e in self.encoders:
            e._deferred_status = self._box
        return self

    def __exit__(self, exc_type, exc, tb):
        for e in self.encoders:
            e._deferred_status = None
        if exc_type is None and self._box:
            _raise_status(_or_all(self._box))
        return False


def _or_all(status_tensors) -> int:
    acc = 0
    for v in torch.cat([s.reshape(1) for s in status_tensors]).tolist():  # one D2H copy for all of them
        acc |= int(v)
    return acc


def _raise_status(status: int) -> None:
    """Data errors first (they are properties of the batch: the reference's exceptions), the transient time-out last."""
    if status & 2:
        raise IndexError("index out of range in self")  # nn.Embedding's message (tests/golden/g10_errors.json)
    if status & 1:
        raise RuntimeError("Length of all samples has to be greater than 0, but found an element in 'lengths' "
                           "that is <= 0")  # pack_padded_sequence's message (model.py:55-57)
    if status & 4:
        raise SplitRecurrenceTimeout("libtt: a column-split GRU recurrence gave up waiting for a partner workgroup (bounded "
                                     "hand-off sweep, csrc/gru16x4.hip); the call's outputs are invalid.  Set "
                                     "encoder.one_workgroup = True (TT_ENC_ONE_WORKGROUP) and call again: same bits, no hand-off")


class _EncoderFn(torch.autograd.Function):
    """forward = tt_encoder_forward_f32 (train=1), backward = tt_encoder_backward_f32."""

    @staticmethod
    def forward(ctx, enc: "RNNEncoder", ids: torch.Tensor, *params: torch.Tensor):
        p = enc.dropout if enc.training else 0.0
        # the mask stream is seeded from torch's default CPU generator (torch.manual_seed reproduces it)
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if p > 0.0 else 0
        # a status word handed to a watching optimizer is read in its step(), behind the backward; one that was checked here (or
        # at the end of a deferred_input_checks block) is re-read behind the backward for the backward's time-out bit
        ctx.handed_over = enc.check_inputs and enc._deferred_status is None and enc._status_sink is not None
        out, ws, status = enc._run_forward(ids, train=True, dropout_p=p, dropout_seed=seed)
        ctx.enc, ctx.ids, ctx.ws, ctx.status = enc, ids, ws, status
        ctx.dropout = (p, seed)
        ctx.opts = enc._opts_bwd()
        ctx.n_params = len(params)
        return out

    @staticmethod
    def backward(ctx, d_out: torch.Tensor):
        enc = ctx.enc
        grads = enc._run_backward(ctx.ids, ctx.ws, d_out.contiguous(), *ctx.dropout, status=ctx.status, opts=ctx.opts)
        ctx.ws = None
        if enc.check_inputs and not ctx.handed_over and enc._backward_may_time_out(ctx.ids.shape[0], ctx.opts):
            # the backward ORs bit 2 into the forward's word (include/tt.h); nobody else reads it on this path
            _raise_status(int(ctx.status.item()) & 4)
        return (None, None, *grads)


class RNNEncoder(nn.Module):
    """Text encoder: embedding gather -> (stacked / bidirectional) GRU | LSTM | RNN -> final hidden -> L2-normalise."""

    def __init__(self, vocab_size: int, embed_dim: int, hidden_dim: int,
                 pretrained_embeddings: Optional[np.ndarray] = None, rnn_type: str = "GRU", num_layers: int = 1,
                 dropout: float = 0.0, bidirectional: bool = False, normalize_output: bool = True, arith: str = "split16"):
        """The reference's signature (backend/model.py:11-22) plus `arith`: "split16" (default: GRU towers with H = 128 / 256
        take every product as three f16 MFMAs on fp16 hi/lo splits, fp32-grade) or "f32" (every product on the fp32-MFMA
        kernels: nn.GRU's own fp32 arithmetic, include/tt.h TT_ENC_F32; also settable later as `encoder.arith`)."""
        super().__init__()
        if arith not in ("split16", "f32"):
            raise ValueError(f"arith={arith!r} (supported: 'split16', 'f32')")
        self.arith = arith
        self.embedding = nn.Embedding(vocab_size, embed_dim, padding_idx=0)  # parameter owner only
        if pretrained_embeddings is not None:
            # every row is copied, row 0 included (it is the GloVe word "the"), and the table is frozen
            self.embedding.weight.data.copy_(torch.from_numpy(np.asarray(pretrained_embeddings)))
            self.embedding.weight.requires_grad = False
        self.rnn_type = rnn_type.upper()
        if self.rnn_type not in _CELLS:  # the reference's getattr(nn, rnn_type.upper()) accepts exactly these three
            raise AttributeError(f"module 'torch.nn' has no attribute {self.rnn_type!r} usable as RNN_TYPE "
                                 "(supported: GRU, LSTM, RNN)")
        self._cell, gates = _CELLS[self.rnn_type]
        self.bidirectional = bidirectional
        self.num_layers = num_layers
        self.hidden_dim = hidden_dim
        self.dropout = float(dropout) if num_layers > 1 else 0.0
        self.rnn = _GRUParams(embed_dim, hidden_dim, num_layers, bidirectional, gates)
        self.normalize_output = normalize_output
        self.projection = _LinearParams(hidden_dim * 2, hidden_dim) if bidirectional else None
        self.check_inputs = True  # read the device status word after each call (one 4-byte D2H sync)
        # inference keeps the weights in kernel form (tt_encoder_prepare_f32) until they change: see _prepared_weights
        self.cache_prepared = True
        self._prep: dict = {}
        # inference, layer 0: the input projection of EVERY vocabulary row, computed once per weight version (the projected
        # table, include/tt.h: tt_encoder_project_table_f32) and gathered by the recurrence kernels, instead of a GEMM over the
        # batch's tokens in every call.  None = when it pays and is safe by construction: the table is frozen (GloVe loaded,
        # backend/model.py:25-27), the configuration has one (GRU, H = 128 / 256) and it fits PROJECTED_MAX_BYTES
        # (V x 3H x 4 bytes per direction: 1.23 GB for the north-star tower).  True / False force it on (any table: its version
        # counter is part of the cache key) / off.  Results are bit-identical either way.
        self.projected_table: Optional[bool] = None
        self._proj: dict = {}
        self._deferred_status = None  # a list while a caller (trainer.train_step) batches the status reads
        # a list owned by an optimizer that watches this encoder (FusedClipAdam.watch): TRAINING calls hand their status words
        # over instead of raising here, and optimizer.step() decides -- together with the other ranks -- whether the step happens
        self._status_sink = None
        # keep every recurrence of this encoder's calls on the one-workgroup kernels (TT_ENC_ONE_WORKGROUP, include/tt.h)
        self.one_workgroup = False
        self.one_workgroup_bwd = None  # None: as one_workgroup; True / False: the reverse-time recurrence alone (any mix is valid)

    # ---- plumbing ---------------------------------------------------------------
    def __getstate__(self):
        """copy.deepcopy / pickling of the module leave the per-device caches behind (kernel-form weights, the 1.2 GB projected
        table): the copy derives its own on first use."""
        state = self.__dict__.copy()
        state["_prep"], state["_proj"] = {}, {}
        return state

    def _flat_params(self):
        ps = [w for quad in self.rnn.quads() for w in quad]
        if self.projection is not None:
            ps += [self.projection.weight, self.projection.bias]
        return ps

    def _arith_bit(self) -> int:
        return _lib.TT_ENC_F32 if self.arith == "f32" else 0

    def _opts(self) -> int:
        return (_lib.TT_ENC_ONE_WORKGROUP if self.one_workgroup else 0) | self._arith_bit()

    def _opts_bwd(self) -> int:
        one = self.one_workgroup if self.one_workgroup_bwd is None else self.one_workgroup_bwd
        return (_lib.TT_ENC_ONE_WORKGROUP if one else 0) | self._arith_bit()

    def split_workgroups(self, B: int) -> int:
        """CUs the column-split recurrence of a B-row call of this encoder occupies (0: it runs the one-workgroup kernels)."""
        if self.arith == "f32":
            return 0
        return _lib.lib().tt_encoder_split_workgroups(int(B), self.hidden_dim, int(self.bidirectional), self._cell)

    def _backward_may_time_out(self, B: int, opts: int) -> bool:
        if opts & (_lib.TT_ENC_ONE_WORKGROUP | _lib.TT_ENC_F32):
            return False
        return _lib.lib().tt_encoder_split_workgroups(B, self.hidden_dim, int(self.bidirectional), self._cell) > 0

    def _check_device(self, x: torch.Tensor):
        if not x.is_cuda:
            raise RuntimeError("twotowermlretrieval_amd.RNNEncoder runs only on an AMD GPU via libtt.so "
                               f"(got ids on {x.device}; there is no CPU fallback)")
        for p in self.parameters():
            if p.device != x.device:
                raise RuntimeError(f"parameter on {p.device} but ids on {x.device}: call model.to(device) first")
            if p.dtype != torch.float32:
                raise TypeError("parameters must be float32")

    def invalidate_prepared(self) -> None:
        """Drop the cached kernel-form weights.  The cache is keyed on every weight's (address, tensor version counter), so
        in-place torch ops ON THE PARAMETER (`w.mul_()`, `w.copy_()` under no_grad), load_state_dict, .to() and
        FusedClipAdam.step / DataParallelTrainer.broadcast_parameters are noticed by themselves.  NOT noticed -- call this
        afterwards: writes through `w.data` (`w.data.copy_(...)` leaves the counter alone), through another view of the same
        storage (an optimizer's flat buffer: FusedClipAdam.mark_params_changed does the bump for its own), or through raw
        pointers."""
        self._prep = {}
        self._proj = {}

    PROJECTED_MAX_BYTES = 8 << 30

    def _projected_table(self, device: torch.device, quads, wptr, prepared: torch.Tensor) -> Optional[torch.Tensor]:
        """P[V, 3H] per direction = table W_ih^T + b_ih of layer 0, keyed on the table's and every weight's (address, version)."""
        if self.projected_table is False or (self.projected_table is None and self.embedding.weight.requires_grad):
            return None
        table = self.embedding.weight
        key = (table.data_ptr(), table._version) + tuple((w.data_ptr(), w._version) for w in quads)
        ent = self._proj.get(device)
        if ent is not None and ent[0] == key:
            return ent[1]
        if torch.cuda.is_current_stream_capturing():
            return None
        L = _lib.lib()
        V, E = table.shape
        need = L.tt_encoder_projected_bytes(V, E, self.hidden_dim, int(self.bidirectional), self._cell)
        if need == 0 or (self.projected_table is None and need > self.PROJECTED_MAX_BYTES):
            return None
        with _PREP_LOCK:
            ent = self._proj.get(device)
            if ent is not None and ent[0] == key:
                return ent[1]
            # (the table this one replaces is dropped here: every call records its stream on the table it reads -- _run_forward --
            #  so the caching allocator hands the 1.2 GB out again only when the calls queued on it have finished)
            self._proj.pop(device, None)
            del ent
            blob = torch.empty(need, dtype=torch.uint8, device=device)
            with torch.cuda.device(device):
                _lib.check(L.tt_encoder_project_table_f32(table.detach().data_ptr(), V, E, self.hidden_dim, self.num_layers,
                                                          int(self.bidirectional), self._cell, wptr, prepared.data_ptr(),
                                                          blob.data_ptr(), blob.numel(), _stream(device)))
                torch.cuda.current_stream(device).synchronize()  # once per weight version: any stream may read it now
            self._proj[device] = (key, blob)
        return blob

    def _prepared_weights(self, device: torch.device, quads, wptr) -> Optional[torch.Tensor]:
        """The weights converted once into what the forward kernels read (W_ih fp16 hi/lo fragment stream, packed W_hh,
        scale words): a serving process (query_inferencer.py:51-75) loads its weights once, and re-deriving them is six
        small launches, a quarter of a query-tower call.  Keyed on every weight's (address, version counter)."""
        key = tuple((w.data_ptr(), w._version) for w in quads)
        ent = self._prep.get(device)
        if ent is not None and ent[0] == key:
            return ent[1]
        if torch.cuda.is_current_stream_capturing():
            return None  # (no synchronisation inside a capture: this call derives them in its workspace)
        with _PREP_LOCK:
            ent = self._prep.get(device)
            if ent is not None and ent[0] == key:
                return ent[1]
            L = _lib.lib()
            E, H = self.embedding.weight.shape[1], self.hidden_dim
            need = L.tt_encoder_prepared_bytes(E, H, self.num_layers, int(self.bidirectional), self._cell)
            blob = torch.empty(max(need, 256), dtype=torch.uint8, device=device)
            with torch.cuda.device(device):
                _lib.check(L.tt_encoder_prepare_f32(E, H, self.num_layers, int(self.bidirectional), self._cell, wptr,
                                                    blob.data_ptr(), blob.numel(), _stream(device)))
                torch.cuda.current_stream(device).synchronize()  # once per weight version: any stream may read it now
            # the buffer it replaces stays referenced until the next replacement: calls queued on other streams may still read it
            self._prep[device] = (key, blob, ent[1] if ent is not None else None)
        return blob

    def _run_forward(self, x: torch.Tensor, train: bool, dropout_p: float = 0.0, dropout_seed: int = 0, opts: Optional[int] = None,
                     sync=None, phase: int = 0, resume=None):
        """sync: an _lib.EncSync (tt_enc_sync_t) that orders this call's recurrences against another stream's (training calls).
        phase: 0 = the whole call; _lib.TT_ENC_PHASE_BEGIN = everything in front of the first recurrence launch (returns the
        (out, ws, status) triple to pass back as `resume`); _lib.TT_ENC_PHASE_FINISH with resume = the rest (train mode only)."""
        L = _lib.lib()
        opts = self._opts() if opts is None else opts
        if isinstance(dropout_seed, torch.Tensor):   # a device int64: the kernels read the seed when they run (captured steps)
            seed_arg, opts = int(dropout_seed.data_ptr()), opts | _lib.TT_ENC_SEED_ON_DEVICE
        else:
            seed_arg = int(dropout_seed)
        ids = x.contiguous()
        if ids.dtype != torch.int64:
            ids = ids.to(torch.int64)
        if ids.dim() != 2:
            raise ValueError(f"expected ids [B,T], got {tuple(ids.shape)}")
        B, T = ids.shape
        V, E = self.embedding.weight.shape
        H = self.hidden_dim
        drop = int(train and dropout_p > 0.0 and self.num_layers > 1)
        # train = 2: the workspace also holds the gradient w.r.t. the gathered vectors (trainable embedding table)
        train_mode = (2 if self.embedding.weight.requires_grad else 1) if train else 0
        quads = [w.detach().contiguous() for quad in self.rnn.quads() for w in quad]
        wptr = _ptr_array(quads)
        prepared = self._prepared_weights(ids.device, quads, wptr) if (
            train_mode == 0 and self.cache_prepared and resume is None and not (opts & _lib.TT_ENC_F32)) else None
        projected = self._projected_table(ids.device, quads, wptr, prepared) if prepared is not None else None
        need = L.tt_encoder_workspace_bytes(B, max(T, 1), E, H, self.num_layers, int(self.bidirectional), self._cell,
                                            train_mode | (_lib.TT_ENC_PROJECTED if projected is not None else 0), drop)
        # One workspace per call, from torch's caching allocator (a cached block: microseconds).  In train mode the
        # autograd node owns it; in eval mode a per-call buffer keeps concurrent callers apart -- the reference serves
        # queries from a thread pool (frontend/main.py:103), and a buffer shared across threads or streams would be
        # overwritten by the next call's kernels while this call's are still queued.
        if resume is not None:
            out, ws, status = resume
        else:
            ws = torch.empty(max(need, 256), dtype=torch.uint8, device=ids.device)
            out = torch.empty((B, H), dtype=torch.float32, device=ids.device)
            # (not zeroed: every call that returns TT_OK WRITES the word on the stream -- include/tt.h, `status` -- and a call that
            #  does not raises below before anyone reads it; the fill was one more ~5 us launch in front of every tower call)
            status = torch.empty(1, dtype=torch.int32, device=ids.device)
        pw = self.projection.weight.detach().contiguous() if self.projection is not None else None
        pb = self.projection.bias.detach().contiguous() if self.projection is not None else None
        table = self.embedding.weight.detach()
        with torch.cuda.device(ids.device):
            if projected is not None:
                projected.record_stream(torch.cuda.current_stream(ids.device))
                _lib.check(L.tt_encoder_forward_projected_f32(
                    ids.data_ptr(), B, T, projected.data_ptr(), V, E, H, self.num_layers, int(self.bidirectional),
                    self._cell, wptr, prepared.data_ptr(), pw.data_ptr() if pw is not None else None,
                    pb.data_ptr() if pb is not None else None, int(self.normalize_output), opts, out.data_ptr(),
                    ws.data_ptr(), ws.numel(), status.data_ptr(), _stream(ids.device)))
            elif prepared is not None:
                _lib.check(L.tt_encoder_forward_prepared_f32(
                    ids.data_ptr(), B, T, table.data_ptr(), V, E, H, self.num_layers, int(self.bidirectional),
                    self._cell, wptr, prepared.data_ptr(), pw.data_ptr() if pw is not None else None,
                    pb.data_ptr() if pb is not None else None, int(self.normalize_output), opts, out.data_ptr(),
                    ws.data_ptr(), ws.numel(), status.data_ptr(), _stream(ids.device)))
            else:
                _lib.check(L.tt_encoder_forward_f32(
                    ids.data_ptr(), B, T, table.data_ptr(), V, E, H, self.num_layers, int(self.bidirectional),
                    self._cell, wptr, pw.data_ptr() if pw is not None else None,
                    pb.data_ptr() if pb is not None else None, int(self.normalize_output), train_mode | opts | phase,
                    float(dropout_p), seed_arg, out.data_ptr(), ws.data_ptr(), ws.numel(), status.data_ptr(),
                    C.byref(sync) if sync is not None else None, _stream(ids.device)))
        if phase == _lib.TT_ENC_PHASE_BEGIN:
            return out, ws, status   # (the status word is handed over / read when the FINISH half has been issued)
        if self.check_inputs:
            if self._deferred_status is not None:
                self._deferred_status.append(status)  # the caller reads them once, after enqueuing its other calls
            elif train and self._status_sink is not None:
                self._status_sink.append(status)  # the watching optimizer reads them, reduced over the ranks, in step()
            else:
                st = int(status.item())
                if (st & 7) == 4 and not train and not (opts & _lib.TT_ENC_ONE_WORKGROUP):
                    # inference: a transient time-out of the column-split recurrence and nothing wrong with the data -- the
                    # same call on the one-workgroup kernels (same bits, nothing to wait for)
                    return self._run_forward(x, train, dropout_p, dropout_seed, opts | _lib.TT_ENC_ONE_WORKGROUP, sync)
                _raise_status(st)
        return out, ws, status

    def _run_backward(self, ids: torch.Tensor, ws: torch.Tensor, d_out: torch.Tensor, dropout_p: float = 0.0,
                      dropout_seed: int = 0, into: Optional[list] = None, status: Optional[torch.Tensor] = None,
                      opts: Optional[int] = None, sync=None):
        """into: contiguous float32 tensors, one per _flat_params() entry, that receive the gradients (the C entry point
        OVERWRITES its gradient buffers): trainer.train_step hands over the optimizer's gradient views.
        status: the forward call's status word; a time-out of the column-split backward recurrence ORs bit 2 into it."""
        L = _lib.lib()
        opts = self._opts_bwd() if opts is None else opts
        if isinstance(dropout_seed, torch.Tensor):
            seed_arg, opts = int(dropout_seed.data_ptr()), opts | _lib.TT_ENC_SEED_ON_DEVICE
        else:
            seed_arg = int(dropout_seed)
        B, T = ids.shape
        V, E = self.embedding.weight.shape
        H = self.hidden_dim
        params = self._flat_params()
        grads = into if into is not None else [torch.empty_like(p, memory_format=torch.contiguous_format) for p in params]
        table = self.embedding.weight
        g_table = torch.empty_like(table, memory_format=torch.contiguous_format) if table.requires_grad else None
        nq = 4 * self.num_layers * (2 if self.bidirectional else 1)
        quads = [p.detach().contiguous() for p in params[:nq]]
        wptr, gptr = _ptr_array(quads), _ptr_array(grads[:nq])
        pw = params[nq].detach().contiguous() if self.projection is not None else None
        pb = params[nq + 1].detach().contiguous() if self.projection is not None else None
        with torch.cuda.device(ids.device):
            _lib.check(L.tt_encoder_backward_f32(
                ids.contiguous().data_ptr(), B, T, self.embedding.weight.detach().data_ptr(), V, E, H,
                self.num_layers, int(self.bidirectional), self._cell, wptr, pw.data_ptr() if pw is not None else None,
                pb.data_ptr() if pb is not None else None, int(self.normalize_output), float(dropout_p),
                seed_arg, d_out.data_ptr(), gptr,
                grads[nq].data_ptr() if pw is not None else None,
                grads[nq + 1].data_ptr() if pb is not None else None,
                g_table.data_ptr() if g_table is not None else None, ws.data_ptr(), ws.numel(), opts,
                status.data_ptr() if status is not None else None, C.byref(sync) if sync is not None else None,
                _stream(ids.device)))
        return grads + ([g_table] if g_table is not None else [])

    # ---- public -----------------------------------------------------------------
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        self._check_device(x)
        params = self._flat_params()
        if self.embedding.weight.requires_grad:
            # no GloVe vectors: the reference trains nn.Embedding(padding_idx=0) (model.py:23-27); the table is the
            # last differentiable input of the autograd node (dense [V,E] gradient, row 0 stays zero)
            params = params + [self.embedding.weight]
        needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        if needs_grad:
            return _EncoderFn.apply(self, x, *params)
        out, _, _ = self._run_forward(x, train=False)
        return out


class TwoTowerModel(nn.Module):
    """Two independent RNNEncoder towers (no weight sharing; each owns its embedding table copy)."""

    def __init__(self, config: Dict, pretrained_embeddings: Optional[np.ndarray] = None):
        super().__init__()
        encoder_args = {
            "vocab_size": config["VOCAB_SIZE"],
            "embed_dim": config["EMBED_DIM"],
            "hidden_dim": config["HIDDEN_DIM"],
            "pretrained_embeddings": pretrained_embeddings,
            "rnn_type": config.get("RNN_TYPE", "GRU"),
            "num_layers": config.get("NUM_LAYERS", 1),
            "dropout": config.get("DROPOUT", 0.0),
            "bidirectional": config.get("BIDIRECTIONAL", False),
            "normalize_output": config.get("NORMALIZE_OUTPUT", True),
            "arith": config.get("ARITH", "split16"),   # (not a reference key: "f32" = every product on the fp32-MFMA kernels)
        }
        self.query_encoder = RNNEncoder(**encoder_args)
        self.doc_encoder = RNNEncoder(**encoder_args)

    def encode_query(self, query: torch.Tensor) -> torch.Tensor:
        return self.query_encoder(query)

    def encode_document(self, document: torch.Tensor) -> torch.Tensor:
        return self.doc_encoder(document)

    def forward(self, query: torch.Tensor, document: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        return self.encode_query(query), self.encode_document(document)


class _TripletFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, p, n, margin: float):
        L = _lib.lib()
        q, p, n = q.contiguous(), p.contiguous(), n.contiguous()
        B, H = q.shape
        loss = torch.empty((), dtype=torch.float32, device=q.device)
        dq, dp, dn = torch.empty_like(q), torch.empty_like(p), torch.empty_like(n)
        rows = torch.empty(B, dtype=torch.float32, device=q.device)
        with torch.cuda.device(q.device):
            _lib.check(L.tt_triplet_loss_f32(q.data_ptr(), p.data_ptr(), n.data_ptr(), B, H, float(margin),
                                             loss.data_ptr(), dq.data_ptr(), dp.data_ptr(), dn.data_ptr(),
                                             rows.data_ptr(), _stream(q.device)))
        ctx.save_for_backward(dq, dp, dn)
        return loss

    @staticmethod
    def backward(ctx, g):
        dq, dp, dn = ctx.saved_tensors
        return g * dq, g * dp, g * dn, None


def triplet_loss_cosine(triplet: Tuple[torch.Tensor, torch.Tensor, torch.Tensor], margin: float = 0.2) -> torch.Tensor:
    """mean(clamp(cos(q,n) - cos(q,p) + margin, min=0)), differentiable (fused forward+gradient kernel)."""
    query, pos_doc, neg_doc = triplet
    for t in (query, pos_doc, neg_doc):
        if not t.is_cuda:
            raise RuntimeError("triplet_loss_cosine runs only on an AMD GPU via libtt.so (no CPU fallback)")
        if t.dtype != torch.float32 or t.dim() != 2 or t.shape != query.shape:
            raise ValueError("triplet_loss_cosine wants three float32 [B,H] tensors of one shape")
    return _TripletFn.apply(query, pos_doc, neg_doc, margin)
