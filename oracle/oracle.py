"""numpy front-end of the CPU oracle (oracle/tt_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never from the product package.  Parity pin: see
the header of tt_oracle.c (goldens generated from the reference itself by
tests/golden/gen_golden.py).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_SO = _HERE / "_build" / "libtt_oracle.so"

O_OK, O_ERR_BAD_SHAPE, O_ERR_BAD_INDEX, O_ERR_ZERO_LENGTH, O_ERR_NOMEM = range(5)


class OracleError(RuntimeError):
    def __init__(self, code: int, where: str):
        super().__init__(f"oracle {where} failed with code {code}")
        self.code = code


def build(force: bool = False) -> Path:
    """Compile tt_oracle.c with gcc (seconds).  Building the checker is not using it."""
    src = _HERE / "tt_oracle.c"
    if force or not _SO.exists() or _SO.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(_HERE), "-s"], check=True)
    return _SO


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(str(_SO))
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _check(rc: int, where: str):
    if rc != O_OK:
        raise OracleError(rc, where)


def lengths(ids: np.ndarray) -> np.ndarray:
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    B, T = ids.shape
    out = np.zeros(B, dtype=np.int32)
    _check(lib().o_lengths(_p(ids), B, T, _p(out)), "o_lengths")
    return out


def _weight_ptrs(weights):
    """weights: list over (layer, dir) of (W_ih, W_hh, b_ih, b_hh) -> (ptr array, keepalive)."""
    keep = []
    arr = (C.c_void_p * (4 * len(weights)))()
    for i, quad in enumerate(weights):
        for j, w in enumerate(quad):
            w = _f32(w)
            keep.append(w)
            arr[4 * i + j] = w.ctypes.data
    return arr, keep


CELLS = {"GRU": 0, "LSTM": 1, "RNN": 2}


def encoder_forward(ids, table, weights, hidden_dim, num_layers=1, bidirectional=False,
                    proj_w=None, proj_b=None, normalize=True, dropout_p=0.0, dropout_seed=0,
                    rnn_type="GRU") -> np.ndarray:
    """RNNEncoder.forward.  weights: [(W_ih,W_hh,b_ih,b_hh)] per (layer,dir); rnn_type GRU / LSTM / RNN."""
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    table = _f32(table)
    B, T = ids.shape
    V, E = table.shape
    H = int(hidden_dim)
    wp, keep = _weight_ptrs(weights)
    pw = _f32(proj_w) if proj_w is not None else None
    pb = _f32(proj_b) if proj_b is not None else None
    out = np.zeros((B, H), dtype=np.float32)
    rc = lib().o_encoder_forward_cell(CELLS[rnn_type.upper()], _p(ids), B, T, _p(table), C.c_int64(V), E, H,
                                      int(num_layers), int(bool(bidirectional)), wp, _p(pw), _p(pb),
                                      int(bool(normalize)), C.c_float(dropout_p), C.c_uint64(dropout_seed), _p(out))
    _check(rc, "o_encoder_forward")
    return out


def encoder_backward(ids, table, weights, hidden_dim, d_out, num_layers=1, bidirectional=False,
                     proj_w=None, proj_b=None, normalize=True, dropout_p=0.0, dropout_seed=0, table_grad=False,
                     rnn_type="GRU"):
    """Returns (grads, g_proj_w, g_proj_b); grads mirrors `weights`.  table_grad=True (the reference's trainable
    embedding table, model.py:23 without GloVe) appends the [V,E] table gradient as a fourth element."""
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    table = _f32(table)
    B, T = ids.shape
    V, E = table.shape
    H = int(hidden_dim)
    wp, keep = _weight_ptrs(weights)
    grads = [tuple(np.zeros_like(_f32(w)) for w in quad) for quad in weights]
    gp = (C.c_void_p * (4 * len(grads)))()
    for i, quad in enumerate(grads):
        for j, g in enumerate(quad):
            gp[4 * i + j] = g.ctypes.data
    pw = _f32(proj_w) if proj_w is not None else None
    pb = _f32(proj_b) if proj_b is not None else None
    gpw = np.zeros_like(pw) if pw is not None else None
    gpb = np.zeros_like(pb) if pb is not None else None
    d_out = _f32(d_out)
    gt = np.zeros_like(table) if table_grad else None
    rc = lib().o_encoder_backward_cell(CELLS[rnn_type.upper()], _p(ids), B, T, _p(table), C.c_int64(V), E, H,
                                       int(num_layers), int(bool(bidirectional)), wp, _p(pw), _p(pb),
                                       int(bool(normalize)), C.c_float(dropout_p), C.c_uint64(dropout_seed),
                                       _p(d_out), gp, _p(gpw), _p(gpb), _p(gt))
    _check(rc, "o_encoder_backward")
    if table_grad:
        return grads, gpw, gpb, gt
    return grads, gpw, gpb


def dropout_mask(seed, layer, n, p):
    out = np.zeros(n, dtype=np.float32)
    lib().o_dropout_mask(C.c_uint64(seed), int(layer), C.c_int64(n), C.c_float(p), _p(out))
    return out


def score_topk(Q, D, k, idx_offset=0):
    Q = _f32(Q)
    D = _f32(D)
    B, d = Q.shape
    N = D.shape[0]
    val = np.zeros((B, k), dtype=np.float32)
    idx = np.zeros((B, k), dtype=np.int64)
    rc = lib().o_score_topk(_p(Q), B, d, _p(D), C.c_int64(N), int(k), C.c_int64(idx_offset),
                            _p(val), _p(idx))
    _check(rc, "o_score_topk")
    return val, idx


def topk_merge(vals, idx, k):
    vals = _f32(vals)
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    B, M = vals.shape
    ov = np.zeros((B, k), dtype=np.float32)
    oi = np.zeros((B, k), dtype=np.int64)
    _check(lib().o_topk_merge(_p(vals), _p(idx), B, M, int(k), _p(ov), _p(oi)), "o_topk_merge")
    return ov, oi


def score_rank(Q, D, target):
    Q = _f32(Q)
    D = _f32(D)
    target = np.ascontiguousarray(target, dtype=np.int64)
    B, d = Q.shape
    rank = np.zeros(B, dtype=np.int64)
    _check(lib().o_score_rank(_p(Q), B, d, _p(D), C.c_int64(D.shape[0]), _p(target), _p(rank)),
           "o_score_rank")
    return rank


def triplet_loss(q, p, n, margin=0.2, with_grads=True):
    q, p, n = _f32(q), _f32(p), _f32(n)
    B, H = q.shape
    loss = C.c_float(0.0)
    if with_grads:
        dq, dp, dn = np.zeros_like(q), np.zeros_like(p), np.zeros_like(n)
    else:
        dq = dp = dn = None
    rc = lib().o_triplet_loss(_p(q), _p(p), _p(n), B, H, C.c_float(margin), C.byref(loss),
                              _p(dq), _p(dp), _p(dn))
    _check(rc, "o_triplet_loss")
    return float(loss.value), dq, dp, dn


def clip_adam_step(p, g, m, v, step, lr, betas=(0.9, 0.999), eps=1e-8, max_norm=1.0):
    """In-place on flat fp32 arrays p, g, m, v.  Returns the pre-clip total grad norm."""
    for a in (p, g, m, v):
        assert a.dtype == np.float32 and a.flags.c_contiguous and a.ndim == 1
    total = C.c_float(0.0)
    rc = lib().o_clip_adam_step(_p(p), _p(g), _p(m), _p(v), C.c_int64(p.size), C.c_int64(step),
                                C.c_float(lr), C.c_float(betas[0]), C.c_float(betas[1]),
                                C.c_float(eps), C.c_float(max_norm), C.byref(total))
    _check(rc, "o_clip_adam_step")
    return float(total.value)


def cpu_threads() -> int:
    return os.cpu_count() or 1
