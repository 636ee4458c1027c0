"""Row-parallel driver of the CPU oracle for the bench-size checks (tests only).

oracle.encoder_forward / encoder_backward are single-threaded C; rows of a batch are independent in the forward and the
parameter gradients are sums over rows, so a batch is cut into row chunks that run on a thread pool (ctypes releases the
GIL) and the chunk gradients are added in float64, in chunk order.  The arithmetic of every row is the oracle's own."""
from __future__ import annotations

import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np


def _threads():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 32))


def _chunks(B, n):
    n = max(1, min(n, B))
    edges = np.linspace(0, B, n + 1).astype(int)
    return [(int(a), int(b)) for a, b in zip(edges[:-1], edges[1:]) if b > a]


def _trim(ids):
    """Drop all-padding trailing columns of a row chunk (the oracle's cost follows T)."""
    nz = np.nonzero((ids != 0).any(axis=0))[0]
    T = int(nz[-1]) + 1 if nz.size else 1
    return np.ascontiguousarray(ids[:, :T])


def forward(oracle, ids, table, quads, H, **kw):
    ids = np.ascontiguousarray(ids)
    # long rows first so the pool drains evenly; results go back to their own rows
    order = np.argsort(-(ids != 0).sum(axis=1), kind="stable")
    parts = _chunks(len(order), 4 * _threads())
    out = np.zeros((ids.shape[0], H), dtype=np.float32)

    def run(ab):
        rows = order[ab[0]:ab[1]]
        return rows, oracle.encoder_forward(_trim(ids[rows]), table, quads, H, **kw)
    with ThreadPoolExecutor(_threads()) as ex:
        for rows, y in ex.map(run, parts):
            out[rows] = y
    return out


def backward(oracle, ids, table, quads, H, d_out, **kw):
    """Returns the list of (dW_ih, dW_hh, db_ih, db_hh) per (layer, dir), float32, summed over row chunks in float64."""
    ids = np.ascontiguousarray(ids)
    d_out = np.ascontiguousarray(d_out, dtype=np.float32)
    order = np.argsort(-(ids != 0).sum(axis=1), kind="stable")
    parts = _chunks(len(order), 4 * _threads())

    def run(ab):
        rows = order[ab[0]:ab[1]]
        g, _, _ = oracle.encoder_backward(_trim(ids[rows]), table, quads, H, d_out[rows], **kw)
        return g
    acc = None
    with ThreadPoolExecutor(_threads()) as ex:
        for g in ex.map(run, parts):
            if acc is None:
                acc = [[x.astype(np.float64) for x in quad] for quad in g]
            else:
                for qa, qg in zip(acc, g):
                    for a, x in zip(qa, qg):
                        a += x
    return [tuple(a.astype(np.float32) for a in quad) for quad in acc]
