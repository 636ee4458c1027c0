"""The reference's own CPU idiom for the scoring path, restated with stock PyTorch calls
(the reference's arithmetic IS these torch calls: backend/evaluators.py:185-186).

TEST INFRASTRUCTURE ONLY (see tt_oracle.c header): used by bench.py's cpu_baseline leg and by
tests; never imported by the product package.
"""
from __future__ import annotations

import time

import torch


def scoring_idiom(q: torch.Tensor, docs: torch.Tensor, k: int):
    """sim = matmul(q, D.t()); topk(sim, k)  -- materialises the full [B,N] score matrix."""
    sim = torch.matmul(q, docs.t())
    return torch.topk(sim, k)


def time_scoring_idiom(q: torch.Tensor, docs: torch.Tensor, k: int, warmup: int = 1, reps: int = 3) -> float:
    """Median wall seconds of one scoring_idiom call on CPU tensors."""
    assert not q.is_cuda and not docs.is_cuda
    for _ in range(warmup):
        scoring_idiom(q, docs, k)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        scoring_idiom(q, docs, k)
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]
