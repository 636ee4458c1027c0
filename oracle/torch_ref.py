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


# ---------------------------------------------------------------------------------------------
# Encoder / training idiom (backend/model.py:48-75, :109-114; backend/main.py:244-259), restated with the same
# stock torch modules the reference builds (nn.Embedding, nn.GRU(batch_first=True), F.normalize,
# F.cosine_similarity, clip_grad_norm_, Adam).  CPU baseline legs of bench.py only.
# ---------------------------------------------------------------------------------------------
class TorchTower(torch.nn.Module):
    """1-layer unidirectional GRU tower with a frozen table: the north-star model shape (E=300, H=256)."""

    def __init__(self, table: torch.Tensor, hidden_dim: int, seed: int = 0):
        super().__init__()
        torch.manual_seed(seed)
        V, E = table.shape
        self.embedding = torch.nn.Embedding(V, E, padding_idx=0)
        self.embedding.weight.data.copy_(table)           # model.py:25-27: every row copied, then frozen
        self.embedding.weight.requires_grad = False
        self.rnn = torch.nn.GRU(E, hidden_dim, num_layers=1, batch_first=True)

    def forward(self, x: torch.Tensor) -> torch.Tensor:  # model.py:48-75
        embedded = self.embedding(x)
        lengths = (x != 0).sum(dim=1).cpu()
        packed = torch.nn.utils.rnn.pack_padded_sequence(embedded, lengths, batch_first=True, enforce_sorted=False)
        _, hidden = self.rnn(packed)
        return torch.nn.functional.normalize(hidden[-1], p=2, dim=1)


def triplet_loss_cosine(q, p, n, margin):  # model.py:109-114
    cs = torch.nn.functional.cosine_similarity
    return torch.clamp(cs(q, n) - cs(q, p) + margin, min=0).mean()


def _median_time(fn, warmup: int, reps: int) -> float:
    for _ in range(warmup):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]


def time_tower_forward(tower: TorchTower, ids: torch.Tensor, warmup: int = 1, reps: int = 3) -> float:
    tower.eval()
    with torch.no_grad():
        return _median_time(lambda: tower(ids), warmup, reps)


def time_train_step(qt: TorchTower, dt: TorchTower, q, p, n, margin: float = 0.5, lr: float = 5e-5,
                    warmup: int = 1, reps: int = 3) -> float:
    """main.py:244-259: zero_grad, three forwards, loss, backward, clip_grad_norm_(1.0), Adam step."""
    params = [w for w in list(qt.parameters()) + list(dt.parameters()) if w.requires_grad]
    opt = torch.optim.Adam(params, lr=lr)
    qt.train()
    dt.train()

    def step():
        opt.zero_grad()
        loss = triplet_loss_cosine(qt(q), dt(p), dt(n), margin)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, max_norm=1.0)
        opt.step()
        return float(loss.item())  # main.py:261 syncs on the loss every step

    return _median_time(step, warmup, reps)
