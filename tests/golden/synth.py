"""Deterministic synthetic weights / inputs shared by the golden generator and the tests.

Fixtures store seeds + inputs + EXPECTED OUTPUTS only; weights are regenerated
from the seed with numpy's legacy RandomState stream (bit-stable across numpy
versions), so the .npz files stay small.
"""
from __future__ import annotations

import numpy as np


def make_table(seed: int, V: int, E: int) -> np.ndarray:
    """GloVe-layout table: every row non-zero, INCLUDING row 0 (the word "the")."""
    rs = np.random.RandomState(seed)
    return (rs.standard_normal((V, E)) * 0.3).astype(np.float32)


def make_encoder_state(seed: int, E: int, H: int, num_layers: int = 1, bidirectional: bool = False,
                       prefix: str = "", gates: int = 3) -> dict:
    """state_dict-keyed numpy weights, nn.GRU default init U(-1/sqrt(H), 1/sqrt(H)).
    gates = rows / H of the recurrent tensors: 3 GRU, 4 LSTM, 1 vanilla RNN."""
    rs = np.random.RandomState(seed)
    k = 1.0 / np.sqrt(H)
    sd = {}
    ndir = 2 if bidirectional else 1
    for layer in range(num_layers):
        I = E if layer == 0 else ndir * H
        for d in range(ndir):
            sfx = f"_l{layer}" + ("_reverse" if d == 1 else "")
            sd[f"{prefix}rnn.weight_ih{sfx}"] = rs.uniform(-k, k, (gates * H, I)).astype(np.float32)
            sd[f"{prefix}rnn.weight_hh{sfx}"] = rs.uniform(-k, k, (gates * H, H)).astype(np.float32)
            sd[f"{prefix}rnn.bias_ih{sfx}"] = rs.uniform(-k, k, (gates * H,)).astype(np.float32)
            sd[f"{prefix}rnn.bias_hh{sfx}"] = rs.uniform(-k, k, (gates * H,)).astype(np.float32)
    if bidirectional:
        kp = 1.0 / np.sqrt(2 * H)
        sd[f"{prefix}projection.weight"] = rs.uniform(-kp, kp, (H, 2 * H)).astype(np.float32)
        sd[f"{prefix}projection.bias"] = rs.uniform(-kp, kp, (H,)).astype(np.float32)
    return sd


def weight_quads(sd: dict, num_layers: int = 1, bidirectional: bool = False, prefix: str = ""):
    """[(W_ih, W_hh, b_ih, b_hh)] ordered (layer, dir), the layout oracle/ and the C ABI take."""
    out = []
    for layer in range(num_layers):
        for d in range(2 if bidirectional else 1):
            sfx = f"_l{layer}" + ("_reverse" if d == 1 else "")
            out.append(tuple(sd[f"{prefix}rnn.{n}{sfx}"] for n in
                             ("weight_ih", "weight_hh", "bias_ih", "bias_hh")))
    return out


def make_ids(seed: int, B: int, T: int, V: int, min_len: int = 1, zero_inside: float = 0.0,
             full_row: bool = True) -> np.ndarray:
    """Right-padded id batch [B,T] int64.  zero_inside = probability that an interior token is
    id 0 (the word "the", which also shortens the row: SURVEY 8a quirk a2-1).  Every row keeps
    at least one non-zero id; row 0 is full length when full_row."""
    rs = np.random.RandomState(seed)
    ids = np.zeros((B, T), dtype=np.int64)
    for b in range(B):
        L = T if (full_row and b == 0) else int(rs.randint(min_len, T + 1))
        row = rs.randint(1, V, size=L)
        if zero_inside > 0 and L > 1:
            mask = rs.uniform(size=L) < zero_inside
            mask[0] = False
            row = np.where(mask, 0, row)
        ids[b, :L] = row
    return ids


def unit_rows(seed: int, n: int, d: int) -> np.ndarray:
    """randn rows, L2-normalised in fp32 (the encoder's output distribution, model.py:74)."""
    rs = np.random.RandomState(seed)
    x = rs.standard_normal((n, d)).astype(np.float32)
    x /= np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-12).astype(np.float32)
    return x.astype(np.float32)
