"""The two host-side pieces the reference's training loop feeds the towers with (backend/main.py:33-56):

    TripletDataset(data, tokenizer)[i] -> (query ids, positive ids, negative ids) as int64 tensors     main.py:33-48
    collate_fn(batch) -> three right-padded int64 matrices (padding id 0, width = the batch maximum)      main.py:50-56

so that `DataLoader(TripletDataset(triplets, tokenizer), batch_size=..., collate_fn=collate_fn)` (main.py:216-222) works
unchanged in front of the HIP towers.  `collate_fn` pads with one copy per row into a preallocated matrix instead of three
`pad_sequence` calls; `encode_batch` (tokenizer.py) is the bulk form used by the index build.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch

__all__ = ["TripletDataset", "collate_fn"]


class TripletDataset(torch.utils.data.Dataset):
    """Dataset for (query, positive_doc, negative_doc) string triplets."""

    def __init__(self, data: Sequence[Tuple[str, str, str]], tokenizer):
        self.data = data
        self.tokenizer = tokenizer

    def __len__(self) -> int:
        return len(self.data)

    def __getitem__(self, idx: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        query, pos_doc, neg_doc = self.data[idx]
        enc = self.tokenizer.encode
        return (torch.tensor(enc(query), dtype=torch.long), torch.tensor(enc(pos_doc), dtype=torch.long),
                torch.tensor(enc(neg_doc), dtype=torch.long))


def _pad(rows: Sequence[torch.Tensor]) -> torch.Tensor:
    width = max((int(r.numel()) for r in rows), default=0)
    out = torch.zeros((len(rows), width), dtype=torch.long)
    for i, r in enumerate(rows):
        out[i, :r.numel()] = r
    return out


def collate_fn(batch: List[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Pads sequences to the max length in a batch (padding id 0, right-padded, like pad_sequence(batch_first=True))."""
    queries, pos_docs, neg_docs = zip(*batch)
    return _pad(queries), _pad(pos_docs), _pad(neg_docs)
