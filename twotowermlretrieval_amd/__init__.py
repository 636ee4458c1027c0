"""MI355X-native (gfx950) implementation of the two-tower retrieval hot path of
jpe17/TwoTowerMLRetrieval: fused brute-force scoring + top-k, GRU encoder towers,
triplet-loss training.  Python host over a C-ABI HIP library (libtt.so, include/tt.h)."""
from .index import BruteForceIndex, ShardedIndex, score_rank, score_topk, shard_bounds, topk_merge

__all__ = ["BruteForceIndex", "ShardedIndex", "score_topk", "topk_merge", "score_rank", "shard_bounds"]
__version__ = "0.1.0"
