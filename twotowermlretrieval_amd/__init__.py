"""MI355X-native (gfx950) implementation of the two-tower retrieval hot path of
jpe17/TwoTowerMLRetrieval: fused brute-force scoring + top-k, GRU encoder towers,
triplet-loss training.  Python host over a C-ABI HIP library (libtt.so, include/tt.h)."""
from . import collective, evaluators, hybrid, model, query_inferencer, tokenizer, trainer
from .index import (BruteForceIndex, GraphedSearch, PendingSearch, ShardedIndex, StreamedIndex, score_all, score_rank, score_topk, shard_bounds,
                    topk_merge)
from .hybrid import HybridSearcher, SimpleHybridRetriever
from .model import RNNEncoder, TwoTowerModel, triplet_loss_cosine
from .query_inferencer import QueryInferencer
from .tokenizer import PretrainedTokenizer
from .trainer import DataParallelTrainer, FusedClipAdam, GraphedTrainStep, train_step

__all__ = ["BruteForceIndex", "GraphedSearch", "ShardedIndex", "PendingSearch", "StreamedIndex", "score_topk", "topk_merge", "score_rank", "score_all", "shard_bounds",
           "RNNEncoder", "TwoTowerModel", "triplet_loss_cosine", "QueryInferencer", "PretrainedTokenizer", "HybridSearcher", "SimpleHybridRetriever",
           "FusedClipAdam", "DataParallelTrainer", "train_step", "GraphedTrainStep", "model", "trainer", "tokenizer", "query_inferencer",
           "evaluators", "hybrid", "collective"]
__version__ = "0.1.0"
