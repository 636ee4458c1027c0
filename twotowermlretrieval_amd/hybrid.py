"""Hybrid rerank stage of the reference's /search handler (frontend/main.py:102-210) with the ChromaDB
HNSW lookup replaced by the exact GPU top-50.

    1. query string -> unit vector (QueryInferencer / the HIP query tower)
    2. exact top-`n_candidates` (50) passages by cosine over the resident embedding matrix
    3. TF-IDF cosine of the query against those candidates (sklearn, CPU: sparse string work, out of the
       GPU path's scope -- the fitted vectorizer is the reference's `tfidf_artifacts.pkl`)
    4. final = alpha * dense + (1 - alpha) * tfidf, sort, top `n_results` (10)

Dense score definition.  The reference stores embeddings in a Chroma collection created WITHOUT
`hnsw:space` (save_to_chromaDB.ipynb), i.e. squared-L2 distances, and uses `1 - dist` as the semantic
score (frontend/main.py:162); for unit vectors that is 2*cos - 1, not the cosine.  `dense_score="cosine"`
(default) uses the cosine itself; `dense_score="chroma_l2"` reproduces the reference's 2*cos - 1 so the
blend weights match its behaviour exactly.  alpha == 0 is the reference's pure keyword search over the
whole corpus (frontend/main.py:119-147) and never touches the GPU.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np
import torch

from .index import BruteForceIndex, score_all


class HybridSearcher:
    def __init__(self, inferencer, documents: Sequence[str], doc_embeddings: torch.Tensor = None, tfidf_vectorizer=None,
                 doc_tfidf_matrix=None, n_candidates: int = 50, dense_score: str = "cosine", index: BruteForceIndex = None):
        """documents: documents.pkl's list (row i of the embeddings <-> documents[i]); any object with __len__ and __getitem__
        is used as it is (a corpus too large for a Python list: an mmap-backed or generated sequence), other iterables are
        listed.  index: an existing BruteForceIndex over the embedding matrix instead of doc_embeddings (one is built otherwise)."""
        if dense_score not in ("cosine", "chroma_l2"):
            raise ValueError("dense_score must be 'cosine' or 'chroma_l2'")
        if (index is None) == (doc_embeddings is None):
            raise ValueError("HybridSearcher wants doc_embeddings or index (exactly one)")
        self.inferencer = inferencer
        indexable = hasattr(documents, "__getitem__") and hasattr(documents, "__len__") and not isinstance(documents, (str, bytes))
        self.documents = documents if indexable else list(documents)
        # single-query searches stream the fp16 shadow corpus
        self.index = index if index is not None else BruteForceIndex(doc_embeddings, screen=True)
        if len(self.documents) != self.index.ntotal:
            raise ValueError(f"{len(self.documents)} documents for {self.index.ntotal} embedding rows")
        self.n_candidates = int(n_candidates)
        self.dense_score = dense_score
        if tfidf_vectorizer is None:  # same construction as backend/main.py:142-143
            from sklearn.feature_extraction.text import TfidfVectorizer
            tfidf_vectorizer = TfidfVectorizer(stop_words="english", max_features=20000)
            doc_tfidf_matrix = tfidf_vectorizer.fit_transform(self.documents)
        self.tfidf = tfidf_vectorizer
        self.doc_tfidf = doc_tfidf_matrix

    def search(self, query: str, alpha: float = 0.5, n_results: int = 10) -> List[Dict]:
        from sklearn.metrics.pairwise import cosine_similarity
        if alpha == 0.0:
            sims = cosine_similarity(self.tfidf.transform([query]), self.doc_tfidf).flatten()
            order = np.argsort(-sims, kind="stable")[:n_results]
            return [{"doc": self.documents[i], "index": int(i), "score": float(sims[i]), "dense_score": 0.0,
                     "tfidf_score": float(sims[i])} for i in order if sims[i] > 1e-5]
        q = torch.from_numpy(self.inferencer.get_query_embedding(query)).to(self.index.docs.device)
        k = min(self.n_candidates, self.index.ntotal)
        vals, idx = self.index.search(q, k)
        cos = vals.cpu().numpy()
        cand = [int(i) for i in idx.cpu().tolist() if i >= 0]
        dense = cos[:len(cand)] if self.dense_score == "cosine" else 2.0 * cos[:len(cand)] - 1.0
        q_tfidf = self.tfidf.transform([query])
        if q_tfidf.nnz > 0:
            tf = np.nan_to_num(cosine_similarity(q_tfidf, self.tfidf.transform([self.documents[i] for i in cand]))[0])
        else:
            tf = np.zeros(len(cand))
        final = alpha * dense + (1.0 - alpha) * tf
        order = np.argsort(-final, kind="stable")[:n_results]
        return [{"doc": self.documents[cand[i]], "index": cand[i], "score": float(final[i]),
                 "dense_score": float(dense[i]), "tfidf_score": float(tf[i])} for i in order]


class SimpleHybridRetriever:
    """Drop-in for backend/simple_hybrid.py:13-67 (same constructor, `fit`, `search` and return types): TF-IDF + dense
    scores of EVERY document, combined = alpha * dense + (1 - alpha) * tfidf, descending argsort, top_k (doc, score)
    pairs.  As in the reference the corpus is embedded with the SAME (query) encoder (:37-41) and the vectorizer is
    TfidfVectorizer(stop_words='english', max_features=10000) (:24).  The dense scores come from the HIP path
    (query tower + tt_score_all_f32); the TF-IDF half is sklearn on the CPU, as in the reference.
    RESTRICTION: the dense term is the dot product of the tower outputs, which equals the reference's
    sklearn `cosine_similarity` only for unit rows, i.e. with NORMALIZE_OUTPUT = true (the reference's default and the
    only configuration G11 pins); with NORMALIZE_OUTPUT = false the blend differs from the reference's."""

    def __init__(self, artifacts_path: str, alpha: float = 0.5, device=None):
        from sklearn.feature_extraction.text import TfidfVectorizer
        from .query_inferencer import QueryInferencer
        self.dense_retriever = QueryInferencer(artifacts_path, device=device)
        self.alpha = alpha
        self.tfidf = TfidfVectorizer(stop_words="english", max_features=10000)
        self.documents: List[str] = []
        self.doc_embeddings = None

    def fit(self, documents: Sequence[str]) -> None:
        self.documents = list(documents)
        self.tfidf_matrix = self.tfidf.fit_transform(self.documents)
        # one batched call of the query tower instead of the reference's per-document loop: same rows
        self.doc_embeddings = self.dense_retriever.get_query_embeddings(self.documents)

    def search(self, query: str, top_k: int = 10):
        from sklearn.metrics.pairwise import cosine_similarity
        tfidf_scores = cosine_similarity(self.tfidf.transform([query]), self.tfidf_matrix)[0]
        q = torch.from_numpy(self.dense_retriever.get_query_embedding(query)).to(self.doc_embeddings.device)
        dense_scores = score_all(q, self.doc_embeddings).cpu().numpy().astype(np.float64)
        combined = self.alpha * dense_scores + (1 - self.alpha) * tfidf_scores
        top = np.argsort(combined)[::-1][:top_k]
        return [(self.documents[i], combined[i]) for i in top]
