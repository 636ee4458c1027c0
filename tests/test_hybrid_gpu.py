"""Hybrid rerank (twotowermlretrieval_amd.hybrid.HybridSearcher) pinned to the reference's own blend:
tests/golden/g11_hybrid.npz holds backend/simple_hybrid.py:28-67 run on a synthetic artifacts directory --
documents embedded with the query tower, TfidfVectorizer(stop_words='english', max_features=10000),
combined = alpha * dense + (1 - alpha) * tfidf, descending argsort -- for alpha in {0.3, 0.5, 1.0}."""
import numpy as np
import pytest
import torch

from conftest import assert_fwd_close
from test_inferencer_gpu import _artifacts

pytestmark = pytest.mark.gpu


def test_hybrid_blend_matches_reference_simple_hybrid(tmp_path, golden):
    from sklearn.feature_extraction.text import TfidfVectorizer
    from twotowermlretrieval_amd.hybrid import HybridSearcher
    from twotowermlretrieval_amd.query_inferencer import QueryInferencer
    g = golden("g11_hybrid.npz")
    docs = [str(d) for d in g["docs"]]
    inf = QueryInferencer(str(_artifacts(tmp_path, g)))
    # the reference embeds the corpus with the SAME (query) encoder (simple_hybrid.py:37-41)
    emb = inf.get_query_embeddings(docs)
    assert_fwd_close(emb.cpu().numpy(), g["doc_emb"])
    tfidf = TfidfVectorizer(stop_words="english", max_features=10000)  # simple_hybrid.py:24
    mat = tfidf.fit_transform(docs)
    hs = HybridSearcher(inf, docs, emb, tfidf_vectorizer=tfidf, doc_tfidf_matrix=mat, n_candidates=len(docs),
                        dense_score="cosine")
    for ai, alpha in enumerate(g["alphas"]):
        for qi, q in enumerate(g["queries"]):
            want = g["combined"][ai, qi]                 # the reference's score of EVERY document
            res = hs.search(str(q), alpha=float(alpha), n_results=10)
            assert len(res) == 10
            for r in res:                                 # every returned score is the reference's score of that document
                assert abs(r["score"] - want[r["index"]]) < 1e-5, (alpha, q, r)
            # same ranking wherever the reference's own adjacent gap exceeds the tolerance
            ref_order = g["top10"][ai, qi]
            ref_sorted = np.sort(want)[::-1]
            for pos in range(10):
                lo_gap = ref_sorted[pos] - ref_sorted[pos + 1]
                hi_gap = ref_sorted[pos - 1] - ref_sorted[pos] if pos else np.inf
                if lo_gap > 2e-5 and hi_gap > 2e-5:
                    assert res[pos]["index"] == int(ref_order[pos]), (alpha, q, pos)
            # and the returned set is a valid top-10: nothing outside it beats the 10th score by more than the tolerance
            tenth = min(r["score"] for r in res)
            outside = np.delete(want, [r["index"] for r in res])
            assert outside.max() <= tenth + 2e-5


def test_simple_hybrid_retriever_drop_in_matches_reference(tmp_path, golden):
    """hybrid.SimpleHybridRetriever mirrors backend/simple_hybrid.py:13-67 (same constructor, fit, search): for every alpha
    and query of G11 the (document, score) pairs it returns carry the reference's scores, in the reference's order."""
    from twotowermlretrieval_amd.hybrid import SimpleHybridRetriever
    from twotowermlretrieval_amd import score_all
    g = golden("g11_hybrid.npz")
    docs = [str(d) for d in g["docs"]]
    art = str(_artifacts(tmp_path, g))
    for ai, alpha in enumerate(g["alphas"]):
        r = SimpleHybridRetriever(art, alpha=float(alpha))
        r.fit(docs)
        assert_fwd_close(r.doc_embeddings.cpu().numpy(), g["doc_emb"])
        for qi, q in enumerate(g["queries"]):
            want = g["combined"][ai, qi]
            res = r.search(str(q), top_k=10)
            assert len(res) == 10 and all(isinstance(d, str) for d, _ in res)
            ref_sorted = np.sort(want)[::-1]
            for pos, (doc, score) in enumerate(res):
                assert abs(score - ref_sorted[pos]) < 1e-5, (alpha, q, pos)        # the reference's pos-th best score
                assert abs(score - want[docs.index(doc)]) < 1e-5                    # ... and it belongs to this document
    # the all-scores entry point itself: the same fp32 chain the top-k kernels select from
    emb = r.doc_embeddings
    S = score_all(emb[:5], emb)
    v, i = __import__("twotowermlretrieval_amd").score_topk(emb[:5], emb, 10)
    assert torch.equal(torch.gather(S, 1, i), v)
