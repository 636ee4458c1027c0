"""BASELINE configs[0] as plumbing: a synthetic MS-MARCO-shaped tiny set (1k queries / 10k passages over a GloVe-
layout vocabulary, index 0 = "the"), the reference's default model shape (backend/config.json: E=200, HIDDEN_DIM 256,
2-layer bidirectional GRU, dropout 0.2, margin 0.5, lr 5e-5 -> 1e-3 here so a few steps move the metrics), its train
loop shape (backend/main.py:244-259), BatchEvaluator, artifact export, QueryInferencer, hybrid search -- all on the
HIP path.  Learnable signal: a query shares three rare words with its positive passage."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _make_corpus(rs, n_docs, n_queries, words):
    docs = [" ".join(words[i] for i in rs.randint(50, len(words), rs.randint(20, 60))) + " ." for _ in range(n_docs)]
    queries, pos = [], []
    for qi in range(n_queries):
        p = int(rs.randint(0, n_docs))
        toks = docs[p].split()
        pick = [toks[i] for i in sorted(rs.choice(len(toks) - 1, 3, replace=False))]
        queries.append("what is the " + " ".join(pick) + " ?")
        pos.append(p)
    return docs, queries, np.array(pos)


def test_tiny_marco_shaped_train_eval_export_serve(tmp_path):
    import twotowermlretrieval_amd as tt
    from twotowermlretrieval_amd.evaluators import BatchEvaluator, corpus_recall_hit, save_inference_artifacts
    from twotowermlretrieval_amd.hybrid import HybridSearcher
    dev = torch.device("cuda")
    rs = np.random.RandomState(0)
    words = ["the", ",", ".", "of", "and", "what", "is", "?"] + [f"w{i}" for i in range(8, 1000)]
    tok = tt.PretrainedTokenizer(word2idx={w: i for i, w in enumerate(words)})
    docs, queries, pos = _make_corpus(rs, 10_000, 1_000, words)
    cfg = {"HIDDEN_DIM": 256, "RNN_TYPE": "GRU", "NUM_LAYERS": 2, "BIDIRECTIONAL": True, "DROPOUT": 0.2,
           "BATCH_SIZE": 64, "LR": 1e-3, "MARGIN": 0.5, "NORMALIZE_OUTPUT": True}
    E = 200
    table = (rs.standard_normal((tok.vocab_size(), E)) * 0.3).astype(np.float32)
    torch.manual_seed(0)
    model = tt.TwoTowerModel({**cfg, "VOCAB_SIZE": tok.vocab_size(), "EMBED_DIM": E}, table).to(dev)

    def batches(idx):
        for i in range(0, len(idx), cfg["BATCH_SIZE"]):
            b = idx[i:i + cfg["BATCH_SIZE"]]
            neg = (pos[b] + 1 + rs.randint(0, len(docs) - 1, len(b))) % len(docs)
            yield (tok.encode_batch([queries[j] for j in b]), tok.encode_batch([docs[pos[j]] for j in b]),
                   tok.encode_batch([docs[j] for j in neg]))

    train_idx, val_idx = np.arange(0, 896), np.arange(0, 128)   # metrics on triplets it trained on: this is a plumbing
    # test (the pipeline optimises what it is given), not a claim about generalisation from 896 synthetic triplets
    ev = BatchEvaluator(top_k=[1, 5, 10])
    m0, l0 = ev.evaluate(model, batches(val_idx), dev, cfg)
    opt = tt.FusedClipAdam(model.parameters(), lr=cfg["LR"], max_norm=1.0)
    losses = []
    for epoch in range(6):
        model.train()
        for q, p, n in batches(rs.permutation(train_idx)):
            losses.append(tt.train_step(model, opt, q.to(dev), p.to(dev), n.to(dev), margin=cfg["MARGIN"]))
    losses = torch.stack(losses).cpu().numpy()
    m1, l1 = ev.evaluate(model, batches(val_idx), dev, cfg)
    print("loss first/last", losses[:5].mean(), losses[-5:].mean(), "val", l0, l1, m0, m1)
    assert np.isfinite(losses).all() and losses[-5:].mean() < 0.85 * losses[:5].mean()     # it trains
    assert l1 < 0.8 * l0 and m1["MRR"] >= m0["MRR"] - 0.01                                  # eval loss follows; metrics are computed

    emb = save_inference_artifacts(tmp_path, model, cfg, tok, docs, dev)                    # main.py:92-138
    assert emb.shape == (10_000, 256) and np.allclose(np.linalg.norm(emb, axis=1), 1.0, atol=1e-5)
    inf = tt.QueryInferencer(str(tmp_path))                                                 # a fresh process would do the same
    D = torch.from_numpy(emb).to(dev)
    qe = torch.stack([torch.from_numpy(inf.get_query_embedding(queries[j])) for j in val_idx[:64]]).to(dev)
    with torch.no_grad():
        model.eval()
        assert torch.equal(qe, model.encode_query(tok.encode_batch([queries[j] for j in val_idx[:64]]).to(dev)))
    ranks = tt.score_rank(qe, D, torch.from_numpy(pos[val_idx[:64]]).to(dev)).cpu().numpy()   # rank of the positive among 10k
    print("median rank of the positive passage among 10k:", np.median(ranks))
    assert np.median(ranks) < 2500                                                          # chance: 5000
    hit = corpus_recall_hit(qe[0], D, [int(pos[val_idx[0]])])
    assert set(hit) == {"Recall@1", "Hit@1", "Recall@5", "Hit@5", "Recall@10", "Hit@10"}
    res = HybridSearcher(inf, docs, D).search(queries[int(val_idx[0])], alpha=0.5, n_results=10)
    assert len(res) == 10 and res[0]["score"] >= res[-1]["score"]
