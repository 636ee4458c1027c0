#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE on CPU.

Only runs where /root/reference exists (the dev container).  It imports the
reference's own backend/model.py, backend/tokenizer.py, backend/evaluators.py
and backend/query_inferencer.py and records their outputs for seeded inputs.
Nothing of the reference (source or bytecode) is written into this repo: the
fixtures hold inputs, seeds and expected outputs only.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py

Fixtures:
  g1_encoder_uni.npz      RNNEncoder.forward, 1-layer unidirectional (model.py:48-75)
  g2_encoder_bi.npz       2-layer bidirectional + projection, eval mode (config.json shape)
  g3_encoder_nonorm.npz   NORMALIZE_OUTPUT = False
  g4_triplet.npz          triplet_loss_cosine value + autograd gradients (model.py:109-114)
  g5_clip_adam.npz        clip_grad_norm_(1.0) + Adam(lr=5e-5) steps (main.py:222,257,259)
  g6_scoring.npz          matmul + topk (evaluators.py:185-186) for k in {5,10,50}
  g7_batch_eval.npz       BatchEvaluator metrics for fixed embeddings (evaluators.py:48-76)
  g8_tokenizer.json       PretrainedTokenizer.encode (tokenizer.py:41-43)
  g9_inferencer.npz       QueryInferencer.get_query_embedding (query_inferencer.py:59-75)
  g12_table_grad.npz      RNNEncoder WITHOUT GloVe vectors: trainable nn.Embedding(padding_idx=0) (model.py:23-27),
                          autograd gradient of the table and of the GRU weights, 1-layer uni and 2-layer bidirectional
  g13_lstm_rnn.npz        RNN_TYPE = LSTM / RNN (model.py:30,59-62): forward outputs and autograd gradients of every tensor,
                          1-layer unidirectional and 2-layer bidirectional
  g14_corpus_eval.json    CorpusEvaluator.evaluate (evaluators.py:83-209): Recall@k / Hit@k over seeded query samples, stub towers
  g11_hybrid.npz          SimpleHybridRetriever.fit/search blend alpha*dense + (1-alpha)*tfidf (simple_hybrid.py:28-67)
  g10_errors.json         error behaviour (all-zero row, empty row, interior zeros)
"""
from __future__ import annotations

import json
import os
import pickle
import sys
import tempfile
from pathlib import Path

sys.dont_write_bytecode = True
HERE = Path(__file__).resolve().parent
REF = Path("/root/reference/backend")
if not REF.exists():
    sys.exit("reference not present; fixtures are generated in the dev container only")
sys.path.insert(0, str(REF))
sys.path.insert(0, str(HERE))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import model as refmodel  # noqa: E402  (reference backend/model.py)
import tokenizer as reftok  # noqa: E402
import evaluators as refeval  # noqa: E402
import synth  # noqa: E402

torch.set_num_threads(1)  # reproducible summation order for the recorded outputs


def ref_encoder(V, E, H, table, sd, num_layers=1, bidirectional=False, normalize=True):
    enc = refmodel.RNNEncoder(V, E, H, pretrained_embeddings=table, num_layers=num_layers,
                              bidirectional=bidirectional, normalize_output=normalize)
    full = {"embedding.weight": torch.from_numpy(table)}
    full.update({k: torch.from_numpy(v) for k, v in sd.items()})
    enc.load_state_dict(full)
    enc.eval()
    return enc


def g1():
    out = {}
    # small case with every quirk: trailing pads, interior id 0, length-1 row, full row
    V, E, H, seed = 64, 16, 32, 101
    table = synth.make_table(seed, V, E)
    sd = synth.make_encoder_state(seed + 1, E, H)
    ids = synth.make_ids(seed + 2, B=12, T=9, V=V, zero_inside=0.25)
    ids[1, :] = 0
    ids[1, 0] = 7                       # length-1 row
    ids[2, :] = [5, 0, 0, 9, 3, 0, 0, 0, 0]   # interior zeros: len 3 -> positions 0..2 run, 9 and 3 partly dropped
    ids[3, :] = [5, 0, 0, 11, 4, 0, 0, 0, 0]  # differs from row 2 only beyond position len-1 -> same output
    enc = ref_encoder(V, E, H, table, sd)
    with torch.no_grad():
        y = enc(torch.from_numpy(ids)).numpy()
    out.update(small_ids=ids, small_out=y, small_dims=np.array([V, E, H, seed]))
    # north-star shape E=300, H=256
    V, E, H, seed = 128, 300, 256, 202
    table = synth.make_table(seed, V, E)
    sd = synth.make_encoder_state(seed + 1, E, H)
    ids = synth.make_ids(seed + 2, B=6, T=14, V=V, zero_inside=0.1)
    enc = ref_encoder(V, E, H, table, sd)
    with torch.no_grad():
        y = enc(torch.from_numpy(ids)).numpy()
    out.update(big_ids=ids, big_out=y, big_dims=np.array([V, E, H, seed]))
    np.savez_compressed(HERE / "g1_encoder_uni.npz", **out)


def g2():
    V, E, H, seed = 80, 20, 32, 303
    table = synth.make_table(seed, V, E)
    sd = synth.make_encoder_state(seed + 1, E, H, num_layers=2, bidirectional=True)
    ids = synth.make_ids(seed + 2, B=10, T=11, V=V, zero_inside=0.15)
    enc = ref_encoder(V, E, H, table, sd, num_layers=2, bidirectional=True)
    with torch.no_grad():
        y = enc(torch.from_numpy(ids)).numpy()
    np.savez_compressed(HERE / "g2_encoder_bi.npz", ids=ids, out=y, dims=np.array([V, E, H, seed]))


def g3():
    V, E, H, seed = 64, 16, 32, 404
    table = synth.make_table(seed, V, E)
    sd = synth.make_encoder_state(seed + 1, E, H)
    ids = synth.make_ids(seed + 2, B=8, T=7, V=V)
    enc = ref_encoder(V, E, H, table, sd, normalize=False)
    with torch.no_grad():
        y = enc(torch.from_numpy(ids)).numpy()
    np.savez_compressed(HERE / "g3_encoder_nonorm.npz", ids=ids, out=y, dims=np.array([V, E, H, seed]))


def _two_tower(V, E, H, seed, num_layers=1, bidirectional=False):
    cfg = dict(VOCAB_SIZE=V, EMBED_DIM=E, HIDDEN_DIM=H, NUM_LAYERS=num_layers,
               BIDIRECTIONAL=bidirectional, DROPOUT=0.0)
    table = synth.make_table(seed, V, E)
    m = refmodel.TwoTowerModel(cfg, table)
    sd = {}
    for i, tower in enumerate(("query_encoder.", "doc_encoder.")):
        sd[tower + "embedding.weight"] = torch.from_numpy(table)
        for k, v in synth.make_encoder_state(seed + 10 + i, E, H, num_layers, bidirectional,
                                             prefix=tower).items():
            sd[k] = torch.from_numpy(v)
    m.load_state_dict(sd)
    return m, cfg, table


def g4():
    out = {}
    for tag, (layers, bi) in {"uni": (1, False), "bi": (2, True)}.items():
        V, E, H, seed = 64, 12, 32, 505 + (7 if bi else 0)
        m, cfg, table = _two_tower(V, E, H, seed, layers, bi)
        m.train()
        q = synth.make_ids(seed + 1, B=8, T=5, V=V)
        p = synth.make_ids(seed + 2, B=8, T=9, V=V, zero_inside=0.1)
        n = synth.make_ids(seed + 3, B=8, T=8, V=V, zero_inside=0.1)
        n[5] = 0
        n[5, :p.shape[1] - 1] = p[5, :p.shape[1] - 1][:n.shape[1]]  # near-duplicate negative
        for margin in (0.5, 0.2):
            m.zero_grad()
            e = (m.encode_query(torch.from_numpy(q)), m.encode_document(torch.from_numpy(p)),
                 m.encode_document(torch.from_numpy(n)))
            for t in e:
                t.retain_grad()
            loss = refmodel.triplet_loss_cosine(e, margin=margin)
            loss.backward()
            mt = f"{tag}_m{int(margin * 10)}"
            out[f"{mt}_loss"] = np.float32(loss.item())
            out[f"{mt}_hinge"] = (torch.clamp(
                torch.nn.functional.cosine_similarity(e[0], e[2]) -
                torch.nn.functional.cosine_similarity(e[0], e[1]) + margin, min=0) > 0).numpy()
            for nm, t in zip("qpn", e):
                out[f"{mt}_emb_{nm}"] = t.detach().numpy()
                out[f"{mt}_demb_{nm}"] = t.grad.numpy()
            for k, prm in m.named_parameters():
                if prm.requires_grad:
                    out[f"{mt}_grad_{k}"] = prm.grad.numpy().copy()
        out.update({f"{tag}_q": q, f"{tag}_p": p, f"{tag}_n": n,
                    f"{tag}_dims": np.array([V, E, H, seed, layers, int(bi)])})
    np.savez_compressed(HERE / "g4_triplet.npz", **out)


def g5():
    rs = np.random.RandomState(606)
    shapes = [(24, 10), (24, 8), (24,), (24,)]
    p0 = [rs.standard_normal(s).astype(np.float32) * 0.1 for s in shapes]
    params = [torch.nn.Parameter(torch.from_numpy(a.copy())) for a in p0]
    opt = torch.optim.Adam(params, lr=5e-5)
    out = {f"p0_{i}": a for i, a in enumerate(p0)}
    scales = [3.0, 0.01, 1.0]  # step 1 clips hard, step 2 does not clip, step 3 borderline
    for step in range(3):
        gs = [rs.standard_normal(s).astype(np.float32) * scales[step] for s in shapes]
        for prm, g in zip(params, gs):
            prm.grad = torch.from_numpy(g.copy())
        tn = torch.nn.utils.clip_grad_norm_(params, max_norm=1.0)
        opt.step()
        out[f"norm_{step}"] = np.float32(tn.item())
        for i, (prm, g) in enumerate(zip(params, gs)):
            out[f"g{step}_{i}"] = g
            out[f"p{step + 1}_{i}"] = prm.detach().numpy().copy()
    np.savez_compressed(HERE / "g5_clip_adam.npz", **out)


def g6():
    Q = synth.unit_rows(707, 32, 256)
    D = synth.unit_rows(708, 4096, 256)
    tq, td = torch.from_numpy(Q), torch.from_numpy(D)
    s = torch.matmul(tq, td.t())
    out = {"seed_q": np.int64(707), "seed_d": np.int64(708)}
    for k in (5, 10, 50):
        v, i = torch.topk(s, k)
        out[f"val_k{k}"] = v.numpy()
        out[f"idx_k{k}"] = i.numpy()
    srt = torch.sort(s, dim=1, descending=True).values[:, :51]
    out["min_gap_top51"] = (srt[:, :-1] - srt[:, 1:]).min(dim=1).values.numpy()
    # single-query call site (evaluators.py:185-186): [1,H] x [H,N] -> squeeze -> topk
    s1 = torch.matmul(tq[:1], td.t()).squeeze(0)
    v1, i1 = torch.topk(s1, 10)
    out["single_val"], out["single_idx"] = v1.numpy(), i1.numpy()
    # constructed exact ties: documents 7, 99 and 3000 are bitwise copies of one row, so their
    # scores tie exactly.  torch.topk's order among them is unspecified; record what it did.
    D2 = D.copy()
    D2[99] = D2[7]
    D2[3000] = D2[7]
    q2 = D2[7:8].copy()
    s2 = torch.matmul(torch.from_numpy(q2), torch.from_numpy(D2).t())
    v2, i2 = torch.topk(s2, 5)
    out["tie_val"], out["tie_idx_torch_unspecified"] = v2.numpy(), i2.numpy()
    np.savez_compressed(HERE / "g6_scoring.npz", **out)


def g7():
    """BatchEvaluator (evaluators.py:18-79) with a stub model returning fixed embeddings."""
    Qe = synth.unit_rows(808, 48, 32)
    De = synth.unit_rows(809, 48, 32)
    De[:24] = (0.6 * Qe[:24] + 0.4 * De[:24])           # make half the positives rank high
    De /= np.linalg.norm(De, axis=1, keepdims=True)
    De = De.astype(np.float32)
    Ne = synth.unit_rows(810, 48, 32)

    class Stub:
        def __init__(self):
            self.qi = self.di = 0

        def eval(self):
            pass

        def encode_query(self, x):
            return torch.from_numpy(Qe[x[:, 0].numpy()])

        def encode_document(self, x):
            tag = x[:, 1].numpy()
            rows = x[:, 0].numpy()
            return torch.from_numpy(np.where(tag[:, None] == 0, De[rows], Ne[rows]))

    loader = []
    for s in range(0, 48, 16):
        r = torch.arange(s, s + 16)
        qb = torch.stack([r, torch.zeros_like(r)], 1)
        pb = torch.stack([r, torch.zeros_like(r)], 1)
        nb = torch.stack([r, torch.ones_like(r)], 1)
        loader.append((qb, pb, nb))
    metrics, val_loss = refeval.BatchEvaluator().evaluate(Stub(), loader, torch.device("cpu"),
                                                          {"MARGIN": 0.5})
    np.savez_compressed(HERE / "g7_batch_eval.npz", q=Qe, d=De, n=Ne,
                        recall1=np.float64(metrics["Recall@1"]), recall5=np.float64(metrics["Recall@5"]),
                        recall10=np.float64(metrics["Recall@10"]), mrr=np.float64(metrics["MRR"]),
                        val_loss=np.float64(val_loss))


def _vocab():
    words = ["the", ",", ".", "of", "and", "w5", "w6", "w7", "machine", "learning", "what", "is",
             "don", "t", "e", "mail", "!", "?", ";"]
    return {w: i for i, w in enumerate(words)}


def g8():
    with tempfile.TemporaryDirectory() as td:
        pth = Path(td) / "word_to_idx.pkl"
        with open(pth, "wb") as f:
            pickle.dump(_vocab(), f)
        tok = reftok.PretrainedTokenizer(str(pth))
        cases = ["The w5, of W7! don't e-mail", "", None, "what is machine learning?",
                 "  multiple   spaces\tand\nnewlines ; ok", "unknown zzz words", "the the", "123 w5_w6"]
        res = [{"text": c, "ids": tok.encode(c)} for c in cases]
        doc = {"vocab": _vocab(), "unk_id": tok.unk_token_id, "vocab_size": tok.vocab_size(), "cases": res}
    with open(HERE / "g8_tokenizer.json", "w") as f:
        json.dump(doc, f, indent=1)


def g9():
    V0, E, H, seed = len(_vocab()), 20, 32, 909
    with tempfile.TemporaryDirectory() as td:
        td = Path(td)
        (td / "frontend").mkdir()
        art = td / "artifacts" / "run-x"
        art.mkdir(parents=True)
        (td / "frontend" / "config.json").write_text(json.dumps({"ARTIFACTS_PATH": str(art)}))
        with open(art / "word_to_idx.pkl", "wb") as f:
            pickle.dump(_vocab(), f)
        V = V0 + 1  # tokenizer appends <UNK>
        m, cfg, table = _two_tower(V, E, H, seed)
        torch.save(m.state_dict(), art / "model.pth")
        cfg_saved = {"HIDDEN_DIM": H, "RNN_TYPE": "GRU", "NUM_LAYERS": 1, "BIDIRECTIONAL": False,
                     "DROPOUT": 0.0, "NORMALIZE_OUTPUT": True, "EMBED_DIM": E, "VOCAB_SIZE": V}
        (art / "config.json").write_text(json.dumps(cfg_saved))
        cwd = os.getcwd()
        os.chdir(td)
        try:
            import query_inferencer as refqi  # reads frontend/config.json at import (query_inferencer.py:15)
            inf = refqi.QueryInferencer(str(art), device=torch.device("cpu"))
            queries = ["what is machine learning", "w5 of w6 , w7 .", "zzz", "What IS the e-mail?"]
            embs = np.stack([inf.get_query_embedding(q) for q in queries])
            empty = inf.get_query_embedding("")
            err = ""
            try:
                inf.get_query_embedding("the the")
            except RuntimeError as e:
                err = type(e).__name__ + ": " + str(e).splitlines()[0]
        finally:
            os.chdir(cwd)
    np.savez_compressed(HERE / "g9_inferencer.npz", queries=np.array(queries), embs=embs, empty=empty,
                        the_the_error=np.array(err), dims=np.array([V, E, H, seed]),
                        vocab_json=np.array(json.dumps(_vocab())))


def g12():
    """The reference trains the embedding table when no pretrained vectors are passed (model.py:23-27).  One encoder,
    loss = sum(out * c) for a fixed random c; records d loss / d embedding.weight (row 0 = padding_idx gets none) and
    the GRU weight gradients."""
    out = {}
    for tag, (layers, bi) in {"uni": (1, False), "bi": (2, True)}.items():
        V, E, H, seed = 48, 12, 32, 1212 + (5 if bi else 0)
        table = synth.make_table(seed, V, E)
        sd = synth.make_encoder_state(seed + 1, E, H, layers, bi)
        enc = refmodel.RNNEncoder(V, E, H, pretrained_embeddings=None, num_layers=layers, bidirectional=bi)
        full = {"embedding.weight": torch.from_numpy(table)}
        full.update({k: torch.from_numpy(v) for k, v in sd.items()})
        enc.load_state_dict(full)
        assert enc.embedding.weight.requires_grad
        enc.train()
        ids = synth.make_ids(seed + 2, B=7, T=9, V=V, zero_inside=0.15)
        c = np.random.RandomState(seed + 3).standard_normal((7, H)).astype(np.float32)
        y = enc(torch.from_numpy(ids))
        (y * torch.from_numpy(c)).sum().backward()
        out[f"{tag}_ids"], out[f"{tag}_c"], out[f"{tag}_out"] = ids, c, y.detach().numpy()
        out[f"{tag}_dims"] = np.array([V, E, H, seed, layers, int(bi)])
        for k, prm in enc.named_parameters():
            out[f"{tag}_grad_{k}"] = prm.grad.numpy().copy()
    np.savez_compressed(HERE / "g12_table_grad.npz", **out)


def g13():
    """The reference builds getattr(nn, RNN_TYPE.upper()) (model.py:30): nn.LSTM (h_n is kept, :59-60) and nn.RNN
    (tanh).  loss = sum(out * c) for a fixed random c; forward outputs and d loss / d every parameter."""
    out = {}
    for cell, gates in (("LSTM", 4), ("RNN", 1)):
        for tag, (layers, bi) in {"uni": (1, False), "bi": (2, True)}.items():
            V, E, H, seed = 40, 12, 32, 1313 + gates + (5 if bi else 0)
            table = synth.make_table(seed, V, E)
            sd = synth.make_encoder_state(seed + 1, E, H, layers, bi, gates=gates)
            enc = refmodel.RNNEncoder(V, E, H, pretrained_embeddings=table, rnn_type=cell, num_layers=layers,
                                      bidirectional=bi)
            full = {"embedding.weight": torch.from_numpy(table)}
            full.update({k: torch.from_numpy(v) for k, v in sd.items()})
            enc.load_state_dict(full)
            enc.train()
            ids = synth.make_ids(seed + 2, B=6, T=10, V=V, zero_inside=0.1)
            c = np.random.RandomState(seed + 3).standard_normal((6, H)).astype(np.float32)
            y = enc(torch.from_numpy(ids))
            (y * torch.from_numpy(c)).sum().backward()
            key = f"{cell}_{tag}"
            out[f"{key}_ids"], out[f"{key}_c"], out[f"{key}_out"] = ids, c, y.detach().numpy()
            out[f"{key}_dims"] = np.array([V, E, H, seed, layers, int(bi), gates])
            for k, prm in enc.named_parameters():
                if prm.requires_grad:
                    out[f"{key}_grad_{k}"] = prm.grad.numpy().copy()
    np.savez_compressed(HERE / "g13_lstm_rnn.npz", **out)


def _hybrid_docs():
    """40 short passages over the synthetic vocabulary (none empty, none all-"the": both would raise or zero out)."""
    rs = np.random.RandomState(1109)
    words = [w for w in _vocab() if w.isalnum()]
    docs = []
    for i in range(40):
        n = int(rs.randint(3, 12))
        toks = [words[int(j)] for j in rs.randint(0, len(words), n)]
        if all(t == "the" for t in toks):
            toks[0] = "w5"
        docs.append(" ".join(toks) + (" ." if i % 3 == 0 else ""))
    return docs


def g11():
    """backend/simple_hybrid.py on the g9 artifacts: documents embedded with the SAME (query) encoder (:37-41),
    TfidfVectorizer(stop_words='english', max_features=10000) (:24), combined = alpha*dense + (1-alpha)*tfidf (:56),
    argsort descending (:59)."""
    V0, E, H, seed = len(_vocab()), 20, 32, 909
    docs = _hybrid_docs()
    queries = ["what is machine learning", "w5 of w6 , w7 .", "e mail w7 w7", "learning machine w6"]
    alphas = [0.3, 0.5, 1.0]
    with tempfile.TemporaryDirectory() as td:
        td = Path(td)
        (td / "frontend").mkdir()
        art = td / "artifacts" / "run-x"
        art.mkdir(parents=True)
        (td / "frontend" / "config.json").write_text(json.dumps({"ARTIFACTS_PATH": str(art)}))
        with open(art / "word_to_idx.pkl", "wb") as f:
            pickle.dump(_vocab(), f)
        V = V0 + 1
        m, cfg, table = _two_tower(V, E, H, seed)
        torch.save(m.state_dict(), art / "model.pth")
        cfg_saved = {"HIDDEN_DIM": H, "RNN_TYPE": "GRU", "NUM_LAYERS": 1, "BIDIRECTIONAL": False,
                     "DROPOUT": 0.0, "NORMALIZE_OUTPUT": True, "EMBED_DIM": E, "VOCAB_SIZE": V}
        (art / "config.json").write_text(json.dumps(cfg_saved))
        cwd = os.getcwd()
        os.chdir(td)
        try:
            import simple_hybrid as refsh  # imports query_inferencer, which reads frontend/config.json at import
            combined, top = [], []
            doc_emb = None
            for a in alphas:
                r = refsh.SimpleHybridRetriever(str(art), alpha=a)
                r.fit(list(docs))
                doc_emb = np.asarray(r.doc_embeddings, dtype=np.float32)
                per_q, per_top = [], []
                for q in queries:
                    # the reference's search() returns only (doc, score) pairs; its intermediate arithmetic is
                    # re-run here through the SAME objects to record the full score vector as well
                    q_tfidf = r.tfidf.transform([q])
                    from sklearn.metrics.pairwise import cosine_similarity
                    tfidf_scores = cosine_similarity(q_tfidf, r.tfidf_matrix)[0]
                    q_emb = r.dense_retriever.get_query_embedding(q)
                    dense = cosine_similarity([q_emb], r.doc_embeddings)[0]
                    comb = r.alpha * dense + (1 - r.alpha) * tfidf_scores
                    res = r.search(q, top_k=10)
                    order = [docs.index(d) for d, _ in res]
                    assert np.allclose([sc for _, sc in res], comb[order])
                    per_q.append(comb)
                    per_top.append(order)
                combined.append(per_q)
                top.append(per_top)
        finally:
            os.chdir(cwd)
    np.savez_compressed(HERE / "g11_hybrid.npz", docs=np.array(docs), queries=np.array(queries),
                        alphas=np.array(alphas, dtype=np.float64), combined=np.array(combined, dtype=np.float64),
                        top10=np.array(top, dtype=np.int64), doc_emb=doc_emb, dims=np.array([V, E, H, seed]),
                        vocab_json=np.array(json.dumps(_vocab())))


def g10():
    V, E, H, seed = 64, 16, 32, 101
    table = synth.make_table(seed, V, E)
    sd = synth.make_encoder_state(seed + 1, E, H)
    enc = ref_encoder(V, E, H, table, sd)
    res = {}
    for name, ids in {"all_zero_row": [[3, 4, 0], [0, 0, 0]], "empty_T0": [[]],
                      "index_out_of_range": [[1, 64, 2]]}.items():
        try:
            with torch.no_grad():
                enc(torch.tensor(ids, dtype=torch.long))
            res[name] = "no error"
        except Exception as e:  # noqa: BLE001
            res[name] = type(e).__name__ + ": " + str(e).splitlines()[0]
    with open(HERE / "g10_errors.json", "w") as f:
        json.dump(res, f, indent=1)



def g14():
    """CorpusEvaluator (evaluators.py:83-209) with stub towers returning fixed embeddings: the metric bookkeeping, the
    candidate pool and the seeded query sample.  Texts are 'q<i>' / 'd<j>'; the stub tokenizer maps a text to one id."""
    import random
    import evaluators as refeval  # reference backend/evaluators.py
    nq, nd, dim = 40, 120, 32
    Qe = synth.unit_rows(1401, nq, dim)
    De = synth.unit_rows(1402, nd, dim)
    rs = np.random.RandomState(1403)
    triplets = []
    for i in range(nq):
        for pos in rs.choice(nd, size=1 + i % 3, replace=False):
            triplets.append((i, int(pos), int(rs.randint(nd))))
    for i, p_, _ in triplets[::2]:                      # half the positives are made to rank high
        De[p_] = 0.55 * Qe[i] + 0.45 * De[p_]
    De /= np.linalg.norm(De, axis=1, keepdims=True)
    De = De.astype(np.float32)
    val_data = [(f"q{i}", f"d{p_}", f"d{n_}") for i, p_, n_ in triplets]

    class Tok:
        def encode(self, text):
            return [1 + int(text[1:]) + (1000 if text[0] == "d" else 0)]

    class Stub:
        def eval(self):
            pass

        def encode_query(self, x):
            return torch.from_numpy(Qe[x[:, 0].numpy() - 1])

        def encode_document(self, x):
            return torch.from_numpy(De[x[:, 0].numpy() - 1001])

    cases = []
    for seed, max_q, top_k in ((1234, 25, [1, 5, 10]), (7, 50, [1, 3]), (99, 8, [2, 10])):
        random.seed(seed)
        ev = refeval.CorpusEvaluator(top_k=top_k, max_candidates=1000, max_queries=max_q)
        import io, contextlib
        with contextlib.redirect_stdout(io.StringIO()):
            m = ev.evaluate(Stub(), val_data, Tok(), torch.device("cpu"))
        cases.append({"seed": seed, "max_queries": max_q, "top_k": top_k, "metrics": {k: float(v) for k, v in m.items()}})
    doc = {"q": Qe.tolist(), "d": De.tolist(), "triplets": triplets, "cases": cases}
    with open(HERE / "g14_corpus_eval.json", "w") as f:
        json.dump(doc, f)

if __name__ == "__main__":
    only = set(sys.argv[1:])
    for fn in (g1, g2, g3, g4, g5, g6, g7, g8, g9, g10, g11, g12, g13, g14):
        if only and fn.__name__ not in only:
            continue
        fn()
        print("wrote", fn.__name__)
