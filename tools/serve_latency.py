#!/usr/bin/env python3
"""End-to-end serving latency of one query (the reference's /search dense half, frontend/main.py:150-156):
string -> tokens -> query tower -> exact top-10 over a 10M x 256 resident corpus.  Synthetic artifacts."""
import sys, json, time, tempfile
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import bench
import twotowermlretrieval_amd as tt
from twotowermlretrieval_amd.evaluators import save_inference_artifacts
dev = torch.device("cuda:0")
V, E, H = 50_000, 300, 256
words = ["the", ",", ".", "of", "and"] + [f"w{i}" for i in range(5, V)]
tok = tt.PretrainedTokenizer(word2idx={w: i for i, w in enumerate(words)})
cfg = {"HIDDEN_DIM": H, "NUM_LAYERS": 1, "BIDIRECTIONAL": False, "BATCH_SIZE": 64}
table = (np.random.RandomState(1).standard_normal((tok.vocab_size(), E)) * 0.3).astype(np.float32)
m = tt.TwoTowerModel({**cfg, "VOCAB_SIZE": tok.vocab_size(), "EMBED_DIM": E}, table).to(dev)
with tempfile.TemporaryDirectory() as d:
    save_inference_artifacts(d, m, cfg, tok, ["w5 w6"], dev)
    inf = tt.QueryInferencer(d)
docs = bench.gen_rows(0, bench.N_DOCS, dev)
ix = tt.BruteForceIndex(docs, screen=True)
rs = np.random.RandomState(0)
queries = [" ".join(words[i] for i in rs.randint(5, V, rs.randint(3, 12))) for _ in range(200)]
lat, enc = [], []
for i, qs in enumerate(queries):
    t0 = time.perf_counter()
    qv = torch.from_numpy(inf.get_query_embedding(qs)).to(dev)     # the reference's API: numpy vector out
    t1 = time.perf_counter()
    v, idx = ix.search(qv, 10)
    top = idx.tolist()                                             # result on the host
    t2 = time.perf_counter()
    if i >= 20:
        enc.append(t1 - t0); lat.append(t2 - t0)
lat, enc = np.array(lat) * 1e3, np.array(enc) * 1e3
print(json.dumps(dict(what="one query end to end over 10M docs (host clock, result on host)", p50_ms=round(float(np.median(lat)), 3),
                      p99_ms=round(float(np.percentile(lat, 99)), 3), encode_p50_ms=round(float(np.median(enc)), 3),
                      search_p50_ms=round(float(np.median(lat - enc)), 3))), flush=True)
