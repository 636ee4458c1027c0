#!/usr/bin/env python3
"""Index build from STRINGS: synthetic MS-MARCO-shaped passages -> tokenise -> doc tower -> [N,256].
Reports the host front end alone (Python loop vs native), and the pipelined build."""
import sys, json, time, random
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import twotowermlretrieval_amd as tt
from twotowermlretrieval_amd.evaluators import embed_corpus
n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
V = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000
words = ["the", ",", ".", "of", "and"] + [f"w{i}" for i in range(5, V)]
tok = tt.PretrainedTokenizer(word2idx={w: i for i, w in enumerate(words)})
rs = np.random.RandomState(0)
zipf = np.minimum(rs.zipf(1.07, size=n_docs * 80) - 1, V - 1)
lens = np.clip(rs.poisson(70, n_docs), 10, 250)
docs, p = [], 0
for L in lens:
    docs.append(" ".join(words[i] for i in zipf[p:p + L]))
    p += L
n_tok = int(lens.sum())
t = time.time(); tok.encode_batch(docs[:20000], native=False); t_py = (time.time() - t) / 20000 * n_docs
tok.encode_batch(docs[:20000])   # (warm-up: the native table is built on first use, the per-thread scratch is faulted in)
t = time.time(); tok.encode_batch(docs[:20000]); tok.encode_batch(docs[20000:40000]); t_na = (time.time() - t) / 40000 * n_docs
print(json.dumps(dict(what="host front end alone", docs=n_docs, tokens=n_tok, python_tok_per_s=round(n_tok / t_py), native_tok_per_s=round(n_tok / t_na))), flush=True)
dev = torch.device("cuda:0")
E, H = 300, 256
table = (np.random.RandomState(1).standard_normal((tok.vocab_size(), E)) * 0.3).astype(np.float32)
m = tt.TwoTowerModel({"HIDDEN_DIM": H, "VOCAB_SIZE": tok.vocab_size(), "EMBED_DIM": E}, table).to(dev).eval()
embed_corpus(m, tok, docs[:20000], dev)
torch.cuda.synchronize()
from twotowermlretrieval_amd.tokenizer import host_cores
for prod, tpp in [tuple(map(int, x.split("x"))) for x in (sys.argv[3].split(",") if len(sys.argv) > 3 else "0x0,1x0,1x8,2x0,4x0,2x16,3x16,0x0".split(","))]:
    st = {}
    embed_corpus(m, tok, docs, dev, producers=prod, threads_per_producer=tpp)
    torch.cuda.synchronize()
    t = time.time(); emb = embed_corpus(m, tok, docs, dev, producers=prod, stats=st, threads_per_producer=tpp); torch.cuda.synchronize(); dt = time.time() - t
    print(json.dumps(dict(what="pipelined index build from strings", producers_arg=prod, **st, docs=n_docs, tokens=n_tok, s=round(dt, 3),
                          docs_per_s=round(n_docs / dt), tok_per_s=round(n_tok / dt), shape=list(emb.shape))), flush=True)
print(json.dumps(dict(host_cores=host_cores(), affinity=len(__import__("os").sched_getaffinity(0)),
                      cpu_max=open("/sys/fs/cgroup/cpu.max").read().strip() if __import__("os").path.exists("/sys/fs/cgroup/cpu.max") else None)))
