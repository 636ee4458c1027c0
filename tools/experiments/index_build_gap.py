#!/usr/bin/env python3
"""Where the index build from strings loses its last 15 % against the GPU-only rate: the same 16 batches of 16 384 passages
(a) resident ids, encode only; (b) ids in pinned memory, copied on a side stream as embed_corpus does, no tokenising;
(c) as (b) plus the [N, H] result copy; (d) embed_corpus itself."""
import sys, json, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np, torch
import twotowermlretrieval_amd as tt
from twotowermlretrieval_amd.evaluators import embed_corpus
import bench

dev = torch.device("cuda:0")
inp = bench.make_encoder_inputs(dev, with_index_batch=False)
model = inp["model"].eval()
V = bench.ENC_V
words = ["the", ",", ".", "of", "and"] + [f"w{i}" for i in range(5, V - 1)]
tok = tt.PretrainedTokenizer(word2idx={w: i for i, w in enumerate(words)})
rs = np.random.RandomState(3)
n_docs, bs = 262_144, 16384
lens = np.clip(rs.poisson(70, n_docs), 10, 250)
z = rs.zipf(1.07, int(lens.sum())) % (V - 1)
docs, p0 = [], 0
for L_ in lens:
    docs.append(" ".join(map(words.__getitem__, z[p0:p0 + L_]))); p0 += L_
host = [tok.encode_batch(docs[i:i + bs], pin=True) for i in range(0, n_docs, bs)]
resident = [h.to(dev) for h in host]
res = torch.empty((n_docs, 256), dtype=torch.float32, device=dev)


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best


def a():
    with torch.no_grad():
        for x in resident:
            model.encode_document(x)


def b(copy_out=False):
    cs, cur = torch.cuda.Stream(device=dev), torch.cuda.current_stream(dev)
    with torch.no_grad():
        for k, h in enumerate(host):
            with torch.cuda.stream(cs):
                ids = h.to(dev, non_blocking=True)
            cur.wait_stream(cs)
            ids.record_stream(cur)
            emb = model.encode_document(ids)
            if copy_out:
                res[k * bs:k * bs + emb.shape[0]].copy_(emb)


for name, fn in (("a resident ids", a), ("b pinned ids + side-stream copies", b), ("c b + result copy", lambda: b(True)),
                 ("d embed_corpus", lambda: embed_corpus(model, tok, docs, dev)),
                 ("d0 embed_corpus, copies not ahead", lambda: embed_corpus(model, tok, docs, dev, copy_ahead=0)),
                 ("d embed_corpus again", lambda: embed_corpus(model, tok, docs, dev)),
                 ("d0 again", lambda: embed_corpus(model, tok, docs, dev, copy_ahead=0)),
                 ("d 14 threads", lambda: embed_corpus(model, tok, docs, dev, threads_per_producer=14)),
                 ("d 12 threads", lambda: embed_corpus(model, tok, docs, dev, threads_per_producer=12)),
                 ("d 8 threads", lambda: embed_corpus(model, tok, docs, dev, threads_per_producer=8)),
                 ("d 2 producers x 7", lambda: embed_corpus(model, tok, docs, dev, producers=2, threads_per_producer=7)),
                 ("d 16 threads", lambda: embed_corpus(model, tok, docs, dev)), ("a again", a)):
    t = timed(fn)
    print(json.dumps({"what": name, "ms": round(t * 1e3, 2), "docs_per_s": round(n_docs / t)}), flush=True)
