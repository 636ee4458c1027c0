#!/usr/bin/env python3
"""Tower forward / train step at the reference's default config.json shape: E=200, HIDDEN_DIM 256, 2 layers, bidirectional,
dropout 0.2 (backend/config.json:13-17) -- BASELINE configs[0]'s model on synthetic MS-MARCO-shaped batches."""
import sys, json, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import twotowermlretrieval_amd as tt
from bench import make_ids

def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters

def main():
    dev = torch.device("cuda"); rs = np.random.RandomState(0)
    V, E, H = 400003, 200, 256
    table = (rs.standard_normal((V, E)) * 0.3).astype(np.float32)
    torch.manual_seed(0)
    m = tt.TwoTowerModel({"VOCAB_SIZE": V, "EMBED_DIM": E, "HIDDEN_DIM": H, "NUM_LAYERS": 2, "BIDIRECTIONAL": True,
                          "DROPOUT": 0.2}, table).to(dev)
    for enc in (m.query_encoder, m.doc_encoder): enc.check_inputs = False
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512   # (64 = config.json's own BATCH_SIZE)
    q, qt = make_ids(rs, B, 6, 1, 30, V); p, pt = make_ids(rs, B, 70, 10, 250, V); n, nt = make_ids(rs, B, 70, 10, 250, V)
    q, p, n = q.to(dev), p.to(dev), n.to(dev)
    m.eval()
    with torch.no_grad():
        t_doc = timeit(lambda: m.encode_document(p))
        t_q = timeit(lambda: m.encode_query(q))
        big, bt = make_ids(rs, 8192, 70, 10, 250, V); big = big.to(dev)
        t_big = timeit(lambda: m.encode_document(big), iters=3, warm=1)
    print(json.dumps(dict(what="config.json shape: doc tower fwd", B=B, tokens=pt, ms=round(t_doc * 1e3, 3), tokens_per_s=round(pt / t_doc))), flush=True)
    print(json.dumps(dict(what="config.json shape: query tower fwd", B=B, tokens=qt, ms=round(t_q * 1e3, 3))), flush=True)
    print(json.dumps(dict(what="config.json shape: index build fwd", B=8192, tokens=bt, ms=round(t_big * 1e3, 3), tokens_per_s=round(bt / t_big))), flush=True)
    m.train()
    opt = tt.FusedClipAdam(m.parameters(), lr=5e-5, max_norm=1.0)
    t_tr = timeit(lambda: tt.train_step(m, opt, q, p, n, margin=0.5, concurrent_towers=True), iters=20, warm=3)
    print(json.dumps(dict(what="config.json shape: train step", triplets=B, tokens=qt + pt + nt, ms=round(t_tr * 1e3, 3), triplets_per_s=round(B / t_tr))), flush=True)

if __name__ == "__main__":
    main()
