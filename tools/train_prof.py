#!/usr/bin/env python3
"""Only the training step of tools/encoder_bench.py (512 triplets, sequential towers), for a per-kernel profile:
   rocprofv3 --kernel-trace --stats -d gpurun_out/prof_train -- python3 tools/train_prof.py"""
import sys, json, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import twotowermlretrieval_amd as tt
from bench import make_ids

def main():
    dev = torch.device("cuda"); rs = np.random.RandomState(0)
    V, E, H, B = 400003, 300, 256, 512
    cfg = {"VOCAB_SIZE": V, "EMBED_DIM": E, "HIDDEN_DIM": H}
    if len(sys.argv) > 2 and sys.argv[2] == "config1":   # the reference's default model (backend/config.json:13-17)
        E = 200
        cfg.update(EMBED_DIM=E, NUM_LAYERS=2, BIDIRECTIONAL=True, DROPOUT=0.2)
        B = int(sys.argv[3]) if len(sys.argv) > 3 else B
    table = (rs.standard_normal((V, E)) * 0.3).astype(np.float32)
    torch.manual_seed(0)
    m = tt.TwoTowerModel(cfg, table).to(dev)
    for enc in (m.query_encoder, m.doc_encoder): enc.check_inputs = False
    q, qt = make_ids(rs, B, 6, 1, 30, V); p, pt = make_ids(rs, B, 70, 10, 250, V); n, nt = make_ids(rs, B, 70, 10, 250, V)
    q, p, n = q.to(dev), p.to(dev), n.to(dev)
    m.train()
    opt = tt.FusedClipAdam(m.parameters(), lr=5e-5, max_norm=1.0)
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    for conc in (False, True):
        for _ in range(2): tt.train_step(m, opt, q, p, n, margin=0.5, concurrent_towers=conc)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(iters): tt.train_step(m, opt, q, p, n, margin=0.5, concurrent_towers=conc)
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / iters
        print(json.dumps(dict(what="train step", concurrent_towers=conc, triplets=B, tokens=qt + pt + nt, ms=round(t * 1e3, 3))), flush=True)

if __name__ == "__main__":
    main()
