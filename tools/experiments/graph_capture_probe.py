#!/usr/bin/env python3
"""Which form of the direct train step survives torch.cuda.graph capture on this ROCm, and what a replay costs: each variant in
its own process (hipStreamEndCapture segfaulted inside the runtime for the two-stream capture: a crash only ends that process).
   python tools/experiments/graph_capture_probe.py            -> runs every variant as a subprocess
   python tools/experiments/graph_capture_probe.py VARIANT    -> one variant in this process"""
import json, subprocess, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
VARIANTS = ["eager2", "two", "eager2_qone", "two_qone", "eager2", "two"]


def one(variant):
    import numpy as np, torch
    import twotowermlretrieval_amd as tt
    from twotowermlretrieval_amd import trainer as T
    from bench import make_ids
    dev = torch.device("cuda"); rs = np.random.RandomState(0)
    V, E, H, B = 400003, 300, 256, 512
    table = (rs.standard_normal((V, E)) * 0.3).astype(np.float32)
    torch.manual_seed(0)
    m = tt.TwoTowerModel({"VOCAB_SIZE": V, "EMBED_DIM": E, "HIDDEN_DIM": H}, table).to(dev).train()
    q, _ = make_ids(rs, B, 6, 1, 30, V); p, _ = make_ids(rs, B, 70, 10, 250, V); n, _ = make_ids(rs, B, 70, 10, 250, V)
    q, p, n = q.to(dev), p.to(dev), n.to(dev)
    opt = tt.FusedClipAdam(m.parameters(), lr=5e-5, max_norm=1.0)
    opt.check = False
    if variant.endswith("_qone"):     # the first co-residency plan: the query tower on the one-workgroup recurrences, no ordering
        m.query_encoder.one_workgroup = True
        variant = variant[:-5]
    rows = {m.query_encoder: B, m.doc_encoder: 2 * B}
    if variant == "two_norecord":
        torch.Tensor.record_stream = lambda self, s: None

    def run():
        with T._towers_in_flight(m, opt, rows) as plan:
            if variant.startswith("fwd"):
                with torch.no_grad():
                    pass
                # forward halves only: the two tower calls, on two streams or one
                cur = torch.cuda.current_stream(dev)
                ss = T._tower_streams(dev)[:2] if variant == "fwd_two" else (cur, cur)
                outs = []
                for enc, ids, s in zip((m.doc_encoder, m.query_encoder), (p, q), ss):
                    s.wait_stream(cur)
                    with torch.cuda.stream(s):
                        outs.append(enc._run_forward(ids, train=True)[0])
                for s in ss:
                    cur.wait_stream(s)
                opt._pending_status.clear()
                return outs[0]
            return T._train_step_direct(m, opt, q, p, n, 0.5, join_on_caller=(variant != "eager2"), plan=plan)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            run()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    if variant == "eager2":
        fn = run
    else:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            run()
        fn = g.replay
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    print(json.dumps(dict(variant=variant, ok=True, ms=round((time.perf_counter() - t0) / 20 * 1e3, 3))), flush=True)


if len(sys.argv) > 1:
    one(sys.argv[1])
else:
    for v in VARIANTS:
        r = subprocess.run([sys.executable, __file__, v], capture_output=True, text=True, timeout=300)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        print(lines[-1] if lines else json.dumps(dict(variant=v, ok=False, rc=r.returncode, err=r.stderr.strip().splitlines()[-1:] if r.stderr.strip() else None)), flush=True)
