#!/usr/bin/env python3
"""The stream choreography of the direct train step with trivial kernels, to find which fork / join pattern hipStreamEndCapture
cannot take on this ROCm (the full two-stream step segfaults inside the runtime at capture end; a two-stream forward alone is
fine).  Every variant in its own process.   python tools/experiments/graph_pattern_probe.py [VARIANT]"""
import json, subprocess, sys
VARIANTS = ["refork_same", "refork_fresh", "join_via_origin", "one_cycle", "refork_same_events"]


def one(v):
    import torch
    dev = torch.device("cuda")
    a, b, c = (torch.zeros(1 << 16, device=dev) for _ in range(3))
    s1, s2, s3 = (torch.cuda.Stream() for _ in range(3))
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1): a.add_(1)          # "document tower forward"
        with torch.cuda.stream(s2): b.add_(1)          # "query tower forward"
        if v == "one_cycle":
            cur.wait_stream(s1); cur.wait_stream(s2)
        elif v == "join_via_origin":
            cur.wait_stream(s1); cur.wait_stream(s2)
            c.copy_(a + b)                              # "loss" on the origin stream
            s1.wait_stream(cur); s2.wait_stream(cur)
            with torch.cuda.stream(s1): a.add_(c)
            with torch.cuda.stream(s2): b.add_(c)
            cur.wait_stream(s1); cur.wait_stream(s2)
            c.add_(a)                                   # "optimizer"
        else:
            s1.wait_stream(s2)                          # join for the loss
            with torch.cuda.stream(s1): c.copy_(a + b)
            sq = s3 if v == "refork_fresh" else s2
            if v == "refork_same_events":
                e = torch.cuda.Event(); e.record(s1); sq.wait_event(e)
            else:
                sq.wait_stream(s1)                      # fork again for the backwards
            with torch.cuda.stream(s1): a.add_(c)
            with torch.cuda.stream(sq): b.add_(c)
            s1.wait_stream(sq)
            with torch.cuda.stream(s1): c.add_(a)      # "optimizer"
            cur.wait_stream(s2); cur.wait_stream(s1)
            if v == "refork_fresh":
                cur.wait_stream(s3)
    g.replay(); g.replay()
    torch.cuda.synchronize()
    print(json.dumps(dict(variant=v, ok=True, a=float(a[0]), b=float(b[0]), c=float(c[0]))), flush=True)


if len(sys.argv) > 1:
    one(sys.argv[1])
else:
    for v in VARIANTS:
        r = subprocess.run([sys.executable, __file__, v], capture_output=True, text=True, timeout=200)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        print(lines[-1] if lines else json.dumps(dict(variant=v, ok=False, rc=r.returncode)), flush=True)
