#!/usr/bin/env python3
"""Launch the K4 kernel a few times at the two bench shapes so a rocprofv3 --pmc pass can read
per-dispatch counters.   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT -- python3 tools/pmc_probe.py [N_DOCS [B ...]]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else bench.N_DOCS
dev = torch.device("cuda:0")
docs = bench.gen_rows(0, n, dev)
q = bench.gen_queries(bench.BATCH, dev)
shapes = [(int(b), 10, 2) for b in sys.argv[2:]] or [(32, 10, 3), (1024, 10, 2)]
for B, k, it in shapes:
    ms, ms_br = bench.kernel_only_ms(q[:B].contiguous(), docs, k, iters=it, warm=1)
    print(f"B={B} k={k} N={n} kernel_ms={ms:.4f} with_sample_pass_ms={ms_br:.4f}", flush=True)
