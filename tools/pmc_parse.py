#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc csv output: per kernel name and counter, mean value per dispatch.
   python tools/pmc_parse.py gpurun_out/pmc_fetch [more dirs...]"""
import csv, sys, collections
from pathlib import Path
acc = collections.defaultdict(list)
for d in sys.argv[1:]:
    for f in Path(d).rglob("*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            name = row.get("Kernel_Name", "")
            if not any(t in name for t in ("score_topk", "merge", "gru", "screen", "sgemm")):
                continue
            grid = row.get("Grid_Size", "")
            acc[(name[:70], grid, row["Counter_Name"])].append(float(row["Counter_Value"]))
for (name, grid, ctr), vals in sorted(acc.items()):
    print(f"{name:70s} grid={grid:>9s} {ctr:12s} n={len(vals):3d} mean={sum(vals)/len(vals):.1f}")
