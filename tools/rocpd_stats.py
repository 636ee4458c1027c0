#!/usr/bin/env python3
"""Per-kernel summary (calls, total / average / min / max duration) from a rocprofv3 rocpd database
(`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd` writes DIR/NAME_results.db on this ROCm): the same table
`--output-format csv` would put in *_kernel_stats.csv, with ONE ROW PER (kernel, grid) -- a bench run launches the same
kernel at several shapes (10M-document step, 2M-document leg, predicated no-op launches) and an average over all of them
says nothing about any.   python tools/rocpd_stats.py DB [out.csv]"""
import csv
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
grid = "grid_x, grid_y" if "grid_x" in cols else "0, 0"
rows = db.execute(f"select {name_col}, start, end, {grid} from kernels").fetchall()
agg = {}
for name, s, e, gx, gy in rows:
    name = re.sub(r"\s+", " ", name)
    a = agg.setdefault((name, gx, gy), [0, 0, 1 << 62, 0])
    d = e - s
    a[0] += 1
    a[1] += d
    a[2] = min(a[2], d)
    a[3] = max(a[3], d)
total = sum(a[1] for a in agg.values()) or 1
out = sorted(agg.items(), key=lambda kv: -kv[1][1])
w = csv.writer(open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout)
w.writerow(["Name", "GridX", "GridY", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
for (name, gx, gy), (n, tot, mn, mx) in out:
    w.writerow([name, gx, gy, n, tot, round(tot / n, 1), round(100.0 * tot / total, 3), mn, mx])
