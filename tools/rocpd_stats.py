#!/usr/bin/env python3
"""Per-kernel summary (calls, total / average / min / max duration) from a rocprofv3 rocpd database
(`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd` writes DIR/NAME_results.db on this ROCm): the table
`--output-format csv` would put in *_kernel_stats.csv, with ONE ROW PER (kernel, grid, duration cluster).  A bench run
launches the same kernel at several shapes -- the 10M-document step, the 2M-document encoder-corpus leg, the screened
search's predicated no-op launches of the exact kernel, all with the same grid -- and an average over all of them says
nothing about any: durations of one (kernel, grid) are sorted and a new cluster starts wherever one is more than 1.6x the
one before.   python tools/rocpd_stats.py DB [out.csv]"""
import csv
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
grid = "grid_x, grid_y" if "grid_x" in cols else "0, 0"
rows = db.execute(f"select {name_col}, start, end, {grid} from kernels").fetchall()
by = {}
for name, s, e, gx, gy in rows:
    by.setdefault((re.sub(r"\s+", " ", name), gx, gy), []).append(e - s)
out = []
for (name, gx, gy), ds in by.items():
    ds.sort()
    cl = [[ds[0]]]
    for d in ds[1:]:
        if d > 1.6 * cl[-1][-1]:
            cl.append([])
        cl[-1].append(d)
    for i, c in enumerate(cl):
        out.append((name, gx, gy, f"{i + 1}/{len(cl)}", len(c), sum(c), min(c), max(c)))
total = sum(o[5] for o in out) or 1
out.sort(key=lambda o: -o[5])
w = csv.writer(open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout)
w.writerow(["Name", "GridX", "GridY", "Cluster", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
for name, gx, gy, cid, n, tot, mn, mx in out:
    w.writerow([name, gx, gy, cid, n, tot, round(tot / n, 1), round(100.0 * tot / total, 3), mn, mx])
