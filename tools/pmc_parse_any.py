#!/usr/bin/env python3
"""Mean counter values per dispatch by kernel and grid from rocprofv3 --pmc csv output:  python tools/pmc_parse_any.py DIR [substring ...]"""
import csv, sys, collections, json
from pathlib import Path
want = sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in Path(sys.argv[1]).rglob("*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        name = row.get("Kernel_Name", "")
        if want and not any(t in name for t in want):
            continue
        acc[(name[:80], row.get("Grid_Size", ""))][row["Counter_Name"]].append(float(row["Counter_Value"]))
for (name, grid), ctrs in sorted(acc.items()):
    print(name, "grid=" + grid, json.dumps({k: round(sum(v) / len(v)) for k, v in sorted(ctrs.items())}), "n=%d" % len(next(iter(ctrs.values()))))
