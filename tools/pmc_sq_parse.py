#!/usr/bin/env python3
"""One rocprofv3 --pmc pass (SQ / GRBM counters, --kernel-trace --output-format csv) -> per kernel: mean counters per dispatch,
mean duration, and for MFMA kernels the two derived figures DESIGN quotes:
  clock_MHz  = GRBM_GUI_ACTIVE / 8 / duration        (rocprofv3 sums the counter over the 8 XCDs: MI355X_MICROARCH.md, DVFS)
  mfma_busy  = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)   (the counter counts cycles: 16 per 16x16x32 MFMA)
    python tools/pmc_sq_parse.py DIR [kernel-name substring ...]   (prints text; the last line is a JSON object of the first match)"""
import collections, csv, json, sys
from pathlib import Path

root, want = Path(sys.argv[1]), sys.argv[2:]
ctr = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
key_of = {}
for f in root.rglob("*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        if want and not any(w in name for w in want):
            continue
        k = (name[:90], r.get("Grid_Size", ""))
        ctr[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        key_of[r.get("Dispatch_Id")] = k
for f in root.rglob("*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        k = key_of.get(r.get("Dispatch_Id"))
        if k is not None:
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
first = None
for k, cs in sorted(ctr.items(), key=lambda kv: -sum(dur.get(kv[0], [0]))):
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    us = sum(dur[k]) / len(dur[k]) if dur.get(k) else float("nan")
    row = {"kernel": k[0], "grid": k[1], "dispatches": len(next(iter(cs.values()))), "mean_us": round(us, 1), **{c: round(v) for c, v in sorted(m.items())}}
    if m.get("GRBM_GUI_ACTIVE") and us == us:
        cyc = m["GRBM_GUI_ACTIVE"] / 8.0
        row["clock_MHz"] = round(cyc / us)
        if m.get("SQ_VALU_MFMA_BUSY_CYCLES"):
            row["mfma_busy"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc), 4)
        if m.get("SQ_WAVE_CYCLES"):
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                if m.get(c) is not None:
                    row[c + "_share_of_wave_cycles"] = round(m[c] / m["SQ_WAVE_CYCLES"], 4)
    print(json.dumps(row))
    first = first or row
if first:
    print(json.dumps(first))
