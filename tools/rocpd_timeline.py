#!/usr/bin/env python3
"""Kernel timeline of ONE iteration from a rocprofv3 rocpd database: the launches between the last two occurrences of a
marker kernel (default clip_adam_kernel = one training step), with start offset, duration, gap to the previous kernel's
end on the same queue and the queue id.   python tools/rocpd_timeline.py DB [marker]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
marker = sys.argv[2] if len(sys.argv) > 2 else "clip_adam_kernel"
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
qcol = next((c for c in ("queue_id", "stream_id", "queue", "stream") if c in cols), None)
rows = db.execute(f"select {name_col}, start, end{', ' + qcol if qcol else ''} from kernels order by start").fetchall()
marks = [i for i, r in enumerate(rows) if marker in r[0]]
if len(marks) < 2:
    sys.exit(f"fewer than two '{marker}' launches; columns: {cols}")
seg = rows[marks[-2] + 1:marks[-1] + 1]
t0 = seg[0][1]
last_end = {}
print(f"# {len(seg)} launches, {(seg[-1][2] - t0) / 1e3:.1f} us from the first start to the marker's end; columns: start_us dur_us gap_us queue name")
for r in seg:
    name = re.sub(r"\s+", " ", r[0])
    name = re.sub(r"\(anonymous namespace\)::", "", name)[:90]
    q = r[3] if qcol else 0
    gap = (r[1] - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = r[2]
    print(f"{(r[1] - t0) / 1e3:9.1f} {(r[2] - r[1]) / 1e3:8.1f} {gap:7.1f} {q!s:>4} {name}")
