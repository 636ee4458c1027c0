#!/usr/bin/env python3
"""profiles/pmc_traffic.json from one round's committed PMC summaries (tools/collect_evidence.sh's TAG_pmc_summary.txt and
tools/pmc_sq_parse.py's TAG_pmc_sq_summary.txt):  python tools/pmc_traffic_update.py profiles/TAG_pmc_summary.txt [profiles/TAG_pmc_sq_summary.txt]
bytes per launch = FETCH_SIZE * 1024 * 2 (gfx950: FETCH_SIZE reports half of a 16-B/lane streaming read, MI355X_MICROARCH.md,
HBM) + WRITE_SIZE * 1024."""
import json, re, sys
from pathlib import Path

src = Path(sys.argv[1])
sect, vals = None, {}
for line in src.read_text().splitlines():
    if line.startswith("== "):
        sect = line[3:].strip()
        continue
    m = re.search(r"(\S.*?)\s+grid=\s*(\d+)\s+(FETCH_SIZE|WRITE_SIZE)\s+n=\s*\d+\s+mean=([\d.]+)", line)
    if m and sect:
        vals[(sect, m.group(1), m.group(3))] = float(m.group(4))


def traffic(sect, needle):
    f = [v for (s, n, c), v in vals.items() if s == sect and needle in n and c == "FETCH_SIZE"]
    w = [v for (s, n, c), v in vals.items() if s == sect and needle in n and c == "WRITE_SIZE"]
    return round(max(f) * 1024 * 2 + max(w) * 1024) if f and w else None


out = {"_source": f"{src} : rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace --output-format csv only; "
                  "tools/collect_evidence.sh) over tools/screen_probe.py 10000000 1024 (screen_kernel<false,4>), tools/screen_probe.py "
                  "10000000 32 (screen_stream_kernel<false,2>) and tools/pmc_probe.py (score_topk_kernel<8,64,false,*> at B=32 and B=1024), "
                  "10M x 256 docs; bytes per launch = FETCH_SIZE*1024*2 (gfx950: FETCH_SIZE reports half of a 16-B/lane streaming read, "
                  "MI355X_MICROARCH.md HBM section) + WRITE_SIZE*1024",
       "b32": traffic("k4", "score_topk_kernel<8, 64, false, true>"),
       "b1024": traffic("k4", "score_topk_kernel<8, 64, false, false>"),
       "screen_b1024": traffic("screen1024", "screen_kernel<false, 4>"),
       "stream_b32": traffic("screen32", "screen_stream_kernel<false, 2>")}
if len(sys.argv) > 2:
    sq = Path(sys.argv[2])
    last = json.loads([l for l in sq.read_text().splitlines() if l.startswith("{")][-1])
    out["_sq_source"] = f"{sq} : one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY " \
                        "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY, --kernel-trace) over tools/screen_probe.py 10000000 1024; tools/pmc_sq_parse.py"
    out["screen_b1024_mfma_busy"] = last.get("mfma_busy")
    out["screen_b1024_clock_MHz"] = last.get("clock_MHz")
    out["screen_b1024_profiled_kernel_us"] = last.get("mean_us")
Path("profiles/pmc_traffic.json").write_text(json.dumps(out, indent=1) + "\n")
print(json.dumps(out, indent=1))
