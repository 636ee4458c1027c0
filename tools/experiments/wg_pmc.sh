# PMC passes over the big wgrad16 launches of tools/train_prof.py (one pass per counter group; --kernel-trace only)
# (a pass with TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_GATE_EN1_sum never finished on this pool: left out)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/wg_pmc; rm -rf $OUT; mkdir -p $OUT
i=0
while read -r grp; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python3 tools/train_prof.py 3 > /dev/null 2>&1 || echo "pass $i failed: $grp"
done <<'LIST'
SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL
TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum
TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum TCC_BUSY_avr
TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum
LIST
python3 - <<'PY'
import csv, collections, json
from pathlib import Path
for d in sorted(Path("gpurun_out/wg_pmc").glob("p*")):
    per = collections.defaultdict(dict)
    for f in d.rglob("*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            if "wgrad16" in row["Kernel_Name"]:
                per[(row["Kernel_Name"][28:48], row["Dispatch_Id"])][row["Counter_Name"]] = float(row["Counter_Value"])
    by = collections.defaultdict(list)
    for (k, _), c in per.items():
        by[k].append(c)
    for k, lst in sorted(by.items()):
        key = sorted(lst[0])[0]
        lst.sort(key=lambda c: -max(c.values()))
        big = lst[:max(1, len(lst) // 5)]
        print(d.name, k, json.dumps({n: round(sum(c[n] for c in big) / len(big)) for n in sorted(big[0])}), "n=%d of %d" % (len(big), len(lst)))
PY
rm -rf $OUT
