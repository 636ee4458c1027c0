#!/bin/bash
# Evidence of one build (run on the MI355X box from the repo root):  bash tools/collect_evidence.sh TAG
#   gpurun_out/TAG_bench_kernel_stats.csv   rocprofv3 --kernel-trace --stats over `bench.py --steps 5 --warmup 2 --no-cpu-baseline`
#                                           (tools/rocpd_stats.py: one row per kernel, grid and duration cluster)
#   gpurun_out/TAG_bench_profiled.json      the bench line of that same run
#   gpurun_out/TAG_pmc_summary.txt          FETCH_SIZE / WRITE_SIZE per launch (separate --pmc passes, --kernel-trace only) over
#                                           tools/screen_probe.py (B=1024; B=32) and tools/pmc_probe.py (K4 at B=32 and B=1024)
#   gpurun_out/TAG_pmc_sq_summary.txt       SQ / GRBM pass over tools/screen_probe.py (B=1024): mfma_busy, clock_MHz (tools/pmc_sq_parse.py)
#   then: python tools/pmc_traffic_update.py profiles/TAG_pmc_summary.txt profiles/TAG_pmc_sq_summary.txt  -> profiles/pmc_traffic.json
TAG=${1:-evidence}
OUT=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf $OUT/${TAG}_prof $OUT/${TAG}_pmc_*
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof -o bench -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_bench_profiled.json 2> $OUT/${TAG}_bench_profiled.err || exit 2
DB=$(ls $OUT/${TAG}_prof/*results.db $OUT/${TAG}_prof/*/*results.db 2>/dev/null | head -1)
python3 tools/rocpd_stats.py "$DB" $OUT/${TAG}_bench_kernel_stats.csv || exit 3
echo "kernel stats done" 
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_${ctr}_screen1024 -- python3 tools/screen_probe.py 10000000 1024 > /dev/null 2>&1 || exit 4
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_${ctr}_screen32 -- python3 tools/screen_probe.py 10000000 32 > /dev/null 2>&1 || exit 5
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_${ctr}_k4 -- python3 tools/pmc_probe.py > /dev/null 2>&1 || exit 6
  echo "$ctr passes done"
done
{
  for probe in screen1024 screen32 k4; do
    echo "== $probe"
    python3 tools/pmc_parse.py $OUT/${TAG}_pmc_FETCH_SIZE_$probe $OUT/${TAG}_pmc_WRITE_SIZE_$probe
  done
} > $OUT/${TAG}_pmc_summary.txt
rm -rf $OUT/${TAG}_pmc_FETCH_SIZE_* $OUT/${TAG}_pmc_WRITE_SIZE_* $OUT/${TAG}_prof
# matrix-pipe utilisation and clock of the headline kernel: ONE pass of SQ / GRBM counters (their own run, --kernel-trace only)
rm -rf $OUT/${TAG}_sq
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/${TAG}_sq -- python3 tools/screen_probe.py 10000000 1024 > /dev/null 2>&1 || exit 7
python3 tools/pmc_sq_parse.py $OUT/${TAG}_sq screen_kernel > $OUT/${TAG}_pmc_sq_summary.txt
rm -rf $OUT/${TAG}_sq
tail -1 $OUT/${TAG}_pmc_sq_summary.txt | cut -c1-400
cat $OUT/${TAG}_pmc_summary.txt | grep -v "grid=.*mean=0.0$" | cut -c1-150
