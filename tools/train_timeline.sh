#!/bin/bash
# Kernel timeline of one training step:  bash tools/train_timeline.sh TAG [config1 [B]]  ->  gpurun_out/TAG_train_timeline.txt, TAG_train_step_kernel_stats.csv
TAG=${1:-tl}
OUT=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf $OUT/${TAG}_tprof
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_tprof -o tr -- python3 tools/train_prof.py 6 $2 $3 > $OUT/${TAG}_train_prof.log 2>&1 || exit 2
DB=$(ls $OUT/${TAG}_tprof/*results.db $OUT/${TAG}_tprof/*/*results.db 2>/dev/null | head -1)
python3 tools/rocpd_timeline.py "$DB" > $OUT/${TAG}_train_timeline.txt && python3 tools/rocpd_stats.py "$DB" $OUT/${TAG}_train_step_kernel_stats.csv
rm -rf $OUT/${TAG}_tprof
grep "^{" $OUT/${TAG}_train_prof.log
cut -c1-${COLS:-118} $OUT/${TAG}_train_timeline.txt
