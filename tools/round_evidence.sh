#!/bin/bash
# The evidence of one build in ONE gpurun call (run on the MI355X box from the repo root):  bash tools/round_evidence.sh TAG
#   gpurun_out/TAG_tests.log                 pytest -m gpu (all) tail
#   gpurun_out/TAG_bench_n1.json             python bench.py (defaults: full line with cpu_baseline)
#   gpurun_out/TAG_bench_profiled.json, TAG_bench_kernel_stats.csv, TAG_pmc_summary.txt      tools/collect_evidence.sh TAG
#   gpurun_out/TAG_train_timeline.txt, TAG_train_step_kernel_stats.csv    rocprofv3 --kernel-trace --stats over tools/train_prof.py 6
# Steps are joined with && : a step that fails or is killed ends the call.
TAG=${1:-evidence}
OUT=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_tests.log 2>&1; rc=$?; tail -3 $OUT/${TAG}_tests.log; [ $rc = 0 ] || exit 2
cp $OUT/tolerance_report.json $OUT/${TAG}_tolerance_report.json 2>/dev/null
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/${TAG}_smoke.log 2>&1 || { tail -5 $OUT/${TAG}_smoke.log; exit 6; }
tail -1 $OUT/${TAG}_smoke.log
timeout -k 10 900 python3 bench.py > $OUT/${TAG}_bench_n1.json 2> $OUT/${TAG}_bench_n1.err || exit 3
python3 -c "
import json; d = json.load(open('$OUT/${TAG}_bench_n1.json')); L = d['roofline']['legs']
print('value', d['value'], 'ms/step', d['ms_per_step'], 'roofline.frac', d['roofline']['frac'], 'hbm_screen', L['hbm_screen']['frac'], 'hbm_exact', L['hbm_exact_f32']['frac'], 'mfma_exact', L['mfma_exact_f32']['frac'])
print('train', L['train']['ms_per_step'], L['train'].get('graphed'))
print('streamed_bf16', L.get('streamed_bf16', {}).get('achieved'), 'dp_train', L.get('dp_train', {}).get('ms_per_step'), 'serve_b1', {k: L.get('serve_b1', {}).get(k) for k in ('p50_ms', 'p99_ms', 'stage_p50_ms', 'error')})
print('encoder', {k: (v.get('ms'), v.get('docs_per_s')) for k, v in L['encoder'].items() if isinstance(v, dict)})
print('index_build_from_strings', L.get('index_build_from_strings'))
print('cpu', d.get('cpu_baseline', {}).get('value'))"
bash tools/collect_evidence.sh $TAG || exit 4
rm -rf $OUT/${TAG}_tprof
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_tprof -o tr -- python3 tools/train_prof.py 6 > $OUT/${TAG}_train_prof.log 2>&1 || exit 5
DB=$(ls $OUT/${TAG}_tprof/*results.db $OUT/${TAG}_tprof/*/*results.db 2>/dev/null | head -1)
python3 tools/rocpd_timeline.py "$DB" > $OUT/${TAG}_train_timeline.txt && python3 tools/rocpd_stats.py "$DB" $OUT/${TAG}_train_step_kernel_stats.csv
rm -rf $OUT/${TAG}_tprof
head -3 $OUT/${TAG}_train_timeline.txt; grep "^{" $OUT/${TAG}_train_prof.log
