# Upper bound of "feed wgrad16's B operand from pre-split fp16 hi/lo images" (VERDICT r03 item 4), before building it: a
# -DTT_WG_EXP_B_PRESPLIT library (tools/build_variant.py wgpre -DTT_WG_EXP_B_PRESPLIT) takes the 16 bytes it loads per float4 as
# 4 hi + 4 lo halves -- same bytes per element as the images would have, no conversion instructions for B; results are wrong,
# only the kernel durations matter.  Interleaved with the product library on one box.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cp twotowermlretrieval_amd/libtt.so /tmp/libtt_keep.so
trap 'cp /tmp/libtt_keep.so twotowermlretrieval_amd/libtt.so' EXIT   # (whatever ends the script: the package never stays on the experiment's wrong-results library)
for rep in 1 2; do
  for v in product wgpre; do
    if [ $v = product ]; then cp /tmp/libtt_keep.so twotowermlretrieval_amd/libtt.so; else cp ab/libtt_$v.so twotowermlretrieval_amd/libtt.so; fi
    rm -rf gpurun_out/prof_wgp
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_wgp -o tr -- python3 tools/train_prof.py 10 > gpurun_out/prof_wgp.log 2>&1
    echo "== $v (rep $rep): $(grep concurrent_towers.*true gpurun_out/prof_wgp.log)"
    python3 tools/rocpd_stats.py gpurun_out/prof_wgp/tr_results.db gpurun_out/prof_wgp.csv > /dev/null 2>&1
    python3 -c "
import csv
for r in csv.DictReader(open('gpurun_out/prof_wgp.csv')):
    if 'wgrad16' in r['Name'] and int(r['Calls']) >= 10: print('   ', r['Name'][-45:], 'grid', r['GridX'], 'calls', r['Calls'], 'avg_us', round(float(r['AverageNs']) / 1e3, 1))
"
  done
done
rm -rf gpurun_out/prof_wgp
cp /tmp/libtt_keep.so twotowermlretrieval_amd/libtt.so
