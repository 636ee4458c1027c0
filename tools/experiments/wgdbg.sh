# phase clocks of the weight-gradient kernel: a -DTT_WG_DBG build (tools/build_variant.py wgdbg -DTT_WG_DBG) run under the train step
cd $GRAFT_REPO_ROOT
cp twotowermlretrieval_amd/libtt.so /tmp/libtt_keep.so
for v in ${WG_VARIANTS:-wgdbg}; do
  cp ab/libtt_$v.so twotowermlretrieval_amd/libtt.so
  echo "== $v"; python3 tools/train_prof.py 6 2>&1 | grep -E "wgdbg|ms" | tail -3
done
cp /tmp/libtt_keep.so twotowermlretrieval_amd/libtt.so
# wall time of the instrumented kernel (to turn clocks into a frequency)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cp ab/libtt_wgdbg.so twotowermlretrieval_amd/libtt.so
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_wgd -o tr -- python3 tools/train_prof.py 4 > /dev/null 2>&1
python3 tools/rocpd_stats.py gpurun_out/prof_wgd/tr_results.db 2>&1 | grep -i "wgrad" | head -4
rm -rf gpurun_out/prof_wgd
cp /tmp/libtt_keep.so twotowermlretrieval_amd/libtt.so
