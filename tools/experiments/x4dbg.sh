# phase clocks of the split forward recurrence: a -DTT_X4_DBG build (tools/build_variant.py x4dbg -DTT_X4_DBG) under the train step
cd $GRAFT_REPO_ROOT
cp twotowermlretrieval_amd/libtt.so /tmp/libtt_keep.so
cp ab/libtt_x4dbg.so twotowermlretrieval_amd/libtt.so
python3 tools/train_prof.py 8 2>&1 | grep -E "x4dbg|xbdbg|ms" | tail -5
cp /tmp/libtt_keep.so twotowermlretrieval_amd/libtt.so
