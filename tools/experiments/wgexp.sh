cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cp twotowermlretrieval_amd/libtt.so /tmp/libtt_keep.so
for v in 0 v6 v0 exp32 exp33; do
  if [ $v != 0 ]; then cp ab/libtt_wg$v.so twotowermlretrieval_amd/libtt.so; fi
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_wge$v -o tr -- python3 tools/train_prof.py 4 > /dev/null 2>&1
  python3 - <<PY
import sqlite3
db = sqlite3.connect("gpurun_out/prof_wge$v/tr_results.db")
rows = db.execute("select name, end - start from kernels where name like '%wgrad16%'").fetchall()
for tag in ("<5>", "<4>"):
    d = sorted(x[1] for x in rows if tag in x[0])
    big = [x for x in d if x > d[-1] * 0.5]
    print("exp=$v wgrad16%s large launches: n=%d median %.1f us" % (tag, len(big), big[len(big) // 2] / 1e3))
PY
  rm -rf gpurun_out/prof_wge$v
done
cp /tmp/libtt_keep.so twotowermlretrieval_amd/libtt.so
