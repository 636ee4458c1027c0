cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 200 python3 tools/config1_bench.py 64 > gpurun_out/r04_z_config1_b64.log 2>&1 || exit 2
grep "^{" gpurun_out/r04_z_config1_b64.log
rm -rf gpurun_out/c1prof
rocprofv3 --kernel-trace --stats -d gpurun_out/c1prof -o tr -- python3 tools/config1_bench.py 512 > gpurun_out/r04_z_config1_b512.log 2>&1 || exit 3
grep "^{" gpurun_out/r04_z_config1_b512.log
DB=$(ls gpurun_out/c1prof/*results.db gpurun_out/c1prof/*/*results.db 2>/dev/null | head -1)
python3 tools/rocpd_stats.py "$DB" gpurun_out/r04_z_config1_kernel_stats.csv > /dev/null
rm -rf gpurun_out/c1prof
python3 - <<'P'
import csv
rows = list(csv.DictReader(open('gpurun_out/r04_z_config1_kernel_stats.csv')))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
for r in rows[:28]:
    print(r['Name'][:70].ljust(70), 'grid', r['GridX'].rjust(8), r['GridY'], 'calls', r['Calls'].rjust(4), 'avg_us', round(float(r['AverageNs']) / 1e3, 1))
P
