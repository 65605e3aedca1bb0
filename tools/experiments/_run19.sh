set -x
mkdir -p gpurun_out/r19
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
for v in 0 7; do
  SIDLSG_ATTN_XCD=$v timeout 600 rocprofv3 --kernel-trace --stats -d $root/gpurun_out/r19/s$v -o s --output-format csv -- python $root/tools/bench_kernels.py attn > $root/gpurun_out/r19/s$v.log 2>&1
  cp $root/gpurun_out/r19/s$v/s_kernel_stats.csv $root/gpurun_out/r19/stats_xcd$v.csv; rm -rf $root/gpurun_out/r19/s$v
done
cd $root
python - <<'PY'
import csv
a={r['Name']:r for r in csv.DictReader(open('gpurun_out/r19/stats_xcd0.csv'))}
b={r['Name']:r for r in csv.DictReader(open('gpurun_out/r19/stats_xcd7.csv'))}
for k in sorted(a):
    if 'attn' in k and k in b: print(f"{k[:60]:60s} n={a[k]['Calls']:>4} xcd0 {float(a[k]['AverageNs'])/1e3:8.1f} us  xcd7 {float(b[k]['AverageNs'])/1e3:8.1f} us")
PY
for rep in 1 2; do
for cfg in 1 3 7 0; do
  SIDLSG_ATTN_XCD=$cfg timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('attn_xcd=$cfg', d['ms_per_step'], d['value'], d['loss_check'])" >> gpurun_out/r19_ab.log
done; done
cat gpurun_out/r19_ab.log
