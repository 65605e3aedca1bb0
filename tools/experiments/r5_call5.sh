#!/bin/bash
# round 5, call 5: gemm_s64_kernel (tests, per-launch A/B under rocprofv3, step A/B) + the distributed tests after the flush fix
root=$(pwd)
out=$root/gpurun_out/r5c5
mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_dist.py tests/test_gpu_grouped.py tests/test_gpu_fp8.py -x -q -m gpu > $out/tests_ops.log 2>&1
tail -3 $out/tests_ops.log
timeout 1500 python -m pytest tests/test_gpu_unet.py -x -q -m gpu -k "forward_backward or iteration_matches or glue or graphed" > $out/tests_unet.log 2>&1
tail -3 $out/tests_unet.log
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  SIDLSG_GEMM_S64=$v rocprofv3 --kernel-trace --stats -d /tmp/sg_$v -o a --output-format csv -- python $root/tools/ab/small_gemm.py > $out/small_$v.log 2>&1
  tail -1 $out/small_$v.log
  grep -E "gemm_" /tmp/sg_$v/a_kernel_stats.csv > $out/small_$v_stats.csv
  python3 - /tmp/sg_$v/a_kernel_trace.csv <<'PY'
import sys, csv, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    n = r['Kernel_Name']
    if 'gemm' not in n: continue
    key = (n[:40], r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size', ''))
    agg.setdefault(key, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000)
for k, v in agg.items():
    v = sorted(v)
    print(f'  {k[0]:40s} grid {k[1]:>8s} n={len(v)} median {v[len(v)//2]:7.1f} us')
PY
done
cd $root
for i in 1 2 3; do
  for d in 0 1; do
    SIDLSG_GEMM_S64=$d SIDLSG_BENCH_DETAIL=/tmp/d.json python bench.py --no-cpu-baseline --no-kernel-timing --steps 12 --warmup 4 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('step S64=$d', round(d['ms_per_step'],2), 'ms', d['loss_check'])"
  done
done
