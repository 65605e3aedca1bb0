#!/bin/bash
# round 5, call 3: attention numbering / rotated-walk A/B (kernel averages + step), tests of the finer exchange segments and the shared
# time-embedding gradient buffer
root=$(pwd)
out=$root/gpurun_out/r5c3
mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_unet.py tests/test_gpu_dist.py tests/test_gpu_grouped.py tests/test_gpu_fp32.py -x -q -m gpu -k "forward_backward or segments or iteration_matches or two_rank or rccl or grouped or side_streams or graphed or fp32" > $out/tests.log 2>&1
tail -4 $out/tests.log
cd /tmp && export TMPDIR=/tmp
run() {  # name, XCD, ROT
  SIDLSG_ATTN_XCD=$2 SIDLSG_ATTN_ROT=$3 rocprofv3 --kernel-trace --stats -d /tmp/prof_$1 -o a --output-format csv -- python $root/tools/ab/attn_rot.py /tmp/attn_ref.pt > $out/$1.log 2>&1
  echo "== $1 (XCD=$2 ROT=$3) $(grep -c 'max diff' $out/$1.log) compared, $(tail -1 $out/$1.log)"
  grep -E "attn_(q|dkdv)_kernel" /tmp/prof_$1/a_kernel_stats.csv > $out/$1_stats.csv
  python3 - $out/$1_stats.csv <<'PY'
import sys, csv
for r in csv.reader(open(sys.argv[1])):
    print(f'  {r[0][:56]:56s} n={r[1]} avg {float(r[3])/1000:8.1f} us')
PY
}
run base 7 0
run xcd15 15 0
run xcd15_rot6 15 6
run xcd15_rot7 15 7
run base2 7 0
cd $root
for i in 1 2; do
  for cfg in "7 0" "15 6" "15 7"; do
    set -- $cfg
    SIDLSG_ATTN_XCD=$1 SIDLSG_ATTN_ROT=$2 SIDLSG_BENCH_DETAIL=/tmp/d.json python bench.py --no-cpu-baseline --no-kernel-timing --steps 12 --warmup 4 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('step XCD=$1 ROT=$2', round(d['ms_per_step'],2), 'ms', d['loss_check'])"
  done
done
