#!/bin/bash
# round 5, call 2: new tests + attention numbering / rotated-walk A/B (rocprofv3 kernel averages)
root=$(pwd)
out=$root/gpurun_out/r5c2
mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_bench_parity.py tests/test_gpu_unet.py -x -q -m gpu -k "bench_workload or cpu_oracle_at_the_bench or stale_unzeroed or old_flat_layout or config4_768px or trace" -s > $out/tests.log 2>&1
tail -5 $out/tests.log
cd /tmp && export TMPDIR=/tmp
run() {  # name, XCD, ROT
  SIDLSG_ATTN_XCD=$2 SIDLSG_ATTN_ROT=$3 rocprofv3 --kernel-trace --stats -d $out/$1 -o a --output-format csv -- python $root/tools/ab/attn_rot.py $out/ref.pt > $out/$1.log 2>&1
  tail -4 $out/$1.log
  echo "== $1 (XCD=$2 ROT=$3)"; grep -E "attn_(q|dkdv)_kernel" $out/$1/a_kernel_stats.csv | awk -F'","' '{printf "%-60s calls %s avg_us %.1f\n", substr($1,2,58), $2, $4/1000}'
  rm -f $out/$1/a_kernel_trace.csv
}
run base 7 0
run xcd15 15 0
run xcd15_rot6 15 6
run xcd15_rot7 15 7
run xcd7_rot1 7 1
run base2 7 0
