#!/bin/bash
# round 5, call 21: the wide layers' weight gradients grouped too (160 x 160 tiles): tests, step A/B
root=$(pwd)
out=$root/gpurun_out/r5c21
mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "wgrad or feed_forward or geglu" > $out/tests_ops.log 2>&1
tail -3 $out/tests_ops.log
timeout 1500 python -m pytest tests/test_gpu_unet.py tests/test_gpu_dist.py -x -q -m gpu -k "forward_backward or iteration_matches or segments or graphed or unzeroed or side_streams or two_rank or stale or foreign or rccl or ddp" > $out/tests_unet.log 2>&1
tail -3 $out/tests_unet.log
b() { python bench.py --no-cpu-baseline --no-kernel-timing --steps 12 --warmup 4 "$@" 2>$out/err.log | grep '^{' | tail -1; }
show() { python3 -c "import sys,json
t=sys.stdin.read().strip()
try:
    d=json.loads(t); print('$1', round(d['ms_per_step'],2), 'ms', d['loss_check'])
except Exception as e:
    print('$1 FAILED', t[:200]); print(open('$out/err.log').read()[-2000:])"; }
for i in 1 2 3; do
  SIDLSG_WGRAD_GROUP=0 SIDLSG_BENCH_DETAIL=/tmp/d.json b | show "WGRAD_GROUP=0"
  SIDLSG_WGRAD_GROUP160=0 SIDLSG_BENCH_DETAIL=/tmp/d.json b | show "WGRAD_GROUP=1, wide layers alone"
  SIDLSG_BENCH_DETAIL=/tmp/d.json b | show "WGRAD_GROUP=1, wide layers grouped"
done
