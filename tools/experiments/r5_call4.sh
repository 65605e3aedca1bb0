#!/bin/bash
# round 5, call 4: deferred dgamma / dbeta reductions (tests + step A/B), attention defaults XCD=15 ROT=7
root=$(pwd)
out=$root/gpurun_out/r5c4
mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_dist.py tests/test_gpu_blocks.py -x -q -m gpu > $out/tests_ops.log 2>&1
tail -3 $out/tests_ops.log
timeout 1500 python -m pytest tests/test_gpu_unet.py -x -q -m gpu -k "forward_backward or segments or iteration_matches or side_streams or graphed or unzeroed or two_rank or foreign" > $out/tests_unet.log 2>&1
tail -3 $out/tests_unet.log
for i in 1 2 3; do
  for d in 0 1; do
    SIDLSG_DEFER_REDUCE=$d SIDLSG_BENCH_DETAIL=/tmp/d.json python bench.py --no-cpu-baseline --no-kernel-timing --steps 12 --warmup 4 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('step DEFER=$d', round(d['ms_per_step'],2), 'ms', d['loss_check'])"
  done
done
