#!/bin/bash
# round 5, call 7: in-launch split-K combine (test + step A/B); forced exchange (fine / coarse segments, grouped or not)
root=$(pwd)
out=$root/gpurun_out/r5c7
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "split_k_in_launch or test_gemm or conv" > $out/tests.log 2>&1
tail -3 $out/tests.log
b() { python bench.py --no-cpu-baseline --no-kernel-timing --steps 12 --warmup 4 "$@" 2>$out/err.log | tail -1; }
show() { python3 -c "import sys,json
t=sys.stdin.read().strip()
try:
    d=json.loads(t); print('$1', round(d['ms_per_step'],2), 'ms', d['loss_check'], 'grouped', d['grouped_frozen_pass'], json.dumps(d.get('comm')) if d.get('comm') else '')
except Exception as e:
    print('$1 FAILED', t[:200]); print(open('$out/err.log').read()[-1500:])"; }
for i in 1 2 3; do
  for t in 0 2; do SIDLSG_SPLITK_TICKET=$t SIDLSG_BENCH_DETAIL=/tmp/d.json b | show "step TICKET=$t"; done
done
for i in 1 2; do
  for f in 1 0; do SIDLSG_FINE_SEGMENTS=$f SIDLSG_BENCH_DETAIL=/tmp/d.json b --force-exchange | show "forced-exchange FINE=$f"; done
  SIDLSG_GROUPED_FROZEN=1 SIDLSG_BENCH_DETAIL=/tmp/d.json b --force-exchange | show "forced-exchange FINE=1 GROUPED=1"
done
