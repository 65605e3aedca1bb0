#!/bin/bash
out=gpurun_out/r5c19; mkdir -p $out
b() { python bench.py --no-cpu-baseline --steps 12 --warmup 4 "$@" 2>$out/err.log | grep '^{' | tail -1; }
show() { python3 -c "import sys,json
t=sys.stdin.read().strip()
try:
    d=json.loads(t); print('$1', round(d['ms_per_step'],2), 'ms', d['loss_check'], d.get('roofline',{}).get('launches_timed'))
except Exception as e:
    print('$1 FAILED', t[:200])"; }
for i in 1 2 3; do
  SIDLSG_BENCH_DETAIL=/tmp/d.json b | show "kernel timing on (default strides)"
  SIDLSG_BENCH_DETAIL=/tmp/d.json b --no-kernel-timing | show "--no-kernel-timing"
  SIDLSG_BENCH_TRACE_STRIDE_MUL=4 SIDLSG_BENCH_DETAIL=/tmp/d.json b | show "strides x4"
done
