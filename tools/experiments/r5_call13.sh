#!/bin/bash
# round 5, call 13: with the grouped frozen pass as the default, are the other stream tricks still wins?
root=$(pwd)
out=$root/gpurun_out/r5c13
mkdir -p $out
b() { python bench.py --no-cpu-baseline --no-kernel-timing --steps 12 --warmup 4 "$@" 2>$out/err.log | grep '^{' | tail -1; }
show() { python3 -c "import sys,json
t=sys.stdin.read().strip()
try:
    d=json.loads(t); print('$1', round(d['ms_per_step'],2), 'ms', d['loss_check'])
except Exception as e:
    print('$1 FAILED', t[:200]); print(open('$out/err.log').read()[-2000:])"; }
for i in 1 2; do
  SIDLSG_BENCH_DETAIL=/tmp/d.json b | show "default (grouped)"
  SIDLSG_WGRAD_STREAM=0 SIDLSG_BENCH_DETAIL=/tmp/d.json b | show "WGRAD_STREAM=0"
  SIDLSG_EARLY_GFWD=0 SIDLSG_BENCH_DETAIL=/tmp/d.json b | show "EARLY_GFWD=0"
  SIDLSG_BENCH_PREFETCH=0 SIDLSG_BENCH_DETAIL=/tmp/d.json b | show "BENCH_PREFETCH=0"
  SIDLSG_SEG_OPT=1 SIDLSG_BENCH_DETAIL=/tmp/d.json b | show "SEG_OPT=1"
  SIDLSG_BENCH_DETAIL=/tmp/d.json b --graph | show "--graph"
done
