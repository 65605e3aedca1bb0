#!/bin/bash
# round 5, call 9: where do the +6-10 ms of the forced exchange come from?  hardware queues (7 HIP streams on 4 queues by default)
root=$(pwd)
out=$root/gpurun_out/r5c9
mkdir -p $out
b() { python bench.py --no-cpu-baseline --no-kernel-timing --steps 12 --warmup 4 "$@" 2>$out/err.log | grep '^{' | tail -1; }
show() { python3 -c "import sys,json
t=sys.stdin.read().strip()
try:
    d=json.loads(t); print('$1', round(d['ms_per_step'],2), 'ms', d['loss_check'], 'grouped', d['grouped_frozen_pass'], ('exposed %.2f ms' % d['comm']['comm_exposed_ms']) if d.get('comm') else '')
except Exception as e:
    print('$1 FAILED', t[:200]); print(open('$out/err.log').read()[-2000:])"; }
for i in 1 2; do
  SIDLSG_BENCH_DETAIL=/tmp/d.json b | show "plain"
  SIDLSG_BENCH_DETAIL=/tmp/d.json b --force-exchange | show "forced-exchange"
  GPU_MAX_HW_QUEUES=8 SIDLSG_BENCH_DETAIL=/tmp/d.json b --force-exchange | show "forced-exchange HWQ=8"
  GPU_MAX_HW_QUEUES=8 SIDLSG_BENCH_DETAIL=/tmp/d.json b | show "plain HWQ=8"
  SIDLSG_OVERLAP_G=0 SIDLSG_BENCH_DETAIL=/tmp/d.json b --force-exchange | show "forced-exchange OVERLAP_G=0"
done
