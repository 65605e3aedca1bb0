#!/bin/bash
# round 5, call 8: forced exchange on RCCL world 1 (fine / coarse segments; grouped or not), grouped frozen pass at batch_gpu 8 (3 alternations)
root=$(pwd)
out=$root/gpurun_out/r5c8
mkdir -p $out
b() { python -X faulthandler bench.py --no-cpu-baseline --no-kernel-timing --steps 12 --warmup 4 "$@" 2>$out/err.log | grep '^{' | tail -1; echo "rc=${PIPESTATUS[0]}" >> $out/err.log; }
show() { python3 -c "import sys,json
t=sys.stdin.read().strip()
try:
    d=json.loads(t); print('$1', round(d['ms_per_step'],2), 'ms', d['loss_check'], 'grouped', d['grouped_frozen_pass'], json.dumps(d.get('comm')) if d.get('comm') else '')
except Exception as e:
    print('$1 FAILED', t[:200]); print(open('$out/err.log').read()[-3000:])"; }
SIDLSG_BENCH_DETAIL=/tmp/d.json b --force-exchange | show "forced-exchange FINE=1"
SIDLSG_FINE_SEGMENTS=0 SIDLSG_BENCH_DETAIL=/tmp/d.json b --force-exchange | show "forced-exchange FINE=0"
SIDLSG_GROUPED_FROZEN=1 SIDLSG_BENCH_DETAIL=/tmp/d.json b --force-exchange | show "forced-exchange FINE=1 GROUPED=1"
SIDLSG_BENCH_DETAIL=/tmp/d.json b --force-exchange | show "forced-exchange FINE=1"
SIDLSG_FINE_SEGMENTS=0 SIDLSG_BENCH_DETAIL=/tmp/d.json b --force-exchange | show "forced-exchange FINE=0"
SIDLSG_GROUPED_FROZEN=1 SIDLSG_BENCH_DETAIL=/tmp/d.json b --force-exchange | show "forced-exchange FINE=1 GROUPED=1"
for i in 1 2 3; do
  for g in auto 1; do SIDLSG_GROUPED_FROZEN=$g SIDLSG_BENCH_DETAIL=/tmp/d.json b | show "step GROUPED=$g"; done
done
