#!/bin/bash
# round 5, call 11: forced-exchange overhead: process group alone? choreography alone (no process group)? which part of the choreography?
root=$(pwd)
out=$root/gpurun_out/r5c11
mkdir -p $out
b() { python bench.py --no-cpu-baseline --no-kernel-timing --steps 12 --warmup 4 "$@" 2>$out/err.log | grep '^{' | tail -1; }
show() { python3 -c "import sys,json
t=sys.stdin.read().strip()
try:
    d=json.loads(t); print('$1', round(d['ms_per_step'],2), 'ms', d['loss_check'])
except Exception as e:
    print('$1 FAILED', t[:200]); print(open('$out/err.log').read()[-2000:])"; }
export SIDLSG_BENCH_COMM_TIMING=0
for i in 1 2; do
  SIDLSG_BENCH_DETAIL=/tmp/d.json b | show "plain"
  SIDLSG_BENCH_PG_ONLY=1 SIDLSG_BENCH_DETAIL=/tmp/d.json b --force-exchange | show "process group only"
  SIDLSG_BENCH_NO_PG=1 SIDLSG_EXCHANGE_DRYRUN=1 SIDLSG_BENCH_DETAIL=/tmp/d.json b --force-exchange | show "choreography only (no process group)"
  SIDLSG_BENCH_NO_PG=1 SIDLSG_EXCHANGE_DRYRUN=1 SIDLSG_OVERLAP_G=0 SIDLSG_BENCH_DETAIL=/tmp/d.json b --force-exchange | show "choreography only, no overlap"
  SIDLSG_BENCH_NO_PG=1 SIDLSG_EXCHANGE_DRYRUN=1 SIDLSG_FINE_SEGMENTS=0 SIDLSG_BENCH_DETAIL=/tmp/d.json b --force-exchange | show "choreography only, 3 segments"
done
