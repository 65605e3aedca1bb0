#!/bin/bash
# round 5, call 10: decomposition of the forced-exchange overhead (dry run = stream choreography only; no comm timing)
root=$(pwd)
out=$root/gpurun_out/r5c10
mkdir -p $out
b() { python bench.py --no-cpu-baseline --no-kernel-timing --steps 12 --warmup 4 "$@" 2>$out/err.log | grep '^{' | tail -1; }
show() { python3 -c "import sys,json
t=sys.stdin.read().strip()
try:
    d=json.loads(t); print('$1', round(d['ms_per_step'],2), 'ms', d['loss_check'], ('exposed %.2f ms' % d['comm']['comm_exposed_ms']) if d.get('comm') else '')
except Exception as e:
    print('$1 FAILED', t[:200]); print(open('$out/err.log').read()[-2000:])"; }
for i in 1 2; do
  SIDLSG_BENCH_DETAIL=/tmp/d.json b | show "plain"
  SIDLSG_BENCH_DETAIL=/tmp/d.json b --force-exchange | show "forced-exchange"
  SIDLSG_EXCHANGE_DRYRUN=1 SIDLSG_BENCH_DETAIL=/tmp/d.json b --force-exchange | show "forced-exchange DRYRUN"
  SIDLSG_BENCH_COMM_TIMING=0 SIDLSG_BENCH_DETAIL=/tmp/d.json b --force-exchange | show "forced-exchange no-timing"
  SIDLSG_BENCH_COMM_TIMING=0 SIDLSG_EXCHANGE_DRYRUN=1 SIDLSG_BENCH_DETAIL=/tmp/d.json b --force-exchange | show "forced-exchange DRYRUN no-timing"
done
