#!/bin/bash
# round 5, call 6: whole GPU suite on the current tree; grouped frozen pass at batch_gpu 8 (A/B); exposed exchange time with the
# collectives forced on one rank (RCCL world 1), fine vs coarse exchange segments
root=$(pwd)
out=$root/gpurun_out/r5c6
mkdir -p $out
b() { python bench.py --no-cpu-baseline --no-kernel-timing --steps 12 --warmup 4 "$@" 2>/dev/null | tail -1; }
for i in 1 2; do
  for g in auto 1; do
    SIDLSG_GROUPED_FROZEN=$g SIDLSG_BENCH_DETAIL=/tmp/d.json b | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('step GROUPED=$g', round(d['ms_per_step'],2), 'ms', d['loss_check'], 'grouped', d['grouped_frozen_pass'])"
  done
done
for i in 1 2; do
  for f in 1 0; do
    SIDLSG_FINE_SEGMENTS=$f SIDLSG_BENCH_DETAIL=/tmp/d.json b --force-exchange | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('step forced-exchange FINE_SEGMENTS=$f', round(d['ms_per_step'],2), 'ms', d['loss_check'], json.dumps(d.get('comm')))"
  done
done
timeout 2400 python -m pytest tests -x -q -m gpu > $out/gpu_suite.txt 2>&1
tail -5 $out/gpu_suite.txt
