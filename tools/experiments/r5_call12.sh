#!/bin/bash
# round 5, call 12: a live RCCL process group alone costs ~3 ms per iteration (call 11): which part of it?
root=$(pwd)
out=$root/gpurun_out/r5c12
mkdir -p $out
b() { python bench.py --no-cpu-baseline --no-kernel-timing --steps 12 --warmup 4 "$@" 2>$out/err.log | grep '^{' | tail -1; }
show() { python3 -c "import sys,json
t=sys.stdin.read().strip()
try:
    d=json.loads(t); print('$1', round(d['ms_per_step'],2), 'ms', d['loss_check'])
except Exception as e:
    print('$1 FAILED', t[:200]); print(open('$out/err.log').read()[-2000:])"; }
export SIDLSG_BENCH_COMM_TIMING=0 SIDLSG_BENCH_PG_ONLY=1
for i in 1 2; do
  SIDLSG_BENCH_PG_ONLY=0 SIDLSG_BENCH_DETAIL=/tmp/d.json b | show "plain"
  SIDLSG_BENCH_DETAIL=/tmp/d.json b --force-exchange | show "pg only"
  TORCH_NCCL_ENABLE_MONITORING=0 SIDLSG_BENCH_DETAIL=/tmp/d.json b --force-exchange | show "pg only, monitoring off"
  TORCH_NCCL_ASYNC_ERROR_HANDLING=0 TORCH_NCCL_ENABLE_MONITORING=0 SIDLSG_BENCH_DETAIL=/tmp/d.json b --force-exchange | show "pg only, async error handling + monitoring off"
  TORCH_NCCL_AVOID_RECORD_STREAMS=1 SIDLSG_BENCH_DETAIL=/tmp/d.json b --force-exchange | show "pg only, avoid record streams"
  NCCL_LAUNCH_MODE=PARALLEL RCCL_MSCCL_ENABLE=0 SIDLSG_BENCH_DETAIL=/tmp/d.json b --force-exchange | show "pg only, msccl off"
done
