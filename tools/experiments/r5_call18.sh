#!/bin/bash
out=gpurun_out/r5c18; mkdir -p $out
b() { python bench.py --no-cpu-baseline --no-kernel-timing --steps 12 --warmup 4 "$@" 2>$out/err.log | grep '^{' | tail -1; }
show() { python3 -c "import sys,json
t=sys.stdin.read().strip()
try:
    d=json.loads(t); print('$1', round(d['ms_per_step'],2), 'ms', d['loss_check'])
except Exception as e:
    print('$1 FAILED', t[:200])"; }
for i in 1 2; do
  SIDLSG_BENCH_DETAIL=/tmp/d.json b | show "default"
  SIDLSG_V3_DIRECT_TILES=513 SIDLSG_BENCH_DETAIL=/tmp/d.json b | show "V3_DIRECT_TILES=513 (512-tile launches on 64-row tiles)"
  SIDLSG_V3_DIRECT_TILES=1025 SIDLSG_BENCH_DETAIL=/tmp/d.json b | show "V3_DIRECT_TILES=1025"
  SIDLSG_GEGLU_FUSE_MIN_K=320 SIDLSG_BENCH_DETAIL=/tmp/d.json b | show "GEGLU_FUSE_MIN_K=320"
  SIDLSG_GEMM_AS_MIN_N=960 SIDLSG_BENCH_DETAIL=/tmp/d.json b | show "GEMM_AS_MIN_N=960"
done
