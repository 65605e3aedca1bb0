#!/bin/bash
# tools/fp8_ab.sh : SD2.1-base kappa 1.5 with --teacher-weights fp8-frozen, before (SIDLSG_P8_GEGLU=0 SIDLSG_FF_G2=0 SIDLSG_GEGLU_FUSE_MIN_K=640) and after the GEGLU changes of round 6, next to bf16, alternating
out=gpurun_out/fp8ab; mkdir -p $out
for i in 1 2; do
  SIDLSG_P8_GEGLU=0 SIDLSG_FF_G2=0 SIDLSG_GEGLU_FUSE_MIN_K=640 python bench.py --arch sd21-base --kappa 1.5 --teacher-weights fp8-frozen --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 > $out/base_$i.json
  python bench.py --arch sd21-base --kappa 1.5 --teacher-weights fp8-frozen --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 > $out/new_$i.json
  python bench.py --arch sd21-base --kappa 1.5 --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 > $out/bf16_$i.json
done
for f in $out/*.json; do python - $f <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d['value'], 'images/s', d['ms_per_step'], 'ms')
PY
done | sort
