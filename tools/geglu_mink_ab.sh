#!/bin/bash
# tools/geglu_mink_ab.sh OUT : in the step: fused FF-in + GEGLU (256x320 kernel) at K = 320 with h kept (default since round 6) against A-stationary GEMM + stand-alone GEGLU (SIDLSG_GEGLU_FUSE_MIN_K=640),
# and both against the tree without the p8 GEGLU kernels and grouped fusions (SIDLSG_P8_GEGLU=0 SIDLSG_FF_G2=0 SIDLSG_GEGLU_FUSE_MIN_K=640)
out=${1:-gpurun_out/mink}; mkdir -p $out
for i in 1 2 3; do
  SIDLSG_P8_GEGLU=0 SIDLSG_FF_G2=0 SIDLSG_GEGLU_FUSE_MIN_K=640 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 > $out/base_$i.json
  SIDLSG_GEGLU_FUSE_MIN_K=640 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 > $out/k640_$i.json
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 > $out/k320_$i.json
done
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $out/k320_full_line.json
for f in $out/*.json; do python - $f <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d['value'], 'images/s', d['ms_per_step'], 'ms', d.get('loss_check'), d.get('teacher_pass'), d.get('frozen_pair_pass'))
PY
done | sort | tee $out/summary.txt
