#!/bin/bash
# tools/ffg2_ab.sh OUT : FeedForward of the grouped frozen pass, in the step, alternating:
#   base   = SIDLSG_FF_G2=0 SIDLSG_P8_GEGLU=0 : unfused grouped chain (grouped GEMMs by the p8 rule + stand-alone GEGLU kernels), single-set fusions on v3   (the round-5 / early round-6 state)
#   g2v3   = SIDLSG_P8_GEGLU=0                : grouped GEGLU fusions, all fusions on the v3 kernel
#   g2p8   = default                           : grouped GEGLU fusions, all fusions on the 256 x 320 kernel where the cost model says so
out=${1:-gpurun_out/ffg2}; mkdir -p $out
for i in 1 2 3; do
  SIDLSG_FF_G2=0 SIDLSG_P8_GEGLU=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 > $out/base_$i.json
  SIDLSG_P8_GEGLU=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 > $out/g2v3_$i.json
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 > $out/g2p8_$i.json
done
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $out/g2p8_full_line.json
for f in $out/*.json; do python - $f <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d['value'], 'images/s', d['ms_per_step'], 'ms', d.get('loss_check'), d.get('teacher_pass'), d.get('frozen_pair_pass'))
PY
done | sort | tee $out/summary.txt
