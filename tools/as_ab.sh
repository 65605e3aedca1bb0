#!/bin/bash
# tools/as_ab.sh OUT : the A-stationary K = 320 kernel (FF-in at the 64x64 stage) against the rule's choice (p8 / v3) for the same calls, in the step
out=${1:-gpurun_out/as_ab}; mkdir -p $out
for i in 1 2; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 > $out/default_$i.json
  SIDLSG_GEMM_AS=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 > $out/noas_$i.json
done
for f in $out/*.json; do python - $f <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d['value'], 'images/s', d['ms_per_step'], 'ms', d.get('loss_check'))
PY
done | sort | tee $out/summary.txt
