#!/bin/bash
# tools/p8_margin_ab.sh OUT : in-step A/B of the p8 rule's admission margin (SIDLSG_P8_MARGIN, % predicted gain a call must show; default 3)
out=${1:-gpurun_out/p8margin}; mkdir -p $out
for i in 1 2; do for m in 3 -5 10 -15; do
  SIDLSG_P8_MARGIN=$m python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 > $out/m${m}_$i.json
done; done
for f in $out/*.json; do python - $f <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d['value'], 'images/s', d['ms_per_step'], 'ms', d.get('loss_check'))
PY
done | sort | tee $out/summary.txt
