#!/bin/bash
# tools/prio_ab.sh OUT: step time with the weight-gradient streams at the lowest HIP stream priority (SIDLSG_WGRAD_PRIO=low) against the default, alternating
out=${1:-gpurun_out/prio}; mkdir -p $out
for i in 1 2 3; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > $out/default_$i.json 2> $out/default_$i.err
  SIDLSG_WGRAD_PRIO=low python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > $out/low_$i.json 2> $out/low_$i.err
done
for f in $out/*.json; do python - $f <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d['value'], 'images/s', d['ms_per_step'], 'ms', d.get('loss_check'))
PY
done | tee $out/summary.txt
