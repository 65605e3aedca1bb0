#!/bin/bash
# tools/other_configs.sh OUT.jsonl : bench lines of the configurations BASELINE.json lists beside the headline one (parity-tested in tests/; not bench lines of the driver)
out=${1:-gpurun_out/other_configs.jsonl}
: > $out
run() { python bench.py --steps 8 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | tail -1 >> $out; }
run --kappa 4.5
run --arch sd21-base --kappa 1.5
run --arch sd21-base --kappa 1.5 --teacher-weights fp8
run --arch sd21-base --kappa 1.5 --teacher-weights fp8-frozen
run --arch sd21-base --resolution 768 --kappa 2 --batch-gpu 4
run --teacher-weights fp8-frozen
run --batch-gpu 16
python - $out <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    print(d['metric'], '|', d['config']['workload'][:70], '|', d['config'].get('teacher_weights'), '|', d['value'], 'images/s', d['ms_per_step'], 'ms', d.get('loss_check'), d.get('teacher_pass'))
PY
