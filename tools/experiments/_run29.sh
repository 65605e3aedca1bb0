set -x
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_unet.py -x -q -k "another_flat_layout" 2>&1 | tail -n 3
rm -f gpurun_out/r29_other.jsonl
run() { timeout 900 python bench.py --no-cpu-baseline --steps 12 --warmup 3 $@ 2>/dev/null | tail -n 1 >> gpurun_out/r29_other.jsonl; }
run --arch sd21-base --resolution 768 --kappa 2 --batch-gpu 4
run --kappa 4.5
run --arch sd21-base --kappa 1.5
run --arch sd21-base --kappa 1.5 --teacher-weights fp8
run --arch sd21-base --kappa 1.5 --teacher-weights fp8-frozen
run --batch-gpu 1
run --batch-gpu 2
python - <<'PY'
import json
for l in open('gpurun_out/r29_other.jsonl'):
    d=json.loads(l); c=d['config']
    print(c['workload'][:90], '|', c.get('teacher_weights'), '| images/s', round(d['value'],2), 'ms', round(d['ms_per_step'],1), 'loss_check', d.get('loss_check'), 'grouped', d.get('grouped_frozen_pass'))
PY
