# bench lines of the other BASELINE configurations on the final tree (12 timed iterations each)
set -x
mkdir -p gpurun_out; O=gpurun_out/r47_other_configs.jsonl; rm -f $O
B="python bench.py --no-cpu-baseline --no-kernel-timing --steps 12 --warmup 3"
timeout 600 $B --arch sd21-base --resolution 768 --kappa 2 --batch-gpu 4 2>/dev/null | tail -n 1 >> $O
timeout 600 $B --kappa 4.5 2>/dev/null | tail -n 1 >> $O
timeout 600 $B --arch sd21-base --kappa 1.5 2>/dev/null | tail -n 1 >> $O
timeout 600 $B --arch sd21-base --kappa 1.5 --teacher-weights fp8 2>/dev/null | tail -n 1 >> $O
timeout 600 $B --arch sd21-base --kappa 1.5 --teacher-weights fp8-frozen 2>/dev/null | tail -n 1 >> $O
timeout 600 $B --batch-gpu 1 2>/dev/null | tail -n 1 >> $O
timeout 600 $B --batch-gpu 2 2>/dev/null | tail -n 1 >> $O
python - <<'P'
import json
for l in open('gpurun_out/r47_other_configs.jsonl'):
    d = json.loads(l); print(d['config']['workload'][:70], d.get('config', {}).get('teacher_weights'), round(d['value'], 2), round(d['ms_per_step'], 1), d.get('loss_check'))
P
