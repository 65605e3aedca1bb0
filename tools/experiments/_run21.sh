set -x
mkdir -p gpurun_out
for rep in 1 2; do
for cfg in "1" ""; do
  SIDLSG_EXP_SKIP_TW=$cfg timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('skip_tw=$cfg', d['ms_per_step'], d['value'], d['loss_check'])" >> gpurun_out/r21_ab.log
done; done
cat gpurun_out/r21_ab.log
