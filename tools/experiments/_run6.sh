set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_grouped.py tests/test_gpu_blocks.py -x -q -k "groupnorm or norm_fork or grouped_groupnorm or blocks" > gpurun_out/r6_gn.log 2>&1
tail -n 4 gpurun_out/r6_gn.log
for rep in 1 2 3; do
for cfg in "1 0" "0 0" "1 1"; do
  set -- $cfg
  SIDLSG_GN_ONEPASS=$1 SIDLSG_WGRAD_TN160_DENSE=$2 timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gn_onepass=$1 tn160dense=$2', d['ms_per_step'], d['value'], d['loss_check'])" >> gpurun_out/r6_ab.log
done; done
cat gpurun_out/r6_ab.log
timeout 2400 python -m pytest tests/ -x -q -m gpu --durations=25 > gpurun_out/r6_suite.log 2>&1
tail -n 40 gpurun_out/r6_suite.log
