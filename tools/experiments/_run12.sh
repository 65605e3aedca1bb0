set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "wgrad or linear" > gpurun_out/r12_wgrad.log 2>&1
tail -n 4 gpurun_out/r12_wgrad.log
for rep in 1 2 3; do
for cfg in "1" "0"; do
  SIDLSG_WGRAD_SQ160=$cfg timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wgrad_sq160=$cfg', d['ms_per_step'], d['value'], d['loss_check'])" >> gpurun_out/r12_ab.log
done; done
cat gpurun_out/r12_ab.log
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/r12_suite.log 2>&1
tail -n 8 gpurun_out/r12_suite.log
