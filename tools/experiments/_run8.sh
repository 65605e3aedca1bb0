set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_grouped.py tests/test_gpu_blocks.py -x -q -k "groupnorm or norm_fork or blocks" > gpurun_out/r8_gn.log 2>&1
tail -n 3 gpurun_out/r8_gn.log
timeout 900 python -m pytest tests/test_gpu_unet.py -x -q -s -k "config5" 2>&1 | grep -E "kappa|agreement|passed|failed" > gpurun_out/r8_config5.log
cat gpurun_out/r8_config5.log
for rep in 1 2 3 4; do
for cfg in "1" "0"; do
  SIDLSG_GN_ONEPASS=$cfg timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gn_onepass=$cfg', d['ms_per_step'], d['value'], d['loss_check'])" >> gpurun_out/r8_ab.log
done; done
cat gpurun_out/r8_ab.log
