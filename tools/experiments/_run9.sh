set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -s -k "linear_geglu" > gpurun_out/r9_geglu.log 2>&1
tail -n 15 gpurun_out/r9_geglu.log
timeout 900 python -m pytest tests/test_gpu_unet.py tests/test_gpu_grouped.py -x -q -k "forward_backward or sid_iteration_matches or forward_pair or side_streams or graphed" > gpurun_out/r9_unet.log 2>&1
tail -n 5 gpurun_out/r9_unet.log
for rep in 1 2 3; do
for cfg in "1" "0"; do
  SIDLSG_GEMM_GEGLU=$cfg timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused_geglu=$cfg', d['ms_per_step'], d['value'], d['loss_check'])" >> gpurun_out/r9_ab.log
done; done
cat gpurun_out/r9_ab.log
