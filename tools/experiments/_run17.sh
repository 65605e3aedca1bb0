set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "wgrad or linear or conv_autograd" > gpurun_out/r17_ops.log 2>&1
tail -n 4 gpurun_out/r17_ops.log
timeout 1500 python -m pytest tests/test_gpu_unet.py -x -q -k "not full_size and not 768px and not 50_iteration" > gpurun_out/r17_unet.log 2>&1
tail -n 12 gpurun_out/r17_unet.log
for rep in 1; do
for cfg in "1" "0"; do
  SIDLSG_GRAD_ASSIGN=$cfg timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('grad_assign=$cfg', d['ms_per_step'], d['value'], d['loss_check'], d['loss_fake'], d['loss_G'])" >> gpurun_out/r17_ab.log
done; done
cat gpurun_out/r17_ab.log
