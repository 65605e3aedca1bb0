set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "wgrad_dense or linear_autograd or linear_geglu" > gpurun_out/r11_wgrad.log 2>&1
tail -n 6 gpurun_out/r11_wgrad.log
for v in 0 1; do echo "== SIDLSG_WGRAD_SQ160=$v"; SIDLSG_WGRAD_SQ160=$v python tools/bench_kernels.py wgrad 2>/dev/null | grep dense; done > gpurun_out/r11_wgrad_micro.log
cat gpurun_out/r11_wgrad_micro.log
for rep in 1 2 3; do
for cfg in "1" "0"; do
  SIDLSG_WGRAD_SQ160=$cfg timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wgrad_sq160=$cfg', d['ms_per_step'], d['value'], d['loss_check'])" >> gpurun_out/r11_ab.log
done; done
cat gpurun_out/r11_ab.log
