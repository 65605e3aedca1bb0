set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "conv" > gpurun_out/r13_conv.log 2>&1
tail -n 4 gpurun_out/r13_conv.log
for v in 0 1; do echo "== SIDLSG_WGRAD_CONV160=$v"; SIDLSG_WGRAD_CONV160=$v timeout 600 python tools/ab/wgrad_sweep.py conv 2>/dev/null; done > gpurun_out/r13_conv_sweep.log
cat gpurun_out/r13_conv_sweep.log
for rep in 1 2 3; do
for cfg in "1" "0"; do
  SIDLSG_WGRAD_CONV160=$cfg timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('conv160=$cfg', d['ms_per_step'], d['value'], d['loss_check'])" >> gpurun_out/r13_ab.log
done; done
cat gpurun_out/r13_ab.log
