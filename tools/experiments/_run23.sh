set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "conv" > gpurun_out/r23_conv.log 2>&1
tail -n 3 gpurun_out/r23_conv.log
for v in 0 1; do echo "== SIDLSG_WGRAD_CONV_FAST=$v"; SIDLSG_WGRAD_CONV_FAST=$v timeout 600 python tools/ab/wgrad_sweep.py conv 2>/dev/null; done > gpurun_out/r23_conv_sweep.log
grep -E "==|weighted|B 16 out 64x64   320->  320|B 16 out 32x32   640->  640|B 16 out  8x 8  1280-> 1280 s1|B 16 out 16x16  1280-> 1280 s1 u0|B 16 out 32x32  1920" gpurun_out/r23_conv_sweep.log
for rep in 1 2 3; do
for cfg in "1" "0"; do
  SIDLSG_WGRAD_CONV_FAST=$cfg timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('conv_fast=$cfg', d['ms_per_step'], d['value'], d['loss_check'])" >> gpurun_out/r23_ab.log
done; done
cat gpurun_out/r23_ab.log
