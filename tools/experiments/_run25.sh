set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "conv" > gpurun_out/r25_conv.log 2>&1
tail -n 3 gpurun_out/r25_conv.log
for v in nofast main mainslow; do
  unset SIDLSG_LIB SIDLSG_WGRAD_CONV_FAST
  if [ $v = nofast ]; then export SIDLSG_LIB=$(pwd)/tools/ab/libnofast.so; fi
  if [ $v = mainslow ]; then export SIDLSG_WGRAD_CONV_FAST=0; fi
  echo "== lib $v"; timeout 600 python tools/ab/wgrad_sweep.py conv 2>/dev/null | grep -E "weighted|B 16 out 64x64   320->  320|B 16 out 32x32   640->  640|B 16 out  8x 8  1280-> 1280 s|B 16 out 16x16  1280-> 1280 s1 u0|B 16 out 32x32   320->  320 s2|u1"
done > gpurun_out/r25_sweep.log
cat gpurun_out/r25_sweep.log
unset SIDLSG_LIB SIDLSG_WGRAD_CONV_FAST
for rep in 1 2 3; do
for v in main nofast; do
  if [ $v = nofast ]; then export SIDLSG_LIB=$(pwd)/tools/ab/libnofast.so; else unset SIDLSG_LIB; fi
  timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=$v', d['ms_per_step'], d['value'], d['loss_check'])" >> gpurun_out/r25_ab.log
done; done
cat gpurun_out/r25_ab.log
