set -x
mkdir -p gpurun_out
for rep in 1 2; do
for v in nofast main; do
  if [ $v = nofast ]; then export SIDLSG_LIB=$(pwd)/tools/ab/libnofast.so; else unset SIDLSG_LIB; fi
  echo "== lib $v"; timeout 600 python tools/ab/wgrad_sweep.py conv 2>/dev/null | grep -E "weighted|B 16 out 64x64   320->  320|B 16 out 32x32   640->  640|B 16 out  8x 8  1280-> 1280 s1|B 16 out 16x16  1280-> 1280 s1 u0"
done; done > gpurun_out/r24_sweep.log
cat gpurun_out/r24_sweep.log
for rep in 1 2 3; do
for v in main nofast; do
  if [ $v = nofast ]; then export SIDLSG_LIB=$(pwd)/tools/ab/libnofast.so; else unset SIDLSG_LIB; fi
  timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=$v', d['ms_per_step'], d['value'], d['loss_check'])" >> gpurun_out/r24_ab.log
done; done
cat gpurun_out/r24_ab.log
