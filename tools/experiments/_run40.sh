# does the extra epilogue code of gemm_v3_kernel<0> cost the other GEMMs anything?  (library with / without it, same session)
set -x
mkdir -p gpurun_out; L=gpurun_out/r40_micro.log; rm -f $L gpurun_out/r40_ab.log
for rep in 1 2 3; do
for v in base noepi; do
  lib=$(pwd)/tools/ab/lib$v.so; [ $v = base ] && lib=$(pwd)/sid_lsg_amd/libsidlsg_hip.so
  echo "== $v" >> $L
  SIDLSG_LIB=$lib timeout 300 python tools/bench_kernels.py conv gemm 2>/dev/null | grep -E "aggregate|65536x320x320|65536x960x320|16384x640x2560|65536x320x1280" >> $L
done; done
cat $L
run() { lib=$(pwd)/tools/ab/lib$1.so; [ $1 = base ] && lib=$(pwd)/sid_lsg_amd/libsidlsg_hip.so; SIDLSG_LIB=$lib timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['value'], d['loss_check'])" >> gpurun_out/r40_ab.log; }
run base; run noepi; run base; run noepi
cat gpurun_out/r40_ab.log
