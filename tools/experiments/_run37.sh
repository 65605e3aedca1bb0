# s_setprio A/B (attention clusters, gemm_v3 / wgrad clusters): kernel micro-benchmarks per variant library, then the step
set -x
mkdir -p gpurun_out
L=gpurun_out/r37_micro.log; rm -f $L gpurun_out/r37_ab.log
SIDLSG_LIB=$(pwd)/tools/ab/libprio_a15.so timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "attn or attention" > gpurun_out/r37_attn_tests.log 2>&1
tail -n 3 gpurun_out/r37_attn_tests.log
for rep in 1 2; do
for v in base prio_a3 prio_a4 prio_a12 prio_a15; do
  lib=$(pwd)/tools/ab/lib$v.so; [ $v = base ] && lib=$(pwd)/sid_lsg_amd/libsidlsg_hip.so
  echo "== $v" >> $L
  SIDLSG_LIB=$lib timeout 300 python tools/bench_kernels.py attn 2>/dev/null | grep -E "self" >> $L
done; done
for rep in 1 2; do
for v in base prio_g1 prio_g1u; do
  lib=$(pwd)/tools/ab/lib$v.so; [ $v = base ] && lib=$(pwd)/sid_lsg_amd/libsidlsg_hip.so
  echo "== $v" >> $L
  SIDLSG_LIB=$lib timeout 300 python tools/bench_kernels.py conv gemm 2>/dev/null | grep -E "aggregate|64x64 320->320 s1|32x32 1280->640|65536x320x320|65536x960x320|16384x640x2560" >> $L
done
for v in base prio_g2; do
  lib=$(pwd)/tools/ab/lib$v.so; [ $v = base ] && lib=$(pwd)/sid_lsg_amd/libsidlsg_hip.so
  echo "== $v (wgrad)" >> $L
  SIDLSG_LIB=$lib timeout 300 python tools/bench_kernels.py wgrad 2>/dev/null | grep -E "dense|conv " >> $L
done; done
cat $L
run() { lib=$(pwd)/tools/ab/lib$1.so; [ $1 = base ] && lib=$(pwd)/sid_lsg_amd/libsidlsg_hip.so; SIDLSG_LIB=$lib timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['value'], d['loss_check'])" >> gpurun_out/r37_ab.log; }
run base; run prio_a15; run prio_g1; run base; run prio_a15; run prio_g1u; run prio_g2
cat gpurun_out/r37_ab.log
