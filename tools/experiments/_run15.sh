set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "wgrad or linear or conv_autograd or layer" > gpurun_out/r15_tests.log 2>&1
tail -n 4 gpurun_out/r15_tests.log
for v in 1 0; do echo "== REDUCE_GROUPS=$v"; SIDLSG_WGRAD_REDUCE_GROUPS=$v timeout 600 python tools/ab/wgrad_sweep.py 2>/dev/null | grep -E "sum|N   320 K   320|N   640 K   640|N  1280 K  1280"; done > gpurun_out/r15_reduce_micro.log
cat gpurun_out/r15_reduce_micro.log
for cfg in "" "SIDLSG_LN_FWD_WAVES=8192 SIDLSG_LN_BWD_BLOCKS=2048" "SIDLSG_LN_FWD_WAVES=16384 SIDLSG_LN_BWD_BLOCKS=4096" "SIDLSG_LN_FWD_R8=1" "SIDLSG_LN_FWD_R8=1 SIDLSG_LN_FWD_WAVES=8192"; do echo "== $cfg"; env $cfg timeout 300 python tools/bench_kernels.py norm 2>/dev/null | grep "LN "; done > gpurun_out/r15_ln_micro.log
cat gpurun_out/r15_ln_micro.log
for rep in 1 2 3; do
for cfg in "0" "1"; do
  SIDLSG_WGRAD_REDUCE_GROUPS=$cfg timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('reduce_groups=$cfg', d['ms_per_step'], d['value'], d['loss_check'])" >> gpurun_out/r15_ab.log
done; done
cat gpurun_out/r15_ab.log
