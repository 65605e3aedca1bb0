set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_grouped.py -x -q -k "gemm or conv or linear or grouped or g2" > gpurun_out/r26_tests.log 2>&1
tail -n 3 gpurun_out/r26_tests.log
for v in 0 1 0 1; do echo "== SIDLSG_GEMM_SPEC=$v"; SIDLSG_GEMM_SPEC=$v timeout 600 python tools/bench_kernels.py gemm conv 2>/dev/null | grep -E "aggregate|65536x320x320|65536x960x320|16384x640x640|4096x1280x1280|B16 64x64 320->320 s1|B16 32x32 1280->640|B16 16x16 1280->1280 s1 u0"; done > gpurun_out/r26_micro.log
cat gpurun_out/r26_micro.log
for rep in 1 2 3; do
for cfg in "1" "0"; do
  SIDLSG_GEMM_SPEC=$cfg timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gemm_spec=$cfg', d['ms_per_step'], d['value'], d['loss_check'], d['teacher_pass']['ms'])" >> gpurun_out/r26_ab.log
done; done
cat gpurun_out/r26_ab.log
