# per-group one-pass GroupNorm (gn_group_*): tests, kernel micro-benchmark with / without, step A/B
set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_grouped.py tests/test_gpu_blocks.py -x -q -k "groupnorm or group_norm or blocks or gn" > gpurun_out/r44_tests.log 2>&1
tail -n 8 gpurun_out/r44_tests.log
rm -f gpurun_out/r44_micro.log
for v in 0 3 0 3; do echo "== SIDLSG_GN_GROUP=$v" >> gpurun_out/r44_micro.log; SIDLSG_GN_GROUP=$v timeout 300 python tools/bench_kernels.py norm 2>/dev/null | grep "GN " >> gpurun_out/r44_micro.log; done
cat gpurun_out/r44_micro.log
run() { env $1 timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['value'], d['loss_check'])" >> gpurun_out/r44_ab.log; }
rm -f gpurun_out/r44_ab.log
run SIDLSG_GN_GROUP=3; run SIDLSG_GN_GROUP=0; run SIDLSG_GN_GROUP=3; run SIDLSG_GN_GROUP=0; run SIDLSG_GN_GROUP=1; run SIDLSG_GN_GROUP=2
cat gpurun_out/r44_ab.log
