set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_grouped.py -x -q -k "groupnorm or group_norm" > gpurun_out/r45_tests.log 2>&1
tail -n 4 gpurun_out/r45_tests.log
run() { env $1 timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['value'], d['loss_check'])" >> gpurun_out/r45_ab.log; }
rm -f gpurun_out/r45_ab.log
run SIDLSG_GN_GROUP=2; run SIDLSG_GN_GROUP=0; run SIDLSG_GN_GROUP=2; run SIDLSG_GN_GROUP=0; run SIDLSG_GN_GROUP=2; run SIDLSG_GN_GROUP=0
cat gpurun_out/r45_ab.log
