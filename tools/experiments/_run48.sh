# hardware-queue count A/B (six HIP streams in the step; the runtime multiplexes streams onto GPU_MAX_HW_QUEUES queues, default 4)
set -x
mkdir -p gpurun_out; rm -f gpurun_out/r48_ab.log
run() { env $1 timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['value'], d['loss_check'])" >> gpurun_out/r48_ab.log; }
run X=1; run GPU_MAX_HW_QUEUES=8; run GPU_MAX_HW_QUEUES=2; run X=1; run GPU_MAX_HW_QUEUES=8; run GPU_MAX_HW_QUEUES=6
cat gpurun_out/r48_ab.log
