set -x
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -s -k "dispatch_trace" > gpurun_out/r3_trace.log 2>&1
timeout 600 python -m pytest tests/test_gpu_unet.py -x -q -s -k "50_iteration and fp32" > gpurun_out/r3_drift32.log 2>&1
for rep in 1 2 3; do
for cfg in "0 0" "1 0" "1 1" "0 1"; do
  set -- $cfg
  SIDLSG_GROUPED_FROZEN=$1 SIDLSG_SEG_OPT=$2 timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('grouped=$1 segopt=$2', d['ms_per_step'], d['value'], d['loss_check'])" >> gpurun_out/r3_ab.log
done; done
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r3_prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r3_bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r3_rocprof.err
cd $GRAFT_REPO_ROOT
ls -la gpurun_out/r3_prof | head; find gpurun_out/r3_prof -name "*stats*" | head
cat gpurun_out/r3_ab.log; tail -3 gpurun_out/r3_trace.log gpurun_out/r3_drift32.log
