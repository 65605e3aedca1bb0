set -x
mkdir -p gpurun_out
run() { env $1 timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['value'], d['loss_check'])" >> gpurun_out/r34_ab.log; }
rm -f gpurun_out/r34_ab.log
run "X=1"
run "SIDLSG_EXP_SKIP_GEGLU320=1"
run "X=1"
run "SIDLSG_EXP_SKIP_GEGLU320=1"
cat gpurun_out/r34_ab.log
