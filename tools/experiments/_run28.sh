set -x
mkdir -p gpurun_out
run() { env $1 timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['value'], d['loss_check'])" >> gpurun_out/r28_ab.log; }
run "X=1"
run "SIDLSG_SPLITK_MIN_NK=1000"
run "SIDLSG_SPLITK_MIN_NK=64"
run "SIDLSG_GEMM_MIN_TILES=256"
run "X=1"
run "SIDLSG_SPLITK_MIN_KT=16"
run "SIDLSG_WGRAD_SLOTS=256"
run "SIDLSG_WGRAD_SLOTS=1024"
run "X=1"
cat gpurun_out/r28_ab.log
