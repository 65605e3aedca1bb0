set -x
mkdir -p gpurun_out; rm -f gpurun_out/r43_ab.log
run() { env $1 timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['value'], d['loss_check'])" >> gpurun_out/r43_ab.log; }
run SIDLSG_SPLITK_TARGET=512; run SIDLSG_SPLITK_TARGET=256; run SIDLSG_SPLITK_TARGET=384; run SIDLSG_SPLITK_TARGET=768; run SIDLSG_SPLITK_TARGET=1024; run SIDLSG_SPLITK_TARGET=512; run SIDLSG_SPLITK_TARGET=384; run SIDLSG_SPLITK_TARGET=768
cat gpurun_out/r43_ab.log
