set -x
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_unet.py -x -q -k "pipelined or unzeroed or side_streams or graphed or foreign or loop_end_to_end or iteration_matches_oracle or product_loop_matches" > gpurun_out/r32_tests.log 2>&1
tail -n 15 gpurun_out/r32_tests.log
run() { env $1 timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['value'], d['loss_check'], d['loss_fake'], d['loss_G'])" >> gpurun_out/r32_ab.log; }
rm -f gpurun_out/r32_ab.log
run "X=1"
run "SIDLSG_PIPE_OPT=0"
run "SIDLSG_PIPE_OPT_G=0"
run "X=1"
run "SIDLSG_PIPE_OPT=0"
run "SIDLSG_PIPE_OPT_G=0"
cat gpurun_out/r32_ab.log
