set -x
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/ -x -q -m gpu > gpurun_out/r5_suite.log 2>&1
tail -n 5 gpurun_out/r5_suite.log
for rep in 1 2; do for tw in bf16 fp8 fp8-frozen; do
  timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing --arch sd21-base --teacher-weights $tw 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sd21-base teacher_weights=$tw', d['ms_per_step'], d['value'], d['loss_fake'], d['loss_G'])" >> gpurun_out/r5_fp8.log
done; done
cat gpurun_out/r5_fp8.log
timeout 900 python bench.py > gpurun_out/r5_bench_default.json 2> gpurun_out/r5_bench_default.err
tail -c 3000 gpurun_out/r5_bench_default.json
