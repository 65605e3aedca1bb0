set -x
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_grouped.py -x -q -s > gpurun_out/r2_grouped.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_unet.py -x -q -s -k "50_iteration or per_step_loss or current_ema" > gpurun_out/r2_drift.log 2>&1
SIDLSG_GROUPED_FROZEN=0 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r2_bench_twostream.json 2> gpurun_out/r2_bench_twostream.err
SIDLSG_GROUPED_FROZEN=1 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r2_bench_grouped.json 2> gpurun_out/r2_bench_grouped.err
SIDLSG_GROUPED_FROZEN=0 timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing > gpurun_out/r2_bench_twostream_b.json 2>> gpurun_out/r2_bench_twostream.err
SIDLSG_GROUPED_FROZEN=1 timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing > gpurun_out/r2_bench_grouped_b.json 2>> gpurun_out/r2_bench_grouped.err
for f in gpurun_out/r2_*.log; do echo == $f; tail -n 4 $f; done
