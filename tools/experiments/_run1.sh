set -x
mkdir -p gpurun_out
python tools/make_bench_loss_reference.py > gpurun_out/r1_lossref.log 2>&1
cp tests/golden/bench_loss_reference.json gpurun_out/ 2>/dev/null
timeout 1500 python -m pytest tests/test_gpu_bench_parity.py -x -q -s > gpurun_out/r1_benchparity.log 2>&1
timeout 2400 python -m pytest tests/test_gpu_unet.py -x -q -s -k "batch2 or config5 or 50_iteration or current_ema" > gpurun_out/r1_newunet.log 2>&1
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_dist.py -x -q -s -k "as_accurate or timing_of_the_full or two_ranks" > gpurun_out/r1_ops_dist.log 2>&1
timeout 900 python tools/lossg_ablation.py > gpurun_out/r1_lossg_ablation.md 2> gpurun_out/r1_lossg_ablation.err
timeout 900 python bench.py > gpurun_out/r1_bench.json 2> gpurun_out/r1_bench.err
tail -3 gpurun_out/r1_*.log
