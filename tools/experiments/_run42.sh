# final evidence of the tree: GPU suite, default bench line, the same command under rocprofv3 --kernel-trace --stats
set -x
root=$(pwd); out=$root/gpurun_out/r42; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q --durations=40 > $out/gpu_suite.txt 2>&1
tail -n 4 $out/gpu_suite.txt
python bench.py > $out/bench_default.json 2> $out/bench_default.err
tail -n 1 $out/bench_default.json | cut -c1-400
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $out/prof -o b --output-format csv -- python $root/bench.py --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/prof.log
cd $root
cp $out/prof/b_kernel_stats.csv $out/bench_step_kernel_stats.csv
python tools/exclusive_time.py $out/prof/b_kernel_trace.csv > $out/step_concurrency.txt 2>&1
rm -rf $out/prof
head -12 $out/bench_step_kernel_stats.csv | cut -c1-160
