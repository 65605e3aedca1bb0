#!/bin/bash
# round 5: evidence on the final tree (grouped weight gradients in): bench line, rocprofv3 stats of the same command, whole GPU suite
root=$(pwd)
out=$root/gpurun_out/r05
mkdir -p $out
python bench.py > $out/bench_default.json 2> $out/bench_default.err
cp bench_detail.json $out/bench_detail.json
tail -c 2200 $out/bench_default.json
cd /tmp && export TMPDIR=/tmp
SIDLSG_BENCH_DETAIL=$out/bench_detail_under_rocprof.json rocprofv3 --kernel-trace --stats -d /tmp/bench_prof -o b --output-format csv -- python $root/bench.py --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/bench_prof.log
cp /tmp/bench_prof/b_kernel_stats.csv $out/bench_step_kernel_stats.csv
cd $root
timeout 2700 python -m pytest tests -x -q -m gpu > $out/gpu_suite.txt 2>&1
tail -3 $out/gpu_suite.txt
