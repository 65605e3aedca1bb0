set -x
mkdir -p gpurun_out/r14
root=$(pwd)
python bench.py > gpurun_out/r14/bench_default.json 2> gpurun_out/r14/bench_default.err
tail -c 600 gpurun_out/r14/bench_default.json
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $root/gpurun_out/r14/bench_prof -o b --output-format csv -- python $root/bench.py --no-cpu-baseline > $root/gpurun_out/r14/bench_under_rocprof.json 2> $root/gpurun_out/r14/bench_prof.log
cd $root
cp gpurun_out/r14/bench_prof/b_kernel_stats.csv gpurun_out/r14/bench_step_kernel_stats.csv
python tools/kernel_by_grid.py gpurun_out/r14/bench_prof/b_kernel_trace.csv > gpurun_out/r14/kernel_by_grid.txt 2>&1
rm -f gpurun_out/r14/bench_prof/b_kernel_trace.csv
bash tools/collect_traffic.sh gpurun_out/r14/traffic > gpurun_out/r14/traffic.log 2>&1
tail -n 5 gpurun_out/r14/traffic.log
ls -la gpurun_out/r14
