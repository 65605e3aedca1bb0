set -x
mkdir -p gpurun_out/r22
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace -d $root/gpurun_out/r22/t -o t --output-format csv -- python $root/bench.py --no-cpu-baseline --no-kernel-timing --steps 12 --warmup 3 > $root/gpurun_out/r22/bench.json 2> $root/gpurun_out/r22/prof.log
cd $root
python tools/exclusive_time.py gpurun_out/r22/t/t_kernel_trace.csv 0.45 > gpurun_out/r22/exclusive.txt 2>&1
python tools/union_gaps.py gpurun_out/r22/t/t_kernel_trace.csv 0.45 > gpurun_out/r22/union_gaps.txt 2>&1
rm -rf gpurun_out/r22/t
cat gpurun_out/r22/exclusive.txt
head -30 gpurun_out/r22/union_gaps.txt
