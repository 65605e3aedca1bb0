#!/bin/bash
# tools/collect_profiles.sh TAG : the round's evidence, collected on the GPU box into gpurun_out/TAG/ (copy what is to be judged into
# profiles/): default bench line, rocprofv3 kernel stats of the same command, attention SQ counters, dense-GEMM HBM-side traffic.
# PMC passes are separate runs with --kernel-trace only (MI355X_MICROARCH.md, HBM / rocprofv3 section).
tag=${1:-r03}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p $out
python bench.py > $out/bench_default.json 2> $out/bench_default.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/bench_prof -o b --output-format csv -- python $root/bench.py --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/bench_prof.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/gemmF -o f --output-format csv -- python $root/tools/bench_kernels.py gemm > $out/gemmF.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/gemmW -o w --output-format csv -- python $root/tools/bench_kernels.py gemm > $out/gemmW.log 2>&1
cd $root
python tools/pmc_conv_json.py $out/gemmF/f_counter_collection.csv $out/gemmW/w_counter_collection.csv $out/gemm_pmc.json gemm > $out/gemm_pmc.log 2>&1
bash tools/attn_pmc.sh gpurun_out/$tag/attn_pmc > $out/attn_pmc.log 2>&1
cp $out/bench_prof/b_kernel_stats.csv $out/bench_step_kernel_stats.csv 2>/dev/null
rm -rf $out/bench_prof/b_kernel_trace.csv $out/gemmF $out/gemmW $out/attn_pmc/p1 $out/attn_pmc/p2 $out/attn_pmc/st/st_kernel_trace.csv
ls -la $out
