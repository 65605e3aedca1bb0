#!/bin/bash
# round 5: bench line + rocprofv3 kernel stats of the same command on the final tree (sparser kernel-time sampling)
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
python - <<'PY'
import json
a=json.load(open('gpurun_out/r05/bench_detail.json')); b=json.load(open('gpurun_out/r05/bench_detail_under_rocprof.json'))
for k in ('gemm','conv','attn','attn_bwd','wgrad','conv_wgrad','gn','gn_bwd','ln','ln_bwd'):
    ra, rb = a['roofline_'+k], b['roofline_'+k]
    print(f"{k:10s} plain: frac {ra['frac']:.3f} avg {ra['avg_launch_ms']*1e3:7.1f} us | under rocprofv3: frac {rb['frac']:.3f} avg {rb['avg_launch_ms']*1e3:7.1f} us")
PY
