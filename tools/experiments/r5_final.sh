#!/bin/bash
# round 5: the round's evidence on the final tree -> gpurun_out/r05/ (copied into profiles/ afterwards)
root=$(pwd)
out=$root/gpurun_out/r05
mkdir -p $out
python bench.py > $out/bench_default.json 2> $out/bench_default.err
cp bench_detail.json $out/bench_detail.json
tail -c 2500 $out/bench_default.json
cd /tmp && export TMPDIR=/tmp
SIDLSG_BENCH_DETAIL=$out/bench_detail_under_rocprof.json rocprofv3 --kernel-trace --stats -d /tmp/bench_prof -o b --output-format csv -- python $root/bench.py --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/bench_prof.log
cp /tmp/bench_prof/b_kernel_stats.csv $out/bench_step_kernel_stats.csv
cd $root
bash tools/collect_traffic.sh gpurun_out/r05/traffic "attn norm gemm" > $out/traffic.log 2>&1
cp $out/traffic/traffic_*.json $out/ 2>/dev/null; rm -rf $out/traffic
b() { python bench.py --no-cpu-baseline --no-kernel-timing --steps 8 --warmup 3 "$@" 2>/dev/null | grep '^{' | tail -1; }
{ SIDLSG_BENCH_DETAIL=/tmp/d.json b --kappa 4.5
  SIDLSG_BENCH_DETAIL=/tmp/d.json b --arch sd21-base --resolution 768 --kappa 2 --batch-gpu 4
  SIDLSG_BENCH_DETAIL=/tmp/d.json b --arch sd21-base --kappa 1.5
  SIDLSG_BENCH_DETAIL=/tmp/d.json b --arch sd21-base --kappa 1.5 --teacher-weights fp8
  SIDLSG_BENCH_DETAIL=/tmp/d.json b --arch sd21-base --kappa 1.5 --teacher-weights fp8-frozen
  SIDLSG_BENCH_DETAIL=/tmp/d.json b --batch-gpu 1
  SIDLSG_BENCH_DETAIL=/tmp/d.json b --batch-gpu 2
  SIDLSG_BENCH_DETAIL=/tmp/d.json b --batch-gpu 4
  SIDLSG_BENCH_DETAIL=/tmp/d.json b --graph
  SIDLSG_BENCH_DETAIL=/tmp/d.json b --force-exchange; } > $out/bench_other_configs.jsonl
python3 -c "
import json
for l in open('$out/bench_other_configs.jsonl'):
    d=json.loads(l); print(d['config']['workload'][:70], '|', d['config']['parallelism'], d['config']['teacher_weights'], 'graph' if d.get('graph') else '', '|', round(d['value'],2), 'img/s', round(d['ms_per_step'],1), 'ms', d['loss_check'])"
timeout 2400 python -m pytest tests -x -q -m gpu > $out/gpu_suite.txt 2>&1
tail -3 $out/gpu_suite.txt
ls -la $out
