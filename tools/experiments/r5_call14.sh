#!/bin/bash
root=$(pwd)
out=$root/gpurun_out/r5c14
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_unet.py tests/test_gpu_dist.py -x -q -m gpu -k "graph" -s > $out/tests.log 2>&1
tail -15 $out/tests.log | cut -c1-300
SIDLSG_BENCH_DETAIL=$out/detail_graph.json python bench.py --no-cpu-baseline --no-kernel-timing --steps 6 --warmup 4 --graph 2>$out/err.log | tail -1 | cut -c1-600
python3 -c "import json; d=json.load(open('$out/detail_graph.json')); print(json.dumps(d.get('loss_check_detail'), indent=0)[:1500])"
SIDLSG_GROUPED_FROZEN=0 SIDLSG_BENCH_DETAIL=$out/detail_graph0.json python bench.py --no-cpu-baseline --no-kernel-timing --steps 6 --warmup 4 --graph 2>$out/err.log | tail -1 | cut -c1-300
python3 -c "import json; d=json.load(open('$out/detail_graph0.json')); print(json.dumps(d.get('loss_check_detail'), indent=0)[:1500])"
