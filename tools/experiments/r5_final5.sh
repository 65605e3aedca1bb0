#!/bin/bash
root=$(pwd)
out=$root/gpurun_out/r05
mkdir -p $out
for i in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "dispatch_trace" -s 2>&1 | grep -E "batch of|passed|failed"; done
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu 2>&1 | tail -3
timeout 2700 python -m pytest tests -q -m gpu > $out/gpu_suite.txt 2>&1
tail -4 $out/gpu_suite.txt
