#!/bin/bash
# round 5, call 1: bench-workload fixtures (CPU-seeded weights) + the first compact bench line
set -x
mkdir -p gpurun_out
python tools/dump_bench_it0_inputs.py --out gpurun_out/bench_it0_inputs.npz 2>&1 | tail -3
cp tests/golden/bench_loss_reference.json gpurun_out/bench_loss_reference.json
python tools/make_bench_loss_reference.py --out gpurun_out/bench_loss_reference.json 2>&1 | tail -9
cp gpurun_out/bench_loss_reference.json tests/golden/bench_loss_reference.json
SIDLSG_BENCH_DETAIL=gpurun_out/r05_bench_detail_call1.json python bench.py > gpurun_out/r05_bench_call1.json 2> gpurun_out/r05_bench_call1.err
tail -c 4200 gpurun_out/r05_bench_call1.json; tail -5 gpurun_out/r05_bench_call1.err
