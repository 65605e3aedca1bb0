set -x
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_cli.py -x -q -s > gpurun_out/r4_cli.log 2>&1
# host enqueue: two-stream vs grouped, full host and 1/8 of the cores (what 8 ranks on one host get)
NC=$(nproc); E=$((NC/8)); [ $E -lt 1 ] && E=1
for g in 0 1; do
  SIDLSG_GROUPED_FROZEN=$g timeout 300 python tools/host_ahead_probe.py 2>/dev/null | sed "s/^/grouped=$g cores=$NC: /" >> gpurun_out/r4_host.log
  SIDLSG_GROUPED_FROZEN=$g timeout 300 taskset -c 0-$((E-1)) python tools/host_ahead_probe.py 2>/dev/null | sed "s/^/grouped=$g cores=$E (taskset): /" >> gpurun_out/r4_host.log
done
for g in 0 1; do
  SIDLSG_GROUPED_FROZEN=$g timeout 300 taskset -c 0-$((E-1)) python bench.py --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('taskset $E cores grouped=$g', d['ms_per_step'], d['value'])" >> gpurun_out/r4_host.log
  SIDLSG_GROUPED_FROZEN=$g timeout 300 taskset -c 0-$((E-1)) python bench.py --no-cpu-baseline --no-kernel-timing --graph 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('taskset $E cores grouped=$g GRAPH', d['ms_per_step'], d['value'], d['host_enqueue_ms_per_step'])" >> gpurun_out/r4_host.log
done
# small batch: where the grouped launch should pay
for rep in 1 2; do for b in 1 2 4; do for g in 0 1; do
  SIDLSG_GROUPED_FROZEN=$g timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --batch-gpu $b --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch_gpu=$b grouped=$g', d['ms_per_step'], d['value'])" >> gpurun_out/r4_smallbatch.log
done; done; done
timeout 900 python tools/lossg_ablation.py --samples 6 > gpurun_out/r4_lossg_ablation.md 2> gpurun_out/r4_lossg_ablation.err
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r4_prof -o b --output-format csv -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/r4_bench_under_rocprof.json 2> $R/gpurun_out/r4_rocprof.err
cd $R
cp gpurun_out/r4_prof/b_kernel_stats.csv gpurun_out/r4_bench_step_kernel_stats.csv
python tools/kernel_by_grid.py gpurun_out/r4_prof/b_kernel_trace.csv gn_,ln_,colsum > gpurun_out/r4_norm_by_grid.txt
python tools/kernel_by_grid.py gpurun_out/r4_prof/b_kernel_trace.csv gemm_v3_kernel,gemm_bf16,gemm_as,gemm_finish > gpurun_out/r4_gemm_by_grid.txt
python tools/kernel_by_grid.py gpurun_out/r4_prof/b_kernel_trace.csv wgrad,geglu,concat,attn > gpurun_out/r4_other_by_grid.txt
rm -rf gpurun_out/r4_prof
cat gpurun_out/r4_host.log gpurun_out/r4_smallbatch.log; tail -n 3 gpurun_out/r4_cli.log
