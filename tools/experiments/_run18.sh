set -x
mkdir -p gpurun_out/r18
root=$(pwd)
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "attn or attention" > gpurun_out/r18_attn_tests.log 2>&1
tail -n 3 gpurun_out/r18_attn_tests.log
for v in 0 1; do echo "== SIDLSG_ATTN_XCD=$v"; SIDLSG_ATTN_XCD=$v timeout 300 python tools/bench_kernels.py attn 2>/dev/null | grep -E "self|cross"; done > gpurun_out/r18_attn_micro.log
cat gpurun_out/r18_attn_micro.log
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
for c in FETCH_SIZE WRITE_SIZE; do
  SIDLSG_ATTN_XCD=$v timeout 600 rocprofv3 --kernel-trace --pmc $c -d $root/gpurun_out/r18/x${v}_$c -o p --output-format csv -- python $root/tools/bench_kernels.py attn > $root/gpurun_out/r18/x${v}_$c.log 2>&1
done
done
cd $root
python tools/bench_kernels.py attn --trace-json gpurun_out/r18/alg_attn.json > gpurun_out/r18/alg_attn.log 2>&1
for v in 0 1; do
python tools/pmc_traffic_json.py gpurun_out/r18/x${v}_FETCH_SIZE/p_counter_collection.csv gpurun_out/r18/x${v}_WRITE_SIZE/p_counter_collection.csv gpurun_out/r18/traffic_attn_xcd$v.json attn gpurun_out/r18/alg_attn.json "attn=attn_q_kernel&, 0, " "attn_bwd=attn_dkdv_kernel+attn_q_kernel&, 1, "
python -c "
import json; d=json.load(open('gpurun_out/r18/traffic_attn_xcd$v.json'))
for k,v in d['families'].items(): print('xcd=$v',k,v['calls'],round(v['avg_hbm_side_bytes_per_call']/1e6,1),'MB', v.get('traffic_over_algorithmic'))
"
done
rm -rf gpurun_out/r18/x*_FETCH_SIZE gpurun_out/r18/x*_WRITE_SIZE
for rep in 1 2; do
for cfg in "1" "0"; do
  SIDLSG_ATTN_XCD=$cfg timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('attn_xcd=$cfg', d['ms_per_step'], d['value'], d['loss_check'])" >> gpurun_out/r18_ab.log
done; done
cat gpurun_out/r18_ab.log
