#!/bin/bash
# tools/collect_traffic.sh OUTDIR ["mode ..."] : FETCH_SIZE / WRITE_SIZE passes (separate rocprofv3 --pmc runs, --kernel-trace only) over the kernel
# micro-benchmarks of every family -> OUTDIR/traffic_<mode>.json (tools/pmc_traffic_json.py).  Run on the GPU box.
out=${1:-gpurun_out/traffic}
modes=${2:-gemm conv wgrad attn norm}
root=$(pwd)
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for mode in $modes; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $c -d $root/$out/${mode}_$c -o p --output-format csv -- python $root/tools/bench_kernels.py $mode > $root/$out/${mode}_$c.log 2>&1
  done
done
cd $root
for mode in $modes; do python tools/bench_kernels.py $mode --trace-json $out/alg_$mode.json > $out/alg_$mode.log 2>&1; done
P=tools/pmc_traffic_json.py
[[ " $modes " == *" gemm "* ]] && python $P $out/gemm_FETCH_SIZE/p_counter_collection.csv $out/gemm_WRITE_SIZE/p_counter_collection.csv $out/traffic_gemm.json gemm $out/alg_gemm.json "gemm=GemmParams+gemm_finish"
[[ " $modes " == *" conv "* ]] && python $P $out/conv_FETCH_SIZE/p_counter_collection.csv $out/conv_WRITE_SIZE/p_counter_collection.csv $out/traffic_conv.json conv $out/alg_conv.json "conv=GemmParams+gemm_finish"
[[ " $modes " == *" wgrad "* ]] && python $P $out/wgrad_FETCH_SIZE/p_counter_collection.csv $out/wgrad_WRITE_SIZE/p_counter_collection.csv $out/traffic_wgrad.json wgrad $out/alg_wgrad.json "wgrad=wgrad_v2&<0>|wgrad_v2s_kernel" "conv_wgrad=wgrad_v2&<1>|wgrad_v2f_kernel|wgrad_v2wf_kernel" "wgrad_reduce=wgrad_reduce"
[[ " $modes " == *" attn "* ]] && python $P $out/attn_FETCH_SIZE/p_counter_collection.csv $out/attn_WRITE_SIZE/p_counter_collection.csv $out/traffic_attn.json attn $out/alg_attn.json "attn=attn_q_kernel&, 0, " "attn_bwd=attn_dkdv_kernel+attn_q_kernel&, 1, "
[[ " $modes " == *" norm "* ]] && python $P $out/norm_FETCH_SIZE/p_counter_collection.csv $out/norm_WRITE_SIZE/p_counter_collection.csv $out/traffic_norm.json norm $out/alg_norm.json "gn=gn_apply_kernel|gn_small_fwd+gn_stats_kernel" "gn_bwd=gn_bwd_apply_kernel|gn_small_bwd+gn_bwd_stats_kernel" "ln=ln_fwd_kernel" "ln_bwd=ln_bwd_kernel" "norm_param_grad_reduce=colsum_reduce2"
rm -rf $out/*_FETCH_SIZE $out/*_WRITE_SIZE
ls -la $out
