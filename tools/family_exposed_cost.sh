#!/bin/bash
# tools/family_exposed_cost.sh OUT : step time of bench.py with one kernel family left out at a time (tools/ab/libskip.so, built by
# tools/ab/build_skip_variant.sh) -> what each family costs the ITERATION, as opposed to the sum of its kernels' durations (streams overlap,
# HBM-bound phases share the bandwidth).  Results of the runs are wrong by construction (loss_check FAILED is expected).  Run on the GPU box.
out=${1:-gpurun_out/exposed_cost.txt}
export SIDLSG_LIB=$(pwd)/tools/ab/libskip.so
names=(gemm conv attn_fwd attn_bwd wgrad conv_wgrad gn_fwd gn_bwd ln_fwd ln_bwd)
run() { SIDLSG_EXP_SKIP_FAMILIES=$1 timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-12s mask %-5s ms_per_step %.2f' % ('$2', '$1', d['ms_per_step']))" >> $out; }
rm -f $out
run 0 none
for i in 0 1 2 3 4 5 6 7 8 9; do run $((1 << i)) ${names[$i]}; done
run 0 none
cat $out
