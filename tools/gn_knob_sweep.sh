#!/bin/bash
# tools/gn_knob_sweep.sh OUT : GroupNorm two-kernel path geometry (blocks per launch, threads per block) after the prefetching row loops of round 6
out=${1:-gpurun_out/gn_sweep.txt}; : > $out
for blocks in 512 768 1024 1536; do for threads in 256 512; do
  echo "== SIDLSG_GN_BLOCKS=$blocks SIDLSG_GN_THREADS=$threads" >> $out
  SIDLSG_GN_BLOCKS=$blocks SIDLSG_GN_THREADS=$threads python tools/bench_kernels.py norm 2>/dev/null | grep "GN B16" >> $out
done; done
cat $out
