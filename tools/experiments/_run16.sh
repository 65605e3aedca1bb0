set -x
root=$(pwd)
bash tools/attn_pmc.sh gpurun_out/r16/attn > gpurun_out/r16_attn.log 2>&1; rm -rf gpurun_out/r16/attn/p1 gpurun_out/r16/attn/p2
bash tools/sq_pmc.sh gpurun_out/r16/gemm "gemm_v3_kernel<0>,gemm_as_kernel,gemm_finish,gemm_bf16_kernel" "python tools/bench_kernels.py gemm (dense GEMM shapes of the step at batch 16, kernels alone)" python $root/tools/bench_kernels.py gemm
bash tools/sq_pmc.sh gpurun_out/r16/conv "gemm_v3_kernel<1>,gemm_bf16_kernel,gemm_finish" "python tools/bench_kernels.py conv (3x3 conv shapes of the step at batch 16, kernels alone)" python $root/tools/bench_kernels.py conv
bash tools/sq_pmc.sh gpurun_out/r16/wgrad "wgrad_" "python tools/bench_kernels.py wgrad (weight-gradient shapes at batch 16, kernels alone)" python $root/tools/bench_kernels.py wgrad
ls -la gpurun_out/r16/*
