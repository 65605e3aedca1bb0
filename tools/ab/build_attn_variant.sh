#!/bin/bash
# tools/ab/build_attn_variant.sh NAME [-DFLAG ...] : alternative build of the C ABI with attention.hip recompiled with the flags
# (other objects reused) -> tools/ab/libNAME.so, selected at run time with SIDLSG_LIB (in-session A/B on one GPU box).
set -e
cd "$(dirname "$0")/../../sid_lsg_amd/csrc"
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans "$@" -c attention.hip -o build/attention_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/gemm.o build/norm.o build/attention_$name.o build/elementwise.o build/optim.o build/fp32.o build/trace.o -o ../../tools/ab/lib$name.so
echo tools/ab/lib$name.so
