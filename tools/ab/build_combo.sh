#!/bin/bash
# tools/ab/build_combo.sh NAME "ATTN_FLAGS" "GEMM_FLAGS" : alternative build of the C ABI with attention.hip and / or gemm.hip
# recompiled with the given -D flags (empty string = reuse the product object) -> tools/ab/libNAME.so, selected at run time
# with SIDLSG_LIB (in-session A/B on one GPU box).
set -e
cd "$(dirname "$0")/../../sid_lsg_amd/csrc"
name=$1; aflags=$2; gflags=$3
CC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result"
aobj=build/attention.o; gobj=build/gemm.o
if [ -n "$aflags" ]; then aobj=build/attention_$name.o; $CC -mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans $aflags -c attention.hip -o $aobj & fi
if [ -n "$gflags" ]; then gobj=build/gemm_$name.o; $CC $gflags -c gemm.hip -o $gobj & fi
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $gobj build/norm.o $aobj build/elementwise.o build/optim.o build/fp32.o build/trace.o -o ../../tools/ab/lib$name.so
echo tools/ab/lib$name.so
