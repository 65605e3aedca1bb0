#!/bin/bash
# tools/ab/build_variant.sh NAME [-DFLAG ...] : alternative build of the C ABI (gemm.hip recompiled with the flags, other
# objects reused) -> tools/ab/libNAME.so, selected at run time with SIDLSG_LIB (in-session A/B on one GPU box).
set -e
cd "$(dirname "$0")/../../sid_lsg_amd/csrc"
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result "$@" -c gemm.hip -o build/gemm_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/gemm_$name.o build/norm.o build/attention.o build/elementwise.o build/optim.o build/fp32.o build/trace.o -o ../../tools/ab/lib$name.so
echo tools/ab/lib$name.so
