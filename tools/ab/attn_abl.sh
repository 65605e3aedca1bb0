#!/bin/bash
# tools/ab/attn_abl.sh : ablation builds of the software-pipelined attention forward (attention.hip recompiled with
# -DSIDLSG_SP_ABL=n / -DSIDLSG_SP_OCC=n, other objects reused) -> tools/ab/libattn_*.so; time them with tools/ab/attn_abl.py.
set -e
cd "$(dirname "$0")/../../sid_lsg_amd/csrc"
build() {
  name=$1; shift
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans "$@" -c attention.hip -o build/attention_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/gemm.o build/norm.o build/attention_$name.o build/elementwise.o build/optim.o build/fp32.o -o ../../tools/ab/libattn_$name.so
}
for n in ${ABLS:-1 2 3 4 5}; do build abl$n -DSIDLSG_SP_ABL=$n & done
build occ2 -DSIDLSG_SP_OCC=2 &
build occ4 -DSIDLSG_SP_OCC=4 &
wait
ls ../../tools/ab/libattn_*.so
