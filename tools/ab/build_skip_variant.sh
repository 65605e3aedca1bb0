#!/bin/bash
# tools/ab/build_skip_variant.sh : measurement build of the C ABI with -DSIDLSG_EXP_SKIP (common.h): kernels of the families named by
# $SIDLSG_EXP_SKIP_FAMILIES (bit mask over SIDLSG_FAM_*) are not launched.  WRONG results by construction; the step-time difference against the
# same library with mask 0 is what that family costs the iteration (its exposed time, contention included) -> tools/family_exposed_cost.sh
set -e
cd "$(dirname "$0")/../../sid_lsg_amd/csrc"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -DSIDLSG_EXP_SKIP"
/opt/rocm/bin/hipcc $F -c gemm.hip -o build/gemm_skip.o &
/opt/rocm/bin/hipcc $F -c norm.hip -o build/norm_skip.o &
/opt/rocm/bin/hipcc $F -mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans -c attention.hip -o build/attention_skip.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/gemm_skip.o build/norm_skip.o build/attention_skip.o build/elementwise.o build/optim.o build/fp32.o build/trace.o -o ../../tools/ab/libskip.so
echo tools/ab/libskip.so
