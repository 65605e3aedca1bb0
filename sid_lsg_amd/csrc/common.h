// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the SiD-LSG hot path.
// wave = 64 lanes; MFMA 16x16x32 bf16; LDS 160 KiB/CU.  No CUDA compatibility shims.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define SIDLSG_OK 0
#define SIDLSG_EINVAL (-22)

#define DEVFN __device__ __forceinline__

DEVFN float bf2f(bf16 x) { return (float)x; }
DEVFN bf16 f2bf(float x) { return (bf16)x; }

DEVFN float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// d/dx [x*sigmoid(x)] = s*(1 + x*(1-s))
DEVFN float silu_grad_f(float x) {
    float s = 1.0f / (1.0f + __expf(-x));
    return s * (1.0f + x * (1.0f - s));
}
DEVFN float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
DEVFN float gelu_grad_f(float x) {
    float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
    float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

DEVFN float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
DEVFN float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// 16-byte global load/store of 8 bf16
DEVFN bf16x8 ld8(const bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }
DEVFN void st8(bf16* p, bf16x8 v) { *reinterpret_cast<bf16x8*>(p) = v; }
DEVFN bf16x8 zero8() {
    u32x4 z = {0u, 0u, 0u, 0u};
    return __builtin_bit_cast(bf16x8, z);
}

// Launch-error helper for the extern "C" entry points: returns the HIP error code (0 = ok).
static inline int sidlsg_last_error() { return (int)hipGetLastError(); }
