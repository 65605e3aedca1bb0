// Register-only MFMA streams for the 8-bit formats on gfx950: v_mfma_f32_16x16x32_fp8_fp8 (the non-scaled form the fp8-weight
// kernels use) against v_mfma_scale_f32_16x16x128_f8f6f4 with unit E8M0 scales (e4m3 operands) and the bf16 16x16x32 form.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters, int seed) {
    f32x4 acc[5][4];
    for (int i = 0; i < 5; i++) for (int j = 0; j < 4; j++) acc[i][j] = (f32x4){0, 0, 0, 0};
    if (MODE == 0) {
        bf16x8 a[4], b[5];
        for (int i = 0; i < 4; i++) for (int e = 0; e < 8; e++) a[i][e] = (__bf16)(0.37f * seed * (threadIdx.x % 13 + i + e));
        for (int i = 0; i < 5; i++) for (int e = 0; e < 8; e++) b[i][e] = (__bf16)(0.37f * seed * (threadIdx.x % 7 + i * 3 + e));
        for (int it = 0; it < iters; it++)
#pragma unroll
            for (int i = 0; i < 5; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[i], a[j], acc[i][j], 0, 0, 0);
    } else if (MODE == 1) {
        long a[4], b[5];
        for (int i = 0; i < 4; i++) a[i] = 0x3838383838383838L * seed + (threadIdx.x % 13 + i) * 0x0101010101010101L * seed;
        for (int i = 0; i < 5; i++) b[i] = 0x3030303030303030L * seed + (threadIdx.x % 7 + i) * 0x0101010101010101L * seed;
        for (int it = 0; it < iters; it++)
#pragma unroll
            for (int i = 0; i < 5; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(b[i], a[j], acc[i][j], 0, 0, 0);
    } else {
        i32x8 a[4], b[5];
        for (int i = 0; i < 4; i++) for (int e = 0; e < 8; e++) a[i][e] = (0x38383838 + (threadIdx.x % 13 + i + e) * 0x01010101) * seed;
        for (int i = 0; i < 5; i++) for (int e = 0; e < 8; e++) b[i][e] = (0x30303030 + (threadIdx.x % 7 + i * 3 + e) * 0x01010101) * seed;
        for (int it = 0; it < iters; it++)
#pragma unroll
            for (int i = 0; i < 5; i++)
#pragma unroll
                for (int j = 0; j < 4; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b[i], a[j], acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
    float s = 0;
    for (int i = 0; i < 5; i++) for (int j = 0; j < 4; j++) s += acc[i][j][0] + acc[i][j][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
static void run(const char* name, float* out, int kdim) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int seed : {0, 1}) {
        const int iters = 10000, blocks = 512;
        k<MODE><<<blocks, 256>>>(out, 2000, seed);
        hipEventRecord(e0);
        k<MODE><<<blocks, 256>>>(out, iters, seed);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double fl = (double)blocks * 4 * iters * 20 * 16 * 16 * kdim * 2;
        printf("%-44s %s operands: %8.2f ms  %6.0f TFLOP/s  %.1f ns per MFMA per SIMD\n", name, seed ? "non-zero" : "zero    ", ms, fl / ms / 1e9,
               ms * 1e6 / ((double)blocks * 4 / 1024 * iters * 20));
    }
}
int main() {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    run<0>("v_mfma_f32_16x16x32_bf16", out, 32);
    run<1>("v_mfma_f32_16x16x32_fp8_fp8", out, 32);
    run<2>("v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3)", out, 128);
    return 0;
}
