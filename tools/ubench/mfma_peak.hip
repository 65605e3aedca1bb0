// Practical MFMA ceiling on this box: register-only v_mfma_f32_16x16x32_bf16 streams (random-ish vs zero operands).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256, 2) void k(float* out, int iters, float seed) {
    bf16x8 a[4], b[5];
    for (int i = 0; i < 4; i++) for (int e = 0; e < 8; e++) a[i][e] = (__bf16)(seed * (threadIdx.x % 13 + i + e));
    for (int i = 0; i < 5; i++) for (int e = 0; e < 8; e++) b[i][e] = (__bf16)(seed * (threadIdx.x % 7 + i * 3 + e));
    f32x4 acc[5][4];
    for (int i = 0; i < 5; i++) for (int j = 0; j < 4; j++) acc[i][j] = (f32x4){0, 0, 0, 0};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 5; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[i], a[j], acc[i][j], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 5; i++) for (int j = 0; j < 4; j++) s += acc[i][j][0] + acc[i][j][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (float seed : {0.0f, 0.37f}) for (int blocks : {256, 512, 1024}) {
        const int iters = 20000;
        k<<<blocks, 256>>>(out, 100, seed);
        hipEventRecord(e0);
        k<<<blocks, 256>>>(out, iters, seed);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double fl = (double)blocks * 4 * iters * 20 * 16 * 16 * 32 * 2;
        printf("seed=%.2f blocks=%d: %.2f ms  %.0f TF/s\n", seed, blocks, ms, fl / ms / 1e9);
    }
    return 0;
}
