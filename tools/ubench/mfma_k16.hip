// Issue rate of the legacy K=16 bf16 MFMA (v_mfma_f32_16x16x16_bf16) vs the K=32 one on gfx950: is a 40-wide head
// better served by K=32 + K=16 than by two K=32 instructions?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int K16>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters, float seed) {
    bf16x8 a = {}, b = {};
    for (int e = 0; e < 8; e++) { a[e] = (__bf16)(seed * (threadIdx.x % 13 + e)); b[e] = (__bf16)(seed * (threadIdx.x % 7 + e)); }
    s16x4 a4 = {1, 2, 3, (short)threadIdx.x}, b4 = {4, 5, 6, (short)threadIdx.x};
    f32x4 acc[16];
    for (int i = 0; i < 16; i++) acc[i] = (f32x4){0, 0, 0, 0};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (K16) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[i], 0, 0, 0);
            else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 16; i++) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int K16>
void run(const char* name) {
    float* out; hipMalloc(&out, 1024 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, blocks = 512;
    k<K16><<<blocks, 256>>>(out, 100, 0.37f);
    hipEventRecord(e0);
    k<K16><<<blocks, 256>>>(out, iters, 0.37f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)blocks * 4 * iters * 16;              // wave-instructions
    printf("%s: %.2f ms, %.2f ns per wave-instruction per SIMD (2 waves/SIMD)\n", name, ms, ms * 1e6 / (n / 1024.0));
    hipFree(out);
}
int main() { run<0>("v_mfma_f32_16x16x32_bf16"); run<1>("v_mfma_f32_16x16x16_bf16"); return 0; }
