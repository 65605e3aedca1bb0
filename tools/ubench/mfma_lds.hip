// MFMA rate when the operands stream from LDS (no global traffic, no barriers): what the LDS->VGPR->MFMA path can
// sustain at the GEMM's ratio of 9 ds_read_b128 per 20 v_mfma_f32_16x16x32_bf16 (wave tile 64x80), 2 blocks per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int READS>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters, float seed) {
    __shared__ __attribute__((aligned(16))) __bf16 lds[2][288 * 64];
    for (int i = threadIdx.x; i < 2 * 288 * 64; i += 256) (&lds[0][0])[i] = (__bf16)(seed * ((i * 7 + threadIdx.x) % 23));
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    f32x4 acc[5][4];
    for (int i = 0; i < 5; i++) for (int j = 0; j < 4; j++) acc[i][j] = (f32x4){0, 0, 0, 0};
    bf16x8 fa[2][4], fw[2][5];
    auto rd = [&](int buf, int kk, bf16x8 (&a)[4], bf16x8 (&w)[5]) {
        const int ch = kk * 4 + lg;
#pragma unroll
        for (int mi = 0; mi < 4; mi++) {
            const int r = (wave >> 1) * 64 + mi * 16 + li;
            if (mi < READS) a[mi] = *reinterpret_cast<const bf16x8*>(&lds[buf][r * 64 + ((ch ^ ((r >> 1) & 7)) << 3)]);
        }
#pragma unroll
        for (int ni = 0; ni < 5; ni++) {
            const int r = 128 + (wave & 1) * 80 + ni * 16 + li;
            if (ni + 4 < READS) w[ni] = *reinterpret_cast<const bf16x8*>(&lds[buf][r * 64 + ((ch ^ ((r >> 1) & 7)) << 3)]);
        }
    };
    for (int i = 0; i < 4; i++) { fa[0][i] = fa[1][i] = *reinterpret_cast<const bf16x8*>(&lds[0][(lane + i) * 8]); }
    for (int i = 0; i < 5; i++) { fw[0][i] = fw[1][i] = *reinterpret_cast<const bf16x8*>(&lds[1][(lane + i) * 8]); }
    rd(0, 0, fa[0], fw[0]);
    for (int it = 0; it < iters; it++) {
        rd(it & 1, 1, fa[1], fw[1]);
#pragma unroll
        for (int i = 0; i < 5; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[0][i], fa[0][j], acc[i][j], 0, 0, 0);
        rd((it & 1) ^ 1, 0, fa[0], fw[0]);
#pragma unroll
        for (int i = 0; i < 5; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[1][i], fa[1][j], acc[i][j], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 5; i++) for (int j = 0; j < 4; j++) s += acc[i][j][0] + acc[i][j][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int READS>
void run() {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 10000, blocks = 512;
    k<READS><<<blocks, 256>>>(out, 100, 0.37f);
    hipEventRecord(e0);
    k<READS><<<blocks, 256>>>(out, iters, 0.37f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)blocks * 4 * iters * 40 * 16 * 16 * 32 * 2;
    printf("ds_read_b128 per 20 MFMA = %d : %.2f ms  %.0f TF/s\n", READS, ms, fl / ms / 1e9);
    hipFree(out);
}
int main() { run<0>(); run<3>(); run<6>(); run<9>(); return 0; }
