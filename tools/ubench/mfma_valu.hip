// Co-issue probe (gfx950): how much VALU work (v_exp_f32 / v_fma_f32 / v_cvt_pk_bf16_f32) hides beside an MFMA stream,
//   (a) inside ONE wave's instruction stream (NV VALU ops after every MFMA), at 1 / 2 / 4 waves per SIMD;
//   (b) split by role: in a 512-thread block waves 0-3 issue only MFMAs and waves 4-7 only VALU ops (one of each per SIMD).
// The attention softmax question: is a d=40 flash-attention tile (28 MFMA 16x16x32 + ~150 VALU) bound by max(MFMA, VALU) or
// by their sum?
// hipcc --offload-arch=gfx950 -O3 mfma_valu.hip -o mfma_valu && ./mfma_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

#define MFMA16(acc, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define MFMA32(acc, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))

template <int OP>
__device__ __forceinline__ void valu(float& x, float& y) {
    if (OP == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
    if (OP == 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y));
    if (OP == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x) : "v"(y));
    if (OP == 3) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y));
}

// KIND 0: 16x16x32 (8 independent accumulators), 1: 32x32x16 (4 accumulators).  NM MFMAs, NV VALU ops after each.
// ROLE 0: every wave runs both; 1: waves 0-3 MFMA only, waves 4-7 VALU only (512-thread block)
template <int KIND, int OP, int NV, int ROLE, int WPS>
__global__ __launch_bounds__(ROLE ? 512 : 256, ROLE ? 1 : WPS) void probe(float* out, int iters, float seed) {
    s16x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (short)(0x3f80 + threadIdx.x + i); b[i] = (short)(0x3f00 + i); }
    f32x4 c4[8];
    f32x16 c16[4];
    for (int i = 0; i < 8; i++) c4[i] = (f32x4){seed, seed, seed, seed};
    for (int i = 0; i < 4; i++) for (int j = 0; j < 16; j++) c16[i][j] = seed;
    float e[16];
    for (int i = 0; i < 16; i++) e[i] = seed * 0.001f + i * 0.01f - 1.f;
    float y = seed * 1e-3f;
    const bool do_m = ROLE == 0 || threadIdx.x < 256;
    const bool do_v = ROLE == 0 || threadIdx.x >= 256;
    for (int it = 0; it < iters; it++) {
        if (ROLE == 0) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (KIND == 0) MFMA16(c4[i], a, b); else if (KIND == 1) MFMA32(c16[i & 3], a, b);   // KIND 2: VALU only
#pragma unroll
                for (int j = 0; j < NV; j++) valu<OP>(e[(i * NV + j) & 15], y);
            }
        } else if (do_m) {
#pragma unroll
            for (int i = 0; i < 8; i++) { if (KIND == 0) MFMA16(c4[i], a, b); else MFMA32(c16[i & 3], a, b); }
        } else if (do_v) {
#pragma unroll
            for (int i = 0; i < 8 * NV; i++) valu<OP>(e[i & 15], y);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; i++) s += c4[i][0] + c4[i][3];
    for (int i = 0; i < 4; i++) s += c16[i][0] + c16[i][15];
    for (int i = 0; i < 16; i++) s += e[i];
    if (s == 123.456f) out[threadIdx.x] = s;
}

template <int KIND, int OP, int NV, int ROLE, int WPS>
static void run(const char* tag, float* d) {
    const int iters = 20000;
    const int blocks = 256 * (ROLE ? 1 : WPS);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<KIND, OP, NV, ROLE, WPS><<<blocks, ROLE ? 512 : 256>>>(d, 100, 1.0f);
    hipEventRecord(e0);
    probe<KIND, OP, NV, ROLE, WPS><<<blocks, ROLE ? 512 : 256>>>(d, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // ns per (MFMA + NV VALU) group and per SIMD: each SIMD ran WPS waves (ROLE 0) or one MFMA wave + one VALU wave (ROLE 1)
    const double groups_per_simd = (double)iters * 8 * (ROLE ? 1 : WPS);
    printf("%-44s kind=%s op=%d nv=%d role=%d wps=%d : %7.3f ms  %6.2f ns per MFMA-group per SIMD\n", tag, KIND == 2 ? "none" : (KIND ? "32x32x16" : "16x16x32"), OP, NV, ROLE,
           WPS, ms, ms * 1e6 / groups_per_simd);
}

int main() {
    float* d; hipMalloc(&d, 4096);
#define SWEEP(KIND, OP)                                                   \
    run<KIND, OP, 0, 0, 1>("mfma only, 1 wave/SIMD", d);                  \
    run<KIND, OP, 0, 0, 4>("mfma only, 4 waves/SIMD", d);                 \
    run<KIND, OP, 1, 0, 1>("1 valu per mfma, 1 wave", d);                 \
    run<KIND, OP, 2, 0, 1>("2 valu per mfma, 1 wave", d);                 \
    run<KIND, OP, 4, 0, 1>("4 valu per mfma, 1 wave", d);                 \
    run<KIND, OP, 6, 0, 1>("6 valu per mfma, 1 wave", d);                 \
    run<KIND, OP, 2, 0, 2>("2 valu per mfma, 2 waves", d);                \
    run<KIND, OP, 4, 0, 2>("4 valu per mfma, 2 waves", d);                \
    run<KIND, OP, 2, 0, 4>("2 valu per mfma, 4 waves", d);                \
    run<KIND, OP, 4, 0, 4>("4 valu per mfma, 4 waves", d);                \
    run<KIND, OP, 6, 0, 4>("6 valu per mfma, 4 waves", d);                \
    run<KIND, OP, 2, 1, 1>("role split: 2 valu per mfma", d);             \
    run<KIND, OP, 4, 1, 1>("role split: 4 valu per mfma", d);             \
    run<KIND, OP, 6, 1, 1>("role split: 6 valu per mfma", d);
    printf("== v_exp_f32 beside 16x16x32\n"); SWEEP(0, 0)
    printf("== v_fma_f32 beside 16x16x32\n"); SWEEP(0, 1)
    printf("== v_cvt_pk_bf16_f32 beside 16x16x32\n"); SWEEP(0, 2)
    printf("== v_exp_f32 beside 32x32x16\n"); SWEEP(1, 0)
    printf("== v_fma_f32 beside 32x32x16\n"); SWEEP(1, 1)
    printf("== VALU only (no MFMA), ns per group of NV ops\n");
    run<2, 0, 4, 0, 1>("4 v_exp, 1 wave", d); run<2, 0, 4, 0, 4>("4 v_exp, 4 waves", d);
    run<2, 1, 4, 0, 1>("4 v_fma, 1 wave", d); run<2, 1, 4, 0, 4>("4 v_fma, 4 waves", d);
    run<2, 2, 4, 0, 1>("4 v_cvt_pk, 1 wave", d); run<2, 2, 4, 0, 4>("4 v_cvt_pk, 4 waves", d);
    run<2, 3, 4, 0, 1>("4 v_max3, 1 wave", d); run<2, 3, 4, 0, 4>("4 v_max3, 4 waves", d);
    return 0;
}
