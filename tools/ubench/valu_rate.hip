// Issue-rate microbenchmark for the VALU ops the attention softmax is made of (gfx950).
// hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = seed + threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (OP == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            if (OP == 1) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));
            if (OP == 2) asm volatile("v_max_f32 %0, %0, %0" : "+v"(a[i]));
            if (OP == 4) asm volatile("v_add_f32 %0, %0, %0" : "+v"(a[i]));
            if (OP == 6) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            if (OP == 7) asm volatile("v_exp_f16 %0, %0" : "+v"(a[i]));
            if (OP == 8) asm volatile("v_ldexp_f32 %0, %0, 1" : "+v"(a[i]));
            if (OP == 9) asm volatile("v_lshl_add_u32 %0, %0, 1, %0" : "+v"(a[i]));
        }
        if (OP == 3) {
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                f2 v = {a[i], a[i + 1]};
                asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(v));
                asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(v));
                a[i] = v[0]; a[i + 1] = v[1];
            }
        }
        if (OP == 5) {
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                unsigned r;
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a[i]), "v"(a[i + 1]));
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a[i]), "v"(a[i + 1]));
                a[i] = __uint_as_float(r);
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
void run(const char* name, int waves_per_simd) {
    float* out;
    const int blocks = 256 * waves_per_simd;     // 4 waves per block -> one wave per SIMD per block
    hipMalloc(&out, blocks * 256 * 4);
    const int iters = 4096;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(out, 16, 0.5f);
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(out, iters, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // instructions issued per SIMD = waves_per_simd * iters * 16
    const double ns_per = ms * 1e6 / ((double)waves_per_simd * iters * 16);
    printf("%-22s waves/SIMD=%d : %.3f ns per wave-instruction per SIMD (%.1f cycles @2.4GHz)\n", name, waves_per_simd, ns_per, ns_per * 2.4);
    hipFree(out);
}

int main() {
    for (int w = 1; w <= 4; w *= 2) {
        run<0>("v_exp_f32", w);
        run<7>("v_exp_f16", w);
        run<6>("v_rcp_f32", w);
        run<1>("v_fma_f32", w);
        run<2>("v_max_f32", w);
        run<4>("v_add_f32", w);
        run<3>("v_pk_fma_f32 (x16)", w);
        run<5>("v_cvt_pk_bf16_f32(x16)", w);
        run<8>("v_ldexp_f32", w);
        run<9>("v_lshl_add_u32", w);
    }
    return 0;
}
