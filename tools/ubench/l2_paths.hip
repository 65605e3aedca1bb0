// Per-CU throughput of the two ways a tile gets from L2 into a CU on gfx950:
//   (a) buffer_load_dwordx4 ... lds   (LDS-DMA: 1 KiB per wave-instruction straight into LDS)
//   (b) buffer_load_dwordx4 -> VGPRs  (1 KiB per wave-instruction, 16 rows x 64 B like an MFMA B-operand fragment, or 1 KiB contiguous)
//   (c) both at once (half the instructions each)
// Every block streams a small L2-resident working set (per block 64 KiB, re-read), 256 threads, 2 blocks per CU, no compute.
// Prints B/clk/CU (2.4 GHz nominal) and TB/s for the chip.     hipcc --offload-arch=gfx950 -O3 l2_paths.hip -o l2_paths
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void* lptr_t;
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int ROWS>      // MODE 0: lds dma, 1: vgpr, 2: both.  ROWS: 1 = contiguous 1 KiB, 16 = 16 rows x 64 B (row pitch 1280 B)
__global__ __launch_bounds__(256, 2) void k(const char* src, int* sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const char* base = src + (size_t)blockIdx.x * 65536;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, 65536, 0x00020000);
    const unsigned voff_c = lane * 16;                                             // contiguous
    const unsigned voff_r = (lane & 15) * 1280 + (lane >> 4) * 16;                 // 16 rows x 64 B
    const unsigned voff = ROWS == 1 ? voff_c : voff_r;
    i32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const unsigned so = ((it * 8 + j) * 4 + wave) * 1024 % (ROWS == 1 ? 65536 : 20480);
            const bool dma = MODE == 0 || (MODE == 2 && (j & 1));
            if (dma) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr_t)(smem + (wave * 8 + j) * 1024), 16, voff_c, so, 0, 0);
            else {
                i32x4 v = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, so % (ROWS == 1 ? 65536 : 64), 0));
                acc += v;
            }
        }
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc[0] == 0x12345678) sink[threadIdx.x] = acc[1] + acc[2] + acc[3];
}

template <int MODE, int ROWS>
static void run(const char* name, const char* src, int* sink) {
    const int blocks = 512, iters = 2000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE, ROWS>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, ROWS>), dim3(blocks), dim3(256), 32768, 0, src, sink, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double bytes = (double)blocks * iters * 8 * 4 * 1024;
        if (rep == 2) printf("%-44s %7.2f TB/s  %6.1f B/clk/CU\n", name, bytes / ms / 1e9, bytes / ms / 1e-3 / 256 / 2.4e9);
    }
}

int main() {
    char* src; int* sink;
    hipMalloc(&src, 512 * 65536 + 65536); hipMalloc(&sink, 4096);
    hipMemset(src, 1, 512 * 65536 + 65536);
    run<0, 1>("LDS-DMA, 1 KiB contiguous", src, sink);
    run<1, 1>("to VGPRs, 1 KiB contiguous", src, sink);
    run<1, 16>("to VGPRs, 16 rows x 64 B", src, sink);
    run<2, 1>("half LDS-DMA + half VGPR (contiguous)", src, sink);
    run<2, 16>("half LDS-DMA + half VGPR (16 x 64 B)", src, sink);
    return 0;
}
