// Semantics probe for buffer_load_dwordx4 ... lds on gfx950: out-of-range voffset -> zeros in LDS? soffset outside the range check?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((address_space(3))) void* lptr_t;
__global__ void k(const unsigned short* a, unsigned short* out, int nbytes, int soff) {
    __shared__ __attribute__((aligned(16))) unsigned short s[64 * 8];
    for (int i = threadIdx.x; i < 512; i += 64) s[i] = 0xBEEF;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a), 0, nbytes, 0x00020000);
    int voff = threadIdx.x * 16;
    if (threadIdx.x % 4 == 1) voff = 0x80000000;          // flagged invalid
    if (threadIdx.x % 4 == 2) voff = nbytes - 8;          // straddles the end
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr_t)s, 16, voff, soff, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64) out[i] = s[i];
}
int main() {
    const int n = 4096;   // elements
    std::vector<unsigned short> h(n);
    for (int i = 0; i < n; i++) h[i] = (unsigned short)i;
    unsigned short *d, *o;
    hipMalloc(&d, n * 2); hipMalloc(&o, 1024);
    hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice);
    for (int soff : {0, 2048}) {
        // num_records covers only the first 2048 bytes: with soff = 2048 every access is physically past num_records
        k<<<1, 64>>>(d, o, 2048, soff);
        std::vector<unsigned short> r(512);
        hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
        printf("soff=%d\n", soff);
        for (int lane = 0; lane < 8; lane++) {
            printf("  lane %d:", lane);
            for (int e = 0; e < 8; e++) printf(" %5u", r[lane * 8 + e]);
            printf("\n");
        }
    }
    return 0;
}
