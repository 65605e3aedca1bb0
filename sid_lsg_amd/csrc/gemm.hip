// bf16 MFMA GEMM / implicit-GEMM conv3x3 for gfx950 (MI355X).
//
//   C[M,N] = epilogue( A[M,K] * W[N,K]^T )        A: dense rows or NHWC implicit im2col
//
// Covers every contraction of the SD UNet except attention:
// ResBlock conv3x3 (fwd and dgrad), 1x1 conv / Linear (fwd and dgrad), GEGLU/FF, QKV.
// Layout decisions (DESIGN.md "Data layout"): activations NHWC bf16, so a 1x1 conv and a
// Linear over tokens are the same dense GEMM; conv weights are [Cout][3][3][Cin] (K index =
// tap*Cin + c) so a K-tile of 64 lies inside one tap and an A-tile row is a contiguous
// 128-byte run of channels of one input pixel.
//
// Kernels in this file:
//   gemm_v3_kernel<MODE>      the main kernel (N % 160 == 0, dense rows or conv with Cin % 64 == 0, >= 384 tiles or split-K):
//                             128 x 160 x 64 tile, tiles staged by buffer_load ... lds (direct to LDS), see its header
//   gemm_bf16_kernel<BM,BN,M> every other shape (ragged N, Cin % 64 != 0, small grids): register-staged loads
//   gemm_finish_kernel        epilogue of split-K launches (fp32 slabs -> bias/residual/activation -> bf16)
//   wgrad_v2_kernel / _v2w    weight (+ bias) gradient, direct-to-LDS staging, transpose reads; wgrad_bf16_kernel for
//                             unaligned / ragged operands; wgrad_reduce_kernel folds the pixel-split slabs
// Common tiling: 256 threads = 4 waves (2x2); block tile BM x BN x 64, wave tile 64 x BN/2 as 16x16x32 MFMAs; LDS double
// buffer, 128-byte rows with a 16-byte-chunk XOR swizzle that makes every ds_read_b128 lane group conflict free.
// Loads are branch-free: every operand is read through a buffer resource with the hardware range check -- an
// out-of-image tap / row >= M / k >= K lane simply gets the OOB offset and reads zeros.  Row offsets are computed once
// per tap (a wave-uniform event every Cin/64 tiles).  The MFMA is issued "transposed" (rows = output channel, cols =
// pixel) and W-tile rows are permuted per tile pair so that each lane ends up with 8 consecutive output channels of one
// pixel: the epilogue stores 16 bytes per lane.
#include "common.h"
#include <stdlib.h>
#include <algorithm>
#include <type_traits>

struct GemmParams {
    const bf16* A;        // dense: [M][lda]; conv: NHWC image [B][Hs][Ws][ldx]
    const bf16* W;        // [N][K] row-major (K contiguous)
    void* C;              // [M][ldc] bf16 (or fp32 when OUT_F32)
    const float* bias;    // [N] or null
    const bf16* res;      // [M][ldres] or null
    const float* rowvec;  // [M/rows_per_batch][ldrv] or null (time-embedding broadcast)
    int ldrv;
    int M, N, K;
    int lda, ldc, ldres;
    int rows_per_batch;
    // conv3x3 (pad 1): virtual input H x Wd (after optional nearest x2 upsample), Cin channels
    int H, Wd, Cin, Ho, Wo, stride, ups;
    float alpha;
    int flags;
    unsigned a_bytes, w_bytes;   // sizes of the A / W allocations seen by the kernel (< 2 GiB)
    int kt_per_split;            // split-K: K-tiles per split (0 = no split)
    int group_m;                 // v3: m-tiles per group of the logical tile order (see gemm_v3_kernel)
    float* ws;                   // split-K: fp32 [splits][M][N] partial-sum slabs
    const float* wscale;         // fp8-weight path: per-output-channel dequantisation scale [N] (else null)
    // grouped launch (sidlsg_*_g2: two networks of identical shape evaluated on one stacked activation matrix): rows [0, Mg) are
    // contracted with (W, bias), rows [Mg, M) with (W1, bias1).  Mg = 0: ordinary launch.  Mtot: row count of a split-K slab
    // (= M; kept separately because a block narrows its M to its own set's row range, see group_select)
    const bf16* W1;
    const float* bias1;
    int Mg, Mtot;
    // fused GEGLU (sidlsg_gemm_geglu_bf16: the transformer's FF-in projection): geglu = F = N / 2 > 0 -> an output tile holds 80
    // features of the "a" half (rows [80 t, +80) of W) and the SAME 80 features of the gate half (rows [F + 80 t, +80)), the
    // epilogue writes h (natural column order, C may be null) and Y[m][f] = a * gelu(g)
    bf16* Y;
    int ldy, geglu;
    // fused GEGLU BACKWARD (sidlsg_gemm_geglu_bwd_bf16: the data gradient of the FF-out projection): geglu_bwd = F > 0 -> the output
    // tile is dy[:, n0 .. n0 + 160) = dOut W2 (never stored); the epilogue reads h[:, n] / h[:, F + n] from Hin and writes
    // dH[:, n] = dy * gelu(g), dH[:, F + n] = dy * a * gelu'(g) to Y (both [M][ldy], ldy >= 2 F)
    const bf16* Hin;
    int geglu_bwd;
};

// m-tiles of a launch: each set of a grouped launch starts on a tile boundary of its own (Mg need not be a multiple of BM)
__host__ __device__ inline int m_tiles_rt(const GemmParams& p, int bm) { return p.Mg ? 2 * ((p.Mg + bm - 1) / bm) : (p.M + bm - 1) / bm; }
// the block's set: rewrites p (a by-value kernel argument: plain scalar registers) to that set's weights, bias and row limit
// and returns the block's first row.  Everything downstream (loader range checks, epilogue, stores) keeps using p unchanged.
template <int BM>
__device__ __forceinline__ int group_select(GemmParams& p, int mt) {
    if (!p.Mg) return mt * BM;
    const int tg = (p.Mg + BM - 1) / BM;
    if (mt >= tg) { p.W = p.W1; p.bias = p.bias1; return p.Mg + (mt - tg) * BM; }
    p.M = p.Mg;
    return mt * BM;
}

enum { F_OUT_F32 = 1, F_SILU = 2, F_ACCUM = 4 };

constexpr int BK = 64;
constexpr int NTHREADS = 256;
constexpr unsigned OOB = 0x80000000u;   // any offset >= 2 GiB is out of range for our buffers -> load returns 0

DEVFN bf16x8 buf_ld8(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}

// MODE 0: dense rows.  MODE 1: conv3x3 with Cin % 64 == 0 (a K-tile lies inside one tap).  MODE 2: conv3x3, any Cin % 8 == 0.
// Shared epilogue: lane (lg, li) holds, for pixel row m = mbase + mi*16 + li, 8 (or 4) consecutive channels per tile pair.
//
// Every global load of the epilogue is issued BEFORE the first store.  C is not __restrict__, so a load that follows a store
// in program order stays behind it: the earlier form (load bias / rowvec / residual, add, store, per 16-row block) made
// MT = 4 dependent round trips through a memory system that the co-resident blocks' tile DMAs keep saturated -- 5.5 us per
// 128 x 160 tile in the phase trace of tools/ab/gemm_trace.py, as long as the whole K loop of a K = 320 GEMM.  The bias
// (a function of the column only) can be fetched before the K loop (epilogue_prefetch) and costs nothing at all.
template <int NT>
struct EpiPre {
    float b[(NT + 1) / 2][8];
};

template <int NT>
DEVFN void epilogue_prefetch(const GemmParams& p, EpiPre<NT>& pre, int nbase, int lg) {
#pragma unroll
    for (int pr = 0; pr < (NT + 1) / 2; pr++) {
        const bool paired = (2 * pr + 1) < NT;
        const int cnt = paired ? 8 : 4;
        const int n = nbase + 32 * pr + lg * cnt;
#pragma unroll
        for (int e = 0; e < 8; e++) pre.b[pr][e] = 0.f;
        if (p.bias && (p.N & 7) == 0 && n < p.N) {
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + n);
            pre.b[pr][0] = b0[0]; pre.b[pr][1] = b0[1]; pre.b[pr][2] = b0[2]; pre.b[pr][3] = b0[3];
            if (paired) {
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(p.bias + n + 4);
                pre.b[pr][4] = b1[0]; pre.b[pr][5] = b1[1]; pre.b[pr][6] = b1[2]; pre.b[pr][7] = b1[3];
            }
        }
    }
}

// stage != nullptr (bf16 outputs, N % 8 == 0): the finished values go to an LDS image of the block's output tile
// (row stride ldr elements, origin (tm0, tn0)) instead of global memory; gemm_store_rows then writes it out row-major.
//
// CODE SIZE is a first-order cost here.  One fully unrolled epilogue that tests every flag per 8-channel group (SiLU with its
// exp, fp32 / accumulate outputs, ragged N, residual, row vector) made gemm_v3_kernel 113 KB of code -- 97 KB of it epilogue --
// against a 64 KB instruction cache shared by two CUs whose four resident blocks are all in different phases.  The phase
// trace (tools/ab/gemm_trace.py) showed the epilogue of one 128 x 160 tile taking 5.5 - 7.3 us, as long as the whole K loop
// of a K = 320 GEMM, and 1.5 us with a bias-only build: the time was instruction fetch through a memory system that the
// tile DMAs keep saturated.  Hence: the feature set is decided ONCE per call (FEAT, uniform branch) and each hot
// combination (bf16 output, N % 8 == 0, no activation: bias only / + residual / + row vector) is its own compact
// straight-line instantiation; everything else takes the generic instantiation (FEAT 4, flags read at run time).
template <int MT, int NT, int FEAT>      // FEAT: 0 bias only, 1 + residual, 2 + row vector, 4 generic
DEVFN void epilogue_rows8(const GemmParams& p, f32x4 (&acc)[NT][MT], int mbase, int nbase, int li, int lg,
                          const EpiPre<NT>& pre, bf16* stage, int tm0, int tn0, int ldr) {
    constexpr int PR = (NT + 1) / 2;
    constexpr bool GENERIC = (FEAT & 4) != 0;
    const bool has_res = GENERIC ? p.res != nullptr : (FEAT & 1) != 0;
    const bool has_rv = GENERIC ? p.rowvec != nullptr : (FEAT & 2) != 0;
    const int flags = GENERIC ? p.flags : 0;
    // rowvec (one vector per batch image): the 16*MT <= 64 rows of this wave lie in at most two images when an image has
    // >= 64 rows (always on the SD path: 8 x 8 latents at the coarsest level) -> two vectors, selected per row
    f32x4 rvv[2][PR][2];
    bf16x8 rsv[MT][PR];
    const int img0 = (mbase < p.M ? mbase : 0) / p.rows_per_batch;
    const bool rv_two = has_rv && (!GENERIC || p.rows_per_batch >= 16 * MT);
    if (rv_two) {
        const int last = (min(mbase + 16 * MT, p.M) - 1) / p.rows_per_batch;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const float* rv = p.rowvec + (size_t)(h ? max(last, img0) : img0) * p.ldrv;
#pragma unroll
            for (int pr = 0; pr < PR; pr++) {
                const bool paired = (2 * pr + 1) < NT;
                const int n = nbase + 32 * pr + lg * (paired ? 8 : 4);
                rvv[h][pr][0] = rvv[h][pr][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (mbase < p.M && n < p.N) {
                    rvv[h][pr][0] = *reinterpret_cast<const f32x4*>(rv + n);
                    if (paired) rvv[h][pr][1] = *reinterpret_cast<const f32x4*>(rv + n + 4);
                }
            }
        }
    }
    if (has_res) {
#pragma unroll
        for (int mi = 0; mi < MT; mi++) {
            const int m = mbase + mi * 16 + li;
#pragma unroll
            for (int pr = 0; pr < PR; pr++) {
                const bool paired = (2 * pr + 1) < NT;
                const int n = nbase + 32 * pr + lg * (paired ? 8 : 4);
                rsv[mi][pr] = zero8();
                if (m < p.M && n < p.N) {
                    const bf16* rp = p.res + (size_t)m * p.ldres + n;
                    if (paired) rsv[mi][pr] = ld8(rp);
                    else {
                        const bf16x4 t = *reinterpret_cast<const bf16x4*>(rp);
                        rsv[mi][pr][0] = t[0]; rsv[mi][pr][1] = t[1]; rsv[mi][pr][2] = t[2]; rsv[mi][pr][3] = t[3];
                    }
                }
            }
        }
    }
#pragma unroll
    for (int mi = 0; mi < MT; mi++) {
        const int m = mbase + mi * 16 + li;
        if (m >= p.M) continue;
        const bool first_img = rv_two && (m / p.rows_per_batch == img0);
#pragma unroll
        for (int pr = 0; pr < PR; pr++) {
            const bool paired = (2 * pr + 1) < NT;
            const int cnt = paired ? 8 : 4;
            const int n = nbase + 32 * pr + lg * cnt;
            if (n >= p.N) continue;
            float v[8];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                v[r] = acc[2 * pr][mi][r];
                v[4 + r] = paired ? acc[(2 * pr + 1) < NT ? 2 * pr + 1 : 2 * pr][mi][r] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 8; e++) {
                float x = v[e] * p.alpha + pre.b[pr][e];
                if (rv_two) x += first_img ? rvv[0][pr][e >> 2][e & 3] : rvv[1][pr][e >> 2][e & 3];
                else if (GENERIC && has_rv) x += p.rowvec[(size_t)(m / p.rows_per_batch) * p.ldrv + n + e];
                if (has_res) x += bf2f(rsv[mi][pr][e]);
                if (GENERIC && (flags & F_SILU)) x = silu_f(x);
                v[e] = x;
            }
#ifdef SIDLSG_EXP_NOSTORE      // measurement build: everything but the global stores
            if (v[0] != 123456.75f) continue;
#endif
            if (GENERIC && (flags & F_OUT_F32)) {
                float* c = reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n;
                if (flags & F_ACCUM) {
                    for (int e = 0; e < cnt; e++) c[e] += v[e];
                } else {
                    *reinterpret_cast<f32x4*>(c) = (f32x4){v[0], v[1], v[2], v[3]};
                    if (cnt == 8) *reinterpret_cast<f32x4*>(c + 4) = (f32x4){v[4], v[5], v[6], v[7]};
                }
            } else {
                bf16* c = stage ? stage + (m - tm0) * ldr + (n - tn0) : reinterpret_cast<bf16*>(p.C) + (size_t)m * p.ldc + n;
                if (cnt == 8) {
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; e++) o[e] = f2bf(v[e]);
                    st8(c, o);
                } else {
                    bf16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; e++) o[e] = f2bf(v[e]);
                    *reinterpret_cast<bf16x4*>(c) = o;
                }
            }
        }
    }
}

template <int MT, int NT, bool HOT = true>       // HOT = false: the caller has already peeled the hot combinations off
DEVFN void gemm_epilogue(const GemmParams& p, f32x4 (&acc)[NT][MT], int mbase, int nbase, int li, int lg,
                         const EpiPre<NT>& pre, bf16* stage = nullptr, int tm0 = 0, int tn0 = 0, int ldr = 0) {
    constexpr int PR = (NT + 1) / 2;
    if ((p.N & 7) == 0) {
        // ---- N % 8 == 0: every lane's 8 (4) channels are all in range or all out of range
        const bool plain = HOT && !(p.flags & (F_SILU | F_OUT_F32 | F_ACCUM));
#ifndef SIDLSG_EXP_GENERIC_EPILOGUE
        if (plain && !p.rowvec) {
            if (!p.res) epilogue_rows8<MT, NT, 0>(p, acc, mbase, nbase, li, lg, pre, stage, tm0, tn0, ldr);
            else epilogue_rows8<MT, NT, 1>(p, acc, mbase, nbase, li, lg, pre, stage, tm0, tn0, ldr);
        } else if (plain && !p.res && p.rows_per_batch >= 16 * MT) {
            epilogue_rows8<MT, NT, 2>(p, acc, mbase, nbase, li, lg, pre, stage, tm0, tn0, ldr);
        } else
#endif
        {
            (void)plain;
            epilogue_rows8<MT, NT, 4>(p, acc, mbase, nbase, li, lg, pre, stage, tm0, tn0, ldr);
        }
        return;
    }
    // ---- ragged N: element-wise, guarded
#pragma unroll
    for (int mi = 0; mi < MT; mi++) {
        const int m = mbase + mi * 16 + li;
        if (m >= p.M) continue;
        const float* rv = p.rowvec ? p.rowvec + (size_t)(m / p.rows_per_batch) * p.ldrv : nullptr;
#pragma unroll
        for (int pr = 0; pr < PR; pr++) {
            const bool paired = (2 * pr + 1) < NT;
            const int cnt = paired ? 8 : 4;
            const int n = nbase + 32 * pr + lg * cnt;
            float v[8];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                v[r] = acc[2 * pr][mi][r];
                v[4 + r] = paired ? acc[(2 * pr + 1) < NT ? 2 * pr + 1 : 2 * pr][mi][r] : 0.f;
            }
            for (int e = 0; e < cnt; e++) {
                const int nn = n + e;
                if (nn >= p.N) break;
                float x = v[e] * p.alpha;
                if (p.bias) x += p.bias[nn];
                if (rv) x += rv[nn];
                if (p.res) x += bf2f(p.res[(size_t)m * p.ldres + nn]);
                if (p.flags & F_SILU) x = silu_f(x);
                if (p.flags & F_OUT_F32) {
                    float* c = reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + nn;
                    *c = (p.flags & F_ACCUM) ? *c + x : x;
                } else {
                    reinterpret_cast<bf16*>(p.C)[(size_t)m * p.ldc + nn] = f2bf(x);
                }
            }
        }
    }
}

// Row-major write-out of a BM x BN bf16 output tile staged in LDS by gemm_epilogue(stage=...).  In the MFMA accumulator
// layout one store instruction covers 16 rows x 64 bytes: half cache lines, which the L2 takes at half its write rate
// (FF-in 65536 x 2560 x 320: 335 MB of output left at 2.2 TB/s and were 58 % of the kernel -- the one-K-tile build of
// tools/ab/ffin.py).  From LDS, consecutive lanes write consecutive 16-byte chunks of a row: BN*2-byte contiguous runs.
template <int BM, int BN>
DEVFN void gemm_store_rows(const GemmParams& p, const bf16* stage, int m0, int n0, int ldr, int tid) {
    constexpr int CPR = BN / 8;                 // 16-byte chunks per row
#pragma unroll
    for (int i = 0; i < (BM * CPR + NTHREADS - 1) / NTHREADS; i++) {
        const int c = tid + i * NTHREADS;
        const int row = c / CPR, col = (c - row * CPR) * 8;
        if ((BM * CPR) % NTHREADS != 0 && row >= BM) break;
        const int m = m0 + row, n = n0 + col;
        if (m < p.M && n < p.N) st8(reinterpret_cast<bf16*>(p.C) + (size_t)m * p.ldc + n, *reinterpret_cast<const bf16x8*>(stage + row * ldr + col));
    }
}

template <int BM, int BN, int MODE>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_bf16_kernel(GemmParams p) {   // 2 blocks/CU: <= 256 registers
    constexpr int WMT = BM / 2, WNT = BN / 2;   // wave tile
    constexpr int MT = WMT / 16, NT = WNT / 16;
    constexpr int AR = BM / 32, BR = BN / 32;   // 16B chunks per thread per tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* As = reinterpret_cast<bf16*>(smem);                 // [2][BM][BK]
    bf16* Bs = As + 2 * BM * BK;                              // [2][BN][BK]

    // ---- XCD-aware block -> tile map: blocks that share an A row-panel sit on one XCD (own L2)
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = m_tiles_rt(p, BM);
    const int nblk = tiles_n * tiles_m;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int m0 = group_select<BM>(p, bid / tiles_n);
    const int n0 = (bid % tiles_n) * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave >> 1) * WMT, wn0 = (wave & 1) * WNT;
    const int kc = tid & 7, r0 = tid >> 3;

    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.A), 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.W), 0, (int)p.w_bytes, 0x00020000);

    // ---- per-thread row descriptors (byte offsets; OOB = reads zeros)
    const int Hs = p.ups ? (p.H >> 1) : p.H, Ws = p.ups ? (p.Wd >> 1) : p.Wd;
    unsigned abase[AR];      // dense: row offset (+ chunk); conv: batch image offset (+ chunk)
    int ahi[AR], awi[AR];    // conv: top-left input coordinate of the 3x3 window
    unsigned arow[AR];       // conv MODE 1: current per-tap row offset
#pragma unroll
    for (int i = 0; i < AR; i++) {
        const int m = m0 + r0 + 32 * i;
        const bool ok = m < p.M;
        if (MODE == 0) {
            abase[i] = ok ? (unsigned)m * (unsigned)p.lda * 2u + kc * 16u : OOB;
            ahi[i] = awi[i] = 0;
        } else {
            const int mm = ok ? m : 0;
            const int hw = p.Ho * p.Wo;
            const int b = mm / hw, rem = mm - b * hw;
            const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
            abase[i] = (unsigned)b * (unsigned)(Hs * Ws) * (unsigned)p.lda * 2u + kc * 16u;
            ahi[i] = ok ? ho * p.stride - 1 : -100000;       // pushes every tap out of range
            awi[i] = wo * p.stride - 1;
        }
        arow[i] = OOB;
    }
    unsigned bbase[BR];
#pragma unroll
    for (int i = 0; i < BR; i++) {
        const int n = n0 + r0 + 32 * i;
        bbase[i] = n < p.N ? (unsigned)n * (unsigned)p.K * 2u + kc * 16u : OOB;
    }

    auto tap_rows = [&](int tap) {   // MODE 1: wave-uniform tap -> per-row offsets
        const int dh = tap / 3, dw = tap - dh * 3;
#pragma unroll
        for (int i = 0; i < AR; i++) {
            int hi = ahi[i] + dh, wi = awi[i] + dw;
            const bool ok = hi >= 0 && hi < p.H && wi >= 0 && wi < p.Wd;
            if (p.ups) { hi >>= 1; wi >>= 1; }
            arow[i] = ok ? abase[i] + (unsigned)(hi * Ws + wi) * (unsigned)p.lda * 2u : OOB;
        }
    };

    bf16x8 areg[AR], breg[BR];
    int cur_tap = -1;
    auto load_tile = [&](int kt) {
        const int k0 = kt * BK;
        const bool kok = k0 + kc * 8 < p.K;
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < AR; i++) areg[i] = buf_ld8(ra, kok ? abase[i] + (unsigned)k0 * 2u : OOB);
        } else if (MODE == 1) {
            const int tap = k0 / p.Cin;          // scalar
            const int c0 = k0 - tap * p.Cin;
            if (tap != cur_tap) { tap_rows(tap); cur_tap = tap; }
#pragma unroll
            for (int i = 0; i < AR; i++) areg[i] = buf_ld8(ra, arow[i] + (unsigned)c0 * 2u);
        } else {
            const int k = k0 + kc * 8;
            const int tap = k / p.Cin, c = k - tap * p.Cin;
            const int dh = tap / 3, dw = tap - dh * 3;
#pragma unroll
            for (int i = 0; i < AR; i++) {
                int hi = ahi[i] + dh, wi = awi[i] + dw;
                const bool ok = kok && hi >= 0 && hi < p.H && wi >= 0 && wi < p.Wd;
                if (p.ups) { hi >>= 1; wi >>= 1; }
                // abase already contains kc*16 bytes; replace it by the channel offset of this chunk
                areg[i] = buf_ld8(ra, ok ? abase[i] - kc * 16u + ((unsigned)(hi * Ws + wi) * (unsigned)p.lda + (unsigned)c) * 2u : OOB);
            }
        }
#pragma unroll
        for (int i = 0; i < BR; i++) breg[i] = buf_ld8(rw, kok ? bbase[i] + (unsigned)k0 * 2u : OOB);
    };
    auto store_tile = [&](int buf) {
        bf16* a = As + buf * BM * BK;
        bf16* b = Bs + buf * BN * BK;
#pragma unroll
        for (int i = 0; i < AR; i++) {
            const int r = r0 + 32 * i;
            st8(a + r * BK + ((kc ^ ((r >> 1) & 7)) << 3), areg[i]);
        }
#pragma unroll
        for (int i = 0; i < BR; i++) {
            const int r = r0 + 32 * i;
            st8(b + r * BK + ((kc ^ ((r >> 1) & 7)) << 3), breg[i]);
        }
    };

    f32x4 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; i++)
#pragma unroll
        for (int j = 0; j < MT; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // operand row (within the wave's n range) that lane (l&15) of n-tile ni supplies
    const int li = lane & 15, lg = lane >> 4;
    int wrow[NT];
#pragma unroll
    for (int ni = 0; ni < NT; ni++) {
        const bool paired = (ni | 1) < NT;
        wrow[ni] = wn0 + (paired ? 32 * (ni >> 1) + (li >> 2) * 8 + (ni & 1) * 4 + (li & 3) : 16 * ni + li);
    }

    const int nk_all = (p.K + BK - 1) / BK;
    const int kt_begin = p.kt_per_split ? blockIdx.y * p.kt_per_split : 0;
    const int nk = p.kt_per_split ? min(nk_all, kt_begin + p.kt_per_split) : nk_all;

    auto read_frags = [&](int buf, int kk, bf16x8 (&fa)[MT], bf16x8 (&fw)[NT]) {
        const bf16* a = As + buf * BM * BK;
        const bf16* b = Bs + buf * BN * BK;
        const int ch = kk * 4 + lg;
#pragma unroll
        for (int mi = 0; mi < MT; mi++) {
            const int r = wm0 + mi * 16 + li;
            fa[mi] = *reinterpret_cast<const bf16x8*>(a + r * BK + ((ch ^ ((r >> 1) & 7)) << 3));
        }
#pragma unroll
        for (int ni = 0; ni < NT; ni++) {
            const int r = wrow[ni];
            fw[ni] = *reinterpret_cast<const bf16x8*>(b + r * BK + ((ch ^ ((r >> 1) & 7)) << 3));
        }
    };
    auto mfma_block = [&](const bf16x8 (&fa)[MT], const bf16x8 (&fw)[NT]) {
#pragma unroll
        for (int ni = 0; ni < NT; ni++)
#pragma unroll
            for (int mi = 0; mi < MT; mi++)
                acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[ni], fa[mi], acc[ni][mi], 0, 0, 0);
    };

    // Software pipeline (one barrier per K-tile, no exposed LDS latency):
    //   A: issue the kk=1 fragment reads of the current LDS buffer, run the kk=0 MFMAs on the fragments prefetched earlier
    //   B: commit the staged next tile (global -> registers, in flight since the previous D) to the other LDS buffer
    //   C: barrier   (the current buffer is now fully in registers, the next one fully written)
    //   D: issue the kk=0 fragment reads of the NEXT buffer and the global loads of the tile after it
    //   E: run the kk=1 MFMAs, which cover the latencies of D
    bf16x8 fa0[MT], fw0[NT], fa1[MT], fw1[NT];
    load_tile(kt_begin);
    store_tile(0);
    __syncthreads();
    read_frags(0, 0, fa0, fw0);
    if (kt_begin + 1 < nk) load_tile(kt_begin + 1);
    auto ktile = [&](const int kt, auto has_next, auto fetch) {     // straight-line instantiations: see gemm_v3_kernel
        const int buf = (kt - kt_begin) & 1;
        read_frags(buf, 1, fa1, fw1);          // A
        mfma_block(fa0, fw0);
        if constexpr (decltype(has_next)::value) store_tile(buf ^ 1);         // B
        __syncthreads();                       // C
        if constexpr (decltype(has_next)::value) {                            // D
            read_frags(buf ^ 1, 0, fa0, fw0);
            if constexpr (decltype(fetch)::value) load_tile(kt + 2);
        }
        mfma_block(fa1, fw1);                  // E
    };
    {
        int kt = kt_begin;
        for (; kt + 2 < nk; kt++) ktile(kt, std::true_type{}, std::true_type{});
        if (kt + 1 < nk) { ktile(kt, std::true_type{}, std::false_type{}); kt++; }
        ktile(kt, std::false_type{}, std::false_type{});
    }

    if (p.kt_per_split) {
        // split-K: raw fp32 partial sums into the workspace; gemm_finish_kernel applies the epilogue
#pragma unroll
        for (int mi = 0; mi < MT; mi++) {
            const int m = m0 + wm0 + mi * 16 + li;
            if (m >= p.M) continue;
#pragma unroll
            for (int ni = 0; ni < NT; ni++) {
                const bool paired = (ni | 1) < NT;
                const int nb = n0 + wn0 + (paired ? 32 * (ni >> 1) + lg * 8 + (ni & 1) * 4 : 16 * ni + lg * 4);
                float* dst = p.ws + ((size_t)blockIdx.y * p.Mtot + m) * p.N + nb;     // this split's slab: plain 16-byte stores
                if (nb + 4 <= p.N) *reinterpret_cast<f32x4*>(dst) = acc[ni][mi];
                else
                    for (int r = 0; r < 4; r++)
                        if (nb + r < p.N) dst[r] = acc[ni][mi][r];
            }
        }
        return;
    }
    EpiPre<NT> pre;
    epilogue_prefetch<NT>(p, pre, n0 + wn0, lg);
    gemm_epilogue<MT, NT>(p, acc, m0 + wm0, n0 + wn0, li, lg, pre);
}

// ---------------------------------------------------------------------------------------------
// fp8-weight path (BASELINE.json configs[4]: "fp8 MFMA weights + bf16 accum"; no reference behaviour exists for it --
// SURVEY.md Appendix C -- so the contract is this file's): frozen networks (teacher, fake score under evaluation) keep
// their GEMM / conv weights as OCP e4m3 bytes with one fp32 scale per output channel (sidlsg_quantize_fp8_rows); the
// activations arrive as bf16, are converted to e4m3 in the loader (static unit scale, clamped to +-448 before the conversion: normalised
// activations sit well inside that range, the residual stream of real weights may not) and the contraction runs on v_mfma_f32_16x16x32_fp8_fp8 with fp32 accumulation;
// the epilogue multiplies by the channel scale.  Half the operand bytes through L2 / LDS; the non-scaled fp8 MFMA has
// the bf16 rate on gfx950 (only the MX block-scaled K=128 forms are faster), so this is a bandwidth, not a FLOP, lever.
// 128 x 128 x 64 tile, 4 waves (64 x 64 each), register-staged loads (the bf16 -> fp8 conversion needs registers), LDS
// rows of 64 bytes padded to 80 so the 8-byte fragment reads of a half-wave fall on distinct banks.
typedef long i64_t;
constexpr int F8_T = 128, F8_LD = 80;     // tile edge; LDS row stride in bytes

// (clamp_e4m3 / cvt4_fp8: common.h)

template <int MODE>   // 0 dense rows, 2 conv3x3 (any Cin % 8 == 0)
__global__ __launch_bounds__(NTHREADS, 2) void gemm_fp8w_kernel(GemmParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char As[2][F8_T * F8_LD];
    __shared__ __attribute__((aligned(16))) unsigned char Ws[2][F8_T * F8_LD];
    const int tiles_n = (p.N + F8_T - 1) / F8_T;
    const int nblk = tiles_n * ((p.M + F8_T - 1) / F8_T);
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int m0 = (bid / tiles_n) * F8_T, n0 = (bid % tiles_n) * F8_T;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.A), 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.W), 0, (int)p.w_bytes, 0x00020000);
    const unsigned char* W8 = reinterpret_cast<const unsigned char*>(p.W);
    (void)W8;
    // A loader: 4 chunks of 8 bf16 per thread: chunk c = tid & 7 (k = c*8), rows r = (tid >> 3) + 32 i
    const int kc = tid & 7, r0 = tid >> 3;
    const int Hs = p.ups ? (p.H >> 1) : p.H, Wsrc = p.ups ? (p.Wd >> 1) : p.Wd;
    unsigned abase[4];
    int ahi[4], awi[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int m = m0 + r0 + 32 * i;
        const bool ok = m < p.M;
        if (MODE == 0) { abase[i] = ok ? (unsigned)m * (unsigned)p.lda * 2u : OOB; ahi[i] = awi[i] = 0; }
        else {
            const int mm = ok ? m : 0, hw = p.Ho * p.Wo;
            const int b = mm / hw, rem = mm - b * hw, ho = rem / p.Wo, wo = rem - ho * p.Wo;
            abase[i] = (unsigned)b * (unsigned)(Hs * Wsrc) * (unsigned)p.lda * 2u;
            ahi[i] = ok ? ho * p.stride - 1 : -100000;
            awi[i] = wo * p.stride - 1;
        }
    }
    // W loader: fp8 rows of K bytes: 2 chunks of 16 bytes per thread: chunk c = tid & 3 (k = c*16), rows (tid >> 2) + 64 i
    const int wc = tid & 3, wr0 = tid >> 2;
    bf16x8 areg[4];
    u32x4 wreg[2];
    auto load_tile = [&](int kt) {
        const int k = kt * BK + kc * 8;
        const bool kok = k < p.K;
        int dh = 0, dw = 0, c = k;
        if (MODE != 0) { const int tap = k / p.Cin; c = k - tap * p.Cin; dh = tap / 3; dw = tap - dh * 3; }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            unsigned off = OOB;
            if (kok && abase[i] != OOB) {
                if (MODE == 0) off = abase[i] + (unsigned)k * 2u;
                else {
                    int hi = ahi[i] + dh, wi = awi[i] + dw;
                    const bool ok = hi >= 0 && hi < p.H && wi >= 0 && wi < p.Wd;
                    if (p.ups) { hi >>= 1; wi >>= 1; }
                    if (ok) off = abase[i] + ((unsigned)(hi * Wsrc + wi) * (unsigned)p.lda + (unsigned)c) * 2u;
                }
            }
            areg[i] = buf_ld8(ra, off);
        }
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int n = n0 + wr0 + 64 * i, kb = kt * BK + wc * 16;
            wreg[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, (n < p.N && kb < p.K) ? (unsigned)n * (unsigned)p.K + (unsigned)kb : OOB, 0, 0));
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const bf16x8 v = areg[i];
            u32x2 q = {cvt4_fp8(bf2f(v[0]), bf2f(v[1]), bf2f(v[2]), bf2f(v[3])), cvt4_fp8(bf2f(v[4]), bf2f(v[5]), bf2f(v[6]), bf2f(v[7]))};
            *reinterpret_cast<u32x2*>(&As[buf][(r0 + 32 * i) * F8_LD + kc * 8]) = q;
        }
#pragma unroll
        for (int i = 0; i < 2; i++) *reinterpret_cast<u32x4*>(&Ws[buf][(wr0 + 64 * i) * F8_LD + wc * 16]) = wreg[i];
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int wrow[4];
#pragma unroll
    for (int ni = 0; ni < 4; ni++) wrow[ni] = wn0 + 32 * (ni >> 1) + (li >> 2) * 8 + (ni & 1) * 4 + (li & 3);
    const int nk = (p.K + BK - 1) / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nk; kt++) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            i64_t fa[4], fw[4];
#pragma unroll
            for (int mi = 0; mi < 4; mi++) fa[mi] = *reinterpret_cast<const i64_t*>(&As[buf][(wm0 + mi * 16 + li) * F8_LD + kk * 32 + lg * 8]);
#pragma unroll
            for (int ni = 0; ni < 4; ni++) fw[ni] = *reinterpret_cast<const i64_t*>(&Ws[buf][wrow[ni] * F8_LD + kk * 32 + lg * 8]);
#pragma unroll
            for (int ni = 0; ni < 4; ni++)
#pragma unroll
                for (int mi = 0; mi < 4; mi++)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(fw[ni], fa[mi], acc[ni][mi], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }
    // per-output-channel dequantisation scales, applied in place (lane (lg, li): channels n .. n+7 of each tile pair)
#pragma unroll
    for (int pr = 0; pr < 2; pr++) {
        const int n = n0 + wn0 + 32 * pr + lg * 8;
        float sc[8];
#pragma unroll
        for (int e = 0; e < 8; e++) sc[e] = (n + e < p.N) ? p.wscale[n + e] : 0.f;
#pragma unroll
        for (int mi = 0; mi < 4; mi++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                acc[2 * pr][mi][r] *= sc[r];
                acc[2 * pr + 1][mi][r] *= sc[4 + r];
            }
    }
    EpiPre<4> pre;
    epilogue_prefetch<4>(p, pre, n0 + wn0, lg);
    gemm_epilogue<4, 4>(p, acc, m0 + wm0, n0 + wn0, li, lg, pre);
}

// per-row e4m3 quantisation of a bf16 matrix [rows][cols] (cols % 8 == 0): scale[r] = max|x| / 448 (1 if the row is zero),
// q = cvt_fp8(x / scale).  One wave per row.
__global__ __launch_bounds__(256) void quantize_fp8_rows_kernel(const bf16* __restrict__ src, unsigned char* __restrict__ dst,
                                                                float* __restrict__ scale, int rows, int cols) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const bf16* s = src + (size_t)row * cols;
    float amax = 0.f;
    for (int c = lane * 8; c < cols; c += 512) {
        const bf16x8 v = ld8(s + c);
#pragma unroll
        for (int e = 0; e < 8; e++) amax = fmaxf(amax, fabsf(bf2f(v[e])));
    }
    amax = wave_max(amax);
    const float sc = amax > 0.f ? amax / 448.f : 1.f;
    const float inv = 1.f / sc;
    if (lane == 0) scale[row] = sc;
    for (int c = lane * 8; c < cols; c += 512) {
        const bf16x8 v = ld8(s + c);
        u32x2 q = {cvt4_fp8(bf2f(v[0]) * inv, bf2f(v[1]) * inv, bf2f(v[2]) * inv, bf2f(v[3]) * inv),
                   cvt4_fp8(bf2f(v[4]) * inv, bf2f(v[5]) * inv, bf2f(v[6]) * inv, bf2f(v[7]) * inv)};
        *reinterpret_cast<u32x2*>(dst + (size_t)row * cols + c) = q;
    }
}

// epilogue of a split-K GEMM: C = act(alpha * sum_s slab_s + bias + rowvec + res)
__global__ void gemm_finish_kernel(GemmParams p, int nsp) {      // nsp: number of K splits (slabs)
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= (size_t)p.M * p.N) return;
    const int m = (int)(i / p.N), n = (int)(i - (size_t)m * p.N);      // N % 4 == 0 on this path
    const size_t slab = (size_t)p.M * p.N;
    f32x4 v = *reinterpret_cast<const f32x4*>(p.ws + i);
    int sidx = 1;
    for (; sidx + 3 < nsp; sidx += 4) {                   // four independent loads in flight, fixed summation order
        const f32x4 a = *reinterpret_cast<const f32x4*>(p.ws + (size_t)sidx * slab + i);
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.ws + (size_t)(sidx + 1) * slab + i);
        const f32x4 c = *reinterpret_cast<const f32x4*>(p.ws + (size_t)(sidx + 2) * slab + i);
        const f32x4 d = *reinterpret_cast<const f32x4*>(p.ws + (size_t)(sidx + 3) * slab + i);
        v += a; v += b; v += c; v += d;
    }
    for (; sidx < nsp; sidx++) v += *reinterpret_cast<const f32x4*>(p.ws + sidx * slab + i);
    const float* rv = p.rowvec ? p.rowvec + (size_t)(m / p.rows_per_batch) * p.ldrv : nullptr;
    const float* bias = (p.Mg && m >= p.Mg) ? p.bias1 : p.bias;      // grouped launch: the row's set
#pragma unroll
    for (int e = 0; e < 4; e++) {
        float x = v[e] * p.alpha;
        if (bias) x += bias[n + e];
        if (rv) x += rv[n + e];
        if (p.res) x += bf2f(p.res[(size_t)m * p.ldres + n + e]);
        if (p.flags & F_SILU) x = silu_f(x);
        v[e] = x;
    }
    if (p.flags & F_OUT_F32) {
        float* c = reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n;
        if (p.flags & F_ACCUM) { for (int e = 0; e < 4; e++) c[e] += v[e]; }
        else *reinterpret_cast<f32x4*>(c) = v;
    } else {
        bf16x4 o = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
        *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(p.C) + (size_t)m * p.ldc + n) = o;
    }
}

// Split-K / pixel-split scratch.  The default workspace (sidlsg_set_workspace) serves every stream that has no workspace
// of its own; a stream that runs contractions CONCURRENTLY with another one (the teacher evaluated beside the fake-score
// network, sid_step.py) registers a private one with sidlsg_set_stream_workspace, so the two never share slabs.
struct WsSlot { hipStream_t stream; float* ptr; long long bytes; };
static float* g_ws_default = nullptr;
static long long g_ws_default_bytes = 0;
static WsSlot g_ws_slots[8] = {};
static int g_ws_nslots = 0;
struct Ws { float* ptr; long long bytes; };
static Ws ws_for(hipStream_t s) {
    for (int i = 0; i < g_ws_nslots; i++)
        if (g_ws_slots[i].stream == s) return {g_ws_slots[i].ptr, g_ws_slots[i].bytes};
    return {g_ws_default, g_ws_default_bytes};
}

// ---------------------------------------------------------------------------------------------
// Direct-to-LDS (buffer_load ... lds) kernels.  (A 256 x 160 / 8-wave / 3-stage-ring variant "v2" and a
// 256 x 160 / one-wave-per-SIMD / 32x32x16-MFMA variant "v5" were built and measured -- both correct, both slower
// than v3 on every SD shape; they live in the git history, not in the build.)
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// buffer_load_dwordx4 ... lds as inline assembly (64 lanes x 16 bytes -> LDS at `dst` + 16 * lane; dst wave-uniform).
// Measurement switch (-DSIDLSG_WGRAD_ASM_DMA=1), off in the build.  Background: the compiler's waitcnt pass tracks every
// LDS-DMA issued through __builtin_amdgcn_raw_ptr_buffer_load_lds and puts s_waitcnt vmcnt(0) in front of the next
// ds_read_b64_tr_b16 (always) or aliasing plain LDS read -- in the weight-gradient kernel that is the kk=1 fragment reads of
// the NEXT stage, half a stage after the DMA was issued, so the prefetch distance is half of what the schedule intends (seen
// in the ISA).  With the DMA invisible to the compiler the only wait is the explicit one at the publish barrier: measured
// +2-4 % on the dense weight gradients, -1-2 % on the conv ones (the asm statements pin the address arithmetic between them)
// -- like the early-DMA experiment in gemm_v3_kernel, more prefetch distance buys nothing.  (M0 is reserved: the compiler
// keeps nothing live in it.)
DEVFN void dma16_asm(__amdgpu_buffer_rsrc_t rs, const void* dst, unsigned voff, unsigned soff) {
    const unsigned la = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lptr_t)dst);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(la), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
#ifndef SIDLSG_MX8_SPREAD_DMA
#define SIDLSG_MX8_SPREAD_DMA 1
#endif
#ifndef SIDLSG_WGRAD_ASM_DMA
#define SIDLSG_WGRAD_ASM_DMA 0
#endif

// ---------------------------------------------------------------------------------------------
// "v3": the 128 x 160 x 64 tile / 4 waves / 2 blocks per CU structure of gemm_bf16_kernel, but tiles are staged by
// direct-to-LDS loads (global_load_lds_dwordx4 + zero page, source-side swizzle as in v2): the ds_write commit phase --
// measured at ~25 % of the kernel and not overlapped with MFMA -- and the staging registers disappear.
// Schedule per K-tile (2 LDS buffers, one raw barrier):
//   A: issue the kk=1 fragment reads of the current buffer; kk=0 MFMAs (fragments were prefetched in the previous D)
//   C: s_waitcnt vmcnt(0) (this wave's DMA of tile kt+1 has landed) ; s_barrier
//   D: issue the kk=0 fragment reads of tile kt+1 and the DMA of tile kt+2 into the buffer just vacated
//   E: kk=1 MFMAs (cover D's latencies)
// W-tile LDS swizzle (16-byte chunk index ^= wsw(row)).  A ds_read_b128 is serviced in the lane groups
// {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... (MI355X guide, LDS table), and the W fragment rows of a paired (ni, ni+1)
// column block are li -> 8*(li>>2) + (li&3): with the plain (row>>1)&7 swizzle the lg=0 and lg=1 lanes of one group
// collide 2-way (measured: SQ_LDS_BANK_CONFLICT = 31 % of SQ_LDS_IDX_ACTIVE).  For those rows use
// (q&1) | ((q>>2)&3)<<1 with q = row>>1, which makes the 16 lanes of every group hit 16 distinct 16-byte slots; the
// trailing unpaired 16-row block (rows 64..79 of each wave's 80) keeps the plain swizzle.
DEVFN int wsw(int r) {
    const int rr = r >= 80 ? r - 80 : r;
    const int q = r >> 1;
    return rr < 64 ? ((q & 1) | (((q >> 2) & 3) << 1)) : (q & 7);
}

#ifdef SIDLSG_EXP_TRACE           // measurement build (tools/ab/gemm_trace.py): per-block phase timestamps, 100 MHz wall clock
__device__ unsigned long long* g_trace = nullptr;
extern "C" int sidlsg_exp_set_trace(void* ptr) { return hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &ptr, sizeof(ptr)) == hipSuccess ? 0 : -1; }
#define TRACE(i) do { if (g_trace && threadIdx.x == 0) g_trace[(size_t)blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#define TRACE_HWID() do { if (g_trace && threadIdx.x == 0) { unsigned h, x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h)); asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); g_trace[(size_t)blockIdx.x * 8 + 7] = ((unsigned long long)x << 32) | h; } } while (0)
#define WTRACE(i) do { if (g_trace && threadIdx.x == 0) g_trace[(size_t)blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#else
#define TRACE(i)
#define TRACE_HWID()
#define WTRACE(i)
#endif
#ifndef SIDLSG_CONV_TAP_INNER
#define SIDLSG_CONV_TAP_INNER 1
#endif
#ifndef SIDLSG_V3_SCHED_FENCE
#define SIDLSG_V3_SCHED_FENCE 1
#endif
#ifndef SIDLSG_V3_PRIO_PIN
#define SIDLSG_V3_PRIO_PIN 1   // 1: sched_barriers pin the MFMAs and scalar instructions relative to the flip (VALU / VMEM / LDS may cross)
#endif
#define V3_SETPRIO(v) do { if (SIDLSG_V3_PRIO_PIN) __builtin_amdgcn_sched_barrier(0x3F2); __builtin_amdgcn_s_setprio(v); if (SIDLSG_V3_PRIO_PIN) __builtin_amdgcn_sched_barrier(0x3F2); } while (0)
#ifndef SIDLSG_V3_PRIO
#define SIDLSG_V3_PRIO 0      // A/B knob (bit mask): s_setprio(1) around the MFMA clusters of a K-tile -- 1: gemm_v3 (dense / conv fwd + dgrad), 2: wgrad_v2 (two co-resident blocks per CU in different phases)
#endif
// (A 3-buffer ring with counted vmcnt(9) -- every DMA gets two K-tiles of MFMA time, but 108 KiB of LDS = ONE block per CU
// -- was measured in the same session: 30-50 % SLOWER on every SD shape (e.g. conv 64x64 320->320 124 -> 195 us, FF-in 239 ->
// 355 us).  Two co-resident blocks per CU hide more than a deeper pipeline in one; the variant is not in the build.)
template <int MODE, int STAGES>   // MODE 0 dense, 1 conv3x3 with Cin % 64 == 0
DEVFN void gemm_v3_body(GemmParams& p) {
    constexpr int BM = 128, BN = 160, MT = 4, NT = 5;
    constexpr bool SCHED_FENCE = SIDLSG_V3_SCHED_FENCE;
    constexpr int STAGE = (BM + BN) * BK;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* ring = reinterpret_cast<bf16*>(smem);

    // Work item = (output tile, K split), 1-D grid.  Hardware block b runs on XCD b % 8; the remap gives every XCD a
    // contiguous range of logical items.  Logical order: split-major, and inside a split the tiles in groups of 8
    // m-tiles (m fastest, then n): the ~64 blocks resident on an XCD then form an ~8 x 8 patch of output tiles and
    // fetch 8 A panels + 8 W panels into that XCD's L2.  (Row-major strips fetched 1 A panel + 64 W panels for FF-in
    // at 16x16 -- every XCD streamed the whole weight matrix; measured 154 -> 130 us on 4096x10240x1280 and 90 -> 72
    // us on the 8x8 2560->1280 conv.)
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = m_tiles_rt(p, BM);
#ifdef SIDLSG_EXP_K1             // measurement build: one K-tile per output tile (prologue + epilogue cost)
    const int nk_all = 1;
#else
    const int nk_all = (p.K + BK - 1) / BK;
#endif
    const int nsplit = p.kt_per_split ? (nk_all + p.kt_per_split - 1) / p.kt_per_split : 1;
    const int ntile = tiles_n * tiles_m;
    int bid = blockIdx.x;
    {
        const int nblk = ntile * nsplit;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int split = bid / ntile;
    int mt, nt;
    {
        const int t = bid - split * ntile;
        const int per_group = p.group_m * tiles_n;
        const int g = t / per_group, first_m = g * p.group_m;
        const int gm = min(p.group_m, tiles_m - first_m);       // last group may be short
        const int in_g = t - g * per_group;
        mt = first_m + in_g % gm;
        nt = in_g / gm;
    }
    const int m0 = group_select<BM>(p, mt);
    const int n0 = nt * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 80;
    const int li = lane & 15, lg = lane >> 4;
    const int lrow = lane >> 3, lslot = lane & 7;

    // first output column of this wave's half of the tile (fused GEGLU: the a half / the gate half live F columns apart)
    const int wcol0 = p.geglu ? (wn0 ? p.geglu + nt * 80 : nt * 80) : n0 + wn0;

    // Loader: buffer_load_dwordx4 ... lds.  Per row ONE 32-bit byte offset (VGPR) that already contains the lane's
    // swizzled chunk; the K position of the tile is the wave-uniform soffset.  Invalid rows (m >= M, n >= N, conv halo)
    // carry an out-of-range offset and the buffer unit writes zeros: no pointer arithmetic, selects or zero page in the
    // steady state (the 64-bit global_load_lds version issued ~150 VALU+SALU per 40 MFMAs).
    const int Hs = p.ups ? (p.H >> 1) : p.H, Ws = p.ups ? (p.Wd >> 1) : p.Wd;
    // conv, K-tile order "channel chunk outer, tap inner" (`lin`): the 9 taps of a chunk re-read the same 3 image rows shifted
    // by a pixel; visited back to back the re-reads hit the XCD's L2 (FETCH_SIZE 291 -> 123 MB per launch at 64x64x320 -> 320,
    // 713 -> 143 MB at 640 channels; with the tap-major order the reuse distance was Cin * 256 B = 80-330 KB per block, times
    // ~64 resident blocks per XCD = 5-20 MB against 4 MB of L2).  The weight matrix stays tap-major, only the visiting order
    // changes.  Addressing: the byte offset of tap (dh, dw) is LINEAR in (dh, dw) with wave-uniform strides, so the per-row
    // VGPR holds the CENTRE pixel's offset, the tap goes into the scalar soffset (against a buffer base moved one row + one
    // pixel below A, so soffset >= 0), and the halo is a 9-bit validity mask per row: 3 VALU per row and K-tile instead of a
    // recomputation of the row offsets (which cost 5 % when done per K-tile).  Not with the fused nearest x2 upsampling
    // ((h + dh) >> 1 is not linear): that keeps the tap-major order.
    const unsigned tap_rs = (unsigned)Ws * (unsigned)p.lda * 2u, tap_ps = (unsigned)p.lda * 2u;
    const bool lin = MODE == 1 && SIDLSG_CONV_TAP_INNER && !p.ups && (unsigned long long)p.a_bytes + tap_rs + tap_ps < 0x80000000ull;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16*>(p.A) - (lin ? (size_t)(tap_rs + tap_ps) / 2 : 0), 0, (int)(p.a_bytes + (lin ? tap_rs + tap_ps : 0u)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.W), 0, (int)p.w_bytes, 0x00020000);
    unsigned abase[4], aoff[4];
    int ahi[4], awi[4];
    unsigned cen[4], tmask[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int r = wave * 32 + j * 8 + lrow;
        const int kcs = lslot ^ ((r >> 1) & 7);
        const int m = m0 + r;
        const bool ok = m < p.M;
        if (MODE == 0) {
            aoff[j] = ok ? ((unsigned)m * (unsigned)p.lda + kcs * 8) * 2u : OOB;
            abase[j] = 0; ahi[j] = awi[j] = 0;
        } else {
            const int mm = ok ? m : 0;
            const int hw = p.Ho * p.Wo;
            const int b = mm / hw, rem = mm - b * hw;
            const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
            abase[j] = ((unsigned)(b * Hs * Ws) * (unsigned)p.lda + kcs * 8) * 2u;
            ahi[j] = ok ? ho * p.stride - 1 : -100000;
            awi[j] = wo * p.stride - 1;
            aoff[j] = OOB;
            cen[j] = abase[j] + (unsigned)((ahi[j] + 1) * Ws + awi[j] + 1) * tap_ps;
            unsigned mk = 0;
#pragma unroll
            for (int tp = 0; tp < 9; tp++) {
                const int hi = ahi[j] + tp / 3, wi = awi[j] + tp % 3;
                if (hi >= 0 && hi < p.H && wi >= 0 && wi < p.Wd) mk |= 1u << tp;
            }
            tmask[j] = mk;
        }
    }
    if (MODE == 0) {
#pragma unroll
        for (int j = 0; j < 4; j++) cen[j] = tmask[j] = 0;
    }
    unsigned boff[5];
#pragma unroll
    for (int j = 0; j < 5; j++) {
        const int g = wave + 4 * j;                 // 20 groups of 8 weight rows, 5 per wave
        const int r = g * 8 + lrow;
        const int kcs = lslot ^ wsw(r);
        const int n = p.geglu ? (r < 80 ? nt * 80 + r : p.geglu + nt * 80 + (r - 80)) : n0 + r;      // fused GEGLU: a rows | gate rows
        boff[j] = n < p.N ? ((unsigned)n * (unsigned)p.K + kcs * 8) * 2u : OOB;
    }
    const int kt_begin = p.kt_per_split ? split * p.kt_per_split : 0;
    const int nk = p.kt_per_split ? min(nk_all, kt_begin + p.kt_per_split) : nk_all;
    int cur_tap = -1;

    auto issue = [&](int t, int buf) {          // 9 buffer_load ... lds per wave
        bf16* sa = ring + buf * STAGE;
        bf16* sb = sa + BM * BK;
        int k0 = t * BK;
        int asoff = k0 * 2;
        if (MODE == 1) {
            if (lin) {
                const int cc = t / 9;
                const int tap = t - cc * 9;
                const int dh = tap / 3, dw = tap - dh * 3;
                asoff = cc * BK * 2 + dh * (int)tap_rs + dw * (int)tap_ps;
                k0 = tap * p.Cin + cc * BK;
                const unsigned bit = 1u << tap;
#pragma unroll
                for (int j = 0; j < 4; j++) aoff[j] = (tmask[j] & bit) ? cen[j] : OOB;
                cur_tap = tap;
            }
            const int tap = lin ? cur_tap : k0 / p.Cin;
            if (!lin) asoff = (k0 - tap * p.Cin) * 2;
            if (tap != cur_tap) {               // every Cin/64 tiles: new tap -> new row offsets / halo mask
                cur_tap = tap;
                const int dh = tap / 3, dw = tap - dh * 3;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    int hi = ahi[j] + dh, wi = awi[j] + dw;
                    const bool v = hi >= 0 && hi < p.H && wi >= 0 && wi < p.Wd;
                    if (p.ups) { hi >>= 1; wi >>= 1; }
                    aoff[j] = v ? abase[j] + (unsigned)(hi * Ws + wi) * (unsigned)p.lda * 2u : OOB;
                }
            }
        }
        if (MODE == 0 && k0 + BK > p.K) {       // ragged K tail (dense only): chunks past K must read zeros
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int r = wave * 32 + j * 8 + lrow;
                const int kcs = lslot ^ ((r >> 1) & 7);
                const unsigned o = (k0 + kcs * 8 < p.K) ? aoff[j] : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lptr_t)(sa + (wave * 32 + j * 8) * BK), 16, o, asoff, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 5; j++) {
                const int g = wave + 4 * j;
                const int kcs = lslot ^ wsw(g * 8 + lrow);
                const unsigned o = (k0 + kcs * 8 < p.K) ? boff[j] : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lptr_t)(sb + g * 8 * BK), 16, o, k0 * 2, 0, 0);
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < 4; j++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lptr_t)(sa + (wave * 32 + j * 8) * BK), 16, aoff[j], asoff, 0, 0);
#pragma unroll
        for (int j = 0; j < 5; j++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lptr_t)(sb + (wave + 4 * j) * 8 * BK), 16, boff[j], k0 * 2, 0, 0);
    };

    f32x4 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; i++)
#pragma unroll
        for (int j = 0; j < MT; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int wrow[NT];
#pragma unroll
    for (int ni = 0; ni < NT; ni++) {
        const bool paired = (ni | 1) < NT;
        wrow[ni] = wn0 + (paired ? 32 * (ni >> 1) + (li >> 2) * 8 + (ni & 1) * 4 + (li & 3) : 16 * ni + li);
    }
    auto read_frags = [&](int buf, int kk, bf16x8 (&fa)[MT], bf16x8 (&fw)[NT]) {
        const bf16* a = ring + buf * STAGE;
        const bf16* b = a + BM * BK;
        const int ch = kk * 4 + lg;
#pragma unroll
        for (int mi = 0; mi < MT; mi++) {
            const int r = wm0 + mi * 16 + li;
            fa[mi] = *reinterpret_cast<const bf16x8*>(a + r * BK + ((ch ^ ((r >> 1) & 7)) << 3));
        }
#pragma unroll
        for (int ni = 0; ni < NT; ni++) {
            const int r = wrow[ni];
            fw[ni] = *reinterpret_cast<const bf16x8*>(b + r * BK + ((ch ^ wsw(r)) << 3));
        }
    };
    auto mfma_block = [&](const bf16x8 (&fa)[MT], const bf16x8 (&fw)[NT]) {
        if constexpr ((SIDLSG_V3_PRIO) & 1) V3_SETPRIO(1);
#pragma unroll
        for (int ni = 0; ni < NT; ni++)
#pragma unroll
            for (int mi = 0; mi < MT; mi++)
                acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[ni], fa[mi], acc[ni][mi], 0, 0, 0);
        if constexpr ((SIDLSG_V3_PRIO) & 1) V3_SETPRIO(0);
    };

    bf16x8 fa0[MT], fw0[NT], fa1[MT], fw1[NT];
    EpiPre<NT> pre;
    if constexpr (STAGES == 2) {
    TRACE(0); TRACE_HWID();
    issue(kt_begin, 0);
    if (!p.kt_per_split) epilogue_prefetch<NT>(p, pre, wcol0, lg);        // behind the first tile's DMA, used after the loop
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    TRACE(1);
    read_frags(0, 0, fa0, fw0);
#ifdef SIDLSG_EXP_NOLDS
    read_frags(0, 1, fa1, fw1);
#endif
    if (kt_begin + 1 < nk) issue(kt_begin + 1, 1);
    // One K-tile.  The steady state, the last-but-one tile (nothing left to fetch) and the last tile are separate
    // STRAIGHT-LINE instantiations: with `if (more)` / `if (kt + 2 < nk)` inside one loop body the waitcnt pass has to merge
    // the paths and makes the kk=1 MFMAs of E wait for the fragment reads D has just issued (lgkmcnt(4..0) in the ISA instead
    // of ~9), i.e. the prefetch of the next tile's fragments was serialised in front of the MFMAs meant to cover it.
    auto ktile = [&](const int kt, auto has_next, auto fetch) {
        const int buf = (kt - kt_begin) & 1;
#ifndef SIDLSG_EXP_NOLDS
        read_frags(buf, 1, fa1, fw1);                          // A
#endif
        mfma_block(fa0, fw0);
        // The MFMAs are register-only, so neither the "memory" clobber nor the barrier orders them: without this fence the
        // scheduler hoists the wait + barrier to right behind the FIRST kk=0 MFMA (seen in the ISA), which halves the MFMA
        // time covering the DMA of tile kt+1 (issued in the previous D) and serialises the wait in front of 19 MFMAs.
        if (SCHED_FENCE) __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // C: my share of tile kt+1 is in LDS
#ifndef SIDLSG_EXP_NOBAR
        __builtin_amdgcn_s_barrier();
#endif
        asm volatile("" ::: "memory");
        if (SCHED_FENCE) __builtin_amdgcn_sched_barrier(0);
        if constexpr (decltype(has_next)::value) {             // D
#ifndef SIDLSG_EXP_NOLDS
            read_frags(buf ^ 1, 0, fa0, fw0);
#endif
#ifndef SIDLSG_EXP_NODMA
            if constexpr (decltype(fetch)::value) issue(kt + 2, buf);
#endif
        }
        mfma_block(fa1, fw1);                                  // E
    };
    {
        int kt = kt_begin;
        for (; kt + 2 < nk; kt++) ktile(kt, std::true_type{}, std::true_type{});
        if (kt + 1 < nk) { ktile(kt, std::true_type{}, std::false_type{}); kt++; }
        ktile(kt, std::false_type{}, std::false_type{});
    }

    }

    if (p.kt_per_split) {
#pragma unroll
        for (int mi = 0; mi < MT; mi++) {
            const int m = m0 + wm0 + mi * 16 + li;
            if (m >= p.M) continue;
#pragma unroll
            for (int ni = 0; ni < NT; ni++) {
                const bool paired = (ni | 1) < NT;
                const int nb = n0 + wn0 + (paired ? 32 * (ni >> 1) + lg * 8 + (ni & 1) * 4 : 16 * ni + lg * 4);
                float* dst = p.ws + ((size_t)split * p.Mtot + m) * p.N + nb;
                if (nb + 4 <= p.N) *reinterpret_cast<f32x4*>(dst) = acc[ni][mi];
                else
                    for (int r = 0; r < 4; r++)
                        if (nb + r < p.N) dst[r] = acc[ni][mi][r];
            }
        }
        return;
    }
#ifndef SIDLSG_EXP_DIRECT_STORE
    if (!(p.flags & (F_OUT_F32 | F_ACCUM)) && (p.N & 7) == 0) {
        constexpr int LDR = BN + 8;             // 336-byte rows: the 16 rows of a b128 write phase fall on distinct banks
        __syncthreads();                        // every wave is done with the last K-tile: the ring becomes the tile image
        TRACE(2);
        gemm_epilogue<MT, NT>(p, acc, m0 + wm0, wcol0, li, lg, pre, ring, m0, wcol0 - wn0, LDR);
        __syncthreads();
        TRACE(3);
        if (MODE == 0 && p.geglu) {       // (dense kernel only: the conv instantiation does not carry this code)
            // the staged tile holds h[.., 0:80) = a and h[.., 80:160) = gate of features [80 nt, +80): write both (natural
            // columns; skipped when the caller keeps no h: a pass without backward) and y = a * gelu(gate) -- from the bf16-ROUNDED
            // values, i.e. bit for bit what sidlsg_geglu_fwd computes from the stored h
            constexpr int CPR = 10;                 // 16-byte chunks of the 80 features
            bf16* H = reinterpret_cast<bf16*>(p.C);
#pragma unroll
            for (int i = 0; i < (BM * CPR + NTHREADS - 1) / NTHREADS; i++) {
                const int c = tid + i * NTHREADS;
                const int row = c / CPR, col = (c - row * CPR) * 8;
                const int m = m0 + row;
                if (row < BM && m < p.M) {
                    const bf16x8 av = *reinterpret_cast<const bf16x8*>(ring + row * LDR + col);
                    const bf16x8 gv = *reinterpret_cast<const bf16x8*>(ring + row * LDR + 80 + col);
                    const int f = nt * 80 + col;
                    if (H) {
                        st8(H + (size_t)m * p.ldc + f, av);
                        st8(H + (size_t)m * p.ldc + p.geglu + f, gv);
                    }
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; e++) o[e] = f2bf(bf2f(av[e]) * gelu_t<bf16>(bf2f(gv[e])));
                    st8(p.Y + (size_t)m * p.ldy + f, o);
                }
            }
            return;
        }
#ifndef SIDLSG_EXP_NO_GEGLU_BWD       // (measurement build: the kernel without this epilogue, to price its code size)
        if (MODE == 0 && p.geglu_bwd) {       // (dense kernel only)
            // the staged tile holds the bf16-ROUNDED dy of 160 features: bit for bit what sidlsg_geglu_bwd computes from a stored dy.
            // Every h load of a half is issued before that half's first store (Hin / dH do not alias: __restrict__ locals).
            constexpr int CPR = BN / 8;                             // 16-byte chunks per tile row
            constexpr int ITER = BM * CPR / NTHREADS / 2;           // chunks per thread and half
            static_assert(BM * CPR % (2 * NTHREADS) == 0, "whole passes");
            const int F = p.geglu_bwd;
            const bf16* __restrict__ Hin = p.Hin;
            bf16* __restrict__ dH = p.Y;
#pragma unroll
            for (int half = 0; half < 2; half++) {
                bf16x8 av[ITER], gv[ITER];
#pragma unroll
                for (int i = 0; i < ITER; i++) {
                    const int c = tid + (half * ITER + i) * NTHREADS;
                    const int row = c / CPR, col = (c - row * CPR) * 8;
                    const bool ok = m0 + row < p.M && n0 + col < p.N;
                    const size_t o = (size_t)(m0 + row) * p.ldy + n0 + col;
                    av[i] = ok ? ld8(Hin + o) : zero8();
                    gv[i] = ok ? ld8(Hin + o + F) : zero8();
                }
#pragma unroll
                for (int i = 0; i < ITER; i++) {
                    const int c = tid + (half * ITER + i) * NTHREADS;
                    const int row = c / CPR, col = (c - row * CPR) * 8;
                    if (!(m0 + row < p.M && n0 + col < p.N)) continue;
                    const bf16x8 dv = *reinterpret_cast<const bf16x8*>(ring + row * LDR + col);
                    bf16x8 da, dg;
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        float gl, gd;
                        gelu_pair_t<bf16>(bf2f(gv[i][e]), gl, gd);
                        const float d = bf2f(dv[e]);
                        da[e] = f2bf(d * gl);
                        dg[e] = f2bf(d * bf2f(av[i][e]) * gd);
                    }
                    const size_t o = (size_t)(m0 + row) * p.ldy + n0 + col;
                    st8(dH + o, da);
                    st8(dH + o + F, dg);
                }
            }
            return;
        }
#endif
        gemm_store_rows<BM, BN>(p, ring, m0, n0, LDR, tid);
        TRACE(4);
#ifdef SIDLSG_EXP_TRACE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TRACE(5);
#endif
        return;
    }
    gemm_epilogue<MT, NT, false>(p, acc, m0 + wm0, n0 + wn0, li, lg, pre);
#else
    gemm_epilogue<MT, NT>(p, acc, m0 + wm0, n0 + wn0, li, lg, pre);
#endif
}

// (plain kernels over one body: a kernel template with a second non-type parameter got no host stub from hipcc 7.2)
template <int MODE>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_v3_kernel(GemmParams p) { gemm_v3_body<MODE, 2>(p); }

template <int MODE>
static int launch_gemm_v3(const GemmParams& p, hipStream_t s) {
    const int tiles = m_tiles_rt(p, 128) * ((p.N + 159) / 160);
    const size_t lds = (size_t)2 * (128 + 160) * BK * sizeof(bf16);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_v3_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (128 + 160) * BK * 2);
        attr_done = true;
    }
    const int nk = (p.K + BK - 1) / BK;
    const int splits = p.kt_per_split ? (nk + p.kt_per_split - 1) / p.kt_per_split : 1;
    GemmParams q = p;
    static const int group_env = getenv("SIDLSG_GEMM_GROUP_M") ? atoi(getenv("SIDLSG_GEMM_GROUP_M")) : 4;   // A/B switch (measured: 4 and 8 equivalent, 1 = row-major strips)
    q.group_m = group_env < 1 ? 1 : group_env;
    SIDLSG_LAUNCH((gemm_v3_kernel<MODE>), dim3(tiles * splits), dim3(NTHREADS), lds, s, q);
    if (p.kt_per_split) {
        const size_t n4 = ((size_t)p.M * p.N + 3) / 4;
        SIDLSG_LAUNCH(gemm_finish_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, p, splits);
    }
    return sidlsg_last_error();
}

#include "gemm_p8.h"

// ---------------------------------------------------------------------------------------------------------------------
// MX-fp8 dense GEMM for FROZEN networks:  C[m][n] = wscale[n] * sum_k A8[m][k] * W8[n][k]  (+ the shared epilogue)
// with BOTH operands OCP e4m3 bytes (A8: activations the producing GroupNorm / LayerNorm kernel wrote as e4m3 at unit
// scale; W8 + wscale: sidlsg_quantize_fp8_rows) on v_mfma_scale_f32_16x16x128_f8f6f4 with unit E8M0 block scales -- the only
// 8-bit MFMA form that runs at twice the bf16 rate on gfx950 (tools/ubench/mfma_f8.hip: 4.27 vs 1.83 PFLOP/s register-only;
// the non-scaled 16x16x32 fp8 form of gemm_fp8w_kernel has the bf16 instruction rate).
// A K-tile of 128 e4m3 elements is byte-identical to gemm_v3_kernel's 64 bf16 elements: same 128 x 160 tile, 128-byte LDS
// rows, direct-to-LDS loader with one 32-bit offset per row, the same conflict-free swizzles and the same one-barrier
// schedule.  The MFMA wants 32 consecutive K bytes per lane and operand; which 32 is free as long as both operands agree, so
// lane group lg takes the 16-byte chunks lg and 4 + lg of its row -- exactly the two ds_read_b128 patterns of the bf16 kernel
// (chunks 2 lg, 2 lg + 1 would be 2-way bank conflicted with these swizzles).  Schedule per K-tile, halves by output column:
//   A: reads of the W fragments ni = 2..4 of the current tile; MFMAs ni = 0..1    C: vmcnt(0); s_barrier
//   D: reads of the next tile's A fragments + W fragments ni = 0..1; DMA of tile kt+2    E: MFMAs ni = 2..4
// Requires N % 160 == 0, K % 16 == 0, lda % 16 == 0, bf16 or fp32 output through the shared epilogue.
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4v __attribute__((ext_vector_type(4)));

template <int MODE>      // 0: dense rows.  1: conv3x3, pad 1, stride 1, no upsampling, Cin % 16 == 0 (NHWC e4m3 image)
__global__ __launch_bounds__(NTHREADS, 2) void gemm_mx8_kernel(GemmParams p) {
    constexpr int BM = 128, BN = 160, MT = 4, NT = 5, KB = 128;      // KB: bytes (= e4m3 elements) per K-tile
    constexpr int STAGE = (BM + BN) * KB;                            // bytes
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned char* A8 = reinterpret_cast<const unsigned char*>(p.A);
    const unsigned char* W8 = reinterpret_cast<const unsigned char*>(p.W);
    const int tiles_n = p.N / BN, tiles_m = (p.M + BM - 1) / BM;
    const int ntile = tiles_n * tiles_m;
    const int nk_all = MODE == 1 ? 9 * ((p.Cin + KB - 1) / KB) : (p.K + KB - 1) / KB;
    const int nsplit = p.kt_per_split ? (nk_all + p.kt_per_split - 1) / p.kt_per_split : 1;
    int bid = blockIdx.x;
    {
        const int nblk = ntile * nsplit;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int split = bid / ntile;       // split-K (few-tile launches, see launch_gemm_mx8): fp32 slabs + gemm_finish_kernel
    int mt, nt;
    {
        const int t = bid - split * ntile;
        const int per_group = p.group_m * tiles_n;
        const int g = t / per_group, first_m = g * p.group_m;
        const int gm = min(p.group_m, tiles_m - first_m);
        const int in_g = t - g * per_group;
        mt = first_m + in_g % gm;
        nt = in_g / gm;
    }
    const int m0 = mt * BM, n0 = nt * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 80;
    const int li = lane & 15, lg = lane >> 4;
    const int lrow = lane >> 3, lslot = lane & 7;
    // conv: K-tiles in the order "channel chunk outer, tap inner" with the tap as a wave-uniform soffset against a buffer base one
    // row + one pixel below the image and a 9-bit halo mask per row (see gemm_v3_kernel).  A chunk is 128 channels of one tap;
    // when Cin is not a multiple of 128 (320: 128 + 128 + 64) the lanes whose 16 channels lie past Cin read zeros from A -- W's
    // over-read then meets zeros (it runs into the next tap's weights, finite numbers; past the matrix the buffer returns zeros).
    const unsigned tap_rs = (unsigned)p.Wd * (unsigned)p.lda, tap_ps = (unsigned)p.lda;
    const unsigned shift = MODE == 1 ? tap_rs + tap_ps : 0u;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(A8) - shift, 0, (int)(p.a_bytes + shift), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(W8), 0, (int)p.w_bytes, 0x00020000);
    unsigned aoff[4], boff[5];
    unsigned cen[4], tmask[4], akb[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int r = wave * 32 + j * 8 + lrow;
        const int kcs = lslot ^ ((r >> 1) & 7);
        const int m = m0 + r;
        aoff[j] = m < p.M ? (unsigned)m * (unsigned)p.lda + kcs * 16u : OOB;
        cen[j] = tmask[j] = akb[j] = 0;
        if (MODE == 1) {
            const bool ok = m < p.M;
            const int mm = ok ? m : 0, hw = p.H * p.Wd;
            const int b = mm / hw, rem = mm - b * hw;
            const int ho = rem / p.Wd, wo = rem - ho * p.Wd;
            cen[j] = (unsigned)((b * p.H + ho) * p.Wd + wo) * (unsigned)p.lda + kcs * 16u;
            akb[j] = kcs * 16u;
            unsigned mk = 0;
#pragma unroll
            for (int tp = 0; tp < 9; tp++) {
                const int hi = ho - 1 + tp / 3, wi = wo - 1 + tp % 3;
                if (ok && hi >= 0 && hi < p.H && wi >= 0 && wi < p.Wd) mk |= 1u << tp;
            }
            tmask[j] = mk;
            aoff[j] = OOB;
        }
    }
#pragma unroll
    for (int j = 0; j < 5; j++) {
        const int r = (wave + 4 * j) * 8 + lrow;
        const int kcs = lslot ^ wsw(r);
        boff[j] = (unsigned)(n0 + r) * (unsigned)p.K + kcs * 16u;       // N % 160 == 0: every row exists
    }
    const int kt_begin = p.kt_per_split ? split * p.kt_per_split : 0;
    const int nk = p.kt_per_split ? min(nk_all, kt_begin + p.kt_per_split) : nk_all;
    auto issue = [&](int t, int buf, const int part = 0) {          // 9 buffer_load ... lds per wave (part 1: the 4 A pieces, 2: the 5 W pieces)
        char* sa = smem + buf * STAGE;
        char* sb = sa + BM * KB;
        if (MODE == 1) {
            const int cc = t / 9, tap = t - cc * 9;
            const int dh = tap / 3, dw = tap - dh * 3;
            const unsigned sa_off = (unsigned)cc * KB + dh * tap_rs + dw * tap_ps;
            const unsigned sw_off = (unsigned)(tap * p.Cin + cc * KB);
            const unsigned lim = (unsigned)(p.Cin - cc * KB), bit = 1u << tap;
            if (part != 2) {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lptr_t)(sa + (wave * 32 + j * 8) * KB), 16, ((tmask[j] & bit) && akb[j] < lim) ? cen[j] : OOB, sa_off, 0, 0);
            }
            if (part != 1) {
#pragma unroll
                for (int j = 0; j < 5; j++) __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lptr_t)(sb + (wave + 4 * j) * 8 * KB), 16, boff[j], sw_off, 0, 0);
            }
            return;
        }
        const int k0 = t * KB;
        if (k0 + KB > p.K) {                    // ragged K tail: chunks past K read zeros
            if (part != 2) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int kcs = lslot ^ (((wave * 32 + j * 8 + lrow) >> 1) & 7);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lptr_t)(sa + (wave * 32 + j * 8) * KB), 16, (k0 + kcs * 16 < p.K) ? aoff[j] : OOB, k0, 0, 0);
                }
            }
            if (part != 1) {
#pragma unroll
                for (int j = 0; j < 5; j++) {
                    const int kcs = lslot ^ wsw((wave + 4 * j) * 8 + lrow);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lptr_t)(sb + (wave + 4 * j) * 8 * KB), 16, (k0 + kcs * 16 < p.K) ? boff[j] : OOB, k0, 0, 0);
                }
            }
            return;
        }
        if (part != 2) {
#pragma unroll
            for (int j = 0; j < 4; j++) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lptr_t)(sa + (wave * 32 + j * 8) * KB), 16, aoff[j], k0, 0, 0);
        }
        if (part != 1) {
#pragma unroll
            for (int j = 0; j < 5; j++) __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lptr_t)(sb + (wave + 4 * j) * 8 * KB), 16, boff[j], k0, 0, 0);
        }
    };
    f32x4 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; i++)
#pragma unroll
        for (int j = 0; j < MT; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int wrow[NT];
#pragma unroll
    for (int ni = 0; ni < NT; ni++) {
        const bool paired = (ni | 1) < NT;
        wrow[ni] = wn0 + (paired ? 32 * (ni >> 1) + (li >> 2) * 8 + (ni & 1) * 4 + (li & 3) : 16 * ni + li);
    }
    auto frag = [&](const char* rowp, int sw) -> i32x8 {     // chunks lg and 4 + lg of the row (swizzled positions)
        const i32x4v lo = *reinterpret_cast<const i32x4v*>(rowp + ((lg ^ sw) << 4));
        const i32x4v hi = *reinterpret_cast<const i32x4v*>(rowp + (((4 + lg) ^ sw) << 4));
        return (i32x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    auto read_a = [&](int buf, i32x8 (&fa)[MT]) {
        const char* a = smem + buf * STAGE;
#pragma unroll
        for (int mi = 0; mi < MT; mi++) {
            const int r = wm0 + mi * 16 + li;
            fa[mi] = frag(a + r * KB, (r >> 1) & 7);
        }
    };
    auto read_w = [&](int buf, int ni) -> i32x8 {
        const char* b = smem + buf * STAGE + BM * KB;
        const int r = wrow[ni];
        return frag(b + r * KB, wsw(r));
    };
    auto mfma_col = [&](const i32x8 (&fa)[MT], const i32x8& fw, int ni) {
#pragma unroll
        for (int mi = 0; mi < MT; mi++)
            acc[ni][mi] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fw, fa[mi], acc[ni][mi], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    };
    // K loop: tile kt lives in buffer kt & 1.  Top of a tile: my DMA share of it has landed (vmcnt(0)), the barrier publishes
    // it and tells that every wave is done reading the other buffer, whose refill with tile kt+1 starts right away; the
    // fragment reads of the tile start behind the barrier (their latency is covered by the co-resident block's waves: the
    // double-buffered register version of gemm_v3_kernel's schedule needs two 32-register A-fragment sets and spilled).
    i32x8 fa[MT], fw01[2], fw24[3];
    issue(kt_begin, 0);
    for (int kt = kt_begin; kt < nk; kt++) {
        const int buf = (kt - kt_begin) & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#if SIDLSG_MX8_SPREAD_DMA
        // fragment reads first; the 9 DMA pieces of the next tile are issued in two groups BETWEEN MFMA columns (a piece costs
        // the wave 60-185 cycles of issue: behind queued MFMAs that time is covered, in front of the reads it is not)
        const bool more = kt + 1 < nk;
        read_a(buf, fa);
        fw01[0] = read_w(buf, 0);
        fw01[1] = read_w(buf, 1);
        fw24[0] = read_w(buf, 2);
        fw24[1] = read_w(buf, 3);
        fw24[2] = read_w(buf, 4);
        mfma_col(fa, fw01[0], 0);
        __builtin_amdgcn_sched_barrier(0);
        if (more) issue(kt + 1, buf ^ 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_col(fa, fw01[1], 1);
        mfma_col(fa, fw24[0], 2);
        __builtin_amdgcn_sched_barrier(0);
        if (more) issue(kt + 1, buf ^ 1, 2);
        __builtin_amdgcn_sched_barrier(0);
        mfma_col(fa, fw24[1], 3);
        mfma_col(fa, fw24[2], 4);
#else
        if (kt + 1 < nk) issue(kt + 1, buf ^ 1);
        read_a(buf, fa);
        fw01[0] = read_w(buf, 0);
        fw01[1] = read_w(buf, 1);
        fw24[0] = read_w(buf, 2);
        fw24[1] = read_w(buf, 3);
        fw24[2] = read_w(buf, 4);
        mfma_col(fa, fw01[0], 0);
        mfma_col(fa, fw01[1], 1);
        mfma_col(fa, fw24[0], 2);
        mfma_col(fa, fw24[1], 3);
        mfma_col(fa, fw24[2], 4);
#endif
    }
    if (p.kt_per_split) {                // partial sums (already scaled per channel) -> this split's fp32 slab
#pragma unroll
        for (int ni = 0; ni < NT; ni++) {
            const bool paired = (ni | 1) < NT;
            const int nb = n0 + wn0 + (paired ? 32 * (ni >> 1) + lg * 8 + (ni & 1) * 4 : 16 * ni + lg * 4);
            const f32x4 sc = *reinterpret_cast<const f32x4*>(p.wscale + nb);
#pragma unroll
            for (int mi = 0; mi < MT; mi++) {
                const int m = m0 + wm0 + mi * 16 + li;
                if (m < p.M) *reinterpret_cast<f32x4*>(p.ws + ((size_t)split * p.M + m) * p.N + nb) = acc[ni][mi] * sc;
            }
        }
        return;
    }
    // (the bias is fetched here, not before the loop as in gemm_v3_kernel: its 24 registers would spill the two A-fragment sets)
    EpiPre<NT> pre;
    epilogue_prefetch<NT>(p, pre, n0 + wn0, lg);
    // per-output-channel dequantisation scale (accumulator layout: see gemm_v3_kernel's split-K store)
#pragma unroll
    for (int ni = 0; ni < NT; ni++) {
        const bool paired = (ni | 1) < NT;
        const int nb = n0 + wn0 + (paired ? 32 * (ni >> 1) + lg * 8 + (ni & 1) * 4 : 16 * ni + lg * 4);
        const f32x4 sc = *reinterpret_cast<const f32x4*>(p.wscale + nb);
#pragma unroll
        for (int mi = 0; mi < MT; mi++) acc[ni][mi] *= sc;
    }
    if (!(p.flags & (F_OUT_F32 | F_ACCUM))) {
        constexpr int LDR = BN + 8;
        bf16* ring = reinterpret_cast<bf16*>(smem);
        __syncthreads();
        gemm_epilogue<MT, NT>(p, acc, m0 + wm0, n0 + wn0, li, lg, pre, ring, m0, n0, LDR);
        __syncthreads();
        gemm_store_rows<BM, BN>(p, ring, m0, n0, LDR, tid);
        return;
    }
    gemm_epilogue<MT, NT, false>(p, acc, m0 + wm0, n0 + wn0, li, lg, pre);
}

template <int MODE>
static int launch_gemm_mx8(const GemmParams& p, hipStream_t s) {
    const int tiles = ((p.M + 127) / 128) * (p.N / 160);
    const size_t lds = (size_t)2 * (128 + 160) * 128;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_mx8_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    GemmParams q = p;
    q.group_m = 4;
    // few tiles (8x8 / 16x16 stages) and a long contraction: split K so that ~512 blocks exist (as gemm_v3_kernel does)
    const int nk = MODE == 1 ? 9 * ((p.Cin + 127) / 128) : (p.K + 127) / 128;
    int splits = 1;
    const Ws w = ws_for(s);
    if (tiles < 384 && nk >= 16 && w.ptr && (long long)p.M * p.N * 8 <= w.bytes) {
        splits = (512 + tiles - 1) / tiles;
        const long long cap = w.bytes / ((long long)p.M * p.N * 4);
        if (splits > cap) splits = (int)cap;
        if (splits > nk / 4) splits = nk / 4;
        if (splits > 16) splits = 16;
        if (splits < 2) splits = 1;
    }
    if (splits > 1) {
        q.kt_per_split = (nk + splits - 1) / splits;
        q.ws = w.ptr;
        splits = (nk + q.kt_per_split - 1) / q.kt_per_split;
    }
    SIDLSG_LAUNCH((gemm_mx8_kernel<MODE>), dim3(tiles * splits), dim3(NTHREADS), lds, s, q);
    if (splits > 1) {
        const size_t n4 = ((size_t)p.M * p.N + 3) / 4;
        SIDLSG_LAUNCH(gemm_finish_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, q, splits);
    }
    return sidlsg_last_error();
}

// ---------------------------------------------------------------------------------------------------------------------
// "A-stationary" dense GEMM for the wide short-contraction products of the 64 x 64 stage (K = 320, N >= 2560: the FF-in
// projection).  The shape is memory bound (65536 x 2560 x 320: 42 MB in, 335 MB out, 107 GFLOP = 63 us of HBM time against
// 43 us of MFMA time) and ran at a third of either bound in gemm_v3_kernel: with 5 K-tiles per output tile a block spends as
// long in its prologue (first tile DMA: nothing to overlap it), epilogue and store drain as in the K loop, and every K-tile
// waits a full L2 -> LDS round trip because only one tile is in flight.
// Here a block keeps its 128 x 320 panel of A in LDS for its whole life (80 KiB, loaded once) and walks over the 160-wide
// column tiles of its range; the W tiles stream through a 3-slot ring as ONE continuous sequence of 160 x 64 chunks -- the
// DMA of chunk s+3 is issued when chunk s has been read, so two chunks are always in flight and no tile restarts the
// pipeline -- the epilogue's loads (bias, residual) are issued at the tile's first chunk, and its stores drain behind the
// next tile's MFMAs.  That needs counted waits, which works because EVERY vector-memory operation of the loop is a buffer
// operation that is always issued (out-of-range lanes carry the OOB offset instead of being predicated off): the
// s_waitcnt vmcnt(N) immediates below count them exactly.  One block of 4 waves per CU (LDS: 80 + 60 + 10 KiB).
// Output tile -> global memory through a per-wave 16 x 80 LDS transposition buffer (row-major 160-byte runs; the MFMA
// layout's 64-byte half lines are taken at half rate by the L2, see gemm_store_rows).
// Measured (MI355X, in-session A/B against gemm_v3_kernel): 65536 x 2560 x 320 220 -> 157 us, 32768 x 2560 x 320 86 -> 77 us;
// N = 1280 / 960 tie and N = 320 loses (two tiles do not amortise the panel load with one block per CU) -> dispatched from
// N = 2560.  Variants measured slower and not kept: 8 waves with 32 x 80 wave tiles (1.55x the LDS fragment traffic per MFMA:
// 172 us), and the write-out software-pipelined into the next tile's chunks (its LDS round trips stall the MFMA stream: 185 us).
// Requires K == 64 * NKT, N % 160 == 0, bf16 output, no row vector / activation / fp32 / accumulate flags.
struct AsLoads {          // epilogue operands of one tile, fetched at its first chunk
    f32x4 b[5];
    bf16x8 r[4][3];
};
template <int NKT, bool RES>
__global__ __launch_bounds__(NTHREADS, 1) void gemm_as_kernel(GemmParams p) {
    constexpr int BM = 128, BN = 160, MT = 4, NT = 5, R = 3;
    constexpr int ACH = BM * BK, WST = BN * BK, LDT = 80;
    constexpr int NLD = 5 + (RES ? 12 : 0);     // buffer loads of AsLoads per wave
    constexpr int NST = 12;                      // buffer stores of one tile's epilogue per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* Ap = reinterpret_cast<bf16*>(smem);    // [NKT][128][64]
    bf16* Wr = Ap + NKT * ACH;                   // [R][160][64]
    bf16* Tb = Wr + R * WST;                     // [4 waves][16][LDT]
    const int tiles_n = p.N / BN;
    const int tpi = p.group_m;                   // column tiles per work item
    const int items_per_panel = (tiles_n + tpi - 1) / tpi;
    int bid = blockIdx.x;
    {
        const int nblk = gridDim.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int mt = bid / items_per_panel, it = bid - mt * items_per_panel;
    const int m0 = group_select<BM>(p, mt);
    const int nt0 = it * tpi, ntl = min(tpi, tiles_n - nt0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 80;
    const int li = lane & 15, lg = lane >> 4;
    const int lrow = lane >> 3, lslot = lane & 7;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.A), 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.W), 0, (int)p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, p.bias ? p.N * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.res), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<bf16*>(p.C), 0, 0x7FFFFFFF, 0x00020000);

    // ---- A panel: all NKT chunks at once (4 DMAs per wave and chunk)
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int r = wave * 32 + j * 8 + lrow;
        const int kcs = lslot ^ ((r >> 1) & 7);
        const int m = m0 + r;
        const unsigned off = m < p.M ? ((unsigned)m * (unsigned)p.lda + kcs * 8) * 2u : OOB;
#pragma unroll
        for (int kc = 0; kc < NKT; kc++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lptr_t)(Ap + kc * ACH + (wave * 32 + j * 8) * BK), 16, off, kc * BK * 2, 0, 0);
    }
    // ---- W stream: chunk s = (tile s / NKT of this item, K chunk s % NKT) -> ring slot s % R
    unsigned boff[5];
#pragma unroll
    for (int j = 0; j < 5; j++) {
        const int r = (wave + 4 * j) * 8 + lrow;
        boff[j] = ((unsigned)r * (unsigned)p.K + (lslot ^ wsw(r)) * 8) * 2u;
    }
    const int S = ntl * NKT;
    // (every lambda of this kernel is force-inlined: a closure left out of line drags the kernel arguments into scratch and
    // the buffer descriptors into VGPRs -> waterfall loops around every buffer operation)
    auto issue_w = [&](int s2) __attribute__((always_inline)) {   // always 5 DMAs; past the end of the item they fetch nothing (zeros into a dead slot)
        const int t = s2 / NKT, kc = s2 - t * NKT;
        const int slot = s2 % R;
        const unsigned so = ((unsigned)((nt0 + t) * BN) * (unsigned)p.K + kc * BK) * 2u;
        const bool live = s2 < S;
#pragma unroll
        for (int j = 0; j < 5; j++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lptr_t)(Wr + slot * WST + (wave + 4 * j) * 8 * BK), 16, live ? boff[j] : OOB, live ? so : 0u, 0, 0);
    };
    issue_w(0); issue_w(1); issue_w(2);

    f32x4 acc[NT][MT];
    int wrow[NT];
#pragma unroll
    for (int ni = 0; ni < NT; ni++) {
        const bool paired = (ni | 1) < NT;
        wrow[ni] = wn0 + (paired ? 32 * (ni >> 1) + (li >> 2) * 8 + (ni & 1) * 4 + (li & 3) : 16 * ni + li);
    }
    auto read_frags = [&](int kc, int slot, int kk, bf16x8 (&fa)[MT], bf16x8 (&fw)[NT]) __attribute__((always_inline)) {
        const bf16* a = Ap + kc * ACH;
        const bf16* b = Wr + slot * WST;
        const int ch = kk * 4 + lg;
#pragma unroll
        for (int mi = 0; mi < MT; mi++) {
            const int r = wm0 + mi * 16 + li;
            fa[mi] = *reinterpret_cast<const bf16x8*>(a + r * BK + ((ch ^ ((r >> 1) & 7)) << 3));
        }
#pragma unroll
        for (int ni = 0; ni < NT; ni++) {
            const int r = wrow[ni];
            fw[ni] = *reinterpret_cast<const bf16x8*>(b + r * BK + ((ch ^ wsw(r)) << 3));
        }
    };
    auto mfma_block = [&](const bf16x8 (&fa)[MT], const bf16x8 (&fw)[NT]) __attribute__((always_inline)) {
#pragma unroll
        for (int ni = 0; ni < NT; ni++)
#pragma unroll
            for (int mi = 0; mi < MT; mi++)
                acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[ni], fa[mi], acc[ni][mi], 0, 0, 0);
    };
    // epilogue operands of the tile whose first column is n0: 5 bias loads (+ 12 residual loads), always issued
    auto tile_loads = [&](int n0, AsLoads& L) __attribute__((always_inline)) {
        const int nb = n0 + wn0;
        L.b[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, (unsigned)(nb + lg * 8) * 4u, 0, 0));
        L.b[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, (unsigned)(nb + lg * 8 + 4) * 4u, 0, 0));
        L.b[2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, (unsigned)(nb + 32 + lg * 8) * 4u, 0, 0));
        L.b[3] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, (unsigned)(nb + 32 + lg * 8 + 4) * 4u, 0, 0));
        L.b[4] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, (unsigned)(nb + 64 + lg * 4) * 4u, 0, 0));
        if (RES) {
#pragma unroll
            for (int mi = 0; mi < MT; mi++) {
                const int m = m0 + wm0 + mi * 16 + li;
                const bool ok = m < p.M;
                const unsigned row = ok ? (unsigned)m * (unsigned)p.ldres : 0u;
#pragma unroll
                for (int pr = 0; pr < 3; pr++) {
                    const int n = nb + 32 * pr + lg * (pr < 2 ? 8 : 4);
                    const unsigned off = ok ? (row + (unsigned)n) * 2u : OOB;
                    if (pr < 2) L.r[mi][pr] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rr, off, 0, 0));
                    else {
                        const u32x2 t = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rr, off, 0, 0));
                        L.r[mi][pr] = __builtin_bit_cast(bf16x8, (u32x4){t[0], t[1], 0u, 0u});
                    }
                }
            }
        }
    };
    // acc -> bf16 tile -> global: per 16-row block through this wave's LDS buffer, 3 buffer stores (160-byte rows)
    bf16* tb = Tb + wave * 16 * LDT;
    auto epilogue = [&](int n0, const AsLoads& L) __attribute__((always_inline)) {
#pragma unroll
        for (int mi = 0; mi < MT; mi++) {
#pragma unroll
            for (int pr = 0; pr < 3; pr++) {
                float v[8];
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    v[r] = acc[2 * pr][mi][r] * p.alpha + L.b[pr < 2 ? 2 * pr : 4][r];
                    v[4 + r] = pr < 2 ? acc[pr < 2 ? 2 * pr + 1 : 0][mi][r] * p.alpha + L.b[pr < 2 ? 2 * pr + 1 : 0][r] : 0.f;
                }
                if (RES) {
#pragma unroll
                    for (int e = 0; e < (pr < 2 ? 8 : 4); e++) v[e] += bf2f(L.r[mi][pr][e]);
                }
                if (pr < 2) {
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; e++) o[e] = f2bf(v[e]);
                    st8(tb + li * LDT + 32 * pr + lg * 8, o);
                } else {
                    bf16x4 o = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
                    *reinterpret_cast<bf16x4*>(tb + li * LDT + 64 + lg * 4) = o;
                }
            }
            // wave-private buffer: the LDS counter orders the writes above before the reads below (and the reads before
            // the next block's writes)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const int c = lane + 64 * j;                    // 16 rows x 10 chunks of 16 bytes
                const int row = c / 10, col = (c - row * 10) * 8;
                const int m = m0 + wm0 + mi * 16 + row;
                const bool ok = c < 160 && m < p.M;
                const bf16x8 val = *reinterpret_cast<const bf16x8*>(tb + (c < 160 ? row * LDT + col : 0));
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, val), rc,
                                                       ok ? ((unsigned)m * (unsigned)p.ldc + (unsigned)(n0 + wn0 + col)) * 2u : OOB, 0, 0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
#pragma unroll
        for (int i = 0; i < NT; i++)
#pragma unroll
            for (int j = 0; j < MT; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    };

#pragma unroll
    for (int i = 0; i < NT; i++)
#pragma unroll
        for (int j = 0; j < MT; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 fa0[MT], fw0[NT], fa1[MT], fw1[NT];
    // the A panel and chunk 0 have landed when at most chunks 1 and 2 (10 DMAs) are outstanding
    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    read_frags(0, 0, 0, fa0, fw0);

    // One W chunk (step s = t * NKT + kc).  WAIT = number of vector-memory operations issued AFTER the DMAs of chunk s+1
    // at the point of the wait: chunk s+1 has landed when no more than that many are outstanding.
    auto step = [&](const int s, const int kc, auto wait_c) __attribute__((always_inline)) {
        constexpr int WAIT = decltype(wait_c)::value;
        const int slot = s % R;
        read_frags(kc, slot, 1, fa1, fw1);
        mfma_block(fa0, fw0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        read_frags(kc + 1 == NKT ? 0 : kc + 1, (s + 1) % R, 0, fa0, fw0);     // chunk s+1 (a dead read after the last step)
        issue_w(s + 3);                                                         // into the slot chunk s just vacated
        mfma_block(fa1, fw1);
    };
    // Outstanding operations younger than chunk s+1's DMAs at the wait of step (t, kc) -- program order per tile:
    //   [tile loads NLD] step 0 .. step NKT-1, each ending with 5 DMAs, [epilogue stores NST]
    //   kc = 0: chunk s+2 (5, issued in the previous tile's last step) + previous epilogue's NST stores + this tile's NLD loads
    //   kc = 1: previous stores + loads + chunk s+2 (5, issued in step 0)     kc >= 2: chunk s+2 only
    // first tile of the item: no previous epilogue (kc = 0: chunk 2 + loads; kc = 1: loads + chunk 3)
    using std::integral_constant;
    AsLoads L;
    auto tile = [&](const int t, auto first) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first)::value;
        const int n0 = (nt0 + t) * BN;
        tile_loads(n0, L);
        const int s0 = t * NKT;
        step(s0, 0, integral_constant<int, 5 + NLD + (FIRST ? 0 : NST)>{});
        if constexpr (NKT > 1) step(s0 + 1, 1, integral_constant<int, 5 + NLD + (FIRST ? 0 : NST)>{});
#pragma unroll
        for (int kc = 2; kc < NKT; kc++) step(s0 + kc, kc, integral_constant<int, 5>{});
        epilogue(n0, L);
    };
    tile(0, std::true_type{});
    for (int t = 1; t < ntl; t++) tile(t, std::false_type{});
}

template <int NKT>
static int launch_gemm_as(const GemmParams& p, hipStream_t s) {
    constexpr size_t lds = (size_t)(NKT * 128 * BK + 3 * 160 * BK + 4 * 16 * 80) * sizeof(bf16);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_as_kernel<NKT, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_as_kernel<NKT, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    const int panels = m_tiles_rt(p, 128), tiles_n = p.N / 160;
    // work item = (panel, range of column tiles): whole panels when they fill the chip, else split the column range so that
    // ~512 items exist (an item re-loads its A panel: keep ranges >= 2 tiles when N allows)
    int tpi = tiles_n;
    while (tpi > 2 && (long long)panels * ((tiles_n + tpi - 1) / tpi) < 512) tpi = (tpi + 1) / 2;
    GemmParams q = p;
    q.group_m = tpi;
    const int items = panels * ((tiles_n + tpi - 1) / tpi);
    if (p.res) SIDLSG_LAUNCH((gemm_as_kernel<NKT, true>), dim3(items), dim3(NTHREADS), lds, s, q);
    else SIDLSG_LAUNCH((gemm_as_kernel<NKT, false>), dim3(items), dim3(NTHREADS), lds, s, q);
    return sidlsg_last_error();
}

// (A multi-stage variant of the 64 x 64 fallback below -- "s64": buffer_load ... lds into a 4-slot ring, three K-tiles in flight, counted
// vmcnt waits, the v3 fragment layout -- was built in round 5 on the theory that the ~250 small launches per iteration (text K / V
// projections, 8 x 8 stage Linears: 17 - 20 us each in the step's profile) are bound by one L2 -> LDS round trip per K-tile.  Measured
// alone (tools/ab/small_gemm.py under rocprofv3) they are not: 1232 x 640 x 768 7.4 us with either kernel, 1024 x 1280 x 1280 12.9 ->
// 11.2, 2048 x 1280 x 1280 17.2 -> 18.7, step 209.98 vs 209.87 ms -- their time in the step is chip sharing, not their own latency.
// Correct (the shapes are in tests/test_gpu_ops.py::test_gemm), neutral, not in the build: commit "gemm_s64_kernel" of round 5.)
template <int BM, int BN, int MODE>
static int launch_gemm(const GemmParams& p, hipStream_t s) {
    const int tiles = m_tiles_rt(p, BM) * ((p.N + BN - 1) / BN);
    const size_t lds = (size_t)2 * (BM + BN) * BK * sizeof(bf16);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_kernel<BM, BN, MODE>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    const int nk = (p.K + BK - 1) / BK;
    const int splits = p.kt_per_split ? (nk + p.kt_per_split - 1) / p.kt_per_split : 1;
    SIDLSG_LAUNCH((gemm_bf16_kernel<BM, BN, MODE>), dim3(tiles, splits), dim3(NTHREADS), lds, s, p);
    if (p.kt_per_split) {
        const size_t n4 = ((size_t)p.M * p.N + 3) / 4;
        SIDLSG_LAUNCH(gemm_finish_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, p, splits);
    }
    return sidlsg_last_error();
}

static int g_p8_mode = -2;        // -2: not read yet
static int p8_mode() {
    if (g_p8_mode == -2) g_p8_mode = getenv("SIDLSG_GEMM_P8") ? atoi(getenv("SIDLSG_GEMM_P8")) : -1;
    return g_p8_mode;
}
// Which p8 configuration a call takes when nothing forces one (-1: none, the v3 / bf16 kernels), and with how many K splits.
// A small cost model with constants measured by tools/ubench/p8_bench on MI355X (profiles/r06_p8_ab.txt), microseconds:
//   v3  (128 x 160 tiles, 512 resident):  rounds * (11  + 1.03 * k-tiles per split)
//   p8W (256 x 320 tiles, 256 resident):  rounds * (16  + 1.61 * k-tiles per split)      13 with slab stores instead of the epilogue
//   p8S (256 x 160 tiles, 256 resident):  rounds * (8.5 + 1.03 * k-tiles per split)       7
//   split-K adds the slab round trip + gemm_finish_kernel: (8 * splits + 4) * M * N bytes at ~10 TB/s (the slabs stay in the Infinity Cache)
// i.e. the W tile's K loop runs at 1.67 PFLOP/s against 1.30 (half the staged bytes and 0.7x the fragment reads per MFMA); its
// blocks all finish at the same time (residual reads + output writes of the entire launch with no MFMA work beside them), which
// the row-major epilogue of gemm_p8.h keeps at the HBM time of those bytes.  The S tile has v3's bytes per MFMA and v3's K loop; what
// it saves is epilogue time per output byte -- worth 8-10 % stand-alone on the short-K dense GEMMs (K = 320 ... 640: five to ten
// K-tiles per tile), nothing on the convolutions (+-3 %), and nothing in the step (201.7 vs 201.8 ms per iteration with / without it,
// profiles/r06_p8_step_ab.txt): the rule offers it to dense calls only and only with SIDLSG_P8_S=1.
struct P8Choice { int cfg, splits; };
template <int MODE>
static P8Choice p8_rule(const GemmParams& p, long long ws_bytes) {
    static const int margin_pct = getenv("SIDLSG_P8_MARGIN") ? atoi(getenv("SIDLSG_P8_MARGIN")) : 3;
    static const bool dense_too = !(getenv("SIDLSG_P8_DENSE") && atoi(getenv("SIDLSG_P8_DENSE")) == 0);      // A/B switch: dense GEMMs by the same model
    static const bool s_too = getenv("SIDLSG_P8_S") && atoi(getenv("SIDLSG_P8_S")) != 0;                     // A/B switch: the 256 x 160 configuration for dense GEMMs (off: neutral in the step)
    if ((MODE == 0 && !dense_too) || p.N % 160 || p.K % BK || p.M <= 0) return {-1, 1};      // (nothing below divides by a zero tile count)
    const int nk = p.K / BK;
    const double mn = (double)p.M * p.N;
    auto fin = [&](int s) { return s > 1 ? (8.0 * s + 4.0) * mn / 10e6 : 0.0; };
    const long long cap = ws_bytes > 0 ? ws_bytes / (long long)(mn * 4) : 1;
    // v3 as dispatch_gemm would launch it
    const long long t3 = (long long)m_tiles_rt(p, 128) * (p.N / 160);
    int s3 = 1;
    if (t3 < 384 && nk >= 32 && cap >= 2) {
        s3 = (int)std::min<long long>(std::min<long long>((512 + t3 - 1) / t3, cap), std::min(nk / 8, 16));
        if (s3 < 2) s3 = 1;
    }
    const double v3_us = (double)((t3 * s3 + 511) / 512) * (11.0 + 1.03 * ((nk + s3 - 1) / s3)) + fin(s3);
    P8Choice best{-1, 1};
    double best_us = v3_us * 100.0 / (100 + margin_pct);
    for (int cfg = 1; cfg >= 0; cfg--) {
        if ((cfg == 0 && (MODE != 0 || !s_too)) || !p8_ok<MODE>(p, cfg)) continue;
        const long long t = (long long)m_tiles_rt(p, 256) * (p.N / (cfg ? 320 : 160));
        const double f1 = cfg ? 16.0 : 8.5, fs = cfg ? 13.0 : 7.0, slope = cfg ? 1.61 : 1.03;
        for (int s = 1; s <= 8; s++) {
            if (s > 1 && (s > cap || nk / s < 12 || (t * s > 256 && t * (s - 1) >= 256))) break;
            const double us = (double)((t * s + 255) / 256) * ((s > 1 ? fs : f1) + slope * ((nk + s - 1) / s)) + fin(s);
            if (us < best_us) { best_us = us; best = {cfg, s}; }
        }
    }
    return best;
}

template <int MODE>
static int dispatch_gemm(const GemmParams& pin, hipStream_t s) {
    GemmParams p = pin;
    p.Mtot = p.M;
    // algorithmic bytes: A read once (conv: the image, not its 9-fold im2col), W once per weight set, C written once, + the residual
    const double a_elems = MODE == 0 ? (double)p.M * p.K : (double)p.M / (p.Ho * p.Wo) * (p.ups ? (p.H / 2) * (p.Wd / 2) : p.H * p.Wd) * p.Cin;
    SidlsgTraceScope ts(MODE == 0 ? SIDLSG_FAM_GEMM : SIDLSG_FAM_CONV, 2.0 * p.M * (double)p.N * p.K,
                        2.0 * (a_elems + (double)p.N * p.K * (p.Mg ? 2 : 1) + (double)p.M * p.N * ((p.flags & F_OUT_F32) ? 2 : 1) + (p.res ? (double)p.M * p.N : 0.0)));
    // N multiple of 160 (every SD channel count is a multiple of 320) -> exact 160-wide tiles and, for dense rows /
    // Cin % 64 == 0 convs, the direct-to-LDS kernel (v3); otherwise 128-wide; narrow outputs (conv_out, dgrad of conv_in)
    // -> 64-wide.  Small pixel counts (8x8 / 16x16 stages, small batches) would give < 256 tiles = idle CUs: split K when
    // the contraction is long, else fall back to 64-row and then 64x64 tiles until the grid covers the chip.
    if (p.N <= 64) return launch_gemm<128, 64, MODE>(p, s);
    if constexpr (MODE == 0) {
        static const bool as_on = !(getenv("SIDLSG_GEMM_AS") && atoi(getenv("SIDLSG_GEMM_AS")) == 0);   // A/B switch
        static const int as_min_n = getenv("SIDLSG_GEMM_AS_MIN_N") ? atoi(getenv("SIDLSG_GEMM_AS_MIN_N")) : 2560;   // measured break-even
        if (as_on && p.K == 320 && p.N % 160 == 0 && p.N >= as_min_n && p.M >= 8192 && !p.rowvec && !p.flags && !p.kt_per_split && (p.ldc & 7) == 0 &&
            (!p.res || (p.ldres & 7) == 0) && (long long)p.M * p.ldc < (1ll << 30) && (!p.res || (long long)p.M * p.ldres < (1ll << 30)))
            return launch_gemm_as<5>(p, s);
    }
    auto tiles = [&](int bm, int bn) { return (long long)m_tiles_rt(p, bm) * ((p.N + bn - 1) / bn); };
    // p8 kernels (gemm_p8.h): 256-row tiles, one block per CU.  g_p8_mode: -1 = by rule, 0 = never, 1 / 2 = always the S / W
    // configuration when the call is admissible (A/B switch: SIDLSG_GEMM_P8, sidlsg_debug_set_p8)
    if constexpr (MODE != 2) {
        const int mode = p8_mode();
        const Ws w = ws_for(s);
        const int nk = p.K / BK;
        P8Choice ch{-1, 1};
        if (mode == 1 || mode == 2) {           // forced configuration: split K towards one block per CU
            ch.cfg = p8_ok<MODE>(p, mode - 1) ? mode - 1 : -1;
            const long long t = tiles(256, ch.cfg == 1 ? 320 : 160);
            if (ch.cfg >= 0 && t < 192 && w.ptr && (long long)p.M * p.N * 8 <= w.bytes)
                ch.splits = std::max(1, (int)std::min<long long>(std::min<long long>((256 + t - 1) / t, w.bytes / ((long long)p.M * p.N * 4)), std::min(nk / 12, 16)));
        } else if (mode < 0) {
            ch = p8_rule<MODE>(p, w.ptr ? w.bytes : 0);
        }
        if (ch.cfg >= 0) {
            GemmParams q = p;
            if (ch.splits >= 2) { q.kt_per_split = (nk + ch.splits - 1) / ch.splits; q.ws = w.ptr; }
            return ch.cfg ? launch_gemm_p8<MODE == 2 ? 0 : MODE, 1>(q, s) : launch_gemm_p8<MODE == 2 ? 0 : MODE, 0>(q, s);
        }
    }
    const bool n160 = p.N % 160 == 0;
    static const bool v3_on = !(getenv("SIDLSG_GEMM_V3") && atoi(getenv("SIDLSG_GEMM_V3")) == 0);   // A/B switch
    const bool v3 = n160 && v3_on && MODE != 2;
    // A/B knobs (defaults = the measured best): launches with fewer than MIN_TILES 128-row tiles split K when the
    // contraction has at least MIN_NK k-tiles, keeping at least MIN_KT k-tiles per split
    static const int MIN_TILES = getenv("SIDLSG_GEMM_MIN_TILES") ? atoi(getenv("SIDLSG_GEMM_MIN_TILES")) : 384;
    static const int MIN_NK = getenv("SIDLSG_SPLITK_MIN_NK") ? atoi(getenv("SIDLSG_SPLITK_MIN_NK")) : 32;
    static const int MIN_KT = getenv("SIDLSG_SPLITK_MIN_KT") ? atoi(getenv("SIDLSG_SPLITK_MIN_KT")) : 8;
    static const int V3_DIRECT_TILES = getenv("SIDLSG_V3_DIRECT_TILES") ? atoi(getenv("SIDLSG_V3_DIRECT_TILES")) : 256;   // 256 since the compact epilogue: 4096x1280x1280 28.1 -> 23.9 us
    {
        const long long t = tiles(128, n160 ? 160 : 128);
        const int nk = (p.K + BK - 1) / BK;
        const Ws w = ws_for(s);
        float* const g_ws = w.ptr;
        const long long g_ws_bytes = w.bytes;
        if (t < MIN_TILES && nk >= MIN_NK && (p.N & 3) == 0 && g_ws && (long long)p.M * p.N * 8 <= g_ws_bytes) {
            static const int TARGET = getenv("SIDLSG_SPLITK_TARGET") ? atoi(getenv("SIDLSG_SPLITK_TARGET")) : 512;   // blocks a split launch aims at (2 per CU)
            int splits = (int)((TARGET + t - 1) / t);
            const long long cap = g_ws_bytes / ((long long)p.M * p.N * 4);
            if (splits > cap) splits = (int)cap;
            if (splits > nk / MIN_KT) splits = nk / MIN_KT;
            if (splits > 16) splits = 16;
            if (splits >= 2) {
                GemmParams q = p;
                q.kt_per_split = (nk + splits - 1) / splits;
                q.ws = g_ws;
                if (v3) return launch_gemm_v3<MODE == 2 ? 0 : MODE>(q, s);
                return n160 ? launch_gemm<128, 160, MODE>(q, s) : launch_gemm<128, 128, MODE>(q, s);
            }
        }
    }
    if (tiles(128, n160 ? 160 : 128) >= (v3 ? V3_DIRECT_TILES : 384)) {
        if (v3) return launch_gemm_v3<MODE == 2 ? 0 : MODE>(p, s);
        return n160 ? launch_gemm<128, 160, MODE>(p, s) : launch_gemm<128, 128, MODE>(p, s);
    }
    if (tiles(64, n160 ? 160 : 128) >= 384) return n160 ? launch_gemm<64, 160, MODE>(p, s) : launch_gemm<64, 128, MODE>(p, s);
    return launch_gemm<64, 64, MODE>(p, s);
}

static int check_common(const GemmParams& p) {
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) return SIDLSG_EINVAL;
    if ((p.K & 7) || (p.lda & 7)) return SIDLSG_EINVAL;            // 16-byte chunks
    if (!p.A || !p.W || !p.C) return SIDLSG_EINVAL;
    if (p.rowvec && p.rows_per_batch <= 0) return SIDLSG_EINVAL;
    if ((p.flags & F_ACCUM) && !(p.flags & F_OUT_F32)) return SIDLSG_EINVAL;
    return SIDLSG_OK;
}

// ---------------------------------------------------------------------------------------------
// Weight gradient:  dW[N][K] += sum_m dY[m][N] * A[m][K]     (split over m; partial sums in fp32 slabs)
//
// Both operands are stored pixel-major, i.e. the contraction index m is the ROW of both LDS
// tiles.  MFMA wants 8 consecutive contraction elements per lane, so fragments are fetched
// with gfx950's LDS transpose read (ds_read_b64_tr_b16): a 16-lane group reads a
// [4 m][16 cols] block and lane i receives column i.  Two reads (m+0..3, m+4..7) give the
// 8-element fragment of the 16x16x32 MFMA for both operands, in the same m order.
// LDS rows are 256 B (128 columns); the 32-byte granule index is XOR-swizzled with
// (m&3)|((m>>3)&1)<<2 so the 8 rows a 32-lane half touches fall on 8 distinct granules.
constexpr int WG_T = 128;   // output tile: 128 (n) x 128 (k)
constexpr int WG_MB = 64;   // contraction rows per LDS stage

struct WgradParams {
    const bf16* dY;  // [M][ldy]
    const bf16* A;   // dense [M][lda] or NHWC image
    float* dW;       // [N][K]
    int M, N, K, ldy, lda;
    int H, Wd, Cin, Ho, Wo, stride, ups;
    int m_per_split;
    unsigned a_bytes, y_bytes;
    float* ws;        // [splits][N][K] slabs when splits > 1 and a workspace is available (else fp32 atomics)
    int nsplits;
    float* dB;        // optional: dB[n] += sum_m dY[m][n] (bias gradient), produced by the k-tile-0 blocks
    int slow_gather;  // conv: 1 = per-stage recomputation of the gather offsets also where the constant-advance path applies (A/B switch)
    int assign;       // dW = instead of dW +=: the caller knows dW holds nothing to keep (sidlsg_*wgrad_assign_bf16) -- no read of dW
    int bid0;         // grouped launch (wgrad_v2g_kernel): index of this job's first block (a multiple of 8); 0 otherwise
};

DEVFN int wg_swz(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

template <int MODE>
__global__ __launch_bounds__(NTHREADS) void wgrad_bf16_kernel(WgradParams p) {
    __shared__ __attribute__((aligned(16))) bf16 Ys[2][WG_MB][WG_T];
    __shared__ __attribute__((aligned(16))) bf16 Xs[2][WG_MB][WG_T];
    const int tiles_k = (p.K + WG_T - 1) / WG_T;
    const int tile = blockIdx.x;
    const int n0 = (tile / tiles_k) * WG_T, k0 = (tile % tiles_k) * WG_T;
    const int mbeg = blockIdx.y * p.m_per_split;
    const int mend = min(p.M, mbeg + p.m_per_split);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int wn0 = (wave >> 1) * 64, wk0 = (wave & 1) * 64;
    const int c16 = tid & 15, r0 = tid >> 4;  // chunk column (8 elements) and first row

    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.A), 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.dY), 0, (int)p.y_bytes, 0x00020000);

    // per-thread constants of the A (k) operand: k is fixed for the whole kernel
    const int kA = k0 + c16 * 8;
    const bool kok = kA < p.K;
    int tap = 0, cA = kA, dh = 0, dw = 0;
    if (MODE == 1) { tap = kA / p.Cin; cA = kA - tap * p.Cin; dh = tap / 3; dw = tap - dh * 3; }
    const int nY = n0 + c16 * 8;
    const bool nfull = nY + 8 <= p.N;   // whole 16-byte chunk in range (else: ragged N, element path)
    const bool nany = nY < p.N;
    const int Hs = p.ups ? (p.H >> 1) : p.H, Ws = p.ups ? (p.Wd >> 1) : p.Wd;
    const int hw = (MODE == 1) ? p.Ho * p.Wo : 1;

    // conv: (b, ho, wo) of each of this thread's 4 rows, advanced by WG_MB pixels per step (no per-step divisions)
    int pb[4], pho[4], pwo[4];
    const int d_b = WG_MB / hw, d_rem = WG_MB - d_b * hw;
    const int d_ho = (MODE == 1) ? d_rem / p.Wo : 0, d_wo = (MODE == 1) ? d_rem - d_ho * p.Wo : 0;
    if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int m = mbeg + r0 + 16 * i;
            pb[i] = m / hw;
            const int rem = m - pb[i] * hw;
            pho[i] = rem / p.Wo;
            pwo[i] = rem - pho[i] * p.Wo;
        }
    }
    bf16x8 yreg[4], xreg[4];
    auto load_tile = [&](int mb) {      // called with mb = mbeg, mbeg + WG_MB, ... in order
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int m = mb + r0 + 16 * i;
            const bool mok = m < mend;
            if (nfull) {
                yreg[i] = buf_ld8(ry, mok ? ((unsigned)m * (unsigned)p.ldy + (unsigned)nY) * 2u : OOB);
            } else {
                bf16x8 v = zero8();
                if (nany && mok)
                    for (int e = 0; e < 8; e++) if (nY + e < p.N) v[e] = p.dY[(size_t)m * p.ldy + nY + e];
                yreg[i] = v;
            }
            if (MODE == 0) {
                xreg[i] = buf_ld8(ra, (kok && mok) ? ((unsigned)m * (unsigned)p.lda + (unsigned)kA) * 2u : OOB);
            } else {
                bool ok = kok && mok;
                const int b = pb[i], ho = pho[i], wo = pwo[i];
                int hi = ho * p.stride - 1 + dh, wi = wo * p.stride - 1 + dw;
                ok = ok && hi >= 0 && hi < p.H && wi >= 0 && wi < p.Wd;
                if (p.ups) { hi >>= 1; wi >>= 1; }
                xreg[i] = buf_ld8(ra, ok ? (((unsigned)(b * Hs + hi) * (unsigned)Ws + (unsigned)wi) * (unsigned)p.lda + (unsigned)cA) * 2u : OOB);
                // advance this row by WG_MB pixels: (b, ho, wo) += (d_b, d_ho, d_wo) with single carries
                int nwo = wo + d_wo, nho = ho + d_ho, nb = b + d_b;
                if (nwo >= p.Wo) { nwo -= p.Wo; nho += 1; }
                if (nho >= p.Ho) { nho -= p.Ho; nb += 1; }
                pwo[i] = nwo; pho[i] = nho; pb[i] = nb;
            }
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int r = r0 + 16 * i;
            const int col = ((((c16 >> 1) ^ wg_swz(r)) << 1) | (c16 & 1)) << 3;
            st8(&Ys[buf][r][col], yreg[i]);
            st8(&Xs[buf][r][col], xreg[i]);
        }
    };

    f32x4 acc[4][4];   // [n tile][k tile]
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // bias gradient = dY^T * ones: one extra MFMA per dY fragment in the waves that own k-columns 0..63 of k-tile 0
    const bool do_bias = p.dB != nullptr && k0 == 0 && wk0 == 0;
    f32x4 accb[4];
#pragma unroll
    for (int i = 0; i < 4; i++) accb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bf16x8 ones = {f2bf(1.f), f2bf(1.f), f2bf(1.f), f2bf(1.f), f2bf(1.f), f2bf(1.f), f2bf(1.f), f2bf(1.f)};

    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    auto tr_frag = [&](const bf16* tilebase, int mrow, int colbase) -> bf16x8 {
        // lane t=(li) of group lg addresses row mrow + 8*lg + (t>>2) (+4 for the 2nd read), 4 columns at (t&3)*4
        const int rA = mrow + 8 * lg + (li >> 2);
        const int rB = rA + 4;
        const int g = colbase >> 4;                  // 16-column granule of this fragment
        const int ca = ((g ^ wg_swz(rA)) << 4) + (li & 3) * 4;
        const int cb = ((g ^ wg_swz(rB)) << 4) + (li & 3) * 4;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tilebase + rA * WG_T + ca));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tilebase + rB * WG_T + cb));
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    };

    const int nsteps = (mend - mbeg + WG_MB - 1) / WG_MB;
    if (nsteps <= 0) return;
    load_tile(mbeg);
    store_tile(0);
    __syncthreads();
    for (int st = 0; st < nsteps; st++) {
        const int buf = st & 1;
        if (st + 1 < nsteps) load_tile(mbeg + (st + 1) * WG_MB);
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            bf16x8 fy[4], fx[4];
#pragma unroll
            for (int i = 0; i < 4; i++) fy[i] = tr_frag(&Ys[buf][0][0], kk * 32, wn0 + i * 16);
#pragma unroll
            for (int j = 0; j < 4; j++) fx[j] = tr_frag(&Xs[buf][0][0], kk * 32, wk0 + j * 16);
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fy[i], fx[j], acc[i][j], 0, 0, 0);
            if (do_bias) {
#pragma unroll
                for (int i = 0; i < 4; i++) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fy[i], ones, accb[i], 0, 0, 0);
            }
        }
        if (st + 1 < nsteps) store_tile(buf ^ 1);
        __syncthreads();
    }
    if (do_bias && li == 0) {
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int n = n0 + wn0 + 16 * i + lg * 4 + r;
                if (n < p.N) unsafeAtomicAdd(p.dB + n, accb[i][r]);
            }
    }
    // acc[i][j][r]: n = n0+wn0+16i + lg*4 + r ; k = k0+wk0+16j + li
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int k = k0 + wk0 + 16 * j + li;
            if (k >= p.K) continue;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int n = n0 + wn0 + 16 * i + lg * 4 + r;
                if (n >= p.N) continue;
                const size_t e = (size_t)n * p.K + k;
                if (p.nsplits == 1) p.dW[e] = p.assign ? acc[i][j][r] : p.dW[e] + acc[i][j][r];   // sole owner of this element
                else if (p.ws) p.ws[(size_t)blockIdx.y * p.N * p.K + e] = acc[i][j][r];      // slab, reduced by wgrad_reduce_kernel
                else unsafeAtomicAdd(p.dW + e, acc[i][j][r]);
            }
        }
}


// "wgrad v2": same tile, LDS layout and fragment reads as wgrad_bf16_kernel, but the dY / X tiles are staged by
// direct-to-LDS loads (global_load_lds_dwordx4): no staging registers, no ds_write commit phase.  The LDS swizzle is
// applied on the SOURCE side: lane (row, q) fetches the 16-byte chunk whose swizzled position is q.  Padding (rows past
// the split, columns past N/K, conv halo) reads a zero page.  Schedule per 64-pixel stage (2 LDS buffers, one barrier):
//   A: issue the kk=1 fragment reads of the current buffer; kk=0 MFMAs   C: vmcnt(0)+lgkmcnt(0); s_barrier
//   D: kk=0 fragment reads of the next buffer; DMA of stage st+2 into the buffer just vacated   E: kk=1 MFMAs
// Blocks are numbered split-major and remapped so that one XCD works on consecutive (split, tile) pairs: the tiles of a
// split share the same dY / X rows (the 9 taps re-read X shifted), which then stay in that XCD's L2.
// Requires N % 8 == 0, K % 8 == 0 (conv: Cin % 8 == 0), ldy % 8 == 0, lda % 8 == 0, 16-byte aligned bases.
// TN = n (dY column) extent of the tile: 160 when N % 160 == 0 (every SD channel count: 320 = 2 tiles instead of 3 of 128),
// else 128.  The k (X column) extent stays 128.
template <int TN> DEVFN int wg_yswz(int row) { return TN == 128 ? wg_swz(row) : ((row >> 3) & 1); }

// TK = k (X column) extent of the tile: 128, or (dense only, round 4) 160 -- with N and K multiples of 160 a 160 x 160 tile wastes no
// MFMA work on padding (320 = 2 x 160 instead of 3 x 128) and the operands are re-read 2 + 2 instead of 3 + 3 times per split
// (L2 -> LDS bytes -44 % at 320 x 320); the X tile is then staged exactly like the dY tile (same chunk grid, same swizzle).
// (TAG: hipcc 7.2's host pass rejects the SECOND kernel that instantiates one specialisation of this template -- "no matching function,
// substitution failure", as it did for attn_q_kernel -- so the grouped kernel asks for its own, TAG = 1)
template <int MODE, int TN, int TK, bool CF = false, int TAG = 0>      // CF: conv with constant-advance gather offsets (see fastc below; the host checks the conditions)
DEVFN void wgrad_v2_body(const WgradParams& p) {
    static_assert(TK == WG_T || (TK == 160 && MODE == 0), "160-wide k tiles: dense operands only");
    constexpr int KI = TK / 32;             // 16-column X fragments per wave (wave = TN/2 x TK/2 of the tile)
    constexpr int XC = TK / 8;              // 16-byte chunks per X row (TK = 160 staging)
    constexpr int NI = TN / 32;             // 16-column dY fragments per wave (wave = TN/2 x 64 of the tile)
    constexpr int YI = TN / 32;             // dY DMA instructions per wave and stage (64 rows x TN/8 chunks / 256 lanes)
    constexpr int YC = TN / 8;              // 16-byte chunks per dY row
    constexpr int STAGE = WG_MB * (TN + TK);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* ring = reinterpret_cast<bf16*>(smem);
    const int tiles_k = (p.K + TK - 1) / TK;
    const int tiles = ((p.N + TN - 1) / TN) * tiles_k;
    int bid = blockIdx.x;
    bid -= p.bid0;                           // (bid0: first block of this job in a grouped launch, else 0)
    {
        const int nblk = tiles * p.nsplits;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int split = bid / tiles, tile = bid - split * tiles;
    const int n0 = (tile / tiles_k) * TN, k0 = (tile % tiles_k) * TK;
    const int mbeg = split * p.m_per_split;
    const int mend = min(p.M, mbeg + p.m_per_split);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int wn0 = (wave >> 1) * (TN / 2), wk0 = (wave & 1) * (TK / 2);
    const int q16 = tid & 15, r0 = tid >> 4;            // LDS chunk position and first row of this lane
    const int c16 = (((q16 >> 1) ^ wg_swz(r0)) << 1) | (q16 & 1);   // source chunk (wg_swz(r0 + 16 i) == wg_swz(r0))
    const int kA = k0 + c16 * 8;
    const bool kok = kA < p.K;
    int cA = kA, dh = 0, dw = 0;
    if (MODE == 1) { const int tap = kA / p.Cin; cA = kA - tap * p.Cin; dh = tap / 3; dw = tap - dh * 3; }
    const int Hs = p.ups ? (p.H >> 1) : p.H, Ws = p.ups ? (p.Wd >> 1) : p.Wd;
    const int hw = (MODE == 1) ? p.Ho * p.Wo : 1;
    // buffer_load ... lds with one 32-bit byte offset per row; rows past the split / columns past N,K / the conv halo
    // carry an out-of-range offset (the buffer unit writes zeros).  dY and dense X advance by a wave-uniform soffset.
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.dY), 0, (int)p.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.A), 0, (int)p.a_bytes, 0x00020000);
    // dY: DMA instruction j of wave w fills LDS chunks [(w*YI + j)*64, +64) of the [64][YC] chunk grid; the lane's
    // position (row, cpos) holds the source chunk whose swizzled position it is
    unsigned yoff[YI], xoff[4];
#pragma unroll
    for (int j = 0; j < YI; j++) {
        const int idx = (wave * YI + j) * 64 + lane;
        const int row = idx / YC, cpos = idx - row * YC;
        const int chunk = (((cpos >> 1) ^ wg_yswz<TN>(row)) << 1) | (cpos & 1);
        const int nY = n0 + chunk * 8;
        yoff[j] = nY < p.N ? ((unsigned)(mbeg + row) * (unsigned)p.ldy + (unsigned)nY) * 2u : OOB;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const unsigned m = (unsigned)(mbeg + r0 + 16 * i);
        xoff[i] = (MODE == 0 && kok) ? (m * (unsigned)p.lda + (unsigned)kA) * 2u : OOB;
    }
    unsigned xoffw[TK == WG_T ? 1 : KI];         // TK = 160: the dY scheme for X (KI DMA instructions per wave and stage)
    if constexpr (TK != WG_T) {
#pragma unroll
        for (int j = 0; j < KI; j++) {
            const int idx = (wave * KI + j) * 64 + lane;
            const int row = idx / XC, cpos = idx - row * XC;
            const int chunk = (((cpos >> 1) ^ wg_yswz<TK>(row)) << 1) | (cpos & 1);
            const int kX = k0 + chunk * 8;
            xoffw[j] = kX < p.K ? ((unsigned)(mbeg + row) * (unsigned)p.lda + (unsigned)kX) * 2u : OOB;
        }
    }

    int pb[4], pho[4], pwo[4];
    const int d_b = WG_MB / hw, d_rem = WG_MB - d_b * hw;
    const int d_ho = (MODE == 1) ? d_rem / p.Wo : 0, d_wo = (MODE == 1) ? d_rem - d_ho * p.Wo : 0;
    if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int m = mbeg + r0 + 16 * i;
            pb[i] = m / hw;
            const int rem = m - pb[i] * hw;
            pho[i] = rem / p.Wo;
            pwo[i] = rem - pho[i] * p.Wo;
        }
    }
    // conv fast path (round 4): when a stage's 64 pixels are whole image rows or whole samples (Wo | 64 or Ho * Wo | 64: every SD shape),
    // the stride divides the image (H == Ho * stride) and there is no fused upsampling, the input row of a lane's pixel advances by the
    // SAME number of rows every stage -- also across sample boundaries -- so the gather offset is a per-lane constant plus a wave-uniform
    // soffset (as in the dense kernel) and only the vertical halo test remains per stage: ~9 instead of ~40 VALU per row and stage
    // (the conv weight gradient ran 5.3 VALU instructions per MFMA, profiles/r04_wgrad_sq_pmc.json).  The descriptor's base sits one
    // image row below X so that the offset of a pixel in the top halo row is not negative.
    const unsigned rowb = (unsigned)Ws * (unsigned)p.lda * 2u;
    constexpr bool fastc = CF;      // (as a run-time branch beside the general path the kernel lost 35 %: two gather bodies in the stage loop)
    static_assert(!CF || (MODE == 1 && TK == WG_T), "constant-advance gather: conv, 128-column k tiles");
    const __amdgpu_buffer_rsrc_t rxf = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.A) - (fastc ? (size_t)rowb / 2 : 0), 0,
                                                                         (int)(p.a_bytes + (fastc ? rowb : 0u)), 0x00020000);
    const int dstage = fastc ? (d_b * p.H + d_ho * p.stride) * (int)rowb : 0;
    unsigned xoffc[4];
    if constexpr (fastc) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int wi = pwo[i] * p.stride - 1 + dw;
            const bool wok = kok && wi >= 0 && wi < p.Wd;
            xoffc[i] = wok ? ((unsigned)(pb[i] * p.H + pho[i] * p.stride + dh) * rowb + ((unsigned)wi * (unsigned)p.lda + (unsigned)cA) * 2u) : OOB;
        }
    }
    auto issue = [&](int mb, int buf) {      // called with mb = mbeg, mbeg + WG_MB, ... in order; YI + 4 DMA instructions per wave
        bf16* ys = ring + buf * STAGE;
        bf16* xs = ys + WG_MB * TN;
        const int ysoff = (mb - mbeg) * p.ldy * 2, xsoff = (mb - mbeg) * p.lda * 2;
        const bool tail = mb + WG_MB > mend;          // only the last stage of a split has rows past mend
#pragma unroll
        for (int j = 0; j < YI; j++) {
            const bool mok = !tail || (mb + ((wave * YI + j) * 64 + lane) / YC < mend);
            if (SIDLSG_WGRAD_ASM_DMA) dma16_asm(ry, ys + (wave * YI + j) * 512, mok ? yoff[j] : OOB, ysoff);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(ry, (lptr_t)(ys + (wave * YI + j) * 512), 16, mok ? yoff[j] : OOB, ysoff, 0, 0);
        }
        if constexpr (TK != WG_T) {
#pragma unroll
            for (int j = 0; j < KI; j++) {
                const bool mok = !tail || (mb + ((wave * KI + j) * 64 + lane) / XC < mend);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lptr_t)(xs + (wave * KI + j) * 512), 16, mok ? xoffw[j] : OOB, xsoff, 0, 0);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const bool mok = !tail || (mb + r0 + 16 * i < mend);
            if (MODE == 0) {
                if (SIDLSG_WGRAD_ASM_DMA) dma16_asm(rx, xs + (4 * wave + 16 * i) * WG_T, mok ? xoff[i] : OOB, xsoff);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lptr_t)(xs + (4 * wave + 16 * i) * WG_T), 16, mok ? xoff[i] : OOB, xsoff, 0, 0);
            } else if constexpr (fastc) {
                const int ho = pho[i];
                const int hi = ho * p.stride - 1 + dh;
                const bool ok = mok && hi >= 0 && hi < p.H;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rxf, (lptr_t)(xs + (4 * wave + 16 * i) * WG_T), 16, ok ? xoffc[i] : OOB,
                                                         ((mb - mbeg) / WG_MB) * dstage, 0, 0);
                const int nho = ho + d_ho;
                pho[i] = nho >= p.Ho ? nho - p.Ho : nho;
            } else {
                bool ok = kok && mok;
                const int b = pb[i], ho = pho[i], wo = pwo[i];
                int hi = ho * p.stride - 1 + dh, wi = wo * p.stride - 1 + dw;
                ok = ok && hi >= 0 && hi < p.H && wi >= 0 && wi < p.Wd;
                if (p.ups) { hi >>= 1; wi >>= 1; }
                const unsigned o = ok ? (((unsigned)(b * Hs + hi) * (unsigned)Ws + (unsigned)wi) * (unsigned)p.lda + (unsigned)cA) * 2u : OOB;
                if (SIDLSG_WGRAD_ASM_DMA) dma16_asm(rx, xs + (4 * wave + 16 * i) * WG_T, o, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lptr_t)(xs + (4 * wave + 16 * i) * WG_T), 16, o, 0, 0, 0);
                int nwo = wo + d_wo, nho = ho + d_ho, nb = b + d_b;
                if (nwo >= p.Wo) { nwo -= p.Wo; nho += 1; }
                if (nho >= p.Ho) { nho -= p.Ho; nb += 1; }
                pwo[i] = nwo; pho[i] = nho; pb[i] = nb;
            }
        }
    };

    f32x4 acc[NI][KI];   // [n tile][k tile]
#pragma unroll
    for (int i = 0; i < NI; i++)
#pragma unroll
        for (int j = 0; j < KI; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool do_bias = p.dB != nullptr && k0 == 0 && wk0 == 0;
    f32x4 accb[NI];
#pragma unroll
    for (int i = 0; i < NI; i++) accb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bf16x8 ones = {f2bf(1.f), f2bf(1.f), f2bf(1.f), f2bf(1.f), f2bf(1.f), f2bf(1.f), f2bf(1.f), f2bf(1.f)};

    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    // per-lane LDS offsets of the two transpose reads of a fragment (rows 8*lg + (li>>2) and +4), excluding the
    // 16-column granule g of the fragment, which enters as (g ^ swz) << 4
    const int rA = 8 * lg + (li >> 2), rB = rA + 4;
    // kk*32 does not change the swizzle bits.  X tile: 256-byte rows, 3-bit granule XOR; dY tile: TN*2-byte rows (320 B
    // rows alias the banks of rows 8 apart only -> 1-bit XOR with (row>>3)&1 for TN = 160)
    const int sxA = wg_yswz<TK>(rA), sxB = wg_yswz<TK>(rB), syA = wg_yswz<TN>(rA), syB = wg_yswz<TN>(rB);
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    auto tr_frag = [&](const bf16* tilebase, int pitch, int sA, int sB, int kk, int colbase) -> bf16x8 {
        const int g = colbase >> 4;
        const bf16* base = tilebase + kk * 32 * pitch + (li & 3) * 4;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + rA * pitch + ((g ^ sA) << 4)));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + rB * pitch + ((g ^ sB) << 4)));
        s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    };
    auto read_frags = [&](int buf, int kk, bf16x8 (&fy)[NI], bf16x8 (&fx)[KI]) {
        const bf16* ys = ring + buf * STAGE;
        const bf16* xs = ys + WG_MB * TN;
#pragma unroll
        for (int i = 0; i < NI; i++) fy[i] = tr_frag(ys, TN, syA, syB, kk, wn0 + i * 16);
#pragma unroll
        for (int j = 0; j < KI; j++) fx[j] = tr_frag(xs, TK, sxA, sxB, kk, wk0 + j * 16);
    };
    auto mfma_block = [&](const bf16x8 (&fy)[NI], const bf16x8 (&fx)[KI]) {
        if constexpr ((SIDLSG_V3_PRIO) & 2) V3_SETPRIO(1);
#pragma unroll
        for (int i = 0; i < NI; i++)
#pragma unroll
            for (int j = 0; j < KI; j++)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fy[i], fx[j], acc[i][j], 0, 0, 0);
        if (do_bias) {
#pragma unroll
            for (int i = 0; i < NI; i++) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fy[i], ones, accb[i], 0, 0, 0);
        }
        if constexpr ((SIDLSG_V3_PRIO) & 2) V3_SETPRIO(0);
    };

    const int nsteps = (mend - mbeg + WG_MB - 1) / WG_MB;
    if (nsteps <= 0) return;
    bf16x8 fy0[NI], fx0[KI], fy1[NI], fx1[KI];
    WTRACE(0);
    issue(mbeg, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    WTRACE(1);
    read_frags(0, 0, fy0, fx0);
    if (nsteps > 1) issue(mbeg + WG_MB, 1);
    auto stage = [&](const int st, auto has_next, auto fetch) {           // straight-line instantiations: see gemm_v3_kernel
        const int buf = st & 1;
        read_frags(buf, 1, fy1, fx1);                                     // A
        mfma_block(fy0, fx0);
        if (SIDLSG_V3_SCHED_FENCE) __builtin_amdgcn_sched_barrier(0);    // keep the kk=0 MFMAs in front of the wait
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // C: my DMA share of stage st+1 landed; my reads of buf done
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (SIDLSG_V3_SCHED_FENCE) __builtin_amdgcn_sched_barrier(0);
        if constexpr (decltype(has_next)::value) {                        // D
            read_frags(buf ^ 1, 0, fy0, fx0);
            if constexpr (decltype(fetch)::value) issue(mbeg + (st + 2) * WG_MB, buf);
        }
        mfma_block(fy1, fx1);                                             // E
    };
    {
        int st = 0;
        for (; st + 2 < nsteps; st++) stage(st, std::true_type{}, std::true_type{});
        if (st + 1 < nsteps) { stage(st, std::true_type{}, std::false_type{}); st++; }
        stage(st, std::false_type{}, std::false_type{});
    }
    WTRACE(2);
    if (do_bias && li == 0) {
#pragma unroll
        for (int i = 0; i < NI; i++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int n = n0 + wn0 + 16 * i + lg * 4 + r;
                if (n < p.N) unsafeAtomicAdd(p.dB + n, accb[i][r]);
            }
    }
    // Result write-out.  In the accumulator layout a lane holds single floats of 16-float row segments: 64-80 scalar stores
    // per lane behind three run-time branches each -- the phase trace (tools/ab/wgrad_trace.py) showed 7.5-18 us per block in
    // this epilogue, a third to a half of a dense weight-gradient block.  Instead the tile goes through LDS (the stage ring is
    // free now; fp32 [TN/2][128 + 4], one half of the n rows at a time) and leaves as 16-byte stores along k: 512-byte runs
    // (11.5 -> 9.3 us dense, 9.6 -> 5.9 us conv; code 19.8 -> 11.2 KB).  What remains is a chip-wide write burst: all blocks of
    // a launch have equal work, reach this point together and write tiles x splits x 64 KB of slabs (31 MB for 65536x320x320)
    // with nothing left to overlap it.
    if (p.nsplits == 1 || p.ws) {
        constexpr int LDW = TK + 4;
        float* img = reinterpret_cast<float*>(smem);
        float* dst = p.nsplits == 1 ? p.dW : p.ws + (size_t)split * p.N * p.K;
        const bool accum = p.nsplits == 1 && !p.assign;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            __syncthreads();                             // the ring (or the previous half image) is no longer read
            if ((wave >> 1) == h) {
#pragma unroll
                for (int i = 0; i < NI; i++)
#pragma unroll
                    for (int j = 0; j < KI; j++)
#pragma unroll
                        for (int r = 0; r < 4; r++) img[(16 * i + lg * 4 + r) * LDW + wk0 + 16 * j + li] = acc[i][j][r];
            }
            __syncthreads();
            for (int c = tid; c < (TN / 2) * (TK / 4); c += NTHREADS) {
                const int row = c / (TK / 4), k4 = (c - row * (TK / 4)) * 4;
                const int n = n0 + h * (TN / 2) + row, k = k0 + k4;
                if (n < p.N && k < p.K) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(img + row * LDW + k4);
                    float* d = dst + (size_t)n * p.K + k;
                    if (accum) v += *reinterpret_cast<const f32x4*>(d);
                    *reinterpret_cast<f32x4*>(d) = v;
                }
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < NI; i++)
#pragma unroll
            for (int j = 0; j < KI; j++) {
                const int k = k0 + wk0 + 16 * j + li;
                if (k >= p.K) continue;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int n = n0 + wn0 + 16 * i + lg * 4 + r;
                    if (n < p.N) unsafeAtomicAdd(p.dW + (size_t)n * p.K + k, acc[i][j][r]);
                }
            }
    }
    WTRACE(3);
#ifdef SIDLSG_EXP_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WTRACE(4);
#endif
}

// (two plain kernels over one body: a kernel template with the second non-type parameter did not get a host stub from
// hipcc 7.2 -- undefined symbol at load time, no diagnostic)
template <int MODE>
__global__ __launch_bounds__(NTHREADS, 2) void wgrad_v2_kernel(WgradParams p) { wgrad_v2_body<MODE, 128, WG_T>(p); }
template <int MODE>
__global__ __launch_bounds__(NTHREADS, 2) void wgrad_v2w_kernel(WgradParams p) { wgrad_v2_body<MODE, 160, WG_T>(p); }
__global__ __launch_bounds__(NTHREADS, 2) void wgrad_v2s_kernel(WgradParams p) { wgrad_v2_body<0, 160, 160>(p); }      // dense, 160 x 160 tiles
__global__ __launch_bounds__(NTHREADS, 2) void wgrad_v2f_kernel(WgradParams p) { wgrad_v2_body<1, 128, WG_T, true>(p); }     // conv, constant-advance gather
__global__ __launch_bounds__(NTHREADS, 2) void wgrad_v2wf_kernel(WgradParams p) { wgrad_v2_body<1, 160, WG_T, true>(p); }    // the same with 160-wide n tiles

// ---- grouped dense weight gradients (round 5) --------------------------------------------------------------------------------------
// The square projections of a transformer block (to_out of both attentions, the cross-attention's to_q, proj_in / proj_out: C x C, plus the
// 77-token k|v projection) each need ~56 pixel splits to fill the chip on their own (9 tiles of 128 x 128 at C = 320): per layer 31 MB of
// slabs, a reduce launch, and 49 + 16 us for 13 GFLOP.  wgrad_v2g_kernel runs up to WG_GROUP such layers in ONE grid -- every job keeps its
// own operands, shapes and split count (WgradParams by value in the kernel argument), a block finds its job from its index -- so that the
// group as a whole fills the chip with ~1/5 of the splits per layer, and wgrad_reduce_g_kernel folds all its slabs in one launch.
constexpr int WG_GROUP = 8;
struct WgradGroup { int njobs; int blk0[WG_GROUP + 1]; WgradParams j[WG_GROUP]; };      // blk0: first block of each job (multiples of 8)
__global__ __launch_bounds__(NTHREADS, 2) void wgrad_v2g_kernel(WgradGroup g) {
    const int bx = blockIdx.x;
    int j = 0;
#pragma unroll
    for (int i = 1; i < WG_GROUP; i++)
        if (i < g.njobs && bx >= g.blk0[i]) j = i;
    const WgradParams& p = g.j[j];          // (p.bid0 == g.blk0[j], set by the host)
    const int local = bx - g.blk0[j];
    const int nblk = (((p.N + 127) / 128) * ((p.K + WG_T - 1) / WG_T)) * p.nsplits;
    if (local >= nblk) return;              // (the job's range is padded to a multiple of 8 blocks: XCD = block index mod 8 inside a job too)
    wgrad_v2_body<0, 128, WG_T, false, 1>(p);
}
// the same for the layers of the 160 x 160-tile kernel (q|k|v, FF-in, FF-out: N and K multiples of 160)
__global__ __launch_bounds__(NTHREADS, 2) void wgrad_v2sg_kernel(WgradGroup g) {
    const int bx = blockIdx.x;
    int j = 0;
#pragma unroll
    for (int i = 1; i < WG_GROUP; i++)
        if (i < g.njobs && bx >= g.blk0[i]) j = i;
    const WgradParams& p = g.j[j];
    const int local = bx - g.blk0[j];
    const int nblk = ((p.N / 160) * (p.K / 160)) * p.nsplits;
    if (local >= nblk) return;
    wgrad_v2_body<0, 160, 160, false, 1>(p);
}
struct WgradReduceGroup { int njobs; unsigned blk0[WG_GROUP + 1]; const float* ws[WG_GROUP]; float* dW[WG_GROUP]; unsigned n4[WG_GROUP]; int splits[WG_GROUP], assign[WG_GROUP]; };
__global__ __launch_bounds__(256) void wgrad_reduce_g_kernel(WgradReduceGroup g) {
    int j = 0;
#pragma unroll
    for (int i = 1; i < WG_GROUP; i++)
        if (i < g.njobs && blockIdx.x >= g.blk0[i]) j = i;
    const unsigned i4 = (blockIdx.x - g.blk0[j]) * 256 + threadIdx.x;
    if (i4 >= g.n4[j]) return;
    const size_t i = (size_t)i4 * 4, n = (size_t)g.n4[j] * 4;
    const float* ws = g.ws[j];
    float* dW = g.dW[j];
    const int splits = g.splits[j];
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (!g.assign[j]) v = *reinterpret_cast<const f32x4*>(dW + i);
    int sidx = 0;
    for (; sidx + 3 < splits; sidx += 4) {          // wgrad_reduce_kernel's order: bit-identical sums
        const f32x4 a = *reinterpret_cast<const f32x4*>(ws + (size_t)sidx * n + i);
        const f32x4 b = *reinterpret_cast<const f32x4*>(ws + (size_t)(sidx + 1) * n + i);
        const f32x4 c = *reinterpret_cast<const f32x4*>(ws + (size_t)(sidx + 2) * n + i);
        const f32x4 d = *reinterpret_cast<const f32x4*>(ws + (size_t)(sidx + 3) * n + i);
        v += a; v += b; v += c; v += d;
    }
    for (; sidx < splits; sidx++) v += *reinterpret_cast<const f32x4*>(ws + (size_t)sidx * n + i);
    *reinterpret_cast<f32x4*>(dW + i) = v;
}

// dW[i] += sum_s slab[s][i]  (assign: dW[i] = sum_s slab[s][i], summed in the same order from 0 -- bit-identical to the sum onto a zeroed dW)
__global__ void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dW, size_t n, int splits, int assign) {
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 4 <= n) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (!assign) v = *reinterpret_cast<const f32x4*>(dW + i);
        int sidx = 0;
        for (; sidx + 3 < splits; sidx += 4) {          // four independent slab loads in flight, fixed summation order
            const f32x4 a = *reinterpret_cast<const f32x4*>(ws + (size_t)sidx * n + i);
            const f32x4 b = *reinterpret_cast<const f32x4*>(ws + (size_t)(sidx + 1) * n + i);
            const f32x4 c = *reinterpret_cast<const f32x4*>(ws + (size_t)(sidx + 2) * n + i);
            const f32x4 d = *reinterpret_cast<const f32x4*>(ws + (size_t)(sidx + 3) * n + i);
            v += a; v += b; v += c; v += d;
        }
        for (; sidx < splits; sidx++) v += *reinterpret_cast<const f32x4*>(ws + (size_t)sidx * n + i);
        *reinterpret_cast<f32x4*>(dW + i) = v;
    } else {
        for (size_t j = i; j < n; j++) {
            float v = assign ? 0.f : dW[j];
            for (int sidx = 0; sidx < splits; sidx++) v += ws[(size_t)sidx * n + j];
            dW[j] = v;
        }
    }
}


template <int MODE>
static int launch_wgrad(WgradParams p, hipStream_t s) {
    static const bool v2_on = !(getenv("SIDLSG_WGRAD_V2") && atoi(getenv("SIDLSG_WGRAD_V2")) == 0);
    static const bool tn160_on = !(getenv("SIDLSG_WGRAD_TN160") && atoi(getenv("SIDLSG_WGRAD_TN160")) == 0);   // A/B switch
    const bool aligned = !(p.N & 7) && !(p.K & 7) && !(p.ldy & 7) && !(p.lda & 7) && !(MODE == 1 && (p.Cin & 7)) &&
                         !(((uintptr_t)p.dY | (uintptr_t)p.A) & 15);
    const bool v2 = v2_on && aligned;
    // 160-wide n tiles: measured faster only where they cut the tile count by a third (conv, Cout = 320: 216 -> 178 us);
    // for N = 640 (5 -> 4 tiles) and the dense shapes the leaner 128 kernel wins
    static const bool tn160_dense = getenv("SIDLSG_WGRAD_TN160_DENSE") && atoi(getenv("SIDLSG_WGRAD_TN160_DENSE"));   // A/B switch
    // dense, N and K multiples of 160 (every Linear of the SD transformer blocks): 160 x 160 tiles (wgrad_v2s_kernel)
    static const bool sq160_on = !(getenv("SIDLSG_WGRAD_SQ160") && atoi(getenv("SIDLSG_WGRAD_SQ160")) == 0);       // A/B switch
    // measured per shape (tools/ab/wgrad_sweep.py, MI355X): faster on every non-square layer of the transformer blocks at M >= 4096
    // (2560 x 320: 192 -> 157 us, 5120 x 640: 145 -> 120, 10240 x 1280: 139 -> 112; a backward pass's dense weight gradients 1025 -> 927 us
    // at CFG batch 16), slower on the square C x C layers (fewer, larger tiles need more pixel splits = more slab traffic: 36.7 -> 39.9 us)
    // and at small pixel counts unless the weight is very large
    const bool sq160 = v2 && sq160_on && MODE == 0 && p.N % 160 == 0 && p.K % 160 == 0 && p.N != p.K &&
                       (p.M >= 4096 || (long long)p.N * p.K >= (8ll << 20));
    const int tn = sq160 ? 160 : (v2 && tn160_on && (MODE == 1 || tn160_dense) && p.N == 320) ? 160 : WG_T;
    const int tk = sq160 ? 160 : WG_T;
    const int tiles = ((p.N + tn - 1) / tn) * ((p.K + tk - 1) / tk);
    // Split the pixel contraction so the grid fills the chip in whole rounds of 512 resident blocks (256 CUs x 2).
    // Small cost model (us): rounds * (rows per block * 24 ns + 3 us block overhead) + slab reduction at ~4 TB/s;
    // e.g. 69 tiles: ceil(768/69) = 12 splits ran 1.6 rounds, the model picks a whole number of rounds.
    int splits = 1;
    {
        const int cap = std::min(64, (p.M + 4 * WG_MB - 1) / (4 * WG_MB));
        static const int slots = getenv("SIDLSG_WGRAD_SLOTS") ? atoi(getenv("SIDLSG_WGRAD_SLOTS")) : 512;   // A/B knob: resident blocks a launch may count on
        double best = 1e30;
        const double slab_us = (double)p.N * p.K * 4.0 / 4e6;
        for (int sp = 1; sp <= cap; sp++) {
            const int rounds = (sp * tiles + slots - 1) / slots;
            const double rows = (double)((p.M + sp - 1) / sp);
            const double est = rounds * (rows * 0.024 + 3.0) + (sp > 1 ? sp * slab_us : 0.0);
            if (est < best) { best = est; splits = sp; }
        }
    }
    const int max_splits = (p.M + 4 * WG_MB - 1) / (4 * WG_MB);
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    int mps = (p.M + splits - 1) / splits;
    mps = (mps + WG_MB - 1) / WG_MB * WG_MB;
    splits = (p.M + mps - 1) / mps;
    const long long nk = (long long)p.N * p.K;
    const Ws wsl = ws_for(s);
    float* const g_ws = wsl.ptr;
    const long long g_ws_bytes = wsl.bytes;
    if (splits > 1 && g_ws) {                       // cap the split count by the slab space
        const long long cap = g_ws_bytes / (nk * 4);
        if (cap < 2) { p.ws = nullptr; }
        else {
            if (splits > cap) {
                splits = (int)cap;
                mps = (p.M + splits - 1) / splits;
                mps = (mps + WG_MB - 1) / WG_MB * WG_MB;
                splits = (p.M + mps - 1) / mps;
            }
            p.ws = g_ws;
        }
    }
    p.m_per_split = mps;
    p.nsplits = splits;
    if (p.assign && splits > 1 && !p.ws) {          // fp32-atomic fallback (no slab space): the atomics need a zeroed target
        if (hipMemsetAsync(p.dW, 0, (size_t)nk * sizeof(float), s) != hipSuccess) return sidlsg_last_error();
    }
    if (v2) {
        const size_t lds = (size_t)2 * WG_MB * (tn + tk) * sizeof(bf16);
        static bool attr_done = false;
        if (!attr_done) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_v2_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * WG_MB * (128 + WG_T) * 2);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_v2w_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * WG_MB * (160 + WG_T) * 2);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_v2s_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * WG_MB * (160 + 160) * 2);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_v2f_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * WG_MB * (128 + WG_T) * 2);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_v2wf_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * WG_MB * (160 + WG_T) * 2);
            attr_done = true;
        }
        bool cf = false;
        if (MODE == 1 && !p.slow_gather && !p.ups && p.H == p.Ho * p.stride) {          // conditions of the constant-advance gather (wgrad_v2_body, fastc)
            const int hw = p.Ho * p.Wo, d_b = WG_MB / hw, d_rem = WG_MB - d_b * hw;
            const int d_wo = d_rem - (d_rem / p.Wo) * p.Wo;
            const unsigned long long rowb = (unsigned long long)p.Wd * p.lda * 2ull;
            cf = d_wo == 0 && (unsigned long long)p.a_bytes + rowb < 0x7fffffffull;
        }
        if (sq160) SIDLSG_LAUNCH(wgrad_v2s_kernel, dim3(tiles * splits), dim3(NTHREADS), lds, s, p);
        else if (cf && tn == 160) SIDLSG_LAUNCH(wgrad_v2wf_kernel, dim3(tiles * splits), dim3(NTHREADS), lds, s, p);
        else if (cf) SIDLSG_LAUNCH(wgrad_v2f_kernel, dim3(tiles * splits), dim3(NTHREADS), lds, s, p);
        else if (tn == 160) SIDLSG_LAUNCH((wgrad_v2w_kernel<MODE>), dim3(tiles * splits), dim3(NTHREADS), lds, s, p);
        else SIDLSG_LAUNCH((wgrad_v2_kernel<MODE>), dim3(tiles * splits), dim3(NTHREADS), lds, s, p);
    } else
    SIDLSG_LAUNCH((wgrad_bf16_kernel<MODE>), dim3(tiles, splits), dim3(NTHREADS), 0, s, p);
    if (splits > 1 && p.ws)
        SIDLSG_LAUNCH(wgrad_reduce_kernel, dim3((unsigned)((nk / 4 + 256) / 256)), dim3(256), 0, s, p.ws, p.dW, (size_t)nk, splits, p.assign);
    return sidlsg_last_error();
}

static bool fits31(unsigned long long bytes) { return bytes < 0x7FFFFFFFull; }

// Host side of the grouped dense weight gradient.  Every job must be one the 128 x 128 direct-to-LDS kernel takes (aligned operands);
// the split count is chosen for the GROUP: all jobs together should put ~512 blocks on the chip (whole rounds, like launch_wgrad's
// model): every job gets the same split count `want` = slots / (tiles of all jobs), limited by its own row count (>= 4 stages per split) and by
// what is left of the stream's workspace for its slabs.
static int launch_wgrad_group(WgradParams* jobs, int njobs, hipStream_t s, bool t160) {
    if (njobs < 1 || njobs > WG_GROUP) return SIDLSG_EINVAL;
    long long tiles[WG_GROUP], total_tiles = 0;
    for (int i = 0; i < njobs; i++) {
        const WgradParams& p = jobs[i];
        const bool aligned = !(p.N & 7) && !(p.K & 7) && !(p.ldy & 7) && !(p.lda & 7) && !(((uintptr_t)p.dY | (uintptr_t)p.A) & 15);
        if (!aligned || (t160 && (p.N % 160 || p.K % 160))) return SIDLSG_EINVAL;
        tiles[i] = t160 ? (long long)(p.N / 160) * (p.K / 160) : (long long)((p.N + 127) / 128) * ((p.K + WG_T - 1) / WG_T);
        total_tiles += tiles[i];
    }
    const Ws wsl = ws_for(s);
    static const int slots = getenv("SIDLSG_WGRAD_SLOTS") ? atoi(getenv("SIDLSG_WGRAD_SLOTS")) : 512;
    int want = (int)std::max<long long>(1, slots / std::max<long long>(1, total_tiles));        // splits per job when all rows counts are equal
    WgradGroup g{};
    WgradReduceGroup rg{};
    g.njobs = njobs;
    int blocks = 0, rjobs = 0;
    unsigned rblocks = 0;
    long long ws_used = 0;
    for (int i = 0; i < njobs; i++) {
        WgradParams p = jobs[i];
        const int max_splits = (p.M + 4 * WG_MB - 1) / (4 * WG_MB);
        int splits = std::min(std::min(want, 64), std::max(1, max_splits));
        int mps = (p.M + splits - 1) / splits;
        mps = (mps + WG_MB - 1) / WG_MB * WG_MB;
        splits = (p.M + mps - 1) / mps;
        const long long nk = (long long)p.N * p.K;
        p.ws = nullptr;
        if (splits > 1) {
            // slabs that do not fit what is left of the workspace: as many splits as do fit (launch_wgrad does the same), one only when not even two fit
            const long long fit = (wsl.ptr && !(nk & 3)) ? (wsl.bytes - ws_used) / (nk * 4) : 0;
            if (fit < splits) {
                splits = fit >= 2 ? (int)fit : 1;
                mps = ((p.M + splits - 1) / splits + WG_MB - 1) / WG_MB * WG_MB;
                splits = (p.M + mps - 1) / mps;
            }
            if (splits > 1) { p.ws = reinterpret_cast<float*>(reinterpret_cast<char*>(wsl.ptr) + ws_used); ws_used += ((long long)splits * nk * 4 + 255) / 256 * 256; }
        }
        p.m_per_split = mps;
        p.nsplits = splits;
        g.blk0[i] = blocks;
        p.bid0 = blocks;
        g.j[i] = p;
        blocks += (int)((tiles[i] * splits + 7) / 8 * 8);
        if (splits > 1) {
            rg.blk0[rjobs] = rblocks; rg.ws[rjobs] = p.ws; rg.dW[rjobs] = p.dW; rg.n4[rjobs] = (unsigned)(nk / 4);
            rg.splits[rjobs] = splits; rg.assign[rjobs] = p.assign;
            rblocks += (unsigned)((nk / 4 + 255) / 256);
            rjobs++;
        }
    }
    g.blk0[njobs] = blocks;
    rg.njobs = rjobs; rg.blk0[rjobs] = rblocks;
    const size_t lds = (size_t)2 * WG_MB * (t160 ? 320 : 128 + WG_T) * sizeof(bf16);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_v2g_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * WG_MB * (128 + WG_T) * 2);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_v2sg_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * WG_MB * 320 * 2);
        attr_done = true;
    }
    if (t160) SIDLSG_LAUNCH(wgrad_v2sg_kernel, dim3(blocks), dim3(NTHREADS), lds, s, g);
    else SIDLSG_LAUNCH(wgrad_v2g_kernel, dim3(blocks), dim3(NTHREADS), lds, s, g);
    if (rjobs) SIDLSG_LAUNCH(wgrad_reduce_g_kernel, dim3(rblocks), dim3(256), 0, s, rg);
    return sidlsg_last_error();
}

extern "C" {

// (diagnostic) resident blocks per CU of the weight-gradient kernels with their dynamic LDS: 0 = 128x128, 1 = 160x128, 2 = 160x160
int sidlsg_debug_wgrad_blocks_per_cu(int which) {
    int n = -1;
    const size_t lds = (size_t)2 * WG_MB * ((which ? 160 : 128) + (which == 2 ? 160 : 128)) * 2;
    const void* f = which == 2 ? reinterpret_cast<const void*>(&wgrad_v2s_kernel)
                               : which ? reinterpret_cast<const void*>(&wgrad_v2w_kernel<0>) : reinterpret_cast<const void*>(&wgrad_v2_kernel<0>);
    (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, f, NTHREADS, lds) != hipSuccess) return -1;
    return n;
}

// (A/B switch) which GEMM / conv calls take the 256-row-tile kernels of gemm_p8.h: -1 by rule (default), 0 never, 1 / 2 every
// admissible call on the 256 x 160 / 256 x 320 configuration.  Returns the previous setting.
int sidlsg_debug_set_p8(int mode) {
    const int old = p8_mode();
    g_p8_mode = mode;
    return old;
}

// Optional scratch for split-K GEMMs and weight gradients: `ptr` = device memory of `bytes` bytes, owned by the
// caller, used by every later sidlsg_gemm_bf16 / sidlsg_conv3x3_bf16 call of this process (one stream at a time).
// Pass NULL/0 to disable split-K.
int sidlsg_set_workspace(void* ptr, long long bytes) {
    g_ws_default = (float*)ptr; g_ws_default_bytes = ptr ? bytes : 0;
    return SIDLSG_OK;
}

// Private scratch for one stream (see WsSlot).  ptr = NULL removes the entry.  At most 4 streams.
int sidlsg_set_stream_workspace(void* stream, void* ptr, long long bytes) {
    hipStream_t s = (hipStream_t)stream;
    for (int i = 0; i < g_ws_nslots; i++)
        if (g_ws_slots[i].stream == s) {
            if (ptr) { g_ws_slots[i].ptr = (float*)ptr; g_ws_slots[i].bytes = bytes; }
            else { g_ws_slots[i] = g_ws_slots[--g_ws_nslots]; }
            return SIDLSG_OK;
        }
    if (!ptr) return SIDLSG_OK;
    if (g_ws_nslots >= 8) return SIDLSG_EINVAL;
    g_ws_slots[g_ws_nslots++] = {s, (float*)ptr, bytes};
    return SIDLSG_OK;
}

// Dense GEMM: C[M,N] = act(alpha * A[M,K] W[N,K]^T + bias[N] + rowvec[m/rpb,N] + res[M,N])
int sidlsg_gemm_bf16(const void* A, int lda, const void* W, void* C, int ldc, const float* bias, const void* res,
                     int ldres, const float* rowvec, int ld_rowvec, int rows_per_batch, int M, int N, int K, float alpha,
                     int flags, void* stream) {
    GemmParams p{};
    p.A = (const bf16*)A; p.W = (const bf16*)W; p.C = C; p.bias = bias; p.res = (const bf16*)res; p.rowvec = rowvec;
    p.ldrv = ld_rowvec > 0 ? ld_rowvec : N;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldc = ldc; p.ldres = ldres; p.rows_per_batch = rows_per_batch;
    p.alpha = alpha; p.flags = flags;
    if (int e = check_common(p)) return e;
    const unsigned long long ab = ((unsigned long long)(M - 1) * lda + K) * 2ull, wb = (unsigned long long)N * K * 2ull;
    if (!fits31(ab) || !fits31(wb)) return SIDLSG_EINVAL;
    p.a_bytes = (unsigned)ab; p.w_bytes = (unsigned)wb;
    return dispatch_gemm<0>(p, (hipStream_t)stream);
}

// Grouped dense GEMM: two weight sets on one stacked activation matrix (M even): rows [0, M/2) use (W, bias), rows [M/2, M)
// use (W1, bias1); res / rowvec / C are stacked like A.  One launch fills the chip where each half alone would not (the frozen
// fake-score and teacher passes of phase B read identical inputs: sid_training_loop.py:494-506).  bf16 only.
int sidlsg_gemm_bf16_g2(const void* A, int lda, const void* W, const void* W1, void* C, int ldc, const float* bias, const float* bias1,
                        const void* res, int ldres, const float* rowvec, int ld_rowvec, int rows_per_batch, int M, int N, int K,
                        float alpha, int flags, void* stream) {
    if (!W1 || (M & 1) || (bias != nullptr) != (bias1 != nullptr)) return SIDLSG_EINVAL;
    GemmParams p{};
    p.A = (const bf16*)A; p.W = (const bf16*)W; p.C = C; p.bias = bias; p.res = (const bf16*)res; p.rowvec = rowvec;
    p.W1 = (const bf16*)W1; p.bias1 = bias1; p.Mg = M / 2;
    p.ldrv = ld_rowvec > 0 ? ld_rowvec : N;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldc = ldc; p.ldres = ldres; p.rows_per_batch = rows_per_batch;
    p.alpha = alpha; p.flags = flags;
    if (int e = check_common(p)) return e;
    const unsigned long long ab = ((unsigned long long)(M - 1) * lda + K) * 2ull, wb = (unsigned long long)N * K * 2ull;
    if (!fits31(ab) || !fits31(wb)) return SIDLSG_EINVAL;
    p.a_bytes = (unsigned)ab; p.w_bytes = (unsigned)wb;
    return dispatch_gemm<0>(p, (hipStream_t)stream);
}

// FF-in projection + GEGLU in one kernel: h[M][N2] = A W^T + bias (N2 = 2 F; H may be NULL: not stored),
// Y[M][F] = h[:, :F] * gelu(h[:, F:]).  Runs on the direct-to-LDS kernel only: returns SIDLSG_EINVAL for shapes it would not take
// (F % 80, K % 64, fewer output tiles than the kernel is dispatched for) -- sidlsg_gemm_geglu_ok tells beforehand.
int sidlsg_gemm_geglu_ok(int M, int N2, int K) {
    static const bool on = !(getenv("SIDLSG_GEMM_GEGLU") && atoi(getenv("SIDLSG_GEMM_GEGLU")) == 0);      // A/B switch
    if (!on || M <= 0 || N2 <= 0 || K <= 0 || (N2 % 160) || (K & 7)) return 0;
    return (long long)((M + 127) / 128) * (N2 / 160) >= 256 ? 1 : 0;
}
int sidlsg_gemm_geglu_bf16(const void* A, int lda, const void* W, void* H, int ldh, void* Y, int ldy, const float* bias, int M, int N2,
                           int K, void* stream) {
    if (!sidlsg_gemm_geglu_ok(M, N2, K) || !Y || (ldy & 7) || (ldh & 7) || (lda & 7)) return SIDLSG_EINVAL;
    GemmParams p{};
    p.A = (const bf16*)A; p.W = (const bf16*)W; p.C = H; p.bias = bias; p.Y = (bf16*)Y; p.ldy = ldy; p.geglu = N2 / 2;
    p.ldrv = N2; p.M = M; p.N = N2; p.K = K; p.lda = lda; p.ldc = ldh; p.rows_per_batch = 1; p.alpha = 1.f; p.Mtot = M;
    if (!A || !W) return SIDLSG_EINVAL;
    const unsigned long long ab = ((unsigned long long)(M - 1) * lda + K) * 2ull, wb = (unsigned long long)N2 * K * 2ull;
    if (!fits31(ab) || !fits31(wb)) return SIDLSG_EINVAL;
    p.a_bytes = (unsigned)ab; p.w_bytes = (unsigned)wb;
    SidlsgTraceScope ts(SIDLSG_FAM_GEMM, 2.0 * M * (double)N2 * K, 2.0 * ((double)M * K + (double)N2 * K + (double)M * N2 * (H ? 1.5 : 0.5)));
    if (p8_geglu_takes(p)) return launch_gemm_p8_geglu<1>(p, (hipStream_t)stream);
    return launch_gemm_v3<0>(p, (hipStream_t)stream);
}

// Data gradient of the FF-out projection + GEGLU derivative in one kernel: dy = dOut Wt^T (Wt = the [F][K] backward-data operand of
// the [K][F] FF-out weight; dy [M][F] is never stored), dH[:, :F] = dy * gelu(h[:, F:]), dH[:, F:] = dy * h[:, :F] * gelu'(h[:, F:]).
// Direct-to-LDS kernel only, same admission rule as the forward fusion.
int sidlsg_gemm_geglu_bwd_ok(int M, int F, int K) {
    static const bool on = !(getenv("SIDLSG_GEMM_GEGLU_BWD") && atoi(getenv("SIDLSG_GEMM_GEGLU_BWD")) == 0);      // A/B switch
#ifdef SIDLSG_EXP_NO_GEGLU_BWD
    return 0;
#endif
    if (!on || M <= 0 || F <= 0 || K <= 0 || (F % 160) || (K & 7)) return 0;
    return (long long)((M + 127) / 128) * (F / 160) >= 256 ? 1 : 0;
}
int sidlsg_gemm_geglu_bwd_bf16(const void* dOut, int lda, const void* Wt, const void* H, void* dH, int ldh, int M, int F, int K,
                               void* stream) {
    if (!sidlsg_gemm_geglu_bwd_ok(M, F, K) || !dOut || !Wt || !H || !dH || (ldh & 7) || (lda & 7) || ldh < 2 * F) return SIDLSG_EINVAL;
    GemmParams p{};
    p.A = (const bf16*)dOut; p.W = (const bf16*)Wt; p.C = nullptr; p.Y = (bf16*)dH; p.Hin = (const bf16*)H; p.ldy = ldh; p.geglu_bwd = F;
    p.ldrv = F; p.M = M; p.N = F; p.K = K; p.lda = lda; p.ldc = F; p.rows_per_batch = 1; p.alpha = 1.f; p.Mtot = M;
    const unsigned long long ab = ((unsigned long long)(M - 1) * lda + K) * 2ull, wb = (unsigned long long)F * K * 2ull;
    if (!fits31(ab) || !fits31(wb)) return SIDLSG_EINVAL;
    p.a_bytes = (unsigned)ab; p.w_bytes = (unsigned)wb;
    SidlsgTraceScope ts(SIDLSG_FAM_GEMM, 2.0 * M * (double)F * K, 2.0 * ((double)M * K + (double)F * K + 4.0 * (double)M * F));
    if (p8_geglu_takes(p)) return launch_gemm_p8_geglu<2>(p, (hipStream_t)stream);
    return launch_gemm_v3<0>(p, (hipStream_t)stream);
}

// Grouped variants of the two GEGLU fusions (round 6: the grouped frozen pass of phase B -- 64 of an iteration's 144 sample-passes -- had fallen back
// to the unfused chain: grouped FF-in GEMM, stand-alone GEGLU, grouped FF-out dgrad, stand-alone GEGLU backward).  Rows [0, M/2) use (W, bias) / Wt,
// rows [M/2, M) use (W1, bias1) / Wt1; everything else as in the single-set entry points (group_select narrows each block to its set, the epilogues
// only see the block's rows).
int sidlsg_gemm_geglu_bf16_g2(const void* A, int lda, const void* W, const void* W1, void* H, int ldh, void* Y, int ldy, const float* bias,
                              const float* bias1, int M, int N2, int K, void* stream) {
    if (!sidlsg_gemm_geglu_ok(M, N2, K) || !Y || (ldy & 7) || (ldh & 7) || (lda & 7) || (M & 1) || !W1 || (bias != nullptr) != (bias1 != nullptr)) return SIDLSG_EINVAL;
    GemmParams p{};
    p.A = (const bf16*)A; p.W = (const bf16*)W; p.C = H; p.bias = bias; p.Y = (bf16*)Y; p.ldy = ldy; p.geglu = N2 / 2;
    p.W1 = (const bf16*)W1; p.bias1 = bias1; p.Mg = M / 2;
    p.ldrv = N2; p.M = M; p.N = N2; p.K = K; p.lda = lda; p.ldc = ldh; p.rows_per_batch = 1; p.alpha = 1.f; p.Mtot = M;
    if (!A || !W) return SIDLSG_EINVAL;
    const unsigned long long ab = ((unsigned long long)(M - 1) * lda + K) * 2ull, wb = (unsigned long long)N2 * K * 2ull;
    if (!fits31(ab) || !fits31(wb)) return SIDLSG_EINVAL;
    p.a_bytes = (unsigned)ab; p.w_bytes = (unsigned)wb;
    SidlsgTraceScope ts(SIDLSG_FAM_GEMM, 2.0 * M * (double)N2 * K, 2.0 * ((double)M * K + 2.0 * (double)N2 * K + (double)M * N2 * (H ? 1.5 : 0.5)));
    if (p8_geglu_takes(p)) return launch_gemm_p8_geglu<1>(p, (hipStream_t)stream);
    return launch_gemm_v3<0>(p, (hipStream_t)stream);
}
int sidlsg_gemm_geglu_bwd_bf16_g2(const void* dOut, int lda, const void* Wt, const void* Wt1, const void* H, void* dH, int ldh, int M, int F, int K,
                                  void* stream) {
    if (!sidlsg_gemm_geglu_bwd_ok(M, F, K) || !dOut || !Wt || !Wt1 || !H || !dH || (ldh & 7) || (lda & 7) || ldh < 2 * F || (M & 1)) return SIDLSG_EINVAL;
    GemmParams p{};
    p.A = (const bf16*)dOut; p.W = (const bf16*)Wt; p.C = nullptr; p.Y = (bf16*)dH; p.Hin = (const bf16*)H; p.ldy = ldh; p.geglu_bwd = F;
    p.W1 = (const bf16*)Wt1; p.Mg = M / 2;
    p.ldrv = F; p.M = M; p.N = F; p.K = K; p.lda = lda; p.ldc = F; p.rows_per_batch = 1; p.alpha = 1.f; p.Mtot = M;
    const unsigned long long ab = ((unsigned long long)(M - 1) * lda + K) * 2ull, wb = (unsigned long long)F * K * 2ull;
    if (!fits31(ab) || !fits31(wb)) return SIDLSG_EINVAL;
    p.a_bytes = (unsigned)ab; p.w_bytes = (unsigned)wb;
    SidlsgTraceScope ts(SIDLSG_FAM_GEMM, 2.0 * M * (double)F * K, 2.0 * ((double)M * K + 2.0 * (double)F * K + 4.0 * (double)M * F));
    if (p8_geglu_takes(p)) return launch_gemm_p8_geglu<2>(p, (hipStream_t)stream);
    return launch_gemm_v3<0>(p, (hipStream_t)stream);
}

// Grouped 3x3 convolution: samples [0, B/2) use (W, bias), samples [B/2, B) use (W1, bias1) (B even).
int sidlsg_conv3x3_bf16_g2(const void* X, int ldx, const void* W, const void* W1, void* Y, int ldc, const float* bias, const float* bias1,
                           const void* res, int ldres, const float* rowvec, int ld_rowvec, int B, int H, int Wd, int Cin, int Cout,
                           int stride, int ups, float alpha, int flags, void* stream) {
    if (stride != 1 && stride != 2) return SIDLSG_EINVAL;
    if ((Cin & 7) || !W1 || (B & 1) || (bias != nullptr) != (bias1 != nullptr)) return SIDLSG_EINVAL;
    if (ups && ((H | Wd) & 1)) return SIDLSG_EINVAL;
    GemmParams p{};
    p.A = (const bf16*)X; p.W = (const bf16*)W; p.C = Y; p.bias = bias; p.res = (const bf16*)res; p.rowvec = rowvec;
    p.W1 = (const bf16*)W1; p.bias1 = bias1;
    p.ldrv = ld_rowvec > 0 ? ld_rowvec : Cout;
    p.H = H; p.Wd = Wd; p.Cin = Cin; p.stride = stride; p.ups = ups;
    p.Ho = (H + 2 - 3) / stride + 1; p.Wo = (Wd + 2 - 3) / stride + 1;
    p.M = B * p.Ho * p.Wo; p.N = Cout; p.K = 9 * Cin; p.lda = ldx; p.ldc = ldc; p.ldres = ldres;
    p.Mg = p.M / 2;
    p.rows_per_batch = p.Ho * p.Wo; p.alpha = alpha; p.flags = flags;
    if (int e = check_common(p)) return e;
    const int Hs = ups ? H / 2 : H, Ws = ups ? Wd / 2 : Wd;
    const unsigned long long ab = (((unsigned long long)B * Hs * Ws - 1) * ldx + Cin) * 2ull, wb = (unsigned long long)Cout * 9 * Cin * 2ull;
    if (!fits31(ab) || !fits31(wb)) return SIDLSG_EINVAL;
    p.a_bytes = (unsigned)ab; p.w_bytes = (unsigned)wb;
    if (Cin % 64 == 0) return dispatch_gemm<1>(p, (hipStream_t)stream);
    return dispatch_gemm<2>(p, (hipStream_t)stream);
}

// Implicit-GEMM 3x3 convolution, pad 1, NHWC.  X: [B][Hs][Ws][ldx] (ldx >= Cin, pixel stride),
// W: [Cout][3][3][Cin], Y: [B][Ho][Wo][ldc].  `ups`=1 reads X through a nearest x2 upsample
// (virtual input H x Wd = 2Hs x 2Ws).  stride in {1,2}.
int sidlsg_conv3x3_bf16(const void* X, int ldx, const void* W, void* Y, int ldc, const float* bias, const void* res,
                        int ldres, const float* rowvec, int ld_rowvec, int B, int H, int Wd, int Cin, int Cout, int stride,
                        int ups, float alpha, int flags, void* stream) {
    if (stride != 1 && stride != 2) return SIDLSG_EINVAL;
    if (Cin & 7) return SIDLSG_EINVAL;
    if (ups && ((H | Wd) & 1)) return SIDLSG_EINVAL;
    GemmParams p{};
    p.A = (const bf16*)X; p.W = (const bf16*)W; p.C = Y; p.bias = bias; p.res = (const bf16*)res; p.rowvec = rowvec;
    p.ldrv = ld_rowvec > 0 ? ld_rowvec : Cout;
    p.H = H; p.Wd = Wd; p.Cin = Cin; p.stride = stride; p.ups = ups;
    p.Ho = (H + 2 - 3) / stride + 1; p.Wo = (Wd + 2 - 3) / stride + 1;
    p.M = B * p.Ho * p.Wo; p.N = Cout; p.K = 9 * Cin; p.lda = ldx; p.ldc = ldc; p.ldres = ldres;
    p.rows_per_batch = p.Ho * p.Wo; p.alpha = alpha; p.flags = flags;
    if (int e = check_common(p)) return e;
    const int Hs = ups ? H / 2 : H, Ws = ups ? Wd / 2 : Wd;
    const unsigned long long ab = (((unsigned long long)B * Hs * Ws - 1) * ldx + Cin) * 2ull, wb = (unsigned long long)Cout * 9 * Cin * 2ull;
    if (!fits31(ab) || !fits31(wb)) return SIDLSG_EINVAL;
    p.a_bytes = (unsigned)ab; p.w_bytes = (unsigned)wb;
    if (Cin % 64 == 0) return dispatch_gemm<1>(p, (hipStream_t)stream);
    return dispatch_gemm<2>(p, (hipStream_t)stream);
}

// dW[N][K] += dY[M][N]^T A[M][K]   (dense: Linear / 1x1 conv weight gradient; fp32 accumulate)
static int wgrad_dense_impl(const void* dY, int ldy, const void* A, int lda, float* dW, float* dBias, int M, int N, int K, int assign, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0 || (K & 7) || (lda & 7) || !dY || !A || !dW) return SIDLSG_EINVAL;
    WgradParams p{};
    p.dY = (const bf16*)dY; p.A = (const bf16*)A; p.dW = dW; p.dB = dBias; p.M = M; p.N = N; p.K = K; p.ldy = ldy; p.lda = lda; p.assign = assign;
    const unsigned long long ab = ((unsigned long long)(M - 1) * lda + K) * 2ull, yb = ((unsigned long long)(M - 1) * ldy + N) * 2ull;
    if (!fits31(ab) || !fits31(yb)) return SIDLSG_EINVAL;
    p.a_bytes = (unsigned)ab; p.y_bytes = (unsigned)yb;
    SidlsgTraceScope ts(SIDLSG_FAM_WGRAD, 2.0 * M * (double)N * K, 2.0 * ((double)M * N + (double)M * K) + (assign ? 4.0 : 8.0) * N * K);      // dY, A read once; dW (read +) written (fp32)
    return launch_wgrad<0>(p, (hipStream_t)stream);
}
int sidlsg_wgrad_bf16(const void* dY, int ldy, const void* A, int lda, float* dW, float* dBias, int M, int N, int K, void* stream) {
    return wgrad_dense_impl(dY, ldy, A, lda, dW, dBias, M, N, K, 0, stream);
}
// dW = dY^T A (dBias still +=): for a gradient buffer whose previous contents are dead -- the fused optimizer leaves the weight
// gradients un-zeroed (sidlsg_adam_ema with zero_grad = 0 on those ranges) and the first weight gradient after it overwrites them,
// which saves the optimizer's 4 B / parameter of zero stores and this kernel's (or its reduction's) 4 B / parameter read of dW.
// Same summation order as the accumulating entry point on a zeroed dW: bit-identical results.
int sidlsg_wgrad_assign_bf16(const void* dY, int ldy, const void* A, int lda, float* dW, float* dBias, int M, int N, int K, void* stream) {
    return wgrad_dense_impl(dY, ldy, A, lda, dW, dBias, M, N, K, 1, stream);
}

// Several dense weight gradients as ONE launch (+ one slab-reduction launch): jobs = host array of njobs <= 8 records
//   { const void* dY; const void* A; float* dW; float* dBias; int ldy, lda, M, N, K, assign; int pad[2]; }      (64 bytes each)
// with the meaning of sidlsg_wgrad_bf16 / _assign_bf16 per record.  Every job must satisfy the alignment of the direct-to-LDS kernel
// (N, K, ldy, lda multiples of 8, 16-byte aligned operands), else EINVAL and nothing is launched.  Bit-identical per job to the single
// entry points wherever both choose the same split count; otherwise the fp32 summation order over pixels differs.
struct sidlsg_wgrad_job { const void* dY; const void* A; float* dW; float* dBias; int ldy, lda, M, N, K, assign; int pad[2]; };
static int wgrad_group_impl(const void* jobs, int njobs, void* stream, bool t160) {
    if (!jobs || njobs < 1 || njobs > WG_GROUP) return SIDLSG_EINVAL;
    const sidlsg_wgrad_job* in = (const sidlsg_wgrad_job*)jobs;
    WgradParams ps[WG_GROUP];
    double flop = 0, bytes = 0;
    for (int i = 0; i < njobs; i++) {
        const sidlsg_wgrad_job& q = in[i];
        if (q.M <= 0 || q.N <= 0 || q.K <= 0 || !q.dY || !q.A || !q.dW) return SIDLSG_EINVAL;
        WgradParams p{};
        p.dY = (const bf16*)q.dY; p.A = (const bf16*)q.A; p.dW = q.dW; p.dB = q.dBias; p.M = q.M; p.N = q.N; p.K = q.K; p.ldy = q.ldy; p.lda = q.lda;
        p.assign = q.assign;
        const unsigned long long ab = ((unsigned long long)(q.M - 1) * q.lda + q.K) * 2ull, yb = ((unsigned long long)(q.M - 1) * q.ldy + q.N) * 2ull;
        if (!fits31(ab) || !fits31(yb)) return SIDLSG_EINVAL;
        p.a_bytes = (unsigned)ab; p.y_bytes = (unsigned)yb;
        ps[i] = p;
        flop += 2.0 * q.M * (double)q.N * q.K;
        bytes += 2.0 * ((double)q.M * q.N + (double)q.M * q.K) + (q.assign ? 4.0 : 8.0) * q.N * q.K;
    }
    SidlsgTraceScope ts(SIDLSG_FAM_WGRAD, flop, bytes);
    return launch_wgrad_group(ps, njobs, (hipStream_t)stream, t160);
}
int sidlsg_wgrad_group_bf16(const void* jobs, int njobs, void* stream) { return wgrad_group_impl(jobs, njobs, stream, false); }
// The same on 160 x 160 tiles: every job's N and K must be multiples of 160 (the q|k|v, FF-in and FF-out projections of a transformer block)
int sidlsg_wgrad_group160_bf16(const void* jobs, int njobs, void* stream) { return wgrad_group_impl(jobs, njobs, stream, true); }

// dW[Cout][3][3][Cin] += conv3x3 weight gradient (same geometry arguments as sidlsg_conv3x3_bf16)
static int wgrad_conv_impl(const void* dY, int ldy, const void* X, int ldx, float* dW, float* dBias, int B, int H, int Wd,
                           int Cin, int Cout, int stride, int ups, int assign, void* stream) {
    if ((stride != 1 && stride != 2) || (Cin & 7) || (ldx & 7) || !dY || !X || !dW) return SIDLSG_EINVAL;
    WgradParams p{};
    p.dY = (const bf16*)dY; p.A = (const bf16*)X; p.dW = dW; p.dB = dBias; p.ldy = ldy; p.lda = ldx; p.assign = assign;
    static const int slow_gather = getenv("SIDLSG_WGRAD_CONV_FAST") && atoi(getenv("SIDLSG_WGRAD_CONV_FAST")) == 0;      // A/B switch
    p.slow_gather = slow_gather;
    p.H = H; p.Wd = Wd; p.Cin = Cin; p.stride = stride; p.ups = ups;
    p.Ho = (H + 2 - 3) / stride + 1; p.Wo = (Wd + 2 - 3) / stride + 1;
    p.M = B * p.Ho * p.Wo; p.N = Cout; p.K = 9 * Cin;
    const int Hs = ups ? H / 2 : H, Ws = ups ? Wd / 2 : Wd;
    const unsigned long long ab = (((unsigned long long)B * Hs * Ws - 1) * ldx + Cin) * 2ull, yb = ((unsigned long long)(p.M - 1) * ldy + Cout) * 2ull;
    if (!fits31(ab) || !fits31(yb)) return SIDLSG_EINVAL;
    p.a_bytes = (unsigned)ab; p.y_bytes = (unsigned)yb;
    SidlsgTraceScope ts(SIDLSG_FAM_CONV_WGRAD, 2.0 * p.M * (double)p.N * p.K,
                        2.0 * ((double)p.M * p.N + (double)B * Hs * Ws * Cin) + (assign ? 4.0 : 8.0) * p.N * p.K);
    return launch_wgrad<1>(p, (hipStream_t)stream);
}
int sidlsg_conv3x3_wgrad_bf16(const void* dY, int ldy, const void* X, int ldx, float* dW, float* dBias, int B, int H, int Wd,
                              int Cin, int Cout, int stride, int ups, void* stream) {
    return wgrad_conv_impl(dY, ldy, X, ldx, dW, dBias, B, H, Wd, Cin, Cout, stride, ups, 0, stream);
}
int sidlsg_conv3x3_wgrad_assign_bf16(const void* dY, int ldy, const void* X, int ldx, float* dW, float* dBias, int B, int H, int Wd,
                                     int Cin, int Cout, int stride, int ups, void* stream) {
    return wgrad_conv_impl(dY, ldy, X, ldx, dW, dBias, B, H, Wd, Cin, Cout, stride, ups, 1, stream);
}

// ---- fp8-weight contractions (see gemm_fp8w_kernel).  W8: e4m3 bytes [N][K]; wscale: fp32 [N].  K % 16 == 0.
int sidlsg_quantize_fp8_rows(const void* src_bf16, void* dst_fp8, float* scale, int rows, int cols, void* stream) {
    if (rows <= 0 || cols <= 0 || (cols & 7) || !src_bf16 || !dst_fp8 || !scale) return SIDLSG_EINVAL;
    SIDLSG_LAUNCH(quantize_fp8_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16*)src_bf16,
                       (unsigned char*)dst_fp8, scale, rows, cols);
    return sidlsg_last_error();
}

int sidlsg_gemm_fp8w(const void* A, int lda, const void* W8, const float* wscale, void* C, int ldc, const float* bias, const void* res,
                     int ldres, const float* rowvec, int ld_rowvec, int rows_per_batch, int M, int N, int K, float alpha,
                     int flags, void* stream) {
    GemmParams p{};
    p.A = (const bf16*)A; p.W = (const bf16*)W8; p.wscale = wscale; p.C = C; p.bias = bias; p.res = (const bf16*)res; p.rowvec = rowvec;
    p.ldrv = ld_rowvec > 0 ? ld_rowvec : N;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldc = ldc; p.ldres = ldres; p.rows_per_batch = rows_per_batch;
    p.alpha = alpha; p.flags = flags;
    if (int e = check_common(p)) return e;
    if ((K & 15) || !wscale) return SIDLSG_EINVAL;
    const unsigned long long ab = ((unsigned long long)(M - 1) * lda + K) * 2ull, wb = (unsigned long long)N * K;
    if (!fits31(ab) || !fits31(wb)) return SIDLSG_EINVAL;
    p.a_bytes = (unsigned)ab; p.w_bytes = (unsigned)wb;
    const int tiles = ((M + F8_T - 1) / F8_T) * ((N + F8_T - 1) / F8_T);
    SIDLSG_LAUNCH((gemm_fp8w_kernel<0>), dim3(tiles), dim3(NTHREADS), 0, (hipStream_t)stream, p);
    return sidlsg_last_error();
}

int sidlsg_gemm_mx8(const void* A8, int lda, const void* W8, const float* wscale, void* C, int ldc, const float* bias, const void* res,
                    int ldres, const float* rowvec, int ld_rowvec, int rows_per_batch, int M, int N, int K, float alpha,
                    int flags, void* stream) {
    GemmParams p{};
    p.A = (const bf16*)A8; p.W = (const bf16*)W8; p.wscale = wscale; p.C = C; p.bias = bias; p.res = (const bf16*)res; p.rowvec = rowvec;
    p.ldrv = ld_rowvec > 0 ? ld_rowvec : N;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldc = ldc; p.ldres = ldres; p.rows_per_batch = rows_per_batch;
    p.alpha = alpha; p.flags = flags;
    if (M <= 0 || N <= 0 || K <= 0 || !A8 || !W8 || !C || !wscale) return SIDLSG_EINVAL;
    if ((K & 15) || (lda & 15) || (N % 160) || lda < K) return SIDLSG_EINVAL;
    if (rowvec && rows_per_batch <= 0) return SIDLSG_EINVAL;
    if ((flags & F_ACCUM) && !(flags & F_OUT_F32)) return SIDLSG_EINVAL;
    const unsigned long long ab = (unsigned long long)(M - 1) * lda + K, wb = (unsigned long long)N * K;
    if (!fits31(ab) || !fits31(wb)) return SIDLSG_EINVAL;
    p.a_bytes = (unsigned)ab; p.w_bytes = (unsigned)wb;
    SidlsgTraceScope ts(SIDLSG_FAM_GEMM, 2.0 * p.M * (double)p.N * p.K);
    return launch_gemm_mx8<0>(p, (hipStream_t)stream);
}

int sidlsg_conv3x3_mx8(const void* X8, int ldx, const void* W8, const float* wscale, void* Y, int ldc, const float* bias, const void* res,
                       int ldres, const float* rowvec, int ld_rowvec, int B, int H, int Wd, int Cin, int Cout, float alpha, int flags,
                       void* stream) {
    if (B <= 0 || H <= 0 || Wd <= 0 || Cin <= 0 || Cout <= 0 || !X8 || !W8 || !Y || !wscale) return SIDLSG_EINVAL;
    if ((Cin & 15) || (ldx & 15) || (Cout % 160) || ldx < Cin) return SIDLSG_EINVAL;
    if ((flags & F_ACCUM) && !(flags & F_OUT_F32)) return SIDLSG_EINVAL;
    GemmParams p{};
    p.A = (const bf16*)X8; p.W = (const bf16*)W8; p.wscale = wscale; p.C = Y; p.bias = bias; p.res = (const bf16*)res; p.rowvec = rowvec;
    p.ldrv = ld_rowvec > 0 ? ld_rowvec : Cout;
    p.M = B * H * Wd; p.N = Cout; p.K = 9 * Cin; p.lda = ldx; p.ldc = ldc; p.ldres = ldres; p.rows_per_batch = H * Wd;
    p.H = H; p.Wd = Wd; p.Cin = Cin; p.Ho = H; p.Wo = Wd; p.stride = 1; p.ups = 0;
    p.alpha = alpha; p.flags = flags;
    const unsigned long long ab = ((unsigned long long)B * H * Wd - 1) * ldx + Cin + (unsigned long long)(Wd + 1) * ldx;
    const unsigned long long wb = (unsigned long long)Cout * 9 * Cin;
    if (!fits31(ab) || !fits31(wb)) return SIDLSG_EINVAL;
    p.a_bytes = (unsigned)(ab - (unsigned long long)(Wd + 1) * ldx); p.w_bytes = (unsigned)wb;
    SidlsgTraceScope ts(SIDLSG_FAM_CONV, 2.0 * p.M * (double)p.N * p.K);
    return launch_gemm_mx8<1>(p, (hipStream_t)stream);
}

__global__ __launch_bounds__(256) void cast_fp8_kernel(const bf16* __restrict__ src, unsigned char* __restrict__ dst, size_t n8) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += stride) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(src + i * 8);
        u32x2 q = {cvt4_fp8(bf2f(v[0]), bf2f(v[1]), bf2f(v[2]), bf2f(v[3])), cvt4_fp8(bf2f(v[4]), bf2f(v[5]), bf2f(v[6]), bf2f(v[7]))};
        *reinterpret_cast<u32x2*>(dst + i * 8) = q;
    }
}
int sidlsg_cast_fp8(const void* src_bf16, void* dst_fp8, long long n, void* stream) {
    if (!src_bf16 || !dst_fp8 || n <= 0 || (n & 7)) return SIDLSG_EINVAL;
    const size_t n8 = (size_t)n / 8;
    size_t blocks = (n8 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    SIDLSG_LAUNCH(cast_fp8_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16*)src_bf16, (unsigned char*)dst_fp8, n8);
    return sidlsg_last_error();
}

int sidlsg_conv3x3_fp8w(const void* X, int ldx, const void* W8, const float* wscale, void* Y, int ldc, const float* bias,
                        const void* res, int ldres, const float* rowvec, int ld_rowvec, int B, int H, int Wd, int Cin, int Cout,
                        int stride, int ups, float alpha, int flags, void* stream) {
    if ((stride != 1 && stride != 2) || (Cin & 15) || !wscale) return SIDLSG_EINVAL;
    if (ups && ((H | Wd) & 1)) return SIDLSG_EINVAL;
    GemmParams p{};
    p.A = (const bf16*)X; p.W = (const bf16*)W8; p.wscale = wscale; p.C = Y; p.bias = bias; p.res = (const bf16*)res; p.rowvec = rowvec;
    p.ldrv = ld_rowvec > 0 ? ld_rowvec : Cout;
    p.H = H; p.Wd = Wd; p.Cin = Cin; p.stride = stride; p.ups = ups;
    p.Ho = (H + 2 - 3) / stride + 1; p.Wo = (Wd + 2 - 3) / stride + 1;
    p.M = B * p.Ho * p.Wo; p.N = Cout; p.K = 9 * Cin; p.lda = ldx; p.ldc = ldc; p.ldres = ldres;
    p.rows_per_batch = p.Ho * p.Wo; p.alpha = alpha; p.flags = flags;
    if (int e = check_common(p)) return e;
    const int Hs = ups ? H / 2 : H, Wss = ups ? Wd / 2 : Wd;
    const unsigned long long ab = (((unsigned long long)B * Hs * Wss - 1) * ldx + Cin) * 2ull, wb = (unsigned long long)Cout * 9 * Cin;
    if (!fits31(ab) || !fits31(wb)) return SIDLSG_EINVAL;
    p.a_bytes = (unsigned)ab; p.w_bytes = (unsigned)wb;
    const int tiles = ((p.M + F8_T - 1) / F8_T) * ((p.N + F8_T - 1) / F8_T);
    SIDLSG_LAUNCH((gemm_fp8w_kernel<2>), dim3(tiles), dim3(NTHREADS), 0, (hipStream_t)stream, p);
    return sidlsg_last_error();
}

}  // extern "C"
