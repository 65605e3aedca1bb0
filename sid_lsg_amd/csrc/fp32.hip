// fp32-accurate compute mode for gfx950 (MI355X): the contractions of the SD UNet with fp32 operands, fp32 accumulation
// and fp32 activations, on the f32-input matrix cores (v_mfma_f32_16x16x4_f32: bit-for-bit a k-ordered fmaf chain, at
// the fp32 vector rate = 1/16 of bf16 MFMA).
//
// Purpose (DESIGN.md "fp32 mode"): the reference's default precision is fp32 (training/sid_training_loop.py:205,
// run_sid.sh:63-88) and BASELINE north_star asks for the loss curve within 1e-3 of the reference fp32 CPU path.  The
// bf16 production kernels (gemm.hip / attention.hip) cannot demonstrate that bound; these kernels can.  They are the
// `_f32` entry-point family of include/sidlsg_hip.h, selected per network (HipUNet2DCondition(compute_dtype=float32)).
// Throughput is secondary (simple 64x64 tiles, register-staged loads, no split-K); layouts are the production ones:
// activations NHWC [tokens][channels], conv weights [Cout][3][3][Cin], attention on strided q|k|v views.
//
// MFMA 16x16x4 f32 operand maps (cdna guide section 3): A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15],
// C/D[row = (lane>>4)*4 + reg][col = lane&15].  Everywhere below the A operand carries the index that should end up
// 4-consecutive per lane in the result (output channel / head-dim) and the B operand the pixel / query / key index, so
// results are stored as 16-byte vectors.
#include "common.h"
#include <stdlib.h>

#define MFMA_F32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

enum { F32_SILU = 2, F32_ACCUM = 4 };

struct GemmF32Params {
    const float* A;        // dense [M][lda] or NHWC image [B][Hs][Ws][lda]
    const float* W;        // [N][K]
    float* C;              // [M][ldc]
    const float* bias;     // [N] or null
    const float* res;      // [M][ldres] or null
    const float* rowvec;   // [M / rows_per_batch][ldrv] or null
    int ldrv, M, N, K, lda, ldc, ldres, rows_per_batch;
    int H, Wd, Cin, Ho, Wo, stride, ups;   // conv3x3 (pad 1) geometry, virtual input H x Wd
    float alpha;
    int flags;
};

constexpr int G_BM = 64, G_BN = 64, G_BK = 16, G_LD = G_BK + 2;   // LD = 18: conflict-free ds_read_b32 fragments

// MODE 0: dense rows.  MODE 1: implicit-GEMM conv3x3 (k = tap*Cin + c, Cin % 4 == 0).
template <int MODE>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmF32Params p) {
    __shared__ float Xs[G_BM * G_LD];
    __shared__ float Ws[G_BN * G_LD];
    const int tiles_n = (p.N + G_BN - 1) / G_BN;
    const int m0 = (blockIdx.x / tiles_n) * G_BM, n0 = (blockIdx.x % tiles_n) * G_BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
    const int lrow = tid >> 2, lch = (tid & 3) * 4;          // loader: row of the tile, first of 4 consecutive k
    // loader geometry of this thread's X row
    const int m = m0 + lrow;
    const bool mok = m < p.M;
    int hi0 = 0, wi0 = 0;
    size_t xbase = 0;
    const int Hs = p.ups ? (p.H >> 1) : p.H, Wsrc = p.ups ? (p.Wd >> 1) : p.Wd;
    if (MODE == 0) xbase = (size_t)(mok ? m : 0) * p.lda;
    else {
        const int mm = mok ? m : 0, hw = p.Ho * p.Wo;
        const int b = mm / hw, rem = mm - b * hw, ho = rem / p.Wo, wo = rem - ho * p.Wo;
        xbase = (size_t)b * Hs * Wsrc * p.lda;
        hi0 = ho * p.stride - 1; wi0 = wo * p.stride - 1;
    }
    const int n_ld = n0 + lrow;
    const bool nok = n_ld < p.N;
    auto load_x = [&](int k0) -> f32x4 {
        const int k = k0 + lch;
        if (!mok || k >= p.K) return (f32x4){0, 0, 0, 0};
        if (MODE == 0) return *reinterpret_cast<const f32x4*>(p.A + xbase + k);
        const int tap = k / p.Cin, c = k - tap * p.Cin;
        const int dh = tap / 3, dw = tap - dh * 3;
        int hi = hi0 + dh, wi = wi0 + dw;
        if (hi < 0 || hi >= p.H || wi < 0 || wi >= p.Wd) return (f32x4){0, 0, 0, 0};
        if (p.ups) { hi >>= 1; wi >>= 1; }
        return *reinterpret_cast<const f32x4*>(p.A + xbase + ((size_t)hi * Wsrc + wi) * p.lda + c);
    };
    auto load_w = [&](int k0) -> f32x4 {
        const int k = k0 + lch;
        if (!nok || k >= p.K) return (f32x4){0, 0, 0, 0};
        return *reinterpret_cast<const f32x4*>(p.W + (size_t)n_ld * p.K + k);
    };
    f32x4 acc[2][2];   // [n tile][m tile]
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = (f32x4){0, 0, 0, 0};
    const int nk = (p.K + G_BK - 1) / G_BK;
    f32x4 xr = load_x(0), wr = load_w(0);
    for (int kt = 0; kt < nk; kt++) {
        __syncthreads();                                   // previous tile fully consumed
#pragma unroll
        for (int e = 0; e < 4; e++) { Xs[lrow * G_LD + lch + e] = xr[e]; Ws[lrow * G_LD + lch + e] = wr[e]; }
        __syncthreads();
        if (kt + 1 < nk) { xr = load_x((kt + 1) * G_BK); wr = load_w((kt + 1) * G_BK); }
#pragma unroll
        for (int k4 = 0; k4 < G_BK / 4; k4++) {
            float fw[2], fx[2];
#pragma unroll
            for (int i = 0; i < 2; i++) fw[i] = Ws[(wn0 + i * 16 + li) * G_LD + k4 * 4 + lg];
#pragma unroll
            for (int j = 0; j < 2; j++) fx[j] = Xs[(wm0 + j * 16 + li) * G_LD + k4 * 4 + lg];
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) acc[i][j] = MFMA_F32(fw[i], fx[j], acc[i][j]);
        }
    }
    // epilogue: lane holds, for pixel row mm = .. + li, channels n .. n+3
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int mm = m0 + wm0 + j * 16 + li;
        if (mm >= p.M) continue;
        const float* rv = p.rowvec ? p.rowvec + (size_t)(mm / p.rows_per_batch) * p.ldrv : nullptr;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int n = n0 + wn0 + i * 16 + lg * 4;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                if (n + r >= p.N) break;
                float x = acc[i][j][r] * p.alpha;
                if (p.bias) x += p.bias[n + r];
                if (rv) x += rv[n + r];
                if (p.res) x += p.res[(size_t)mm * p.ldres + n + r];
                if (p.flags & F32_SILU) x = silu_f(x);
                float* c = p.C + (size_t)mm * p.ldc + n + r;
                *c = (p.flags & F32_ACCUM) ? *c + x : x;
            }
        }
    }
}

template <int MODE>
static int launch_gemm_f32(const GemmF32Params& p, hipStream_t s) {
    const int tiles = ((p.M + G_BM - 1) / G_BM) * ((p.N + G_BN - 1) / G_BN);
    hipLaunchKernelGGL((gemm_f32_kernel<MODE>), dim3(tiles), dim3(256), 0, s, p);
    return sidlsg_last_error();
}

// ---------------------------------------------------------------------------------------------
// weight gradient dW[N][K] += dY[M][N]^T A[M][K]: 64 (n) x 64 (k) tile, pixel contraction in LDS stages of 16 rows,
// pixel range split over grid.y (fp32 atomics when split).
struct WgradF32Params {
    const float* dY;  // [M][ldy]
    const float* A;   // dense [M][lda] or NHWC image
    float* dW;        // [N][K]
    int M, N, K, ldy, lda;
    int H, Wd, Cin, Ho, Wo, stride, ups;
    int m_per_split, nsplits;
};
constexpr int WG_LD = 64 + 16;     // row stride: the two k-groups of a half-wave land on disjoint banks

template <int MODE>
__global__ __launch_bounds__(256) void wgrad_f32_kernel(WgradF32Params p) {
    __shared__ float Ys[16 * WG_LD];
    __shared__ float Xs[16 * WG_LD];
    const int tiles_k = (p.K + 63) / 64;
    const int n0 = (blockIdx.x / tiles_k) * 64, k0 = (blockIdx.x % tiles_k) * 64;
    const int mbeg = blockIdx.y * p.m_per_split, mend = min(p.M, mbeg + p.m_per_split);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int wn0 = (wave >> 1) * 32, wk0 = (wave & 1) * 32;
    const int lrow = tid >> 4, lch = (tid & 15) * 4;
    const int Hs = p.ups ? (p.H >> 1) : p.H, Wsrc = p.ups ? (p.Wd >> 1) : p.Wd;
    const int kA = k0 + lch, nY = n0 + lch;
    int cA = kA, dh = 0, dw = 0;
    if (MODE == 1 && kA < p.K) { const int tap = kA / p.Cin; cA = kA - tap * p.Cin; dh = tap / 3; dw = tap - dh * 3; }
    auto load_y = [&](int mb) -> f32x4 {
        const int m = mb + lrow;
        f32x4 v = {0, 0, 0, 0};
        if (m >= mend || nY >= p.N) return v;
        if (nY + 4 <= p.N) return *reinterpret_cast<const f32x4*>(p.dY + (size_t)m * p.ldy + nY);
        for (int e = 0; e < 4 && nY + e < p.N; e++) v[e] = p.dY[(size_t)m * p.ldy + nY + e];
        return v;
    };
    auto load_x = [&](int mb) -> f32x4 {
        const int m = mb + lrow;
        if (m >= mend || kA >= p.K) return (f32x4){0, 0, 0, 0};
        if (MODE == 0) return *reinterpret_cast<const f32x4*>(p.A + (size_t)m * p.lda + kA);
        const int hw = p.Ho * p.Wo;
        const int b = m / hw, rem = m - b * hw, ho = rem / p.Wo, wo = rem - ho * p.Wo;
        int hi = ho * p.stride - 1 + dh, wi = wo * p.stride - 1 + dw;
        if (hi < 0 || hi >= p.H || wi < 0 || wi >= p.Wd) return (f32x4){0, 0, 0, 0};
        if (p.ups) { hi >>= 1; wi >>= 1; }
        return *reinterpret_cast<const f32x4*>(p.A + (((size_t)b * Hs + hi) * Wsrc + wi) * p.lda + cA);
    };
    f32x4 acc[2][2];   // [n tile][k tile]
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = (f32x4){0, 0, 0, 0};
    if (mbeg >= mend) return;
    f32x4 yr = load_y(mbeg), xr = load_x(mbeg);
    for (int mb = mbeg; mb < mend; mb += 16) {
        __syncthreads();
        *reinterpret_cast<f32x4*>(&Ys[lrow * WG_LD + lch]) = yr;
        *reinterpret_cast<f32x4*>(&Xs[lrow * WG_LD + lch]) = xr;
        __syncthreads();
        if (mb + 16 < mend) { yr = load_y(mb + 16); xr = load_x(mb + 16); }
#pragma unroll
        for (int m4 = 0; m4 < 4; m4++) {
            float fy[2], fx[2];
#pragma unroll
            for (int i = 0; i < 2; i++) fy[i] = Ys[(m4 * 4 + lg) * WG_LD + wn0 + i * 16 + li];
#pragma unroll
            for (int j = 0; j < 2; j++) fx[j] = Xs[(m4 * 4 + lg) * WG_LD + wk0 + j * 16 + li];
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) acc[i][j] = MFMA_F32(fy[i], fx[j], acc[i][j]);
        }
    }
    // acc[i][j][r]: n = n0 + wn0 + 16 i + lg*4 + r ; k = k0 + wk0 + 16 j + li
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int k = k0 + wk0 + 16 * j + li;
            if (k >= p.K) continue;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int n = n0 + wn0 + 16 * i + lg * 4 + r;
                if (n >= p.N) continue;
                float* d = p.dW + (size_t)n * p.K + k;
                if (p.nsplits == 1) *d += acc[i][j][r]; else unsafeAtomicAdd(d, acc[i][j][r]);
            }
        }
}

template <int MODE>
static int launch_wgrad_f32(WgradF32Params p, hipStream_t s) {
    const int tiles = ((p.N + 63) / 64) * ((p.K + 63) / 64);
    int splits = (512 + tiles - 1) / tiles;
    const int max_splits = (p.M + 255) / 256;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    int mps = (p.M + splits - 1) / splits;
    mps = (mps + 15) / 16 * 16;
    splits = (p.M + mps - 1) / mps;
    p.m_per_split = mps; p.nsplits = splits;
    hipLaunchKernelGGL((wgrad_f32_kernel<MODE>), dim3(tiles, splits), dim3(256), 0, s, p);
    return sidlsg_last_error();
}

// ---------------------------------------------------------------------------------------------
// Attention, fp32.  Same math and memory contract as attention.hip (strided q|k|v views, LSE in the log2 domain,
// backward = dQ kernel (publishes delta) + dK/dV kernel), computed with f32 MFMAs:
//   S^T[key][q]  = sum_d K[key][d] Q[q][d]          A = K rows (LDS), B = Q (registers)
//   O^T[d][q]   += sum_key V[key][d] P^T[key][q]    A = V^T read straight from the row-major LDS tile (lane = d),
//                                                   B = the score accumulator itself: step r of key tile kt contracts
//                                                   keys {kt*16 + 4g + r}, exactly register r of lane group g
// so P never leaves registers and nothing is transposed.  Softmax statistics are per query = per lane column.
struct AttnF32Params {
    const float *Q, *K, *V, *O, *dO;
    float *Out, *dQ, *dK, *dV;
    float* LSE;
    float* delta;
    int B, H, Nq, Nk, D;
    int ldq, ldk, ldv, ldo;
    long long bsq, bsk, bsv, bso;
    float scale2, scale;
};
constexpr int AF_KT = 32;      // keys (or queries) per LDS tile

// cooperative load of a [rows][D] fp32 tile (rows past nrows -> zeros) into LDS with row stride LD
DEVFN void af_load_tile(float* dst, int LD, const float* src, long long ld, int row0, int nrows, int D, int rows) {
    const int c4 = D >> 2;
    for (int idx = threadIdx.x; idx < rows * c4; idx += 256) {
        const int r = idx / c4, c = (idx - r * c4) * 4;
        f32x4 v = {0, 0, 0, 0};
        if (row0 + r < nrows) v = *reinterpret_cast<const f32x4*>(src + (long long)(row0 + r) * ld + c);
        float* d = dst + r * LD + c;
        d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    }
}

// MODE 0: forward (O, LSE).  MODE 1: dQ (+ delta).  DP = D rounded up to a multiple of 16.
template <int DP, int MODE>
__global__ __launch_bounds__(256) void attn_f32_q_kernel(AttnF32Params p) {
    constexpr int DT = DP / 16, DS = DP / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int LDK = p.D + 2, LDV = p.D + 4;
    float* Ks = smem;                       // [AF_KT][LDK]
    float* Vs = smem + AF_KT * LDK;         // [AF_KT][LDV]
    const int b = blockIdx.z, h = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    const int q = (blockIdx.x * 4 + wave) * 16 + li;
    const bool qok = q < p.Nq;
    const int nds = p.D >> 2;               // d-steps of 4
    const float* Qrow = p.Q + b * p.bsq + (long long)(qok ? q : 0) * p.ldq + (long long)h * p.D;
    float qf[DS], dof[MODE == 1 ? DS : 1];
    float lse = 0.f, dl = 0.f;
#pragma unroll
    for (int s = 0; s < DS; s++) qf[s] = (qok && s < nds) ? Qrow[4 * s + lg] : 0.f;
    if (MODE == 1) {
        const float* dOrow = p.dO + b * p.bso + (long long)(qok ? q : 0) * p.ldo + (long long)h * p.D;
        const float* Orow = p.O + b * p.bso + (long long)(qok ? q : 0) * p.ldo + (long long)h * p.D;
        float dsum = 0.f;
#pragma unroll
        for (int s = 0; s < DS; s++) {
            dof[s] = (qok && s < nds) ? dOrow[4 * s + lg] : 0.f;
            if (qok && s < nds) dsum += dof[s] * Orow[4 * s + lg];
        }
        dsum += __shfl_xor(dsum, 16, 64);
        dsum += __shfl_xor(dsum, 32, 64);
        dl = dsum;
        lse = qok ? p.LSE[((long long)b * p.H + h) * p.Nq + q] : 0.f;
        if (qok && lg == 0) p.delta[((long long)b * p.H + h) * p.Nq + q] = dsum;
    }
    f32x4 o[DT];
#pragma unroll
    for (int i = 0; i < DT; i++) o[i] = (f32x4){0, 0, 0, 0};
    float mrun = -INFINITY, lrun = 0.f;
    const float* Kb = p.K + b * p.bsk + (long long)h * p.D;
    const float* Vb = p.V + b * p.bsv + (long long)h * p.D;
    for (int k0 = 0; k0 < p.Nk; k0 += AF_KT) {
        __syncthreads();
        af_load_tile(Ks, LDK, Kb, p.ldk, k0, p.Nk, p.D, AF_KT);
        af_load_tile(Vs, LDV, Vb, p.ldv, k0, p.Nk, p.D, AF_KT);
        __syncthreads();
        f32x4 st[2];
#pragma unroll
        for (int kt = 0; kt < 2; kt++) {
            f32x4 a = {0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < DS; s++)
                if (s < nds) a = MFMA_F32(Ks[(kt * 16 + li) * LDK + 4 * s + lg], qf[s], a);
            st[kt] = a;
        }
        if (MODE == 0) {
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 2; kt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const bool kok = k0 + kt * 16 + lg * 4 + r < p.Nk;
                    st[kt][r] = kok ? st[kt][r] * p.scale2 : -INFINITY;
                    mx = fmaxf(mx, st[kt][r]);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mn = fmaxf(mrun, mx);
            const float alpha = mrun == -INFINITY ? 0.f : exp2f(mrun - mn);
            mrun = mn;
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; kt++)
#pragma unroll
                for (int r = 0; r < 4; r++) { const float e = exp2f(st[kt][r] - mn); st[kt][r] = e; sum += e; }
            lrun = lrun * alpha + sum;      // per-lane partial (this lane group's keys); groups are summed in the epilogue
#pragma unroll
            for (int i = 0; i < DT; i++) o[i] *= alpha;
        } else {
#pragma unroll
            for (int kt = 0; kt < 2; kt++) {
                f32x4 dp = {0, 0, 0, 0};
#pragma unroll
                for (int s = 0; s < DS; s++)
                    if (s < nds) dp = MFMA_F32(Vs[(kt * 16 + li) * LDV + 4 * s + lg], dof[s], dp);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const bool kok = k0 + kt * 16 + lg * 4 + r < p.Nk;
                    const float pr = kok ? exp2f(st[kt][r] * p.scale2 - lse) : 0.f;
                    st[kt][r] = pr * (dp[r] - dl);
                }
            }
        }
        const float* T2 = MODE == 0 ? Vs : Ks;
        const int LD2 = MODE == 0 ? LDV : LDK;
#pragma unroll
        for (int kt = 0; kt < 2; kt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float* row = T2 + (kt * 16 + lg * 4 + r) * LD2;
#pragma unroll
                for (int dt = 0; dt < DT; dt++) {
                    const int d = dt * 16 + li;
                    const float a = d < p.D ? row[d] : 0.f;
                    o[dt] = MFMA_F32(a, st[kt][r], o[dt]);
                }
            }
    }
    float inv = MODE == 1 ? p.scale : 1.f;
    if (MODE == 0) {
        float lt = lrun;
        lt += __shfl_xor(lt, 16, 64);
        lt += __shfl_xor(lt, 32, 64);
        inv = lt > 0.f ? 1.f / lt : 0.f;
        if (qok && lg == 0 && p.LSE) p.LSE[((long long)b * p.H + h) * p.Nq + q] = mrun + log2f(lt);
    }
    if (!qok) return;
    float* dst = (MODE == 0 ? p.Out + b * p.bso + (long long)q * p.ldo : p.dQ + b * p.bsq + (long long)q * p.ldq) + (long long)h * p.D;
#pragma unroll
    for (int dt = 0; dt < DT; dt++) {
        const int d = dt * 16 + lg * 4;
        if (d + 4 <= p.D) *reinterpret_cast<f32x4*>(dst + d) = o[dt] * inv;
    }
}

// dK, dV: block = 4 waves x 16 keys, loop over 32-query tiles (Q and dO tiles in LDS).
template <int DP>
__global__ __launch_bounds__(256) void attn_f32_dkdv_kernel(AttnF32Params p) {
    constexpr int DT = DP / 16, DS = DP / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int LDQ = p.D + 2;
    float* Qs = smem;                        // [AF_KT][LDQ]
    float* dOs = smem + AF_KT * LDQ;         // [AF_KT][LDQ]
    float* lse_s = dOs + AF_KT * LDQ;        // [AF_KT]
    float* dl_s = lse_s + AF_KT;             // [AF_KT]
    const int b = blockIdx.z, h = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    const int key = (blockIdx.x * 4 + wave) * 16 + li;
    const bool kok = key < p.Nk;
    const int nds = p.D >> 2;
    const float* Krow = p.K + b * p.bsk + (long long)(kok ? key : 0) * p.ldk + (long long)h * p.D;
    const float* Vrow = p.V + b * p.bsv + (long long)(kok ? key : 0) * p.ldv + (long long)h * p.D;
    float kf[DS], vf[DS];
#pragma unroll
    for (int s = 0; s < DS; s++) {
        kf[s] = (kok && s < nds) ? Krow[4 * s + lg] : 0.f;
        vf[s] = (kok && s < nds) ? Vrow[4 * s + lg] : 0.f;
    }
    f32x4 dk[DT], dv[DT];
#pragma unroll
    for (int i = 0; i < DT; i++) { dk[i] = (f32x4){0, 0, 0, 0}; dv[i] = (f32x4){0, 0, 0, 0}; }
    const float* Qb = p.Q + b * p.bsq + (long long)h * p.D;
    const float* dOb = p.dO + b * p.bso + (long long)h * p.D;
    const float* LSEb = p.LSE + ((long long)b * p.H + h) * p.Nq;
    const float* DLb = p.delta + ((long long)b * p.H + h) * p.Nq;
    for (int q0 = 0; q0 < p.Nq; q0 += AF_KT) {
        __syncthreads();
        af_load_tile(Qs, LDQ, Qb, p.ldq, q0, p.Nq, p.D, AF_KT);
        af_load_tile(dOs, LDQ, dOb, p.ldo, q0, p.Nq, p.D, AF_KT);
        if (threadIdx.x < AF_KT) {
            const int q = q0 + threadIdx.x;
            lse_s[threadIdx.x] = q < p.Nq ? LSEb[q] : INFINITY;     // padded queries: p = exp2(-inf) = 0
            dl_s[threadIdx.x] = q < p.Nq ? DLb[q] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int qt = 0; qt < 2; qt++) {
            f32x4 sacc = {0, 0, 0, 0}, dp = {0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < DS; s++)
                if (s < nds) {
                    sacc = MFMA_F32(Qs[(qt * 16 + li) * LDQ + 4 * s + lg], kf[s], sacc);     // S[q][key]: rows q = 4g + r, col key
                    dp = MFMA_F32(dOs[(qt * 16 + li) * LDQ + 4 * s + lg], vf[s], dp);
                }
            float pr[4], ds[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int ql = qt * 16 + lg * 4 + r;
                pr[r] = exp2f(sacc[r] * p.scale2 - lse_s[ql]);
                ds[r] = pr[r] * (dp[r] - dl_s[ql]);
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float* qrow = Qs + (qt * 16 + lg * 4 + r) * LDQ;
                const float* dorow = dOs + (qt * 16 + lg * 4 + r) * LDQ;
#pragma unroll
                for (int dt = 0; dt < DT; dt++) {
                    const int d = dt * 16 + li;
                    const float a_do = d < p.D ? dorow[d] : 0.f, a_q = d < p.D ? qrow[d] : 0.f;
                    dv[dt] = MFMA_F32(a_do, pr[r], dv[dt]);
                    dk[dt] = MFMA_F32(a_q, ds[r], dk[dt]);
                }
            }
        }
    }
    if (!kok) return;
    float* dKp = p.dK + b * p.bsk + (long long)key * p.ldk + (long long)h * p.D;
    float* dVp = p.dV + b * p.bsv + (long long)key * p.ldv + (long long)h * p.D;
#pragma unroll
    for (int dt = 0; dt < DT; dt++) {
        const int d = dt * 16 + lg * 4;
        if (d + 4 <= p.D) {
            *reinterpret_cast<f32x4*>(dKp + d) = dk[dt] * p.scale;
            *reinterpret_cast<f32x4*>(dVp + d) = dv[dt];
        }
    }
}

template <int DP>
static int launch_attn_f32(const AttnF32Params& p, int mode, hipStream_t s) {
    const size_t lds_q = (size_t)AF_KT * (2 * p.D + 6) * sizeof(float);
    const size_t lds_k = (size_t)AF_KT * (2 * (p.D + 2) + 2) * sizeof(float);
    if (mode == 0) hipLaunchKernelGGL((attn_f32_q_kernel<DP, 0>), dim3((p.Nq + 63) / 64, p.H, p.B), dim3(256), lds_q, s, p);
    else if (mode == 1) hipLaunchKernelGGL((attn_f32_q_kernel<DP, 1>), dim3((p.Nq + 63) / 64, p.H, p.B), dim3(256), lds_q, s, p);
    else hipLaunchKernelGGL((attn_f32_dkdv_kernel<DP>), dim3((p.Nk + 63) / 64, p.H, p.B), dim3(256), lds_k, s, p);
    return sidlsg_last_error();
}
static int dispatch_attn_f32(const AttnF32Params& p, int mode, hipStream_t s) {
    if (p.D % 4 || p.D <= 0 || p.D > 160) return SIDLSG_EINVAL;
    switch ((p.D + 15) / 16 * 16) {
        case 16: return launch_attn_f32<16>(p, mode, s);
        case 32: return launch_attn_f32<32>(p, mode, s);
        case 48: return launch_attn_f32<48>(p, mode, s);
        case 64: return launch_attn_f32<64>(p, mode, s);
        case 80: return launch_attn_f32<80>(p, mode, s);
        case 96: return launch_attn_f32<96>(p, mode, s);
        case 128: return launch_attn_f32<128>(p, mode, s);
        case 160: return launch_attn_f32<160>(p, mode, s);
    }
    return SIDLSG_EINVAL;
}

extern "C" {

int sidlsg_gemm_f32(const void* A, int lda, const void* W, void* C, int ldc, const float* bias, const void* res, int ldres,
                    const float* rowvec, int ld_rowvec, int rows_per_batch, int M, int N, int K, float alpha, int flags,
                    void* stream) {
    if (M <= 0 || N <= 0 || K <= 0 || (K & 3) || (lda & 3) || !A || !W || !C) return SIDLSG_EINVAL;
    if (rowvec && rows_per_batch <= 0) return SIDLSG_EINVAL;
    GemmF32Params p{};
    p.A = (const float*)A; p.W = (const float*)W; p.C = (float*)C; p.bias = bias; p.res = (const float*)res; p.rowvec = rowvec;
    p.ldrv = ld_rowvec > 0 ? ld_rowvec : N;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldc = ldc; p.ldres = ldres; p.rows_per_batch = rows_per_batch;
    p.alpha = alpha; p.flags = flags;
    return launch_gemm_f32<0>(p, (hipStream_t)stream);
}

int sidlsg_conv3x3_f32(const void* X, int ldx, const void* W, void* Y, int ldc, const float* bias, const void* res, int ldres,
                       const float* rowvec, int ld_rowvec, int B, int H, int Wd, int Cin, int Cout, int stride, int ups,
                       float alpha, int flags, void* stream) {
    if ((stride != 1 && stride != 2) || (Cin & 3) || (ldx & 3) || !X || !W || !Y) return SIDLSG_EINVAL;
    if (ups && ((H | Wd) & 1)) return SIDLSG_EINVAL;
    GemmF32Params p{};
    p.A = (const float*)X; p.W = (const float*)W; p.C = (float*)Y; p.bias = bias; p.res = (const float*)res; p.rowvec = rowvec;
    p.ldrv = ld_rowvec > 0 ? ld_rowvec : Cout;
    p.H = H; p.Wd = Wd; p.Cin = Cin; p.stride = stride; p.ups = ups;
    p.Ho = (H + 2 - 3) / stride + 1; p.Wo = (Wd + 2 - 3) / stride + 1;
    p.M = B * p.Ho * p.Wo; p.N = Cout; p.K = 9 * Cin; p.lda = ldx; p.ldc = ldc; p.ldres = ldres;
    p.rows_per_batch = p.Ho * p.Wo; p.alpha = alpha; p.flags = flags;
    if (p.M <= 0 || p.N <= 0) return SIDLSG_EINVAL;
    return launch_gemm_f32<1>(p, (hipStream_t)stream);
}

int sidlsg_wgrad_f32(const void* dY, int ldy, const void* A, int lda, float* dW, float* dBias, int M, int N, int K,
                     void* stream) {
    if (M <= 0 || N <= 0 || K <= 0 || (K & 3) || (lda & 3) || (ldy & 3) || !dY || !A || !dW || dBias) return SIDLSG_EINVAL;
    WgradF32Params p{};
    p.dY = (const float*)dY; p.A = (const float*)A; p.dW = dW; p.M = M; p.N = N; p.K = K; p.ldy = ldy; p.lda = lda;
    return launch_wgrad_f32<0>(p, (hipStream_t)stream);
}

int sidlsg_conv3x3_wgrad_f32(const void* dY, int ldy, const void* X, int ldx, float* dW, float* dBias, int B, int H, int Wd,
                             int Cin, int Cout, int stride, int ups, void* stream) {
    if ((stride != 1 && stride != 2) || (Cin & 3) || (ldx & 3) || (ldy & 3) || !dY || !X || !dW || dBias) return SIDLSG_EINVAL;
    WgradF32Params p{};
    p.dY = (const float*)dY; p.A = (const float*)X; p.dW = dW; p.ldy = ldy; p.lda = ldx;
    p.H = H; p.Wd = Wd; p.Cin = Cin; p.stride = stride; p.ups = ups;
    p.Ho = (H + 2 - 3) / stride + 1; p.Wo = (Wd + 2 - 3) / stride + 1;
    p.M = B * p.Ho * p.Wo; p.N = Cout; p.K = 9 * Cin;
    if (p.M <= 0) return SIDLSG_EINVAL;
    return launch_wgrad_f32<1>(p, (hipStream_t)stream);
}

int sidlsg_attn_fwd_f32(const void* Q, const void* K, const void* V, void* O, float* LSE, int B, int H, int Nq, int Nk, int D,
                        int ldq, int ldk, int ldv, int ldo, long long bsq, long long bsk, long long bsv, long long bso,
                        void* stream) {
    if (((ldq | ldk | ldv | ldo) & 3) || ((bsq | bsk | bsv | bso) & 3)) return SIDLSG_EINVAL;
    AttnF32Params p{};
    p.Q = (const float*)Q; p.K = (const float*)K; p.V = (const float*)V; p.Out = (float*)O; p.LSE = LSE;
    p.B = B; p.H = H; p.Nq = Nq; p.Nk = Nk; p.D = D; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
    p.bsq = bsq; p.bsk = bsk; p.bsv = bsv; p.bso = bso;
    p.scale = 1.0f / sqrtf((float)D); p.scale2 = p.scale * 1.4426950408889634f;
    return dispatch_attn_f32(p, 0, (hipStream_t)stream);
}

int sidlsg_attn_bwd_f32(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* LSE, void* dQ,
                        void* dK, void* dV, float* delta, int B, int H, int Nq, int Nk, int D, int ldq, int ldk, int ldv,
                        int ldo, long long bsq, long long bsk, long long bsv, long long bso, void* stream) {
    if (((ldq | ldk | ldv | ldo) & 3) || ((bsq | bsk | bsv | bso) & 3)) return SIDLSG_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    AttnF32Params p{};
    p.Q = (const float*)Q; p.K = (const float*)K; p.V = (const float*)V; p.O = (const float*)O; p.dO = (const float*)dO;
    p.dQ = (float*)dQ; p.dK = (float*)dK; p.dV = (float*)dV; p.LSE = const_cast<float*>(LSE); p.delta = delta;
    p.B = B; p.H = H; p.Nq = Nq; p.Nk = Nk; p.D = D; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
    p.bsq = bsq; p.bsk = bsk; p.bsv = bsv; p.bso = bso;
    p.scale = 1.0f / sqrtf((float)D); p.scale2 = p.scale * 1.4426950408889634f;
    if (int e = dispatch_attn_f32(p, 1, s)) return e;
    if (!dK && !dV) return SIDLSG_OK;          // query gradient only (see sidlsg_attn_bwd)
    if (!dK || !dV) return SIDLSG_EINVAL;
    return dispatch_attn_f32(p, 2, s);
}

}  // extern "C"
