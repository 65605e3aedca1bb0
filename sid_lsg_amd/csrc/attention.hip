// Flash-style multi-head attention for gfx950 (self-attention N in {4096,1024,256,64}, d in
// {40,80,160} for SD1.5 / 64 for SD2.1; cross-attention with 77 keys), forward and backward.
// SURVEY.md section 8 row A5 ("QKV-attention MFMA-tiled with LDS staging").
//
// Orientation trick (no LDS round trip for P): the score tile is computed TRANSPOSED,
//   S^T[key][query] = mfma(A = K rows, B = Q rows)
// so the accumulator layout (lane: column=query, 4 consecutive keys) is exactly the B-operand
// layout of the next MFMA whose contraction runs over keys:
//   O^T[d][query] += mfma(A = V^T (LDS transpose-read of the row-major V tile), B = P^T).
// Two 16-key tiles are packed into one K=32 MFMA.  Softmax statistics are per query = per lane
// (replicated over the 4 lane groups), so the online rescale is a per-lane scalar.
// Head dims that are not multiples of 32 use one trailing 16-wide MFMA (d=40 -> 32+16 padded with
// zeros, d=80 -> 64+16); nothing is padded in HBM.
//
// Backward = two kernels without atomics:
//   attn_dq   : same loop as forward (per query block, over key tiles): dQ^T = K^T dS^T
//   attn_dkdv : per key block, over query tiles, in the non-transposed orientation
//               S[query][key] so that P / dS are B operands of the contractions over queries.
#include "common.h"

struct AttnParams {
    const bf16 *Q, *K, *V, *dO;
    bf16 *O, *dQ, *dK, *dV;
    float* LSE;          // [B][H][Nq], log2 domain: m + log2(l)
    const float* delta;  // [B][H][Nq] = rowsum(dO * O)
    int B, H, Nq, Nk, D;
    int ldq, ldk, ldv, ldo;        // token strides (elements)
    long long bsq, bsk, bsv, bso;  // batch strides (elements)
    float scale2;                  // d^-0.5 * log2(e)
    float scale;                   // d^-0.5
};

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

template <int DP>
struct Frag {
    static constexpr int N32 = DP / 32;
    static constexpr bool TAIL = (DP % 32) != 0;
    bf16x8 w[N32 > 0 ? N32 : 1];
    s16x4 t;
};

template <int DP>
DEVFN void frag_from_lds(Frag<DP>& f, const bf16* row, int lg) {
#pragma unroll
    for (int s = 0; s < Frag<DP>::N32; s++) f.w[s] = *reinterpret_cast<const bf16x8*>(row + s * 32 + lg * 8);
    if (Frag<DP>::TAIL) f.t = *reinterpret_cast<const s16x4*>(row + Frag<DP>::N32 * 32 + lg * 4);
}
template <int DP>
DEVFN void frag_from_global(Frag<DP>& f, const bf16* row, int lg, int D, bool ok) {
#pragma unroll
    for (int s = 0; s < Frag<DP>::N32; s++) {
        const int d0 = s * 32 + lg * 8;
        f.w[s] = (ok && d0 + 8 <= D) ? ld8(row + d0) : zero8();
    }
    if (Frag<DP>::TAIL) {
        const int d0 = Frag<DP>::N32 * 32 + lg * 4;
        s16x4 z = {0, 0, 0, 0};
        f.t = (ok && d0 + 4 <= D) ? *reinterpret_cast<const s16x4*>(row + d0) : z;
    }
}
// NOTE: the K=16 tail gets its OWN accumulator chain and is added with VALU.  Feeding the result of a
// 16x16x32 MFMA directly into the SrcC of a 16x16x16 MFMA (as hipcc 7.2 schedules it, back to back, no
// wait states) produced wrong sums on gfx950 (tests: D=40/80 failed while D=16 and D=32/64/160 passed).
template <int DP>
DEVFN f32x4 mma_d(f32x4 acc, const Frag<DP>& a, const Frag<DP>& b) {
#pragma unroll
    for (int s = 0; s < Frag<DP>::N32; s++) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.w[s], b.w[s], acc, 0, 0, 0);
    if (Frag<DP>::TAIL) {
        const f32x4 tail = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a.t, b.t, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        if (Frag<DP>::N32 > 0) acc += tail; else acc = tail;
    }
    return acc;
}

// A operand = X^T for a row-major LDS tile X[row][LD]: rows r0 + {4g..4g+3} and r0 + 16 + {4g..4g+3},
// columns c0..c0+15 -> lane i gets column c0+i, the 8 rows in the order that matches pack_p().
DEVFN bf16x8 tr_frag32(const bf16* tile, int LD, int r0, int c0, int li, int lg) {
    const bf16* p0 = tile + (r0 + 4 * lg + (li >> 2)) * LD + c0 + (li & 3) * 4;
    const bf16* p1 = p0 + 16 * LD;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p1);
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}
// two accumulator tiles (rows 0-15 and 16-31 of the contraction index) -> one K=32 B operand
DEVFN bf16x8 pack_p(f32x4 a, f32x4 b) {
    bf16x8 o = {f2bf(a[0]), f2bf(a[1]), f2bf(a[2]), f2bf(a[3]), f2bf(b[0]), f2bf(b[1]), f2bf(b[2]), f2bf(b[3])};
    return o;
}

template <int DP>
DEVFN void load_tile_lds(bf16* dst, int LD, const bf16* src, long long ld, int row0, int nrows, int tile_rows, int D) {
    constexpr int C8 = DP / 8;
    for (int idx = threadIdx.x; idx < tile_rows * C8; idx += blockDim.x) {
        const int r = idx / C8, c = (idx - r * C8) * 8;
        const bool ok = (row0 + r) < nrows && c < D;
        st8(dst + r * LD + c, ok ? ld8(src + (long long)(row0 + r) * ld + c) : zero8());
    }
}

constexpr int AT_KT = 64;   // keys per LDS tile
constexpr int AT_QB = 128;  // queries per block (4 waves x 32)

// MODE 0: forward (O, LSE).  MODE 1: dQ.
template <int DP, int MODE>
__global__ __launch_bounds__(256) void attn_q_kernel(AttnParams p) {
    constexpr int LD = DP + 8;
    constexpr int DT = DP / 16;
    __shared__ __attribute__((aligned(16))) bf16 Ks[AT_KT * LD];
    __shared__ __attribute__((aligned(16))) bf16 Vs[AT_KT * LD];
    const int b = blockIdx.z, h = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    const int q0 = blockIdx.x * AT_QB + wave * 32;
    const bf16* Qb = p.Q + b * p.bsq + (long long)h * p.D;
    const bf16* Kb = p.K + b * p.bsk + (long long)h * p.D;
    const bf16* Vb = p.V + b * p.bsv + (long long)h * p.D;

    Frag<DP> fq[2], fdo[2];
    float lse[2] = {0.f, 0.f}, dl[2] = {0.f, 0.f};
#pragma unroll
    for (int qt = 0; qt < 2; qt++) {
        const int q = q0 + qt * 16 + li;
        const bool ok = q < p.Nq;
        frag_from_global<DP>(fq[qt], Qb + (long long)(ok ? q : 0) * p.ldq, lg, p.D, ok);
        if (MODE == 1) {
            const bf16* dOb = p.dO + b * p.bso + (long long)h * p.D;
            frag_from_global<DP>(fdo[qt], dOb + (long long)(ok ? q : 0) * p.ldo, lg, p.D, ok);
            lse[qt] = ok ? p.LSE[((long long)b * p.H + h) * p.Nq + q] : 0.f;
            dl[qt] = ok ? p.delta[((long long)b * p.H + h) * p.Nq + q] : 0.f;
        }
    }
    f32x4 o[DT][2];
#pragma unroll
    for (int i = 0; i < DT; i++) { o[i][0] = (f32x4){0, 0, 0, 0}; o[i][1] = (f32x4){0, 0, 0, 0}; }
    float m[2] = {-INFINITY, -INFINITY}, l[2] = {0.f, 0.f};

    for (int k0 = 0; k0 < p.Nk; k0 += AT_KT) {
        __syncthreads();
        load_tile_lds<DP>(Ks, LD, Kb, p.ldk, k0, p.Nk, AT_KT, p.D);
        load_tile_lds<DP>(Vs, LD, Vb, p.ldv, k0, p.Nk, AT_KT, p.D);
        __syncthreads();
        f32x4 s[4][2];
#pragma unroll
        for (int kt = 0; kt < 4; kt++) {
            Frag<DP> fk;
            frag_from_lds<DP>(fk, Ks + (kt * 16 + li) * LD, lg);
#pragma unroll
            for (int qt = 0; qt < 2; qt++) s[kt][qt] = mma_d<DP>((f32x4){0, 0, 0, 0}, fk, fq[qt]);
        }
        // scores -> probabilities (keys of this lane: k0 + kt*16 + lg*4 + r)
#pragma unroll
        for (int qt = 0; qt < 2; qt++) {
            if (MODE == 0) {
                float mx = -INFINITY;
#pragma unroll
                for (int kt = 0; kt < 4; kt++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int key = k0 + kt * 16 + lg * 4 + r;
                        const float v = key < p.Nk ? s[kt][qt][r] * p.scale2 : -INFINITY;
                        s[kt][qt][r] = v;
                        mx = fmaxf(mx, v);
                    }
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const float mn = fmaxf(m[qt], mx);
                const float msafe = mn == -INFINITY ? 0.f : mn;
                const float alpha = exp2f(m[qt] - msafe);
                float sum = 0.f;
#pragma unroll
                for (int kt = 0; kt < 4; kt++)
#pragma unroll
                    for (int r = 0; r < 4; r++) { const float e = exp2f(s[kt][qt][r] - msafe); s[kt][qt][r] = e; sum += e; }
                l[qt] = l[qt] * alpha + sum;
                m[qt] = mn;
#pragma unroll
                for (int i = 0; i < DT; i++) o[i][qt] *= alpha;
            } else {
#pragma unroll
                for (int kt = 0; kt < 4; kt++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int key = k0 + kt * 16 + lg * 4 + r;
                        s[kt][qt][r] = key < p.Nk ? exp2f(s[kt][qt][r] * p.scale2 - lse[qt]) : 0.f;
                    }
            }
        }
        if (MODE == 1) {
            // dP^T = V dO^T ; dS^T = P^T * (dP^T - delta) * scale   (overwrites s)
#pragma unroll
            for (int kt = 0; kt < 4; kt++) {
                Frag<DP> fv;
                frag_from_lds<DP>(fv, Vs + (kt * 16 + li) * LD, lg);
#pragma unroll
                for (int qt = 0; qt < 2; qt++) {
                    const f32x4 dp = mma_d<DP>((f32x4){0, 0, 0, 0}, fv, fdo[qt]);
#pragma unroll
                    for (int r = 0; r < 4; r++) s[kt][qt][r] = s[kt][qt][r] * (dp[r] - dl[qt]) * p.scale;
                }
            }
        }
        // second contraction over keys: forward uses V, dQ uses K
        const bf16* T2 = MODE == 0 ? Vs : Ks;
#pragma unroll
        for (int kb = 0; kb < 2; kb++) {
            bf16x8 pb[2];
#pragma unroll
            for (int qt = 0; qt < 2; qt++) pb[qt] = pack_p(s[2 * kb][qt], s[2 * kb + 1][qt]);
#pragma unroll
            for (int dt = 0; dt < DT; dt++) {
                const bf16x8 fa = tr_frag32(T2, LD, kb * 32, dt * 16, li, lg);
#pragma unroll
                for (int qt = 0; qt < 2; qt++) o[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, pb[qt], o[dt][qt], 0, 0, 0);
            }
        }
    }
    // epilogue: lane holds, for query li of tile qt, d = dt*16 + lg*4 + r
#pragma unroll
    for (int qt = 0; qt < 2; qt++) {
        const int q = q0 + qt * 16 + li;
        float inv = 1.f;
        if (MODE == 0) {
            float lt = l[qt];
            lt += __shfl_xor(lt, 16, 64);
            lt += __shfl_xor(lt, 32, 64);
            inv = lt > 0.f ? 1.f / lt : 0.f;
            if (q < p.Nq && lg == 0 && p.LSE) p.LSE[((long long)b * p.H + h) * p.Nq + q] = m[qt] + log2f(lt);
        }
        if (q >= p.Nq) continue;
        bf16* dst = (MODE == 0 ? p.O : p.dQ) + b * (MODE == 0 ? p.bso : p.bsq) + (long long)q * (MODE == 0 ? p.ldo : p.ldq) + (long long)h * p.D;
#pragma unroll
        for (int dt = 0; dt < DT; dt++) {
            const int d = dt * 16 + lg * 4;
            if (d + 4 <= p.D) {
                bf16x4 v = {f2bf(o[dt][qt][0] * inv), f2bf(o[dt][qt][1] * inv), f2bf(o[dt][qt][2] * inv), f2bf(o[dt][qt][3] * inv)};
                *reinterpret_cast<bf16x4*>(dst + d) = v;
            }
        }
    }
}

// dK, dV: block = 64 keys (wave w owns keys 16w..16w+15), loop over 32-query tiles.
constexpr int AK_QT = 32;
template <int DP>
__global__ __launch_bounds__(256) void attn_dkdv_kernel(AttnParams p) {
    constexpr int LD = DP + 8;
    constexpr int DT = DP / 16;
    __shared__ __attribute__((aligned(16))) bf16 Qs[AK_QT * LD];
    __shared__ __attribute__((aligned(16))) bf16 dOs[AK_QT * LD];
    __shared__ float lse_s[AK_QT], dl_s[AK_QT];
    const int b = blockIdx.z, h = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    const int key = blockIdx.x * 64 + wave * 16 + li;
    const bool kok = key < p.Nk;
    const bf16* Qb = p.Q + b * p.bsq + (long long)h * p.D;
    const bf16* dOb = p.dO + b * p.bso + (long long)h * p.D;
    Frag<DP> fk, fv;  // B operands: (k = d, col = key)
    frag_from_global<DP>(fk, p.K + b * p.bsk + (long long)(kok ? key : 0) * p.ldk + (long long)h * p.D, lg, p.D, kok);
    frag_from_global<DP>(fv, p.V + b * p.bsv + (long long)(kok ? key : 0) * p.ldv + (long long)h * p.D, lg, p.D, kok);
    f32x4 dk[DT], dv[DT];
#pragma unroll
    for (int i = 0; i < DT; i++) { dk[i] = (f32x4){0, 0, 0, 0}; dv[i] = (f32x4){0, 0, 0, 0}; }

    for (int q0 = 0; q0 < p.Nq; q0 += AK_QT) {
        __syncthreads();
        load_tile_lds<DP>(Qs, LD, Qb, p.ldq, q0, p.Nq, AK_QT, p.D);
        load_tile_lds<DP>(dOs, LD, dOb, p.ldo, q0, p.Nq, AK_QT, p.D);
        if (threadIdx.x < AK_QT) {
            const int q = q0 + threadIdx.x;
            lse_s[threadIdx.x] = q < p.Nq ? p.LSE[((long long)b * p.H + h) * p.Nq + q] : 0.f;
            dl_s[threadIdx.x] = q < p.Nq ? p.delta[((long long)b * p.H + h) * p.Nq + q] : 0.f;
        }
        __syncthreads();
        f32x4 pp[2], ds[2];
#pragma unroll
        for (int qt = 0; qt < 2; qt++) {
            Frag<DP> fq, fdo;  // A operands: (row = query, k = d)
            frag_from_lds<DP>(fq, Qs + (qt * 16 + li) * LD, lg);
            frag_from_lds<DP>(fdo, dOs + (qt * 16 + li) * LD, lg);
            const f32x4 s = mma_d<DP>((f32x4){0, 0, 0, 0}, fq, fk);    // S[q][key]: lane col=key li, rows q = lg*4+r
            const f32x4 dp = mma_d<DP>((f32x4){0, 0, 0, 0}, fdo, fv);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int ql = qt * 16 + lg * 4 + r;
                const bool ok = kok && (q0 + ql) < p.Nq;
                const float pr = ok ? exp2f(s[r] * p.scale2 - lse_s[ql]) : 0.f;
                pp[qt][r] = pr;
                ds[qt][r] = pr * (dp[r] - dl_s[ql]) * p.scale;
            }
        }
        const bf16x8 pb = pack_p(pp[0], pp[1]), dsb = pack_p(ds[0], ds[1]);
#pragma unroll
        for (int dt = 0; dt < DT; dt++) {
            const bf16x8 fa = tr_frag32(dOs, LD, 0, dt * 16, li, lg);   // dO^T (rows d, k = queries)
            dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, pb, dv[dt], 0, 0, 0);
            const bf16x8 fb = tr_frag32(Qs, LD, 0, dt * 16, li, lg);    // Q^T
            dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, dsb, dk[dt], 0, 0, 0);
        }
    }
    if (!kok) return;
    bf16* dKp = p.dK + b * p.bsk + (long long)key * p.ldk + (long long)h * p.D;
    bf16* dVp = p.dV + b * p.bsv + (long long)key * p.ldv + (long long)h * p.D;
#pragma unroll
    for (int dt = 0; dt < DT; dt++) {
        const int d = dt * 16 + lg * 4;
        if (d + 4 <= p.D) {
            bf16x4 a = {f2bf(dk[dt][0]), f2bf(dk[dt][1]), f2bf(dk[dt][2]), f2bf(dk[dt][3])};
            bf16x4 c = {f2bf(dv[dt][0]), f2bf(dv[dt][1]), f2bf(dv[dt][2]), f2bf(dv[dt][3])};
            *reinterpret_cast<bf16x4*>(dKp + d) = a;
            *reinterpret_cast<bf16x4*>(dVp + d) = c;
        }
    }
}

// delta[b][h][q] = sum_d dO[q][h*D+d] * O[q][h*D+d]
__global__ void attn_delta_kernel(const bf16* __restrict__ dO, const bf16* __restrict__ O, float* __restrict__ delta, int B,
                                  int H, int Nq, int D, int ld, long long bs) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)B * H * Nq) return;
    const int q = (int)(idx % Nq); const int h = (int)((idx / Nq) % H); const int b = (int)(idx / ((long long)Nq * H));
    const bf16* a = dO + b * bs + (long long)q * ld + (long long)h * D;
    const bf16* c = O + b * bs + (long long)q * ld + (long long)h * D;
    float t = 0.f;
    for (int d = 0; d < D; d += 8) {
        const bf16x8 x = ld8(a + d), y = ld8(c + d);
#pragma unroll
        for (int e = 0; e < 8; e++) t += bf2f(x[e]) * bf2f(y[e]);
    }
    delta[idx] = t;
}

template <int DP>
static int launch_attn(const AttnParams& p, int mode, hipStream_t s) {
    if (mode == 0) hipLaunchKernelGGL((attn_q_kernel<DP, 0>), dim3((p.Nq + AT_QB - 1) / AT_QB, p.H, p.B), dim3(256), 0, s, p);
    else if (mode == 1) hipLaunchKernelGGL((attn_q_kernel<DP, 1>), dim3((p.Nq + AT_QB - 1) / AT_QB, p.H, p.B), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((attn_dkdv_kernel<DP>), dim3((p.Nk + 63) / 64, p.H, p.B), dim3(256), 0, s, p);
    return sidlsg_last_error();
}
static int dispatch_attn(const AttnParams& p, int mode, hipStream_t s) {
    if (p.D % 8 || p.D <= 0 || p.D > 160) return SIDLSG_EINVAL;
    const int dp = (p.D + 15) / 16 * 16;
    switch (dp) {
        case 16: return launch_attn<16>(p, mode, s);
        case 32: return launch_attn<32>(p, mode, s);
        case 48: return launch_attn<48>(p, mode, s);
        case 64: return launch_attn<64>(p, mode, s);
        case 80: return launch_attn<80>(p, mode, s);
        case 96: return launch_attn<96>(p, mode, s);
        case 128: return launch_attn<128>(p, mode, s);
        case 160: return launch_attn<160>(p, mode, s);
    }
    return SIDLSG_EINVAL;
}

extern "C" {

// O[b][q][h*D..] = softmax(Q K^T * D^-0.5) V.   Q/K/V/O token strides ld*, batch strides bs* (elements).
// LSE: [B][H][Nq] fp32 (log2 domain), may be null when no backward is needed.
int sidlsg_attn_fwd(const void* Q, const void* K, const void* V, void* O, float* LSE, int B, int H, int Nq, int Nk, int D,
                    int ldq, int ldk, int ldv, int ldo, long long bsq, long long bsk, long long bsv, long long bso,
                    void* stream) {
    if ((ldq | ldk | ldv | ldo) & 3) return SIDLSG_EINVAL;
    AttnParams p{};
    p.Q = (const bf16*)Q; p.K = (const bf16*)K; p.V = (const bf16*)V; p.O = (bf16*)O; p.LSE = LSE;
    p.B = B; p.H = H; p.Nq = Nq; p.Nk = Nk; p.D = D; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
    p.bsq = bsq; p.bsk = bsk; p.bsv = bsv; p.bso = bso;
    p.scale = 1.0f / sqrtf((float)D); p.scale2 = p.scale * 1.4426950408889634f;
    return dispatch_attn(p, 0, (hipStream_t)stream);
}

// dQ, dK, dV given dO (same layout as O).  delta: workspace [B][H][Nq] fp32.
int sidlsg_attn_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* LSE, void* dQ,
                    void* dK, void* dV, float* delta, int B, int H, int Nq, int Nk, int D, int ldq, int ldk, int ldv, int ldo,
                    long long bsq, long long bsk, long long bsv, long long bso, void* stream) {
    if ((ldq | ldk | ldv | ldo) & 3) return SIDLSG_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    AttnParams p{};
    p.Q = (const bf16*)Q; p.K = (const bf16*)K; p.V = (const bf16*)V; p.dO = (const bf16*)dO;
    p.dQ = (bf16*)dQ; p.dK = (bf16*)dK; p.dV = (bf16*)dV; p.LSE = const_cast<float*>(LSE); p.delta = delta;
    p.B = B; p.H = H; p.Nq = Nq; p.Nk = Nk; p.D = D; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
    p.bsq = bsq; p.bsk = bsk; p.bsv = bsv; p.bso = bso;
    p.scale = 1.0f / sqrtf((float)D); p.scale2 = p.scale * 1.4426950408889634f;
    const long long n = (long long)B * H * Nq;
    hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const bf16*)dO, (const bf16*)O,
                       delta, B, H, Nq, D, ldo, bso);
    if (int e = dispatch_attn(p, 1, s)) return e;
    return dispatch_attn(p, 2, s);
}

}  // extern "C"
