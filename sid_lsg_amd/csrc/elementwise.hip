// Memory-bound elementwise / layout / reduction kernels of the SiD-LSG step (gfx950).
// All bf16 traffic is 16 bytes per lane; fp32 math inside.
#include "common.h"

#define GRID1D(n, per) dim3((unsigned)((((size_t)(n)) + (per) - 1) / (per)))

// ---- layout ---------------------------------------------------------------------------------
// Kernels below that touch activations are templated on the activation storage type T: bf16 (production) or float
// (the fp32-accurate parity mode; entry points with the _f32 suffix).  Math is fp32 inside either way.
// x_t = s0[b]*x0 + s1[b]*noise (x0 may be null), written as NHWC with Cp (>=C, mult of 8)
// channels (zero padded), replicated `dup` times along batch (CFG: [uncond ; cond] share x_t).
// Also optionally stores x_t in fp32 NCHW (needed later for the x0 prediction).
template <typename T>
__global__ void noisy_input_kernel(const float* __restrict__ x0, const float* __restrict__ noise,
                                   const float* __restrict__ s0, const float* __restrict__ s1,
                                   T* __restrict__ out, float* __restrict__ xt, int B, int C, int HW, int Cp, int dup) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // over B*HW
    if (idx >= B * HW) return;
    const int b = idx / HW, p = idx - b * HW;
    float o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int c = 0; c < C; c++) {
        const size_t i = ((size_t)b * C + c) * HW + p;
        float v = s1[b] * noise[i];
        if (x0) v += s0[b] * x0[i];
        if (xt) xt[i] = v;
        o[c] = v;
    }
    for (int d = 0; d < dup; d++) {
        T* dst = out + ((size_t)(d * B + b) * HW + p) * Cp;
        stv8<T>(dst, o);
        for (int c = 8; c < Cp; c += 8) zerov8<T>(dst + c);
    }
}

// d_x0[b,c,p] = s0[b] * sum_d g[(d*B+b), p, c]   (backward of noisy_input wrt x0); g NHWC
template <typename T>
__global__ void noisy_input_bwd_kernel(const T* __restrict__ g, const float* __restrict__ s0, float* __restrict__ dx0,
                                       int B, int C, int HW, int Cp, int dup, int accumulate) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * HW) return;
    const int b = idx / HW, p = idx - b * HW;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int d = 0; d < dup; d++) {
        float v[8];
        ldv8<T>(g + ((size_t)(d * B + b) * HW + p) * Cp, v);
        for (int c = 0; c < 8; c++) acc[c] += v[c];
    }
    for (int c = 0; c < C; c++) {
        const size_t i = ((size_t)b * C + c) * HW + p;
        const float v = s0[b] * acc[c];
        dx0[i] = accumulate ? dx0[i] + v : v;
    }
}

// CFG combine + optional x0 prediction.  eps: [dup*B][HW][C] fp32 (NHWC, uncond first).
// out (NCHW fp32) = predict_x0 ? (x_t - s1*e)/s0 : e,   e = dup==2 ? u + kappa*(c-u) : eps
__global__ void cfg_x0_kernel(const float* __restrict__ eps, const float* __restrict__ xt, const float* __restrict__ s0,
                              const float* __restrict__ s1, float* __restrict__ out, int B, int C, int HW, int Ce, int dup,
                              float kappa, int predict_x0) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // over B*HW
    if (idx >= B * HW) return;
    const int b = idx / HW, p = idx - b * HW;
    for (int c = 0; c < C; c++) {
        float e = eps[((size_t)b * HW + p) * Ce + c];
        if (dup == 2) { const float cnd = eps[((size_t)(B + b) * HW + p) * Ce + c]; e = e + kappa * (cnd - e); }
        const size_t i = ((size_t)b * C + c) * HW + p;
        out[i] = predict_x0 ? (xt[i] - s1[b] * e) / s0[b] : e;
    }
}

// backward of cfg_x0: d_eps (NHWC [dup*B][HW][Cp], zero padded) and, when predict_x0, d_xt = g/s0 (fp32 NCHW)
template <typename T>
__global__ void cfg_x0_bwd_kernel(const float* __restrict__ g, const float* __restrict__ s0, const float* __restrict__ s1,
                                  T* __restrict__ deps, float* __restrict__ dxt, int B, int C, int HW, int Cp, int dup,
                                  float kappa, int predict_x0) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * HW) return;
    const int b = idx / HW, p = idx - b * HW;
    float du[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int c = 0; c < C; c++) {
        const size_t i = ((size_t)b * C + c) * HW + p;
        const float go = g[i];
        const float ge = predict_x0 ? -go * s1[b] / s0[b] : go;
        if (dxt) dxt[i] = predict_x0 ? go / s0[b] : 0.f;
        if (dup == 2) { du[c] = (1.f - kappa) * ge; dc[c] = kappa * ge; }
        else du[c] = ge;
    }
    T* d0 = deps + ((size_t)b * HW + p) * Cp;
    stv8<T>(d0, du);
    for (int c = 8; c < Cp; c += 8) zerov8<T>(d0 + c);
    if (dup == 2) {
        T* d1 = deps + ((size_t)(B + b) * HW + p) * Cp;
        stv8<T>(d1, dc);
        for (int c = 8; c < Cp; c += 8) zerov8<T>(d1 + c);
    }
}

// ---- timestep embedding: [cos | sin] of t * exp(-ln(1e4) * i/half), [B][dim] ---------------
template <typename T>
__global__ void timestep_embed_kernel(const long long* __restrict__ t, T* __restrict__ out, int B, int dim) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int half = dim >> 1;
    if (idx >= B * half) return;
    const int b = idx / half, i = idx - b * half;
    const float freq = expf(-9.210340371976184f * (float)i / (float)half);
    const float a = (float)t[b] * freq;
    out[(size_t)b * dim + i] = (T)cosf(a);
    out[(size_t)b * dim + half + i] = (T)sinf(a);
}

// ---- activations ----------------------------------------------------------------------------
template <typename T>
__global__ void silu_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, size_t n8) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    float v[8];
    ldv8<T>(x + i * 8, v);
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = silu_t<T>(v[e]);
    stv8<T>(y + i * 8, v);
}
template <typename T>
__global__ void silu_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, size_t n8) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    float v[8], d[8];
    ldv8<T>(x + i * 8, v); ldv8<T>(dy + i * 8, d);
#pragma unroll
    for (int e = 0; e < 8; e++) d[e] *= silu_grad_t<T>(v[e]);
    stv8<T>(dx + i * 8, d);
}

// GEGLU: h [M][2F] -> y [M][F] = h[:, :F] * gelu(h[:, F:])
template <typename T>
__global__ void geglu_fwd_kernel(const T* __restrict__ h, T* __restrict__ y, size_t M, int F) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int F8 = F >> 3;
    if (i >= M * F8) return;
    const size_t m = i / F8; const int c = (int)(i - m * F8) * 8;
    float a[8], g[8];
    ldv8<T>(h + m * 2 * F + c, a); ldv8<T>(h + m * 2 * F + F + c, g);
#pragma unroll
    for (int e = 0; e < 8; e++) a[e] *= gelu_t<T>(g[e]);
    stv8<T>(y + m * F + c, a);
}
template <typename T>
__global__ void geglu_bwd_kernel(const T* __restrict__ h, const T* __restrict__ dy, T* __restrict__ dh, size_t M, int F) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int F8 = F >> 3;
    if (i >= M * F8) return;
    const size_t m = i / F8; const int c = (int)(i - m * F8) * 8;
    float a[8], g[8], d[8], da[8], dg[8];
    ldv8<T>(h + m * 2 * F + c, a); ldv8<T>(h + m * 2 * F + F + c, g); ldv8<T>(dy + m * F + c, d);
#pragma unroll
    for (int e = 0; e < 8; e++) {
        float gl, gd;
        gelu_pair_t<T>(g[e], gl, gd);
        da[e] = d[e] * gl;
        dg[e] = d[e] * a[e] * gd;
    }
    stv8<T>(dh + m * 2 * F + c, da);
    stv8<T>(dh + m * 2 * F + F + c, dg);
}

// ---- channel concat / split (NHWC) ----------------------------------------------------------
// out[m][0:C1]=a[m], out[m][C1:C1+C2]=b[m]   (split = same kernel with to_parts=1)
template <typename T>
__global__ void concat2_kernel(T* __restrict__ a, T* __restrict__ b, T* __restrict__ out, size_t M, int C1, int C2,
                               int to_parts) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int C8 = (C1 + C2) >> 3;
    if (i >= M * C8) return;
    const size_t m = i / C8; const int c = (int)(i - m * C8) * 8;
    T* part = c < C1 ? a + m * C1 + c : b + m * C2 + (c - C1);
    T* full = out + m * (C1 + C2) + c;
    float v[8];
    if (to_parts) { ldv8<T>(full, v); stv8<T>(part, v); } else { ldv8<T>(part, v); stv8<T>(full, v); }
}

// backward of nearest x2 upsample: out[b][h][w][c] = sum of the 2x2 block of g [b][2h..][2w..][c]
template <typename T>
__global__ void sumpool2x2_kernel(const T* __restrict__ g, T* __restrict__ out, int B, int H, int W, int C) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int C8 = C >> 3;
    if (i >= (size_t)B * H * W * C8) return;
    const int c = (int)(i % C8) * 8; size_t r = i / C8;
    const int w = (int)(r % W); r /= W; const int h = (int)(r % H); const int b = (int)(r / H);
    const T* s = g + (((size_t)b * 2 * H + 2 * h) * 2 * W + 2 * w) * C + c;
    float v0[8], v1[8], v2[8], v3[8];
    ldv8<T>(s, v0); ldv8<T>(s + C, v1); ldv8<T>(s + (size_t)2 * W * C, v2); ldv8<T>(s + (size_t)2 * W * C + C, v3);
#pragma unroll
    for (int e = 0; e < 8; e++) v0[e] = v0[e] + v1[e] + v2[e] + v3[e];
    stv8<T>(out + (((size_t)b * H + h) * W + w) * C + c, v0);
}

// zero insertion (backward-data of a stride-2 conv): out [B][H][W][C], out[2h][2w]=g[h][w], else 0
// (H = 2Ho or 2Ho-1: the original input size of the strided conv)
template <typename T>
__global__ void zero_insert2_kernel(const T* __restrict__ g, T* __restrict__ out, int B, int Ho, int Wo, int H, int W, int C) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int C8 = C >> 3;
    if (i >= (size_t)B * H * W * C8) return;
    const int c = (int)(i % C8) * 8; size_t r = i / C8;
    const int w = (int)(r % W); r /= W; const int h = (int)(r % H); const int b = (int)(r / H);
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (!(h & 1) && !(w & 1)) ldv8<T>(g + (((size_t)b * Ho + (h >> 1)) * Wo + (w >> 1)) * C + c, v);
    stv8<T>(out + (((size_t)b * H + h) * W + w) * C + c, v);
}

template <typename T>
__global__ void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ o, size_t n8) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    float x[8], y[8];
    ldv8<T>(a + i * 8, x); ldv8<T>(b + i * 8, y);
#pragma unroll
    for (int e = 0; e < 8; e++) x[e] += y[e];
    stv8<T>(o + i * 8, x);
}

// ---- column sums of g [B][rows][N]: per_batch[b][n] += , total[n] += (fp32 atomics; callers zero per_batch) -----
// grid (row chunks, B) fills the chip; 256 threads walk columns in passes of cpp 8-wide chunks, rows split over 256/cpp lanes.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ g, float* __restrict__ per_batch,
                                                     float* __restrict__ total, int rows_per_batch, int N, int ldg,
                                                     int rows_per_chunk, int ld_pb) {
    __shared__ float sm[256 * 8];
    const int N8 = N >> 3;
    const int cpp = N8 < 256 ? N8 : 256;
    const int rows = 256 / cpp;
    const int ci = threadIdx.x % cpp, rl = threadIdx.x / cpp;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int r0 = chunk * rows_per_chunk, r1 = min(rows_per_batch, r0 + rows_per_chunk);
    for (int c0 = 0; c0 < N8; c0 += cpp) {
        const int cc = c0 + ci;
        float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (rl < rows && cc < N8)
            for (int r = r0 + rl; r < r1; r += rows) {
                float v[8];
                ldv8<T>(g + ((size_t)b * rows_per_batch + r) * ldg + cc * 8, v);
#pragma unroll
                for (int e = 0; e < 8; e++) s[e] += v[e];
            }
        __syncthreads();
        if (rl < rows)
#pragma unroll
            for (int e = 0; e < 8; e++) sm[(rl * cpp + ci) * 8 + e] = s[e];
        __syncthreads();
        for (int i = threadIdx.x; i < cpp * 8; i += 256) {
            const int col = c0 * 8 + i;
            if (col >= N) continue;
            float t = 0.f;
            for (int r = 0; r < rows; r++) t += sm[r * cpp * 8 + i];
            if (per_batch) unsafeAtomicAdd(per_batch + (size_t)b * ld_pb + col, t);
            if (total) unsafeAtomicAdd(total + col, t);
        }
    }
}

// ---- weights: fp32 master -> bf16 compute copies ---------------------------------------------
__global__ void cast_f32_bf16_kernel(const float* __restrict__ x, bf16* __restrict__ y, size_t n) {
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (i + 8 <= n) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(x + i), b = *reinterpret_cast<const f32x4*>(x + i + 4);
        bf16x8 o = {f2bf(a[0]), f2bf(a[1]), f2bf(a[2]), f2bf(a[3]), f2bf(b[0]), f2bf(b[1]), f2bf(b[2]), f2bf(b[3])};
        st8(y + i, o);
    } else {
        for (size_t j = i; j < n; j++) y[j] = f2bf(x[j]);
    }
}
__global__ void cast_bf16_f32_kernel(const bf16* __restrict__ x, float* __restrict__ y, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = bf2f(x[i]);
}

// dgrad weight: src fp32 [N][T][K] -> dst bf16 [K][T][N] with taps reversed (T=1: plain transpose)
template <typename D>
__global__ void transpose_w_kernel(const float* __restrict__ src, D* __restrict__ dst, int N, int K, int T) {
    __shared__ float tile[32][33];
    const int tap = blockIdx.z;
    const int n0 = blockIdx.y * 32, k0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 32 x 8
    for (int j = ty; j < 32; j += 8) {
        const int n = n0 + j, k = k0 + tx;
        tile[j][tx] = (n < N && k < K) ? src[((size_t)n * T + tap) * K + k] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int k = k0 + j, n = n0 + tx;
        if (n < N && k < K) dst[((size_t)k * T + (T - 1 - tap)) * N + n] = (D)tile[tx][j];
    }
}

// Batched variant: one launch refreshes every backward-data operand of a network (~500 weights after each optimizer
// step; launched one by one they were ~3 % of the step, launch- and tail-bound).  jobs[] is a device table sorted by
// blk0 (first 32x32-tile index of the job); a block finds its job by binary search.  Layout = sidlsg_tw_job in the header.
struct TwJob { const float* src; void* dst; int N, K, T, blk0; };
template <typename D>
__global__ void transpose_w_batched_kernel(const TwJob* __restrict__ jobs, int njobs) {
    __shared__ float tile[32][33];
    int lo = 0, hi = njobs - 1;
    const int bid = blockIdx.x;
    while (lo < hi) {                       // last job with blk0 <= bid
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].blk0 <= bid) lo = mid; else hi = mid - 1;
    }
    const TwJob j = jobs[lo];
    const int tk = (j.K + 31) / 32, tn = (j.N + 31) / 32;
    int t = bid - j.blk0;
    const int tap = t / (tk * tn); t -= tap * tk * tn;
    const int n0 = (t / tk) * 32, k0 = (t % tk) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int n = n0 + r, k = k0 + tx;
        tile[r][tx] = (n < j.N && k < j.K) ? j.src[((size_t)n * j.T + tap) * j.K + k] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int k = k0 + r, n = n0 + tx;
        if (n < j.N && k < j.K) reinterpret_cast<D*>(j.dst)[((size_t)k * j.T + (j.T - 1 - tap)) * j.N + n] = (D)tile[tx][r];
    }
}

// bf16 -> bf16 variant: reads the bf16 COMPUTE copies (which the optimizer kernel has just written) instead of the fp32
// masters -- 4 instead of 6 bytes per parameter -- in 64 x 64 tiles with 16-byte global accesses on both sides.  Same job
// record (src then points at bf16 data); blk0 counts 64 x 64 tiles; N and K multiples of 8.
__global__ __launch_bounds__(256) void transpose_w16_batched_kernel(const TwJob* __restrict__ jobs, int njobs) {
    __shared__ unsigned tile[64][33];        // [n][k / 2]: two bf16 per dword, 33-dword rows -> column reads hit distinct banks
    int lo = 0, hi = njobs - 1;
    const int bid = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].blk0 <= bid) lo = mid; else hi = mid - 1;
    }
    const TwJob j = jobs[lo];
    const bf16* src = reinterpret_cast<const bf16*>(j.src);
    bf16* dst = reinterpret_cast<bf16*>(j.dst);
    const int tk = (j.K + 63) / 64, tn = (j.N + 63) / 64;
    int t = bid - j.blk0;
    const int tap = t / (tk * tn); t -= tap * tk * tn;
    const int n0 = (t / tk) * 64, k0 = (t % tk) * 64;
    const int c8 = (threadIdx.x & 7) * 8, r0 = threadIdx.x >> 3;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int n = n0 + r0 + 32 * i, k = k0 + c8;
        u32x4 q = {0u, 0u, 0u, 0u};         // 8 consecutive k of row n = 4 dwords, already (k even | k odd << 16)
        if (n < j.N && k < j.K) q = *reinterpret_cast<const u32x4*>(src + ((size_t)n * j.T + tap) * j.K + k);
#pragma unroll
        for (int d = 0; d < 4; d++) tile[r0 + 32 * i][(c8 >> 1) + d] = q[d];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int kl = r0 + 32 * i, k = k0 + kl, n = n0 + c8;
        if (n < j.N && k < j.K) {
            const int sh = (kl & 1) * 16;
            u32x4 o;
#pragma unroll
            for (int d = 0; d < 4; d++)
                o[d] = ((tile[c8 + 2 * d][kl >> 1] >> sh) & 0xffffu) | (((tile[c8 + 2 * d + 1][kl >> 1] >> sh) & 0xffffu) << 16);
            *reinterpret_cast<u32x4*>(dst + ((size_t)k * j.T + (j.T - 1 - tap)) * j.N + n) = o;
        }
    }
}

// ---- SiD losses: loss value + closed-form gradients -------------------
// per-sample prepass: nan flag over the inputs, and sum |x - y_r|
__global__ void sid_sample_stats_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                                        float* __restrict__ st, int n) {
    // st[s][0] = nan flag, st[s][1] = sum|a-b|   (a=images, b=y_real, c=y_fake; b,c may alias/null)
    __shared__ float sh[2][4];
    const int s = blockIdx.x;
    float flag = 0.f, sum = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float x = a[(size_t)s * n + i];
        const float y = b ? b[(size_t)s * n + i] : 0.f;
        const float z = c ? c[(size_t)s * n + i] : 0.f;
        if (x != x || y != y || z != z) flag = 1.f;
        sum += fabsf(x - y);
    }
    flag = wave_max(flag); sum = wave_sum(sum);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[0][w] = flag; sh[1][w] = sum; }
    __syncthreads();
    if (threadIdx.x == 0) {
        st[s * 2] = fmaxf(fmaxf(sh[0][0], sh[0][1]), fmaxf(sh[0][2], sh[0][3]));
        st[s * 2 + 1] = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
    }
}

// generator loss: sum (yr-yf)*((yr-x) - alpha*(yr-yf))/w * scale over non-NaN samples  (alpha==1 -> (yr-yf)(yf-x)/w)
__global__ void sid_g_loss_kernel(const float* __restrict__ x, const float* __restrict__ yr, const float* __restrict__ yf,
                                  const float* __restrict__ st, float* __restrict__ dx, float* __restrict__ dyr,
                                  float* __restrict__ dyf, float* __restrict__ loss_part, int n, float alpha, float scale) {
    __shared__ float sh[4];
    const int s = blockIdx.y;
    const bool drop = st[s * 2] != 0.f;
    const float w = fmaxf(st[s * 2 + 1] / (float)n, 1e-5f);
    float acc = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const size_t k = (size_t)s * n + i;
        if (drop) { dx[k] = 0.f; dyr[k] = 0.f; dyf[k] = 0.f; continue; }
        const float d = yr[k] - yf[k], e = yr[k] - x[k];
        acc += d * (e - alpha * d) / w;
        dyr[k] = scale * (e + (1.f - 2.f * alpha) * d) / w;
        dyf[k] = scale * (-e + 2.f * alpha * d) / w;
        dx[k] = scale * (-d) / w;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) loss_part[blockIdx.y * gridDim.x + blockIdx.x] = (sh[0] + sh[1] + sh[2] + sh[3]) * scale;
}

// fake-score loss: sum (e - noise)^2 * scale over non-NaN samples; grad wrt e
__global__ void sid_fake_loss_kernel(const float* __restrict__ e, const float* __restrict__ noise, const float* __restrict__ st,
                                     float* __restrict__ de, float* __restrict__ loss_part, int n, float scale) {
    __shared__ float sh[4];
    const int s = blockIdx.y;
    const bool drop = st[s * 2] != 0.f;
    float acc = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const size_t k = (size_t)s * n + i;
        if (drop) { de[k] = 0.f; continue; }
        const float d = e[k] - noise[k];
        acc += d * d;
        de[k] = 2.f * scale * d;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) loss_part[blockIdx.y * gridDim.x + blockIdx.x] = (sh[0] + sh[1] + sh[2] + sh[3]) * scale;
}

__global__ void sum_small_kernel(const float* __restrict__ part, float* __restrict__ out, int n) {
    float t = 0.f;
    for (int i = threadIdx.x; i < n; i += 64) t += part[i];
    t = wave_sum(t);
    if (threadIdx.x == 0) out[0] = t;
}

#include "bias_act_kernel.h"

// ---- typed host launchers (T = activation storage type) + the two C entry-point families ----------------------------
template <typename T>
static int noisy_input_t(const float* x0, const float* noise, const float* s0, const float* s1, void* out, float* xt, int B,
                         int C, int HW, int Cp, int dup, void* stream) {
    if (C > 8 || Cp % 8 || Cp < 8 || dup < 1) return SIDLSG_EINVAL;
    hipLaunchKernelGGL(noisy_input_kernel<T>, GRID1D((size_t)B * HW, 256), dim3(256), 0, (hipStream_t)stream, x0, noise, s0, s1,
                       (T*)out, xt, B, C, HW, Cp, dup);
    return sidlsg_last_error();
}
template <typename T>
static int noisy_input_bwd_t(const void* g, const float* s0, float* dx0, int B, int C, int HW, int Cp, int dup, int accumulate,
                             void* stream) {
    if (C > 8 || Cp % 8) return SIDLSG_EINVAL;
    hipLaunchKernelGGL(noisy_input_bwd_kernel<T>, GRID1D((size_t)B * HW, 256), dim3(256), 0, (hipStream_t)stream, (const T*)g,
                       s0, dx0, B, C, HW, Cp, dup, accumulate);
    return sidlsg_last_error();
}
template <typename T>
static int cfg_x0_bwd_t(const float* g, const float* s0, const float* s1, void* deps, float* dxt, int B, int C, int HW, int Cp,
                        int dup, float kappa, int predict_x0, void* stream) {
    if (C > 8 || Cp % 8) return SIDLSG_EINVAL;
    hipLaunchKernelGGL(cfg_x0_bwd_kernel<T>, GRID1D((size_t)B * HW, 256), dim3(256), 0, (hipStream_t)stream, g, s0, s1,
                       (T*)deps, dxt, B, C, HW, Cp, dup, kappa, predict_x0);
    return sidlsg_last_error();
}
template <typename T>
static int timestep_embed_t(const long long* t, void* out, int B, int dim, void* stream) {
    hipLaunchKernelGGL(timestep_embed_kernel<T>, GRID1D((size_t)B * (dim / 2), 256), dim3(256), 0, (hipStream_t)stream, t,
                       (T*)out, B, dim);
    return sidlsg_last_error();
}
template <typename T>
static int silu_fwd_t(const void* x, void* y, long long n, void* stream) {
    if (n % 8) return SIDLSG_EINVAL;
    hipLaunchKernelGGL(silu_fwd_kernel<T>, GRID1D(n / 8, 256), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)y, (size_t)n / 8);
    return sidlsg_last_error();
}
template <typename T>
static int silu_bwd_t(const void* x, const void* dy, void* dx, long long n, void* stream) {
    if (n % 8) return SIDLSG_EINVAL;
    hipLaunchKernelGGL(silu_bwd_kernel<T>, GRID1D(n / 8, 256), dim3(256), 0, (hipStream_t)stream, (const T*)x, (const T*)dy,
                       (T*)dx, (size_t)n / 8);
    return sidlsg_last_error();
}
template <typename T>
static int geglu_fwd_t(const void* h, void* y, long long M, int F, void* stream) {
    if (F % 8) return SIDLSG_EINVAL;
    hipLaunchKernelGGL(geglu_fwd_kernel<T>, GRID1D((size_t)M * (F / 8), 256), dim3(256), 0, (hipStream_t)stream, (const T*)h,
                       (T*)y, (size_t)M, F);
    return sidlsg_last_error();
}
template <typename T>
static int geglu_bwd_t(const void* h, const void* dy, void* dh, long long M, int F, void* stream) {
    if (F % 8) return SIDLSG_EINVAL;
    hipLaunchKernelGGL(geglu_bwd_kernel<T>, GRID1D((size_t)M * (F / 8), 256), dim3(256), 0, (hipStream_t)stream, (const T*)h,
                       (const T*)dy, (T*)dh, (size_t)M, F);
    return sidlsg_last_error();
}
template <typename T>
static int concat2_t(void* a, void* b, void* out, long long M, int C1, int C2, int to_parts, void* stream) {
    if (C1 % 8 || C2 % 8) return SIDLSG_EINVAL;
    hipLaunchKernelGGL(concat2_kernel<T>, GRID1D((size_t)M * ((C1 + C2) / 8), 256), dim3(256), 0, (hipStream_t)stream, (T*)a,
                       (T*)b, (T*)out, (size_t)M, C1, C2, to_parts);
    return sidlsg_last_error();
}
template <typename T>
static int sumpool2x2_t(const void* g, void* out, int B, int H, int W, int C, void* stream) {
    if (C % 8) return SIDLSG_EINVAL;
    hipLaunchKernelGGL(sumpool2x2_kernel<T>, GRID1D((size_t)B * H * W * (C / 8), 256), dim3(256), 0, (hipStream_t)stream,
                       (const T*)g, (T*)out, B, H, W, C);
    return sidlsg_last_error();
}
template <typename T>
static int zero_insert2_t(const void* g, void* out, int B, int Ho, int Wo, int H, int W, int C, void* stream) {
    if (C % 8 || (H + 1) / 2 != Ho || (W + 1) / 2 != Wo) return SIDLSG_EINVAL;
    hipLaunchKernelGGL(zero_insert2_kernel<T>, GRID1D((size_t)B * H * W * (C / 8), 256), dim3(256), 0, (hipStream_t)stream,
                       (const T*)g, (T*)out, B, Ho, Wo, H, W, C);
    return sidlsg_last_error();
}
template <typename T>
static int add_t(const void* a, const void* b, void* o, long long n, void* stream) {
    if (n % 8) return SIDLSG_EINVAL;
    hipLaunchKernelGGL(add_kernel<T>, GRID1D(n / 8, 256), dim3(256), 0, (hipStream_t)stream, (const T*)a, (const T*)b,
                       (T*)o, (size_t)n / 8);
    return sidlsg_last_error();
}
static int colsum_nchunks(int B, int rows_per_batch) {
    int want = (512 + B - 1) / B; int maxch = (rows_per_batch + 63) / 64;
    int nch = want < maxch ? want : maxch; if (nch < 1) nch = 1; if (nch > 128) nch = 128; return nch;
}
template <typename T>
static int colsum_t(const void* g, int ldg, float* per_batch, float* total, int B, int rows_per_batch, int N, void* stream, int ld_pb = 0) {
    if (N % 8 || N <= 0 || (ld_pb && ld_pb < N)) return SIDLSG_EINVAL;
    const int nch = colsum_nchunks(B, rows_per_batch);
    const int rpc = (rows_per_batch + nch - 1) / nch;
    hipLaunchKernelGGL(colsum_kernel<T>, dim3(nch, B), dim3(256), 0, (hipStream_t)stream, (const T*)g, per_batch, total,
                       rows_per_batch, N, ldg, rpc, ld_pb ? ld_pb : N);
    return sidlsg_last_error();
}
template <typename D>
static int transpose_w_t(const float* src, void* dst, int N, int K, int T, void* stream) {
    hipLaunchKernelGGL(transpose_w_kernel<D>, dim3((K + 31) / 32, (N + 31) / 32, T), dim3(256), 0, (hipStream_t)stream, src,
                       (D*)dst, N, K, T);
    return sidlsg_last_error();
}
template <typename D>
static int transpose_w_batched_t(const void* jobs, int njobs, int nblocks, void* stream) {
    if (!jobs || njobs <= 0 || nblocks <= 0) return SIDLSG_EINVAL;
    hipLaunchKernelGGL(transpose_w_batched_kernel<D>, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, (const TwJob*)jobs, njobs);
    return sidlsg_last_error();
}

// Scaled re-cast of parameter ranges: dst[i] = bf16(scale * src[i]) -- used to fold the attention's D^-1/2 log2(e) factor
// into the q rows of the projection weights' COMPUTE copy (one rounding from the fp32 master, like the plain copy).
// Job record: { const float* src; bf16* dst; int n; int blk0; float scale; int pad; } (32 bytes); 2048 elements per block.
struct ScJob { const float* src; bf16* dst; int n, blk0; float scale; int pad; };
__global__ __launch_bounds__(256) void scale_cast_ranges_kernel(const ScJob* __restrict__ jobs, int njobs) {
    const int bid = blockIdx.x;
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].blk0 <= bid) lo = mid; else hi = mid - 1;
    }
    const ScJob j = jobs[lo];
    const int i0 = ((bid - j.blk0) * 256 + threadIdx.x) * 8;
    if (i0 >= j.n) return;
    if (i0 + 8 <= j.n && (((uintptr_t)(j.src + i0)) & 15) == 0 && (((uintptr_t)(j.dst + i0)) & 15) == 0) {
        float v[8];
        ldv8<float>(j.src + i0, v);
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] *= j.scale;
        stv8<bf16>(j.dst + i0, v);
    } else {
        for (int e = i0; e < j.n && e < i0 + 8; e++) j.dst[e] = f2bf(j.src[e] * j.scale);
    }
}

extern "C" {

int sidlsg_scale_cast_ranges(const void* jobs, int njobs, int nblocks, void* stream) {
    if (!jobs || njobs <= 0 || nblocks <= 0) return SIDLSG_EINVAL;
    hipLaunchKernelGGL(scale_cast_ranges_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, (const ScJob*)jobs, njobs);
    return sidlsg_last_error();
}
int sidlsg_transpose_w16_batched(const void* jobs, int njobs, int nblocks, void* stream) {
    if (!jobs || njobs <= 0 || nblocks <= 0) return SIDLSG_EINVAL;
    hipLaunchKernelGGL(transpose_w16_batched_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, (const TwJob*)jobs, njobs);
    return sidlsg_last_error();
}

#define SIDLSG_BOTH(name, tmpl, params, args) \
    int name params { return tmpl<bf16> args; }  \
    int name##_f32 params { return tmpl<float> args; }

SIDLSG_BOTH(sidlsg_noisy_input, noisy_input_t,
            (const float* x0, const float* noise, const float* s0, const float* s1, void* out, float* xt, int B, int C, int HW, int Cp, int dup, void* stream),
            (x0, noise, s0, s1, out, xt, B, C, HW, Cp, dup, stream))
SIDLSG_BOTH(sidlsg_noisy_input_bwd, noisy_input_bwd_t,
            (const void* g, const float* s0, float* dx0, int B, int C, int HW, int Cp, int dup, int accumulate, void* stream),
            (g, s0, dx0, B, C, HW, Cp, dup, accumulate, stream))
int sidlsg_cfg_x0(const float* eps, const float* xt, const float* s0, const float* s1, float* out, int B, int C, int HW,
                  int Ce, int dup, float kappa, int predict_x0, void* stream) {
    if (Ce < C) return SIDLSG_EINVAL;
    hipLaunchKernelGGL(cfg_x0_kernel, GRID1D((size_t)B * HW, 256), dim3(256), 0, (hipStream_t)stream, eps, xt, s0, s1, out, B,
                       C, HW, Ce, dup, kappa, predict_x0);
    return sidlsg_last_error();
}
SIDLSG_BOTH(sidlsg_cfg_x0_bwd, cfg_x0_bwd_t,
            (const float* g, const float* s0, const float* s1, void* deps, float* dxt, int B, int C, int HW, int Cp, int dup, float kappa, int predict_x0, void* stream),
            (g, s0, s1, deps, dxt, B, C, HW, Cp, dup, kappa, predict_x0, stream))
SIDLSG_BOTH(sidlsg_timestep_embed, timestep_embed_t, (const long long* t, void* out, int B, int dim, void* stream), (t, out, B, dim, stream))
SIDLSG_BOTH(sidlsg_silu_fwd, silu_fwd_t, (const void* x, void* y, long long n, void* stream), (x, y, n, stream))
SIDLSG_BOTH(sidlsg_silu_bwd, silu_bwd_t, (const void* x, const void* dy, void* dx, long long n, void* stream), (x, dy, dx, n, stream))
SIDLSG_BOTH(sidlsg_geglu_fwd, geglu_fwd_t, (const void* h, void* y, long long M, int F, void* stream), (h, y, M, F, stream))
SIDLSG_BOTH(sidlsg_geglu_bwd, geglu_bwd_t, (const void* h, const void* dy, void* dh, long long M, int F, void* stream), (h, dy, dh, M, F, stream))
SIDLSG_BOTH(sidlsg_concat2, concat2_t, (void* a, void* b, void* out, long long M, int C1, int C2, int to_parts, void* stream),
            (a, b, out, M, C1, C2, to_parts, stream))
SIDLSG_BOTH(sidlsg_sumpool2x2, sumpool2x2_t, (const void* g, void* out, int B, int H, int W, int C, void* stream), (g, out, B, H, W, C, stream))
SIDLSG_BOTH(sidlsg_zero_insert2, zero_insert2_t, (const void* g, void* out, int B, int Ho, int Wo, int H, int W, int C, void* stream),
            (g, out, B, Ho, Wo, H, W, C, stream))
int sidlsg_add_bf16(const void* a, const void* b, void* o, long long n, void* stream) { return add_t<bf16>(a, b, o, n, stream); }
int sidlsg_add_f32(const void* a, const void* b, void* o, long long n, void* stream) { return add_t<float>(a, b, o, n, stream); }
// column sums of g [B][rows_per_batch][N] (row stride ldg): per_batch [B][N] (+=, zero it first) and/or total [N] (+=); ws unused
int sidlsg_colsum_nchunks(int B, int rows_per_batch) { return colsum_nchunks(B, rows_per_batch); }
int sidlsg_colsum(const void* g, int ldg, float* per_batch, float* total, float* ws, int B, int rows_per_batch, int N, void* stream) {
    (void)ws;   // kept in the ABI; the reduction is single-pass with atomics
    return colsum_t<bf16>(g, ldg, per_batch, total, B, rows_per_batch, N, stream);
}
int sidlsg_colsum_f32(const void* g, int ldg, float* per_batch, float* total, float* ws, int B, int rows_per_batch, int N, void* stream) {
    (void)ws;
    return colsum_t<float>(g, ldg, per_batch, total, B, rows_per_batch, N, stream);
}
int sidlsg_colsum_strided(const void* g, int ldg, float* per_batch, int ld_pb, float* total, int B, int rows_per_batch, int N, void* stream) {
    return colsum_t<bf16>(g, ldg, per_batch, total, B, rows_per_batch, N, stream, ld_pb);
}
int sidlsg_colsum_strided_f32(const void* g, int ldg, float* per_batch, int ld_pb, float* total, int B, int rows_per_batch, int N, void* stream) {
    return colsum_t<float>(g, ldg, per_batch, total, B, rows_per_batch, N, stream, ld_pb);
}
int sidlsg_cast_f32_bf16(const float* x, void* y, long long n, void* stream) {
    hipLaunchKernelGGL(cast_f32_bf16_kernel, GRID1D((n + 7) / 8, 256), dim3(256), 0, (hipStream_t)stream, x, (bf16*)y, (size_t)n);
    return sidlsg_last_error();
}
int sidlsg_cast_bf16_f32(const void* x, float* y, long long n, void* stream) {
    hipLaunchKernelGGL(cast_bf16_f32_kernel, GRID1D(n, 256), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, y, (size_t)n);
    return sidlsg_last_error();
}
// src fp32 [N][T][K] -> dst [K][T][N], taps reversed (bf16, or fp32 for the _f32 family)
SIDLSG_BOTH(sidlsg_transpose_w, transpose_w_t, (const float* src, void* dst, int N, int K, int T, void* stream), (src, dst, N, K, T, stream))
SIDLSG_BOTH(sidlsg_transpose_w_batched, transpose_w_batched_t, (const void* jobs, int njobs, int nblocks, void* stream), (jobs, njobs, nblocks, stream))
// Generator loss (A7).  x,yr,yf: [S][n] fp32.  ws: >= S*2 + S*GB floats.  loss[0] = value (already * scale).
#define SID_GB 8
int sidlsg_g_loss(const float* x, const float* yr, const float* yf, float* dx, float* dyr, float* dyf, float* loss, float* ws,
                  int S, int n, float alpha, float scale, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    float* st = ws; float* part = ws + 2 * S;
    hipLaunchKernelGGL(sid_sample_stats_kernel, dim3(S), dim3(256), 0, s, x, yr, yf, st, n);
    hipLaunchKernelGGL(sid_g_loss_kernel, dim3(SID_GB, S), dim3(256), 0, s, x, yr, yf, st, dx, dyr, dyf, part, n, alpha, scale);
    hipLaunchKernelGGL(sum_small_kernel, dim3(1), dim3(64), 0, s, part, loss, S * SID_GB);
    return sidlsg_last_error();
}
int sidlsg_fake_loss(const float* e, const float* noise, float* de, float* loss, float* ws, int S, int n, float scale,
                     void* stream) {
    hipStream_t s = (hipStream_t)stream;
    float* st = ws; float* part = ws + 2 * S;
    hipLaunchKernelGGL(sid_sample_stats_kernel, dim3(S), dim3(256), 0, s, e, (const float*)nullptr, (const float*)nullptr, st, n);
    hipLaunchKernelGGL(sid_fake_loss_kernel, dim3(SID_GB, S), dim3(256), 0, s, e, noise, st, de, part, n, scale);
    hipLaunchKernelGGL(sum_small_kernel, dim3(1), dim3(64), 0, s, part, loss, S * SID_GB);
    return sidlsg_last_error();
}
// bias_act: dtype 0 = fp32, 1 = bf16.  grad 0: out = clamp(act(x+b)*gain); grad 1: out = dL/dx given dy (x,b = saved inputs)
int sidlsg_bias_act(const void* x, const void* b, const void* dy, void* out, long long n, int stepB, int sizeB, int act,
                    float alpha, float gain, float clamp, int grad, int dtype, void* stream) {
    if (grad != 0 && grad != 1) return SIDLSG_EINVAL;       // second order: the plugin entry point (plugins/bias_act_plugin.hip)
    return bias_act_launch(x, b, dy, nullptr, out, n, stepB, sizeB, act, alpha, gain, clamp, grad, dtype, stream);
}

}  // extern "C"
