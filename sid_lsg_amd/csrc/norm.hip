// GroupNorm(+SiLU) and LayerNorm, forward and backward, NHWC bf16 activations, fp32 statistics.
// gfx950 / MI355X.  HBM-bound (61 GroupNorms = 45.06 M elements and 48
// LayerNorms = 34.65 M elements per sample-forward at SD1.5 512^2): every kernel reads 16 bytes per
// lane, fully coalesced along the channel axis, and touches each element the minimum number of
// times (stats pass + apply pass; the second read of a <=~100 MB tensor is served by the
// 256 MiB Infinity Cache).
//
// GroupNorm is split in two kernels so that the grid (B x pixel-chunks) fills all 256 CUs even
// at batch 1-2, with a deterministic two-level reduction (no atomics):
//   gn_stats : per (sample, pixel chunk) -> per-group partial (sum, sumsq)
//   gn_apply : folds the partials, y = act((x-mean)*rstd*gamma+beta), writes (mean, rstd)
// Backward:
//   gn_bwd_stats : per (sample, chunk) per-CHANNEL partial (sum dy'*xhat, sum dy') -- these
//                  yield both the group sums needed for dx and dgamma/dbeta
//   gn_bwd_apply : dx = rstd*(dy'*gamma - s1/n - xhat*s2/n)
//   colsum_reduce: dgamma/dbeta (+=) from the per-channel partials
#include "common.h"
#include <stdlib.h>
#include <map>
#include <mutex>
#include <type_traits>
#include <vector>

constexpr int GN_MAX_C = 2560;
#ifndef SIDLSG_GN_U
#define SIDLSG_GN_U 4
#endif
constexpr int GN_U = SIDLSG_GN_U;      // forward kernels: independent loads in flight per thread

// thread layout shared by the GroupNorm kernels: blockDim.x = C8 * rows, thread owns channel chunk
// cc (8 channels) and walks pixels rl, rl+rows, ...
struct GnGeom {
    int C, HW, G, cpg, C8, rows, nch, ppb;  // ppb: pixels per block (chunk)
};

template <typename T>
__global__ void gn_stats_kernel(const T* __restrict__ x, float* __restrict__ part, GnGeom g) {
    extern __shared__ float sm[];  // [rows][C][2]
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int cc = threadIdx.x % g.C8, rl = threadIdx.x / g.C8;
    const int p0 = chunk * g.ppb, p1 = min(g.HW, p0 + g.ppb);
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; e++) s[e] = q[e] = 0.f;
    const T* xb = x + (size_t)b * g.HW * g.C + cc * 8;
    int p = p0 + rl;
    // GN_U independent 16-byte loads in flight per thread (8 measured slower than 4: 37 -> 44 us on 16 x 4096 x 320 -- the
    // chunk of a block is only ~20 pixel rows per thread, the unrolled body then runs twice and leaves a serial tail)
    for (; p + (GN_U - 1) * g.rows < p1; p += GN_U * g.rows) {
        float v[GN_U][8];
#pragma unroll
        for (int u = 0; u < GN_U; u++) ldv8<T>(xb + (size_t)(p + u * g.rows) * g.C, v[u]);
#pragma unroll
        for (int u = 0; u < GN_U; u++)
#pragma unroll
            for (int e = 0; e < 8; e++) { const float f = v[u][e]; s[e] += f; q[e] += f * f; }
    }
    for (; p < p1; p += g.rows) {
        float v[8];
        ldv8<T>(xb + (size_t)p * g.C, v);
#pragma unroll
        for (int e = 0; e < 8; e++) { const float f = v[e]; s[e] += f; q[e] += f * f; }
    }
    float* row = sm + (size_t)rl * g.C * 2;
#pragma unroll
    for (int e = 0; e < 8; e++) { row[(cc * 8 + e) * 2] = s[e]; row[(cc * 8 + e) * 2 + 1] = q[e]; }
    __syncthreads();
    for (int grp = threadIdx.x; grp < g.G; grp += blockDim.x) {
        float a = 0.f, c = 0.f;
        for (int r = 0; r < g.rows; r++)
            for (int ch = grp * g.cpg; ch < (grp + 1) * g.cpg; ch++) {
                a += sm[((size_t)r * g.C + ch) * 2];
                c += sm[((size_t)r * g.C + ch) * 2 + 1];
            }
        float* o = part + (((size_t)b * g.nch + chunk) * g.G + grp) * 2;
        o[0] = a; o[1] = c;
    }
}

// sm[0..G) / sm[G..2G) = sum over the nch chunks of one sample's per-group partial pairs p[k][grp][2].  All threads load
// (a serial loop of nch dependent-latency loads in G threads cost several us per block); the order of the additions is
// fixed by the block size, so the result is deterministic.  sm needs 2*G*GN_FOLD floats; ends with a barrier.
constexpr int GN_FOLD = 8;
DEVFN void gn_fold_partials(const float* __restrict__ p, const GnGeom& g, float* sm) {
    const int slices = min(GN_FOLD, max(1, (int)blockDim.x / g.G));
    float* tmp = sm + 2 * g.G;                                 // [slices][G][2]
    for (int item = threadIdx.x; item < slices * g.G; item += blockDim.x) {
        const int grp = item % g.G, sl = item / g.G;
        float a = 0.f, c = 0.f;
        for (int k = sl; k < g.nch; k += slices) {
            const float2 o = *reinterpret_cast<const float2*>(p + ((size_t)k * g.G + grp) * 2);
            a += o.x; c += o.y;
        }
        tmp[(sl * g.G + grp) * 2] = a; tmp[(sl * g.G + grp) * 2 + 1] = c;
    }
    __syncthreads();
    for (int q = threadIdx.x; q < g.G; q += blockDim.x) {
        float a = 0.f, c = 0.f;
        for (int j = 0; j < slices; j++) { a += tmp[(j * g.G + q) * 2]; c += tmp[(j * g.G + q) * 2 + 1]; }
        sm[q] = a; sm[g.G + q] = c;
    }
    __syncthreads();
}

template <typename T, bool F8 = false>      // F8: y receives e4m3 bytes (same [B][HW][C] geometry)
__global__ void gn_apply_kernel(const T* __restrict__ x, const float* __restrict__ part,
                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                T* __restrict__ y, float* __restrict__ stats, GnGeom g, float eps, int act,
                                const float* __restrict__ gamma1, const float* __restrict__ beta1, int split) {
    extern __shared__ float sm[];  // mean[G], rstd[G]
    const int b = blockIdx.y, chunk = blockIdx.x;
    if (gamma1 && b >= split) { gamma = gamma1; beta = beta1; }      // grouped launch: samples [split, B) belong to the second network
    // fold the nch per-chunk partials of every group: all threads load (one (sum, sumsq) pair each, independent loads), LDS
    // float adds combine them -- a serial loop of nch dependent-latency loads in G threads cost several us per block
    gn_fold_partials(part + (size_t)b * g.nch * g.G * 2, g, sm);
    for (int grp = threadIdx.x; grp < g.G; grp += blockDim.x) {      // each group is read and overwritten by one thread
        const float a = sm[grp], c = sm[g.G + grp];
        const float n = (float)g.cpg * (float)g.HW;
        const float mean = a / n;
        const float var = fmaxf(c / n - mean * mean, 0.f);
        const float rstd = rsqrtf(var + eps);
        sm[grp] = mean; sm[g.G + grp] = rstd;
        if (chunk == 0 && stats) { stats[((size_t)b * g.G + grp) * 2] = mean; stats[((size_t)b * g.G + grp) * 2 + 1] = rstd; }
    }
    __syncthreads();
    const int cc = threadIdx.x % g.C8, rl = threadIdx.x / g.C8;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const int ch = cc * 8 + e, grp = ch / g.cpg;
        sc[e] = sm[g.G + grp] * gamma[ch];
        sh[e] = beta[ch] - sm[grp] * sc[e];
    }
    const int p0 = chunk * g.ppb, p1 = min(g.HW, p0 + g.ppb);
    // Branch-free, prefetching row loop (see ln_fwd_kernel): descriptors over this SAMPLE's [HW][C] slab, pixel rows >= p1 get an
    // out-of-range offset (zeros in, store dropped), and the next GN_U rows are requested before the current ones are stored --
    // "load; normalise; store" per iteration made every iteration's loads wait for the previous iteration's stores.
    const __amdgpu_buffer_rsrc_t rx = mk_buf(x + (size_t)b * g.HW * g.C, (long long)g.HW * g.C * sizeof(T));
    const __amdgpu_buffer_rsrc_t ry = mk_buf(reinterpret_cast<unsigned char*>(y) + (size_t)b * g.HW * g.C * (F8 ? 1 : sizeof(T)),
                                             (long long)g.HW * g.C * (F8 ? 1 : sizeof(T)));
    auto off = [&](int p) { return p < p1 ? (unsigned)p * (unsigned)g.C + cc * 8 : SIDLSG_OOB; };
    // (the wait for the prefetched rows -- the unpack -- sits at the END of the body, behind the stores: at the loop header the
    // compiler would merge the entry state (loads only) with the back edge (loads + stores) into s_waitcnt vmcnt(0); the
    // sched_barrier keeps the scheduler from sinking the requests below the stores)
    Raw8<T> raw[GN_U];
    float v[GN_U][8];
    int p = p0 + rl;
    if (p >= p1) return;
#pragma unroll
    for (int u = 0; u < GN_U; u++) Raw8_bload(raw[u], rx, off(p + u * g.rows));
#pragma unroll
    for (int u = 0; u < GN_U; u++) raw[u].unpack(v[u]);
    do {
        const int pn = p + GN_U * g.rows;
#pragma unroll
        for (int u = 0; u < GN_U; u++) Raw8_bload(raw[u], rx, off(pn + u * g.rows));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < GN_U; u++) {
#pragma unroll
            for (int e = 0; e < 8; e++) {
                float f = v[u][e] * sc[e] + sh[e];
                if (act) f = silu_t<T>(f);
                v[u][e] = f;
            }
            bst8_out<T, F8>(ry, off(p + u * g.rows), v[u]);
        }
        __builtin_amdgcn_sched_barrier(0);       // (... and from hoisting the unpack, i.e. the wait, above the stores)
#pragma unroll
        for (int u = 0; u < GN_U; u++) raw[u].unpack(v[u]);
        p = pn;
    } while (p < p1);
}

template <typename T>
__global__ void gn_bwd_stats_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                    const float* __restrict__ stats, const float* __restrict__ gamma,
                                    const float* __restrict__ beta, float* __restrict__ part, GnGeom g, int act,
                                    const float* __restrict__ gamma1, const float* __restrict__ beta1, int split) {
    extern __shared__ float sm[];  // [rows][C][2]
    const int b = blockIdx.y, chunk = blockIdx.x;
    if (gamma1 && b >= split) { gamma = gamma1; beta = beta1; }
    const int cc = threadIdx.x % g.C8, rl = threadIdx.x / g.C8;
    float mu[8], rs[8], ga[8], be[8], a[8], c[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const int ch = cc * 8 + e, grp = ch / g.cpg;
        mu[e] = stats[((size_t)b * g.G + grp) * 2]; rs[e] = stats[((size_t)b * g.G + grp) * 2 + 1];
        ga[e] = gamma[ch]; be[e] = beta[ch]; a[e] = c[e] = 0.f;
    }
    const int p0 = chunk * g.ppb, p1 = min(g.HW, p0 + g.ppb);
    const size_t base = (size_t)b * g.HW * g.C + cc * 8;
    int p = p0 + rl;
    for (; p + g.rows < p1; p += 2 * g.rows) {            // 4 independent loads (2 pixels x {x, dy}) in flight
        float xv[2][8], dv[2][8];
#pragma unroll
        for (int u = 0; u < 2; u++) { ldv8<T>(x + base + (size_t)(p + u * g.rows) * g.C, xv[u]); ldv8<T>(dy + base + (size_t)(p + u * g.rows) * g.C, dv[u]); }
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const float xh = (xv[u][e] - mu[e]) * rs[e];
                float d = dv[u][e];
                if (act) d *= silu_grad_t<T>(xh * ga[e] + be[e]);
                a[e] += d * xh; c[e] += d;
            }
    }
    for (; p < p1; p += g.rows) {
        float xv[8], dv[8];
        ldv8<T>(x + base + (size_t)p * g.C, xv); ldv8<T>(dy + base + (size_t)p * g.C, dv);
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const float xh = (xv[e] - mu[e]) * rs[e];
            float d = dv[e];
            if (act) d *= silu_grad_t<T>(xh * ga[e] + be[e]);
            a[e] += d * xh; c[e] += d;
        }
    }
    float* row = sm + (size_t)rl * g.C * 2;
#pragma unroll
    for (int e = 0; e < 8; e++) { row[(cc * 8 + e) * 2] = a[e]; row[(cc * 8 + e) * 2 + 1] = c[e]; }
    __syncthreads();
    for (int i = threadIdx.x; i < g.C * 2; i += blockDim.x) {
        float t = 0.f;
        for (int r = 0; r < g.rows; r++) t += sm[(size_t)r * g.C * 2 + i];
        part[((size_t)b * g.nch + chunk) * g.C * 2 + i] = t;
        sm[i] = t;   // row 0 now holds the block totals (each i is owned by exactly one thread)
    }
    __syncthreads();
    // gamma-weighted group partials for this (sample, chunk): what gn_bwd_apply needs
    float* gpart = part + (size_t)gridDim.y * g.nch * g.C * 2;
    for (int grp = threadIdx.x; grp < g.G; grp += blockDim.x) {
        float s1 = 0.f, s2 = 0.f;
        for (int ch = grp * g.cpg; ch < (grp + 1) * g.cpg; ch++) { s2 += gamma[ch] * sm[ch * 2]; s1 += gamma[ch] * sm[ch * 2 + 1]; }
        gpart[(((size_t)b * g.nch + chunk) * g.G + grp) * 2] = s1;
        gpart[(((size_t)b * g.nch + chunk) * g.G + grp) * 2 + 1] = s2;
    }
}

template <typename T, bool ACT>      // ACT (SiLU behind the norm) is a template parameter: a run-time `if (act)` per element is control flow inside the row loop,
__global__ void gn_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ dy,      // behind which the compiler waits s_waitcnt vmcnt(0)
                                    const float* __restrict__ stats, const float* __restrict__ gamma,
                                    const float* __restrict__ beta, const float* __restrict__ part,
                                    const T* __restrict__ add, T* __restrict__ dx, GnGeom g, int act,
                                    const float* __restrict__ gamma1, const float* __restrict__ beta1, int split) {
    if (gamma1 && (int)blockIdx.y >= split) { gamma = gamma1; beta = beta1; }
    // add (may be null): gradient arriving at x through its OTHER consumer (residual / shortcut branch); summed here
    // instead of in a separate elementwise kernel
    extern __shared__ float sm[];  // s1[G], s2[G]
    const int b = blockIdx.y, chunk = blockIdx.x;
    {   // fold the per-chunk group partials with all threads (see gn_apply_kernel)
        const float* gpart = part + (size_t)gridDim.y * g.nch * g.C * 2;
        gn_fold_partials(gpart + (size_t)b * g.nch * g.G * 2, g, sm);
        const float n = (float)g.cpg * (float)g.HW;
        for (int i = threadIdx.x; i < 2 * g.G; i += blockDim.x) sm[i] = sm[i] / n;
    }
    __syncthreads();
    const int cc = threadIdx.x % g.C8, rl = threadIdx.x / g.C8;
    float mu[8], rs[8], ga[8], be[8], m1[8], m2[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const int ch = cc * 8 + e, grp = ch / g.cpg;
        mu[e] = stats[((size_t)b * g.G + grp) * 2]; rs[e] = stats[((size_t)b * g.G + grp) * 2 + 1];
        ga[e] = gamma[ch]; be[e] = beta[ch]; m1[e] = sm[grp]; m2[e] = sm[g.G + grp];
    }
    const int p0 = chunk * g.ppb, p1 = min(g.HW, p0 + g.ppb);
    // branch-free, prefetching row loop (see gn_apply_kernel / ln_fwd_kernel); `add` == null is a descriptor of zero records
    const size_t sb = (size_t)b * g.HW * g.C;
    const long long sbytes = (long long)g.HW * g.C * sizeof(T);
    const __amdgpu_buffer_rsrc_t rx = mk_buf(x + sb, sbytes), rdy = mk_buf(dy + sb, sbytes), rdx = mk_buf(dx + sb, sbytes);
    const __amdgpu_buffer_rsrc_t radd = mk_buf(add ? add + sb : x, add ? sbytes : 0);
    auto off = [&](int p) { return p < p1 ? (unsigned)p * (unsigned)g.C + cc * 8 : SIDLSG_OOB; };
    // (rows stay RAW until they are used -- two raw sets instead of a raw set + unpacked floats: the kernel may run with 1024-thread
    // blocks, i.e. 128 registers, and 48 of them hold the per-channel constants)
    struct Rows { Raw8<T> x[2], dy[2], add[2]; };
    auto request = [&](int p, Rows& r) {
#pragma unroll
        for (int u = 0; u < 2; u++) { const unsigned o = off(p + u * g.rows); Raw8_bload(r.x[u], rx, o); Raw8_bload(r.dy[u], rdy, o); Raw8_bload(r.add[u], radd, o); }
    };
    int p = p0 + rl;
    if (p >= p1) return;
    Rows cur, nxt;
    request(p, cur);
    do {
        const int pn = p + 2 * g.rows;
        request(pn, nxt);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 2; u++) {
            float o[8], xv[8], dv[8], av[8];
            cur.x[u].unpack(xv); cur.dy[u].unpack(dv); cur.add[u].unpack(av);
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const float xh = (xv[e] - mu[e]) * rs[e];
                float d = dv[e];
                if (ACT) d *= silu_grad_t<T>(xh * ga[e] + be[e]);
                o[e] = rs[e] * (d * ga[e] - m1[e] - xh * m2[e]) + av[e];
            }
            bst8_out<T, false>(rdx, off(p + u * g.rows), o);
        }
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
        p = pn;
    } while (p < p1);
}

// dgamma[i] += sum_p part[p][i][0], dbeta[i] += sum_p part[p][i][1] in ONE launch (the partials are (d.xhat, d) pairs):
// these reductions are ~5 us launches with almost no work, two per normalisation layer and backward pass
__global__ void colsum_reduce2_kernel(const float* __restrict__ part, float* __restrict__ out0, float* __restrict__ out1, int P,
                                      size_t pstride, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int per = (P + gridDim.y - 1) / gridDim.y;
    const int p0 = blockIdx.y * per, p1 = min(P, p0 + per);
    float t0 = 0.f, t1 = 0.f;
    for (int p = p0; p < p1; p++) {
        const float2 v = *reinterpret_cast<const float2*>(part + (size_t)p * pstride + (size_t)i * 2);
        t0 += v.x; t1 += v.y;
    }
    if (p1 > p0) { unsafeAtomicAdd(out0 + i, t0); unsafeAtomicAdd(out1 + i, t1); }
}
static dim3 reduce_grid(int n, int P) {
    int split = (P + 15) / 16; if (split > 64) split = 64; if (split < 1) split = 1;
    return dim3((n + 255) / 256, split);
}

// ---- deferred parameter-gradient reductions (round 5) --------------------------------------------------------------------------
// A trainable backward pass holds 48 LayerNorm + ~32 two-kernel GroupNorm layers, each ending in a colsum_reduce2 launch of 1-10
// blocks' worth of work: ~160 launches and 2.2 ms of kernel time per iteration on the critical stream, for sums nobody reads before
// the end of the backward (or the next gradient-exchange marker).  With deferral switched on for a stream (sidlsg_defer_reductions),
// the norm backward entry points QUEUE the reduction (pointers + sizes, host side) instead of launching it, and
// sidlsg_flush_reductions(stream) runs everything queued for that stream as ONE launch: the job table travels as a by-value kernel
// argument (no device table, no copies: graph-capturable like every other launch of this library).  Same block decomposition and
// the same arithmetic per job as colsum_reduce2_kernel.  The caller keeps the partial-sum workspaces alive until the flush.
struct PairJob { const float* part; float* out0; float* out1; unsigned n, P, pstride, blk0, gx, gy; };      // 48 bytes
constexpr int PAIR_JOBS = 72;                                                                              // 72 x 48 = 3456 bytes of kernarg
struct PairBatch { int njobs, pad; PairJob j[PAIR_JOBS]; };

__global__ __launch_bounds__(256) void colsum_reduce2_batched_kernel(PairBatch b) {
    int lo = 0, hi = b.njobs - 1;                       // last job whose first block is <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (b.j[mid].blk0 <= blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const PairJob& q = b.j[lo];
    const unsigned local = blockIdx.x - q.blk0;
    const unsigned bx = local % q.gx, by = local / q.gx;
    const unsigned i = bx * 256 + threadIdx.x;
    if (i >= q.n) return;
    const unsigned per = (q.P + q.gy - 1) / q.gy;
    const unsigned p0 = by * per, p1 = min(q.P, p0 + per);
    float t0 = 0.f, t1 = 0.f;
    for (unsigned p = p0; p < p1; p++) {
        const float2 v = *reinterpret_cast<const float2*>(q.part + (size_t)p * q.pstride + (size_t)i * 2);
        t0 += v.x; t1 += v.y;
    }
    if (p1 > p0) { unsafeAtomicAdd(q.out0 + i, t0); unsafeAtomicAdd(q.out1 + i, t1); }
}

namespace {
struct DeferState {
    std::mutex mu;
    struct Q { bool accepting = false; std::vector<PairJob> jobs; };
    std::map<hipStream_t, Q> q;      // streams that have used deferral -> whether calls queue right now, and their queued jobs
};
DeferState& dst() { static DeferState s; return s; }
}  // namespace

static int flush_pairs_locked(hipStream_t s, std::vector<PairJob>& v) {
    int done = 0;
    for (size_t o = 0; o < v.size(); o += PAIR_JOBS) {
        PairBatch b{};
        b.njobs = (int)std::min<size_t>(PAIR_JOBS, v.size() - o);
        unsigned blocks = 0;
        for (int k = 0; k < b.njobs; k++) {
            b.j[k] = v[o + k];
            b.j[k].blk0 = blocks;
            blocks += b.j[k].gx * b.j[k].gy;
        }
        hipLaunchKernelGGL(colsum_reduce2_batched_kernel, dim3(blocks), dim3(256), 0, s, b);
        done += b.njobs;
    }
    v.clear();
    return done;
}

// true: queued (the caller must not launch the reduction itself)
static bool defer_pairs(hipStream_t s, const float* part, float* out0, float* out1, int P, size_t pstride, int n) {
    DeferState& d = dst();
    std::lock_guard<std::mutex> lk(d.mu);
    auto it = d.q.find(s);
    if (it == d.q.end() || !it->second.accepting) return false;
    const dim3 g = reduce_grid(n, P);
    it->second.jobs.push_back(PairJob{part, out0, out1, (unsigned)n, (unsigned)P, (unsigned)pstride, 0u, g.x, g.y});
    return true;
}
#define SIDLSG_REDUCE_PAIRS(s, ws, o0, o1, P, C)                                                                                   \
    do {                                                                                                                            \
        if (!defer_pairs(s, ws, o0, o1, P, (size_t)(C) * 2, C))                                                                     \
            SIDLSG_LAUNCH(colsum_reduce2_kernel, reduce_grid(C, P), dim3(256), 0, s, ws, o0, o1, P, (size_t)(C) * 2, C);            \
    } while (0)

// ---------------------------------------------------------------------------------------------
// LayerNorm over the last dim C (C % 8 == 0, C <= 2048): one wave per token row, lane owns NCH chunks of 8 channels.
// HBM/latency bound: what matters is bytes in flight per CU.  The kernels are specialised on NCH (1 for C = 320) so
// that they stay under ~64 VGPRs (8 waves per SIMD), and every wave keeps R rows in flight (all loads issued before
// the first reduction).  (The generic 4-chunk version held 160 VGPRs and one row per wave in flight: 2.8 TB/s.)
constexpr int LN_MAXCH = 4;   // chunks of 8 per lane -> C <= 2048

template <typename T, int NCH, int R, bool F8 = false>      // F8: y receives e4m3 bytes
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y,
                                                     float* __restrict__ stats, int rows, int C, float eps, int rpw,
                                                     const float* __restrict__ gamma1, const float* __restrict__ beta1, int split) {
    const int lane = threadIdx.x & 63;
    const int C8 = C >> 3;
    const int rbeg = (blockIdx.x * 4 + (threadIdx.x >> 6)) * rpw;      // rpw rows per wave
    const int rend = min(rows, rbeg + rpw);
    if (rbeg >= rows) return;
    if (gamma1 && rbeg >= split) { gamma = gamma1; beta = beta1; }     // grouped launch (split % rpw == 0): rows [split, rows) = second network
    float ga[NCH][8], be[NCH][8];
#pragma unroll
    for (int i = 0; i < NCH; i++) {
        const int cc = lane + 64 * i;
#pragma unroll
        for (int e = 0; e < 8; e++) { ga[i][e] = cc < C8 ? gamma[cc * 8 + e] : 0.f; be[i][e] = cc < C8 ? beta[cc * 8 + e] : 0.f; }
    }
    const float invC = 1.0f / (float)C;
    // gfx950 retires loads and stores through ONE in-order counter: a load issued behind a row group's stores cannot be waited for
    // before those stores are acknowledged, so "load R rows; normalise; store; next group" paid a load AND a store round trip per
    // group and wave (and with the loads inside `if (valid) ... else zeros` the compiler waited for each ROW at the end of its branch).
    // Now: every access is branch-free (buffer descriptors: out-of-range offsets read zeros / drop the store), the loads stay raw
    // registers until they are unpacked, and the next group is requested BEFORE the current one is stored -- the wait for it then
    // leaves the stores in flight.  The last group is peeled (no wasted prefetch, no branch inside the loop body).
    const __amdgpu_buffer_rsrc_t rx = mk_buf(x, (long long)rows * C * sizeof(T));
    const __amdgpu_buffer_rsrc_t ry = mk_buf(y, (long long)rows * C * (F8 ? 1 : sizeof(T)));
    const __amdgpu_buffer_rsrc_t rst = mk_buf(stats, stats ? (long long)rows * 8 : 0);
    auto load_group = [&](int r0, Raw8<T> (&raw)[R][NCH]) {
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int i = 0; i < NCH; i++) {
                const int cc = lane + 64 * i;
                Raw8_bload(raw[r][i], rx, (r0 + r < rend && cc < C8) ? (unsigned)(r0 + r) * (unsigned)C + cc * 8 : SIDLSG_OOB);
            }
    };
    auto finish_group = [&](int r0, const float (&v)[R][NCH][8]) {
        float mean[R], rstd[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < NCH; i++)
#pragma unroll
                for (int e = 0; e < 8; e++) t += v[r][i][e];
            mean[r] = t;
        }
#pragma unroll
        for (int r = 0; r < R; r++) mean[r] = wave_sum(mean[r]) * invC;
#pragma unroll
        for (int r = 0; r < R; r++) {
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < NCH; i++) {
                const float in = lane + 64 * i < C8 ? 1.f : 0.f;       // (pad lanes hold zeros: their (0 - mean)^2 must not count)
#pragma unroll
                for (int e = 0; e < 8; e++) { const float d = v[r][i][e] - mean[r]; q += in * d * d; }
            }
            rstd[r] = q;
        }
#pragma unroll
        for (int r = 0; r < R; r++) rstd[r] = rsqrtf(wave_sum(rstd[r]) * invC + eps);
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int row = r0 + r;
            const bool okr = row < rend;
            __builtin_amdgcn_raw_buffer_store_b64((u32x2){__float_as_uint(mean[r]), __float_as_uint(rstd[r])}, rst,
                                                  (okr && lane == 0) ? (unsigned)row * 8u : SIDLSG_OOB, 0, 0);
#pragma unroll
            for (int i = 0; i < NCH; i++) {
                const int cc = lane + 64 * i;
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; e++) o[e] = (v[r][i][e] - mean[r]) * rstd[r] * ga[i][e] + be[i][e];
                bst8_out<T, F8>(ry, (okr && cc < C8) ? (unsigned)row * (unsigned)C + cc * 8 : SIDLSG_OOB, o);
            }
        }
    };
    auto unpack = [&](const Raw8<T> (&raw)[R][NCH], float (&v)[R][NCH][8]) {
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int i = 0; i < NCH; i++) raw[r][i].unpack(v[r][i]);
    };
    Raw8<T> raw[R][NCH];
    float v[R][NCH][8];
    load_group(rbeg, raw);
    unpack(raw, v);
    int r0 = rbeg;
    for (; r0 + R < rend; r0 += R) {
        load_group(r0 + R, raw);
        __builtin_amdgcn_sched_barrier(0);       // (keep the scheduler from sinking the requests below the stores)
        finish_group(r0, v);
        __builtin_amdgcn_sched_barrier(0);       // (... and from hoisting the unpack, i.e. the wait, above the stores)
        unpack(raw, v);
    }
    finish_group(r0, v);
}

// dx = rstd*(dy*gamma - mean(dy*gamma) - xhat*mean(dy*gamma*xhat)); per-block partial dgamma/dbeta.
template <typename T, int NCH, int R>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                     const float* __restrict__ stats, const float* __restrict__ gamma,
                                                     const T* __restrict__ add, T* __restrict__ dx,
                                                     float* __restrict__ part, int rows, int C, int rows_per_block,
                                                     const float* __restrict__ gamma1, int split) {
    extern __shared__ float dyn[];  // [4 waves][C][2] for the param-grad partials
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (gamma1 && (int)blockIdx.x * rows_per_block >= split) gamma = gamma1;     // grouped launch (split % rows_per_block == 0)
    const int C8 = C >> 3;
    float ga[NCH][8], pg[NCH][8], pb[NCH][8];
#pragma unroll
    for (int i = 0; i < NCH; i++)
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int cc = lane + 64 * i;
            ga[i][e] = cc < C8 ? gamma[cc * 8 + e] : 0.f;
            pg[i][e] = pb[i][e] = 0.f;
        }
    const float invC = 1.0f / (float)C;
    const int rbeg = blockIdx.x * rows_per_block, rend = min(rows, rbeg + rows_per_block);
    // Same structure as ln_fwd_kernel: branch-free accesses through buffer descriptors (rows >= rend / pad lanes read zeros, their
    // stores are dropped; `add` == null is a descriptor of zero records), every load of a row group -- x, dy, the residual-branch
    // gradient and the statistics -- issued together and kept raw, and the NEXT group requested before the current one is stored.
    // (The former loop fetched `add` chunk by chunk between the stores: each chunk waited for the previous chunk's store.)
    const __amdgpu_buffer_rsrc_t rx = mk_buf(x, (long long)rows * C * sizeof(T)), rdy = mk_buf(dy, (long long)rows * C * sizeof(T));
    const __amdgpu_buffer_rsrc_t radd = mk_buf(add ? add : x, add ? (long long)rows * C * sizeof(T) : 0);
    const __amdgpu_buffer_rsrc_t rdx = mk_buf(dx, (long long)rows * C * sizeof(T)), rst = mk_buf(stats, (long long)rows * 8);
    struct Group { Raw8<T> x[R][NCH], dy[R][NCH], add[R][NCH]; u32x2 st[R]; };
    auto load_group = [&](int r0, Group& g) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            const bool okr = r0 + r < rend;
            g.st[r] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rst, okr ? (unsigned)(r0 + r) * 8u : SIDLSG_OOB, 0, 0));
#pragma unroll
            for (int i = 0; i < NCH; i++) {
                const int cc = lane + 64 * i;
                const unsigned o = (okr && cc < C8) ? (unsigned)(r0 + r) * (unsigned)C + cc * 8 : SIDLSG_OOB;
                Raw8_bload(g.x[r][i], rx, o);
                Raw8_bload(g.dy[r][i], rdy, o);
                Raw8_bload(g.add[r][i], radd, o);
            }
        }
    };
    // (the group stays RAW -- 12 registers per 8 elements of x, dy, add -- and is converted where it is used, twice: two register sets of
    // unpacked floats next to the partial-sum accumulators cost the wide variants their occupancy: 312 registers at NCH = 2)
    auto finish_group = [&](int r0, const Group& g) {
        float s1[R], s2[R], mean[R], rstd[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            mean[r] = __uint_as_float(g.st[r][0]); rstd[r] = __uint_as_float(g.st[r][1]);      // zero rows: mean = rstd = 0 -> xh = 0
            float a = 0.f, c = 0.f;
#pragma unroll
            for (int i = 0; i < NCH; i++) {
                float xv[8], dv[8];
                g.x[r][i].unpack(xv); g.dy[r][i].unpack(dv);
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const float xh = (xv[e] - mean[r]) * rstd[r];
                    const float d = dv[e];
                    pg[i][e] += d * xh; pb[i][e] += d;
                    const float dg = d * ga[i][e];
                    a += dg; c += dg * xh;
                }
            }
            s1[r] = a; s2[r] = c;
        }
#pragma unroll
        for (int r = 0; r < R; r++) { s1[r] = wave_sum(s1[r]) * invC; s2[r] = wave_sum(s2[r]) * invC; }
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int row = r0 + r;
#pragma unroll
            for (int i = 0; i < NCH; i++) {
                const int cc = lane + 64 * i;
                float o[8], xv[8], dv[8], a8[8];
                g.x[r][i].unpack(xv); g.dy[r][i].unpack(dv); g.add[r][i].unpack(a8);
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const float xh = (xv[e] - mean[r]) * rstd[r];
                    o[e] = rstd[r] * (dv[e] * ga[i][e] - s1[r] - xh * s2[r]) + a8[e];
                }
                bst8_out<T, false>(rdx, (row < rend && cc < C8) ? (unsigned)row * (unsigned)C + cc * 8 : SIDLSG_OOB, o);
            }
        }
    };
    int r0 = rbeg + wave * R;
    if (r0 < rend) {
        Group cur, nxt;
        load_group(r0, cur);
        for (; r0 + 4 * R < rend; r0 += 4 * R) {
            load_group(r0 + 4 * R, nxt);
            __builtin_amdgcn_sched_barrier(0);   // (keep the scheduler from sinking the requests below the stores)
            finish_group(r0, cur);
            __builtin_amdgcn_sched_barrier(0);
            cur = nxt;
        }
        finish_group(r0, cur);
    }
    if (part) {
#pragma unroll
        for (int i = 0; i < NCH; i++) {
            const int cc = lane + 64 * i;
            if (cc < C8) {
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    dyn[((size_t)wave * C + cc * 8 + e) * 2] = pg[i][e];
                    dyn[((size_t)wave * C + cc * 8 + e) * 2 + 1] = pb[i][e];
                }
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < C * 2; i += 256)
            part[(size_t)blockIdx.x * C * 2 + i] = dyn[i] + dyn[(size_t)C * 2 + i] + dyn[(size_t)2 * C * 2 + i] + dyn[(size_t)3 * C * 2 + i];
    }
}


// ---------------------------------------------------------------------------------------------
// ONE-PASS GroupNorm for the 32x32 / 16x16 / 8x8 stages (bf16): a block owns the channels of a few WHOLE groups of one sample
// (GPB groups = CW channels, CW % 8 == 0), loads its [HW][CW] slab ONCE into registers, reduces the group statistics inside the
// block (deterministic: per-thread partials -> LDS -> fixed-order sums) and normalises from the registers: x is read once, there is
// one launch instead of gn_stats + gn_apply, and no statistics workspace.  Fits when the slab is small enough for registers
// (P = ceil(HW / pixel rows in flight) 16-byte chunks per thread) AND measured faster: the 16x16 and 8x8 stages (30 of the 61
// GroupNorms of the SD UNet forward, the 8x8 stage and the 1280-channel 16x16 layers backward); 32x32 fits but is slower (the slab
// costs occupancy), 64x64 (1.3 MB per group range) does not fit: both keep the two-kernel path.
// Channel ranges of neighbouring blocks share cache lines (80-byte runs at a 1280-byte pitch): the block order keeps a sample's
// ranges on one XCD (blocks are numbered XCD-contiguously), so the shared lines meet in that XCD's L2.
struct GnSmall {
    int B, HW, C, G, cpg, GPB, CW8, rows, ngr;     // ngr = G / GPB channel ranges per sample
};

DEVFN int gn_small_block(const GnSmall& g, int& b, int& gr) {     // XCD-contiguous logical order: consecutive ids = (sample, range) row-major
    int bid = blockIdx.x;
    const int nblk = g.B * g.ngr;
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    b = bid / g.ngr; gr = bid - b * g.ngr;
    return bid;
}

// deterministic block reduction of per-thread per-channel pairs: sm[(rl * CW + ch) * 2 + {0,1}] -> out pairs per channel in
// sm[ch * 2 + ..] of row 0 (every (channel, stat) summed over the pixel rows by ONE thread, 4 independent chains)
DEVFN void gn_small_fold_rows(float* sm, int CW, int rows) {
    __syncthreads();
    for (int i = threadIdx.x; i < CW * 2; i += blockDim.x) {
        float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
        int r = 0;
        for (; r + 3 < rows; r += 4) {
            t0 += sm[(size_t)r * CW * 2 + i]; t1 += sm[(size_t)(r + 1) * CW * 2 + i];
            t2 += sm[(size_t)(r + 2) * CW * 2 + i]; t3 += sm[(size_t)(r + 3) * CW * 2 + i];
        }
        for (; r < rows; r++) t0 += sm[(size_t)r * CW * 2 + i];
        sm[i] = (t0 + t1) + (t2 + t3);       // row 0 is only read by its own (channel, stat) thread above
    }
    __syncthreads();
}

template <int P, bool F8>
__global__ __launch_bounds__(512) void gn_small_fwd_kernel(const bf16* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           bf16* __restrict__ y, float* __restrict__ stats, GnSmall g, float eps, int act,
                                                           const float* __restrict__ gamma1, const float* __restrict__ beta1, int split) {
    extern __shared__ float sm[];      // [rows][CW][2] partials, then [2 * GPB] mean / rstd behind them
    int b, gr;
    gn_small_block(g, b, gr);
    if (gamma1 && b >= split) { gamma = gamma1; beta = beta1; }
    const int CW = g.CW8 * 8;
    const int cc = threadIdx.x % g.CW8, rl = threadIdx.x / g.CW8;
    const bool live = rl < g.rows;
    const int c0 = gr * CW + cc * 8;                 // first of this thread's 8 channels
    const bf16* xb = x + (size_t)b * g.HW * g.C + c0;
    bf16x8 v[P];
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; e++) s[e] = q[e] = 0.f;
#pragma unroll
    for (int i = 0; i < P; i++) {
        const int p = rl + i * g.rows;
        v[i] = (live && p < g.HW) ? ld8(xb + (size_t)p * g.C) : zero8();
    }
#pragma unroll
    for (int i = 0; i < P; i++)
#pragma unroll
        for (int e = 0; e < 8; e++) { const float f = bf2f(v[i][e]); s[e] += f; q[e] += f * f; }
    if (live) {
        float* row = sm + ((size_t)rl * CW + cc * 8) * 2;
#pragma unroll
        for (int e = 0; e < 8; e++) { row[e * 2] = s[e]; row[e * 2 + 1] = q[e]; }
    }
    gn_small_fold_rows(sm, CW, g.rows);
    float* mr = sm + (size_t)g.rows * CW * 2;        // mean[GPB], rstd[GPB]
    if ((int)threadIdx.x < g.GPB) {
        float a = 0.f, c = 0.f;
        for (int ch = threadIdx.x * g.cpg; ch < ((int)threadIdx.x + 1) * g.cpg; ch++) { a += sm[ch * 2]; c += sm[ch * 2 + 1]; }
        const float n = (float)g.cpg * (float)g.HW;
        const float mean = a / n;
        const float rstd = rsqrtf(fmaxf(c / n - mean * mean, 0.f) + eps);
        mr[threadIdx.x] = mean; mr[g.GPB + threadIdx.x] = rstd;
        if (stats) { float* o = stats + ((size_t)b * g.G + gr * g.GPB + threadIdx.x) * 2; o[0] = mean; o[1] = rstd; }
    }
    __syncthreads();
    if (!live) return;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const int lg = (cc * 8 + e) / g.cpg;
        sc[e] = mr[g.GPB + lg] * gamma[c0 + e];
        sh[e] = beta[c0 + e] - mr[lg] * sc[e];
    }
    const size_t base = (size_t)b * g.HW * g.C + c0;
#pragma unroll
    for (int i = 0; i < P; i++) {
        const int p = rl + i * g.rows;
        if (p >= g.HW) break;
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            float f = bf2f(v[i][e]) * sc[e] + sh[e];
            if (act) f = silu_t<bf16>(f);
            o[e] = f;
        }
        st8_out<bf16, F8>(y, base + (size_t)p * g.C, o);
    }
}

// backward of the same: x and dy of the block's slab are read once; dx = rstd * (d * gamma - s1 / n - xhat * s2 / n) (+ add);
// dgamma / dbeta (trainable networks): per-channel block totals added with one fp32 atomic per channel and block.
template <int P>
__global__ __launch_bounds__(512) void gn_small_bwd_kernel(const bf16* __restrict__ x, const bf16* __restrict__ dy, const float* __restrict__ stats,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const bf16* __restrict__ add, bf16* __restrict__ dx, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, GnSmall g, int act,
                                                           const float* __restrict__ gamma1, const float* __restrict__ beta1, int split) {
    extern __shared__ float sm[];
    int b, gr;
    gn_small_block(g, b, gr);
    if (gamma1 && b >= split) { gamma = gamma1; beta = beta1; }
    const int CW = g.CW8 * 8;
    const int cc = threadIdx.x % g.CW8, rl = threadIdx.x / g.CW8;
    const bool live = rl < g.rows;
    const int c0 = gr * CW + cc * 8;
    const size_t base = (size_t)b * g.HW * g.C + c0;
    float mu[8], rs[8], ga[8], be[8], a[8], c[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const int grp = (c0 + e) / g.cpg;
        mu[e] = stats[((size_t)b * g.G + grp) * 2]; rs[e] = stats[((size_t)b * g.G + grp) * 2 + 1];
        ga[e] = gamma[c0 + e]; be[e] = beta[c0 + e]; a[e] = c[e] = 0.f;
    }
    bf16x8 xv[P], dv[P];
#pragma unroll
    for (int i = 0; i < P; i++) {
        const int p = rl + i * g.rows;
        const bool ok = live && p < g.HW;
        xv[i] = ok ? ld8(x + base + (size_t)p * g.C) : zero8();
        dv[i] = ok ? ld8(dy + base + (size_t)p * g.C) : zero8();
    }
#pragma unroll
    for (int i = 0; i < P; i++)
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const float xh = (bf2f(xv[i][e]) - mu[e]) * rs[e];
            float d = bf2f(dv[i][e]);                                   // rows past HW hold zeros: d = 0
            if (act) d *= silu_grad_t<bf16>(xh * ga[e] + be[e]);
            a[e] += d * xh; c[e] += d;
        }
    if (live) {
        float* row = sm + ((size_t)rl * CW + cc * 8) * 2;
#pragma unroll
        for (int e = 0; e < 8; e++) { row[e * 2] = a[e]; row[e * 2 + 1] = c[e]; }
    }
    gn_small_fold_rows(sm, CW, g.rows);
    if (dgamma && dbeta) {
        for (int i = threadIdx.x; i < CW; i += blockDim.x) {
            unsafeAtomicAdd(dgamma + gr * CW + i, sm[i * 2]);
            unsafeAtomicAdd(dbeta + gr * CW + i, sm[i * 2 + 1]);
        }
    }
    float* ms = sm + (size_t)g.rows * CW * 2;        // s1 / n [GPB], s2 / n [GPB]
    if ((int)threadIdx.x < g.GPB) {
        float s1 = 0.f, s2 = 0.f;
        for (int ch = threadIdx.x * g.cpg; ch < ((int)threadIdx.x + 1) * g.cpg; ch++) {
            const float gm = gamma[gr * CW + ch];
            s2 += gm * sm[ch * 2]; s1 += gm * sm[ch * 2 + 1];
        }
        const float n = (float)g.cpg * (float)g.HW;
        ms[threadIdx.x] = s1 / n; ms[g.GPB + threadIdx.x] = s2 / n;
    }
    __syncthreads();
    if (!live) return;
    float m1[8], m2[8];
#pragma unroll
    for (int e = 0; e < 8; e++) { const int lg = (cc * 8 + e) / g.cpg; m1[e] = ms[lg]; m2[e] = ms[g.GPB + lg]; }
#pragma unroll
    for (int i = 0; i < P; i++) {
        const int p = rl + i * g.rows;
        if (p >= g.HW) break;
        float o[8], av[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (add) ldv8<bf16>(add + base + (size_t)p * g.C, av);
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const float xh = (bf2f(xv[i][e]) - mu[e]) * rs[e];
            float d = bf2f(dv[i][e]);
            if (act) d *= silu_grad_t<bf16>(xh * ga[e] + be[e]);
            o[e] = rs[e] * (d * ga[e] - m1[e] - xh * m2[e]) + av[e];
        }
        stv8<bf16>(dx + base + (size_t)p * g.C, o);
    }
}

// ---------------------------------------------------------------------------------------------
// ONE-PASS GroupNorm for the LARGE stages (64x64, 32x32; bf16): a block owns ONE (sample, group) -- cpg channels of every pixel,
// i.e. D = cpg / 2 consecutive dwords per pixel at a pitch of C / 2 dwords (the group's offset is only 4-byte aligned: 10 channels
// = 20 bytes at C = 320, which is why the 16-byte kernels above cannot take single groups) -- and keeps its slab (HW x cpg bf16:
// 80 KB for a 64x64 x 320 layer) in REGISTERS: thread t holds dword t % D of the pixels t / D + i R, so its two channels are fixed
// (scale / shift / gamma are four registers) and consecutive lanes read consecutive dwords of a pixel, then the next pixel (a wave
// touches ~64 / D pixels per load instruction, each a 2 D-word run inside one or two cache lines).  Statistics: per-thread partials
// -> xor butterfly inside the wave -> one LDS slot per wave -> every thread adds the slots in the same order (deterministic, all
// threads hold bit-identical totals).  B x G blocks (512 at batch 16): x is read ONCE (the two-kernel path reads it twice -- three
// times in the backward, with dy twice), there is one launch instead of two (three with the parameter gradients), no statistics
// workspace.  The group ranges of one sample interleave inside every cache line, so the block order keeps a sample on one XCD.
struct GnGrp {
    int B, HW, C, G, cpg, D, R, NT;      // R pixel rows in flight, NT = D * R live threads (block size = NT rounded up to waves)
};
DEVFN void gn_grp_block(const GnGrp& g, int& b, int& gr) {
    int bid = blockIdx.x;
    const int nblk = g.B * g.G;
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    b = bid / g.G; gr = bid - b * g.G;
    // (scalar registers: the buffer descriptors built from them must not end up in VGPRs -- waterfall loops around every buffer operation)
    b = __builtin_amdgcn_readfirstlane(b); gr = __builtin_amdgcn_readfirstlane(gr);
}
// deterministic block sum of two values; every thread returns the same bits.  red: >= 2 * waves floats of LDS
DEVFN void gn_grp_sum2(float& a, float& b, float* red) {
#pragma unroll
    for (int o = 32; o; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) { red[2 * w] = a; red[2 * w + 1] = b; }
    __syncthreads();
    float ta = 0.f, tb = 0.f;
    for (int i = 0; i < nw; i++) { ta += red[2 * i]; tb += red[2 * i + 1]; }
    a = ta; b = tb;
}
// buffer descriptor over one sample's bytes behind the group's first channel: rows past HW and dead threads (offset 2^31) read zeros /
// store nothing; the per-thread VGPR offset is fixed, the pixel-row step travels in the wave-uniform soffset
DEVFN __amdgpu_buffer_rsrc_t gn_grp_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
DEVFN unsigned gn_ld(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) { return (unsigned)__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0); }
DEVFN void gn_st(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, unsigned v) { __builtin_amdgcn_raw_buffer_store_b32(v, r, voff, soff, 0); }
DEVFN float bflo(unsigned v) { return __uint_as_float(v << 16); }
DEVFN float bfhi(unsigned v) { return __uint_as_float(v & 0xffff0000u); }
DEVFN unsigned bfpack(float lo, float hi) {
    return (unsigned)__builtin_bit_cast(unsigned short, f2bf(lo)) | ((unsigned)__builtin_bit_cast(unsigned short, f2bf(hi)) << 16);
}

template <int P, bool ACT, int MAXT>       // (the activation is a template parameter: a run-time branch per row made the compiler spill the slab)
__global__ __launch_bounds__(MAXT) void gn_group_fwd_kernel(const bf16* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            bf16* __restrict__ y, float* __restrict__ stats, GnGrp g, float eps,
                                                            const float* __restrict__ gamma1, const float* __restrict__ beta1, int split) {
    __shared__ float red[32];
    int b, gr;
    gn_grp_block(g, b, gr);
    if (gamma1 && b >= split) { gamma = gamma1; beta = beta1; }
    const int d = threadIdx.x % g.D, r = threadIdx.x / g.D;
    const bool live = r < g.R;
    const size_t base = (size_t)b * g.HW * g.C + (size_t)gr * g.cpg;                  // first element of the (sample, group)
    const unsigned bytes = (unsigned)(((size_t)g.HW * g.C - (size_t)gr * g.cpg) * 2);   // to the end of the sample: rows >= HW are out of range
    const unsigned voff = live ? ((unsigned)r * (unsigned)g.C + 2u * d) * 2u : 0x80000000u;
    const unsigned step = (unsigned)g.R * (unsigned)g.C * 2u;                            // bytes between a thread's pixel rows
    const __amdgpu_buffer_rsrc_t rx = gn_grp_rsrc(x + base, bytes);
    unsigned v[P];
#pragma unroll
    for (int i = 0; i < P; i++) v[i] = gn_ld(rx, voff, i * step);
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int i = 0; i < P; i++) { const float f0 = bflo(v[i]), f1 = bfhi(v[i]); s += f0 + f1; q += f0 * f0 + f1 * f1; }
    gn_grp_sum2(s, q, red);
    const float n = (float)g.cpg * (float)g.HW;
    const float mean = s / n;
    const float rstd = rsqrtf(fmaxf(q / n - mean * mean, 0.f) + eps);
    if (threadIdx.x == 0 && stats) { float* o = stats + ((size_t)b * g.G + gr) * 2; o[0] = mean; o[1] = rstd; }
    if (!live) return;
    const int c = gr * g.cpg + 2 * d;
    const float sc0 = rstd * gamma[c], sc1 = rstd * gamma[c + 1];
    const float sh0 = beta[c] - mean * sc0, sh1 = beta[c + 1] - mean * sc1;
    const __amdgpu_buffer_rsrc_t ry = gn_grp_rsrc(y + base, bytes);
    // (the slab stays PACKED between the passes: without this the compiler keeps the unpacked floats of the first pass alive
    // -- two extra registers per dword -- instead of re-deriving them with one shift / and)
#pragma unroll
    for (int i = 0; i < P; i++) asm volatile("" : "+v"(v[i]));
#pragma unroll
    for (int i = 0; i < P; i++) {
        float f0 = bflo(v[i]) * sc0 + sh0, f1 = bfhi(v[i]) * sc1 + sh1;
        if (ACT) { f0 = silu_t<bf16>(f0); f1 = silu_t<bf16>(f1); }
        gn_st(ry, voff, i * step, bfpack(f0, f1));          // rows >= HW: dropped by the range check
    }
}

// backward of the same: x of the (sample, group) stays in registers, dy is streamed twice in 8-row pieces (its second read comes out
// of the L2 / Infinity Cache: the block has just fetched those 80 KB; holding dy as well -- 2 x 32 dwords plus the eight rows of
// unpacked operands the scheduler keeps in flight around the exponentials -- spilled at every block size).  dgamma / dbeta (trainable
// networks): the per-channel totals are folded over the pixel rows in LDS (fixed order) and added with one fp32 atomic per channel
// and block.
template <int P, bool ACT, int MAXT>
__global__ __launch_bounds__(MAXT) void gn_group_bwd_kernel(const bf16* __restrict__ x, const bf16* __restrict__ dy, const float* __restrict__ stats,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const bf16* __restrict__ add, bf16* __restrict__ dx, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, GnGrp g,
                                                            const float* __restrict__ gamma1, const float* __restrict__ beta1, int split) {
    static_assert(P % 8 == 0, "8-row pieces");
    __shared__ float red[32];
    extern __shared__ float part[];          // [R][D][4] per-thread channel partials (only with parameter gradients)
    int b, gr;
    gn_grp_block(g, b, gr);
    if (gamma1 && b >= split) { gamma = gamma1; beta = beta1; }
    const int d = threadIdx.x % g.D, r = threadIdx.x / g.D;
    const bool live = r < g.R;
    const size_t base = (size_t)b * g.HW * g.C + (size_t)gr * g.cpg;
    const unsigned bytes = (unsigned)(((size_t)g.HW * g.C - (size_t)gr * g.cpg) * 2);
    const unsigned voff = live ? ((unsigned)r * (unsigned)g.C + 2u * d) * 2u : 0x80000000u;
    const unsigned step = (unsigned)g.R * (unsigned)g.C * 2u;
    const __amdgpu_buffer_rsrc_t rx = gn_grp_rsrc(x + base, bytes), rd = gn_grp_rsrc(dy + base, bytes);
    const float mean = stats[((size_t)b * g.G + gr) * 2], rstd = stats[((size_t)b * g.G + gr) * 2 + 1];
    const int c = gr * g.cpg + 2 * d;
    const float ga0 = gamma[c], ga1 = gamma[c + 1], be0 = beta[c], be1 = beta[c + 1];
    unsigned xv[P];
#pragma unroll
    for (int i = 0; i < P; i++) xv[i] = gn_ld(rx, voff, i * step);
    float a0 = 0.f, a1 = 0.f, c0 = 0.f, c1 = 0.f;        // sum d * xhat, sum d per channel (d = dy * act'(.)); rows >= HW read dy = 0
    unsigned dv[8], dn[8];
#pragma unroll
    for (int j = 0; j < 8; j++) dv[j] = gn_ld(rd, voff, j * step);
#pragma unroll
    for (int i0 = 0; i0 < P; i0 += 8) {
        if (i0 + 8 < P) {            // the next piece is in flight while this one is worked on
#pragma unroll
            for (int j = 0; j < 8; j++) dn[j] = gn_ld(rd, voff, (i0 + 8 + j) * step);
        }
        // (pins the unpacking of these rows to this piece: pure arithmetic on the slab otherwise floats to the top of the kernel --
        // the instruction selector is not bound by sched_barrier -- and the unpacked floats of ALL rows stay live)
#pragma unroll
        for (int j = 0; j < 8; j++) asm volatile("" : "+v"(xv[i0 + j]));
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float h0 = (bflo(xv[i0 + j]) - mean) * rstd, h1 = (bfhi(xv[i0 + j]) - mean) * rstd;
            float d0 = bflo(dv[j]), d1 = bfhi(dv[j]);
            if (ACT) { d0 *= silu_grad_t<bf16>(h0 * ga0 + be0); d1 *= silu_grad_t<bf16>(h1 * ga1 + be1); }
            a0 += d0 * h0; a1 += d1 * h1; c0 += d0; c1 += d1;
        }
        // (and the piece's arithmetic ends here: without the dependency the exponentials sink below the next pieces' loads and all of
        // dy is live at once; "memory": the next piece's loads stay behind it)
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(c0), "+v"(c1) : : "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 8; j++) dv[j] = dn[j];
    }
    if (dgamma && dbeta) {
        if (live) { float* o = part + ((size_t)r * g.D + d) * 4; o[0] = a0; o[1] = c0; o[2] = a1; o[3] = c1; }
        __syncthreads();
        if ((int)threadIdx.x < g.D * 4) {        // (channel pair, stat) column summed over the rows by one thread, 4 independent chains
            float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
            const size_t st = (size_t)g.D * 4;
            int rr = 0;
            for (; rr + 3 < g.R; rr += 4) {
                t0 += part[rr * st + threadIdx.x]; t1 += part[(rr + 1) * st + threadIdx.x];
                t2 += part[(rr + 2) * st + threadIdx.x]; t3 += part[(rr + 3) * st + threadIdx.x];
            }
            for (; rr < g.R; rr++) t0 += part[rr * st + threadIdx.x];
            const float tot = (t0 + t1) + (t2 + t3);
            const int dd = threadIdx.x >> 2, k = threadIdx.x & 3;          // k: 0 a(ch 2dd), 1 c(ch 2dd), 2 a(ch 2dd+1), 3 c(ch 2dd+1)
            const int ch = gr * g.cpg + 2 * dd + (k >> 1);
            unsafeAtomicAdd(((k & 1) ? dbeta : dgamma) + ch, tot);
        }
    }
    float s1 = ga0 * c0 + ga1 * c1, s2 = ga0 * a0 + ga1 * a1;
    gn_grp_sum2(s1, s2, red);
    if (!live) return;
    const float n = (float)g.cpg * (float)g.HW;
    const float m1 = s1 / n, m2 = s2 / n;
    const __amdgpu_buffer_rsrc_t ra = gn_grp_rsrc(add ? add + base : x, add ? bytes : 0u), ro = gn_grp_rsrc(dx + base, bytes);
    unsigned av[8], an[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { dv[j] = gn_ld(rd, voff, j * step); av[j] = gn_ld(ra, voff, j * step); }   // no residual-branch gradient: zero-length buffer -> zeros
#pragma unroll
    for (int i0 = 0; i0 < P; i0 += 8) {
        if (i0 + 8 < P) {
#pragma unroll
            for (int j = 0; j < 8; j++) { dn[j] = gn_ld(rd, voff, (i0 + 8 + j) * step); an[j] = gn_ld(ra, voff, (i0 + 8 + j) * step); }
        }
#pragma unroll
        for (int j = 0; j < 8; j++) asm volatile("" : "+v"(xv[i0 + j]));       // (as above; also keeps the slab packed between the passes)
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float h0 = (bflo(xv[i0 + j]) - mean) * rstd, h1 = (bfhi(xv[i0 + j]) - mean) * rstd;
            float d0 = bflo(dv[j]), d1 = bfhi(dv[j]);
            if (ACT) { d0 *= silu_grad_t<bf16>(h0 * ga0 + be0); d1 *= silu_grad_t<bf16>(h1 * ga1 + be1); }
            const float o0 = rstd * (d0 * ga0 - m1 - h0 * m2) + bflo(av[j]), o1 = rstd * (d1 * ga1 - m1 - h1 * m2) + bfhi(av[j]);
            gn_st(ro, voff, (i0 + j) * step, bfpack(o0, o1));
        }
        asm volatile("" : : : "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 8; j++) { dv[j] = dn[j]; av[j] = an[j]; }
    }
}

// Geometry of the per-group one-pass kernels, or false.  pmax640 / pmax1024: the largest per-thread dword count the caller has a kernel
// for in a block of <= 640 / <= 1024 threads.  mask bit 0: forward, bit 1: backward (SIDLSG_GN_GROUP, A/B switch).
static bool gn_grp_geom(GnGrp& g, int& threads, int& P, int B, int HW, int C, int G, int pmax640, int pmax1024, int dirbit) {
    // Measured (MI355X, B = 16, tools/bench_kernels.py norm, SIDLSG_GN_GROUP=0/3, profiles/r04_gn_group_ab.txt): the BACKWARD wins where a
    // group is at least 40 bytes of a pixel and its slab at most ~128 KB -- 32x32: 640 channels 36 -> 26 us, 1280 channels 63 -> 38 us,
    // 1920 channels 70 -> 68 us -- and loses at 64x64 (320 channels: 51 -> 73 us, 640: 82 -> 107 us): with 20-byte runs every 128-byte
    // line is pulled through the L2 by six different blocks, and a 160 KB slab leaves one block per CU.  The FORWARD ties or loses
    // everywhere (29.5 -> 33 us at 64x64 x 320, 37-41 -> 36.5 us at 32x32 x 1920).  Inside the step the backward under those two limits
    // is neutral (SIDLSG_GN_GROUP=2 vs 0, three alternations: 208.3 vs 208.1 ms): the kernels stay in the library, tested, and OFF by
    // default (mask bit 2 lifts the limits for measurements and for the test).
    static const int mask = getenv("SIDLSG_GN_GROUP") ? atoi(getenv("SIDLSG_GN_GROUP")) : 0;
    static const int min_hw = getenv("SIDLSG_GN_GROUP_MIN_HW") ? atoi(getenv("SIDLSG_GN_GROUP_MIN_HW")) : 1024;
    if (!((mask >> dirbit) & 1) || C % G || (C & 1) || B <= 0 || HW < min_hw) return false;
    const int cpg = C / G;
    if (cpg & 1) return false;
    if (!(mask & 4) && (cpg < 20 || (size_t)HW * cpg * 2 > (128u << 10))) return false;
    g.B = B; g.HW = HW; g.C = C; g.G = G; g.cpg = cpg; g.D = cpg / 2;
    if (g.D > 64 || (size_t)HW * C * 2 >= 0x7fffffffull) return false;
    for (int pass = 0; pass < 2; pass++) {
        const int nt = pass ? 1024 : 640, pmax = pass ? pmax1024 : pmax640;
        int R = nt / g.D; if (R > HW) R = HW; if (R < 1) continue;
        const int p = (HW + R - 1) / R;
        if (p <= pmax) { g.R = R; g.NT = g.D * R; threads = (g.NT + 63) / 64 * 64; P = p; return true; }
    }
    return false;
}

// Geometry of the one-pass kernels for a shape, or false: pmax = the largest per-thread chunk count the caller has a kernel for.
static bool gn_small_geom(GnSmall& g, int& threads, int& P, int B, int HW, int C, int G, int pmax) {
    static const bool on = !(getenv("SIDLSG_GN_ONEPASS") && atoi(getenv("SIDLSG_GN_ONEPASS")) == 0);      // A/B switch
    if (!on || C % 8 || C % G || B <= 0 || HW <= 0) return false;
    g.B = B; g.HW = HW; g.C = C; g.G = G; g.cpg = C / G;
    int gpb = 0;
    for (int t = 1; t <= 8; t *= 2)
        if (G % t == 0 && (t * g.cpg) % 8 == 0) { gpb = t; break; }
    if (!gpb) return false;
    g.GPB = gpb; g.CW8 = gpb * g.cpg / 8; g.ngr = G / gpb;
    if (g.CW8 > 256) return false;
    for (int th = 256; th <= 512; th *= 2) {
        int rows = th / g.CW8; if (rows > HW) rows = HW; if (rows < 1) continue;
        const int p = (HW + rows - 1) / rows;
        if (p <= pmax) {
            g.rows = rows; threads = (g.CW8 * rows + 63) / 64 * 64; P = p;
            return ((size_t)rows * g.CW8 * 8 * 2 + 2 * gpb) * sizeof(float) <= 60000;
        }
    }
    return false;
}

static int gn_geom(GnGeom& g, int B, int HW, int C, int G) {
    if (C % 8 || C % G || C > GN_MAX_C || B <= 0 || HW <= 0) return SIDLSG_EINVAL;
    g.C = C; g.HW = HW; g.G = G; g.cpg = C / G; g.C8 = C / 8;
    static const int gn_threads = getenv("SIDLSG_GN_THREADS") ? atoi(getenv("SIDLSG_GN_THREADS")) : 256;
    g.rows = gn_threads / g.C8; if (g.rows < 1) g.rows = 1;
    if (g.rows > HW) g.rows = HW;
    // enough (sample, chunk) blocks for ~6 blocks per CU (latency bound: bytes in flight per CU are what count; with
    // 2 blocks per CU the kernels ran at 2-3 TB/s), each chunk >= rows*4 pixels
    static const int target = getenv("SIDLSG_GN_BLOCKS") ? atoi(getenv("SIDLSG_GN_BLOCKS")) : 512;
    int want = (target + B - 1) / B;
    int maxch = HW / (g.rows * 4); if (maxch < 1) maxch = 1;
    g.nch = want < maxch ? want : maxch;
    if (g.nch > 128) g.nch = 128;
    g.ppb = (HW + g.nch - 1) / g.nch;
    g.nch = (HW + g.ppb - 1) / g.ppb;
    return SIDLSG_OK;
}

extern "C" {

// workspace sizes (in floats) a caller must provide
int sidlsg_groupnorm_ws_floats(int B, int HW, int C, int G) {
    GnGeom g; if (gn_geom(g, B, HW, C, G)) return -1;
    return B * g.nch * (C * 2 + G * 2);   // bwd: per-channel partials [B][nch][C][2] + group partials [B][nch][G][2]
}
int sidlsg_groupnorm_nchunks(int B, int HW, int C, int G) {
    GnGeom g; if (gn_geom(g, B, HW, C, G)) return -1;
    return g.nch;
}

}  // extern "C" (reopened below)

// y = act(GroupNorm(x)); x,y: [B][HW][C]; stats: [B][G][2] fp32 (mean, rstd) saved for backward
template <typename T, bool F8 = false>
static int groupnorm_fwd_t(const void* x, const float* gamma, const float* beta, void* y, float* stats, float* ws,
                           int B, int HW, int C, int G, float eps, int silu, void* stream, const float* gamma1 = nullptr,
                           const float* beta1 = nullptr) {
    GnGeom g; if (int e = gn_geom(g, B, HW, C, G)) return e;
    if (gamma1 && ((B & 1) || !beta1)) return SIDLSG_EINVAL;
    const double tb_ = (double)B * HW * C * (sizeof(T) + (F8 ? 1 : sizeof(T)));
    SidlsgTraceScope ts(SIDLSG_FAM_GN_FWD, tb_, tb_);      // algorithmic bytes: read x, write y
    hipStream_t s = (hipStream_t)stream;
    if constexpr (std::is_same<T, bf16>::value) {
        GnSmall gs; int th, P;
        // one-pass kernel where it measured faster than stats + apply (tools/bench_kernels.py norm, SIDLSG_GN_ONEPASS=0/1, B = 8 / 16 / 32):
        // <= 12 chunks per thread = the 16x16 and 8x8 stages (13.6 -> 9.7 us, 22.1 -> 13.4 us at B = 16); at 32x32 (21 chunks) the register
        // slab costs occupancy and the two streaming kernels win (19.9 vs 24.2 us)
        if (gn_small_geom(gs, th, P, B, HW, C, G, 12)) {
            const size_t lds = ((size_t)gs.rows * gs.CW8 * 8 * 2 + 2 * gs.GPB) * sizeof(float);
#define GN_SF(PP) SIDLSG_LAUNCH((gn_small_fwd_kernel<PP, F8>), dim3(B * gs.ngr), dim3(th), lds, s, (const bf16*)x, gamma, beta, (bf16*)y, stats, gs, eps, silu, gamma1, beta1, B / 2)
            if (P <= 2) GN_SF(2); else if (P <= 4) GN_SF(4); else if (P <= 8) GN_SF(8); else GN_SF(12);
#undef GN_SF
            return sidlsg_last_error();
        }
    }
    if constexpr (std::is_same<T, bf16>::value && !F8) {
        GnGrp gg; int th, P;
        if (gn_grp_geom(gg, th, P, B, HW, C, G, 64, 64, 0)) {
#define GN_GF3(PP, AA, TT) SIDLSG_LAUNCH((gn_group_fwd_kernel<PP, AA, TT>), dim3(B * G), dim3(th), 0, s, (const bf16*)x, gamma, beta, (bf16*)y, stats, gg, eps, gamma1, beta1, B / 2)
#define GN_GF2(PP, AA) do { if (th <= 640) GN_GF3(PP, AA, 640); else GN_GF3(PP, AA, 1024); } while (0)
#define GN_GF(PP) do { if (silu) GN_GF2(PP, true); else GN_GF2(PP, false); } while (0)
            if (P <= 16) GN_GF(16); else if (P <= 32) GN_GF(32); else GN_GF(64);
#undef GN_GF3
#undef GN_GF2
#undef GN_GF
            return sidlsg_last_error();
        }
    }
    const int threads = g.C8 * g.rows;
    SIDLSG_LAUNCH(gn_stats_kernel<T>, dim3(g.nch, B), dim3(threads), (size_t)g.rows * C * 2 * sizeof(float), s,
                       (const T*)x, ws, g);
    SIDLSG_LAUNCH((gn_apply_kernel<T, F8>), dim3(g.nch, B), dim3(threads), (size_t)2 * G * (1 + GN_FOLD) * sizeof(float), s,
                       (const T*)x, ws, gamma, beta, (T*)y, stats, g, eps, silu, gamma1, beta1, B / 2);
    return sidlsg_last_error();
}

// dx (and optionally dgamma/dbeta +=) of y = act(GroupNorm(x))
template <typename T>
static int groupnorm_bwd_t(const void* x, const void* dy, const float* stats, const float* gamma, const float* beta,
                           const void* dres, void* dx, float* dgamma, float* dbeta, float* ws, int B, int HW, int C, int G,
                           int silu, void* stream, const float* gamma1 = nullptr, const float* beta1 = nullptr) {
    GnGeom g; if (int e = gn_geom(g, B, HW, C, G)) return e;
    if (gamma1 && ((B & 1) || !beta1 || dgamma || dbeta)) return SIDLSG_EINVAL;      // grouped: frozen networks (no parameter gradients)
    const double tb_ = (double)B * HW * C * sizeof(T) * (dres ? 4 : 3);
    SidlsgTraceScope ts(SIDLSG_FAM_GN_BWD, tb_, tb_);                // read x, dy (, dres), write dx
    hipStream_t s = (hipStream_t)stream;
    if constexpr (std::is_same<T, bf16>::value) {
        GnSmall gs; int th, P;
        // one-pass kernel: x and dy read once, one launch (+ atomics for dgamma / dbeta); <= 8 chunks of x AND dy per thread (256 VGPRs
        // without spills): 8x8 stage 22.7 -> 9.0 us, 16x16 at 1280 channels 32.0 -> 15.6 us; beyond that the two-kernel path is faster
        if (gn_small_geom(gs, th, P, B, HW, C, G, 8)) {
            const size_t lds = ((size_t)gs.rows * gs.CW8 * 8 * 2 + 2 * gs.GPB) * sizeof(float);
#define GN_SB(PP) SIDLSG_LAUNCH((gn_small_bwd_kernel<PP>), dim3(B * gs.ngr), dim3(th), lds, s, (const bf16*)x, (const bf16*)dy, stats, gamma, beta, (const bf16*)dres, (bf16*)dx, dgamma, dbeta, gs, silu, gamma1, beta1, B / 2)
            if (P <= 2) GN_SB(2); else if (P <= 4) GN_SB(4); else GN_SB(8);
#undef GN_SB
            return sidlsg_last_error();
        }
    }
    if constexpr (std::is_same<T, bf16>::value) {
        GnGrp gg; int th, P;
        if (gn_grp_geom(gg, th, P, B, HW, C, G, 64, 32, 1)) {      // x in registers (<= 64 dwords per thread in <= 640 threads, 32 in <= 1024), dy streamed twice
            const size_t lds = (dgamma && dbeta) ? (size_t)gg.R * gg.D * 4 * sizeof(float) : 0;
#define GN_GB3(PP, AA, TT) SIDLSG_LAUNCH((gn_group_bwd_kernel<PP, AA, TT>), dim3(B * G), dim3(th), lds, s, (const bf16*)x, (const bf16*)dy, stats, gamma, beta, (const bf16*)dres, (bf16*)dx, dgamma, dbeta, gg, gamma1, beta1, B / 2)
#define GN_GB2(PP, AA) do { if (th <= 640) GN_GB3(PP, AA, 640); else GN_GB3(PP, AA, 1024); } while (0)
#define GN_GB(PP) do { if (silu) GN_GB2(PP, true); else GN_GB2(PP, false); } while (0)
            if (P <= 16) GN_GB(16); else if (P <= 32) GN_GB(32); else GN_GB(64);
#undef GN_GB3
#undef GN_GB2
#undef GN_GB
            return sidlsg_last_error();
        }
    }
    const int threads = g.C8 * g.rows;
    SIDLSG_LAUNCH(gn_bwd_stats_kernel<T>, dim3(g.nch, B), dim3(threads), (size_t)g.rows * C * 2 * sizeof(float), s,
                       (const T*)x, (const T*)dy, stats, gamma, beta, ws, g, silu, gamma1, beta1, B / 2);
    if (silu) SIDLSG_LAUNCH((gn_bwd_apply_kernel<T, true>), dim3(g.nch, B), dim3(threads), (size_t)2 * G * (1 + GN_FOLD) * sizeof(float), s,
                       (const T*)x, (const T*)dy, stats, gamma, beta, ws, (const T*)dres, (T*)dx, g, silu, gamma1, beta1, B / 2);
    else SIDLSG_LAUNCH((gn_bwd_apply_kernel<T, false>), dim3(g.nch, B), dim3(threads), (size_t)2 * G * (1 + GN_FOLD) * sizeof(float), s,
                       (const T*)x, (const T*)dy, stats, gamma, beta, ws, (const T*)dres, (T*)dx, g, silu, gamma1, beta1, B / 2);
    if (dgamma && dbeta) {
        const int P = B * g.nch;
        SIDLSG_REDUCE_PAIRS(s, ws, dgamma, dbeta, P, C);
    }
    return sidlsg_last_error();
}

// y = LayerNorm(x) over C; x,y [rows][C]; stats [rows][2]
template <typename T, bool F8 = false>
static int layernorm_fwd_t(const void* x, const float* gamma, const float* beta, void* y, float* stats, int rows, int C,
                           float eps, void* stream, const float* gamma1 = nullptr, const float* beta1 = nullptr) {
    if (C % 8 || C > 8 * 64 * LN_MAXCH || rows <= 0) return SIDLSG_EINVAL;
    if (gamma1 && ((rows & 1) || !beta1)) return SIDLSG_EINVAL;
    const double tb_ = (double)rows * C * (sizeof(T) + (F8 ? 1 : sizeof(T)));
    SidlsgTraceScope ts(SIDLSG_FAM_LN_FWD, tb_, tb_);
    const int nch = (C / 8 + 63) / 64;
    const int R = nch <= 1 ? 4 : 2;
    // rows per wave: enough waves to fill the chip (>= ~4096), at most 16 rows (amortises the gamma/beta loads)
    int rpw = rows / 4096; rpw = rpw < R ? R : (rpw > 16 ? 16 : rpw); rpw = (rpw + R - 1) / R * R;
    if (gamma1) {       // grouped launch: a wave's row range must not straddle the two networks' halves
        while (rpw > R && (rows / 2) % rpw) rpw -= R;
        if ((rows / 2) % rpw) return SIDLSG_EINVAL;
    }
    const dim3 grid((rows + 4 * rpw - 1) / (4 * rpw));
#define LN_FWD(NCH, RR) SIDLSG_LAUNCH((ln_fwd_kernel<T, NCH, RR, F8>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)x, gamma, \
                                           beta, (T*)y, stats, rows, C, eps, rpw, gamma1, beta1, rows / 2)
    if (nch == 1) LN_FWD(1, 4); else if (nch == 2) LN_FWD(2, 2); else if (nch == 3) LN_FWD(3, 2); else LN_FWD(4, 2);
#undef LN_FWD
    return sidlsg_last_error();
}
static int layernorm_bwd_nblocks(int rows) {
    int nb = (rows + 15) / 16; if (nb > 1024) nb = 1024; if (nb < 1) nb = 1; return nb;
}

// dx, and dgamma/dbeta (+=) when non-null; ws: [nblocks][C][2] floats
template <typename T>
static int layernorm_bwd_t(const void* x, const void* dy, const float* stats, const float* gamma, const void* dres, void* dx,
                           float* dgamma, float* dbeta, float* ws, int rows, int C, void* stream, const float* gamma1 = nullptr) {
    if (C % 8 || C > 8 * 64 * LN_MAXCH || rows <= 0) return SIDLSG_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const double tb_ = (double)rows * C * sizeof(T) * (dres ? 4 : 3);
    SidlsgTraceScope ts(SIDLSG_FAM_LN_BWD, tb_, tb_);
    int nb = layernorm_bwd_nblocks(rows);
    int rpb = (rows + nb - 1) / nb;
    if (gamma1) {       // grouped launch (frozen networks): blocks must not straddle the halves
        if ((rows & 1) || dgamma || dbeta) return SIDLSG_EINVAL;
        const int half = rows / 2;
        int nbh = layernorm_bwd_nblocks(half);
        rpb = (half + nbh - 1) / nbh;
        while (half % rpb) rpb++;
        nb = 2 * (half / rpb);
    }
    const bool pg = dgamma && dbeta;
    const int nch = (C / 8 + 63) / 64;
    const size_t lds = pg ? (size_t)4 * C * 2 * sizeof(float) : 0;
#define LN_BWD(NCH, R) SIDLSG_LAUNCH((ln_bwd_kernel<T, NCH, R>), dim3(nb), dim3(256), lds, s, (const T*)x, (const T*)dy, \
                                          stats, gamma, (const T*)dres, (T*)dx, pg ? ws : nullptr, rows, C, rpb, gamma1, rows / 2)
    if (nch == 1) LN_BWD(1, 2); else if (nch == 2) LN_BWD(2, 2); else if (nch == 3) LN_BWD(3, 1); else LN_BWD(4, 1);
#undef LN_BWD
    if (pg) {
        SIDLSG_REDUCE_PAIRS(s, ws, dgamma, dbeta, nb, C);
    }
    return sidlsg_last_error();
}

extern "C" {

// Deferred parameter-gradient reductions of the normalisation backward kernels (see colsum_reduce2_batched_kernel).
// on = 1: from now on sidlsg_layernorm_bwd* / sidlsg_groupnorm_bwd* called with `stream` queue their dgamma / dbeta reduction instead
// of launching it; the partial sums stay in the caller's `ws` (which must stay allocated and untouched) until
// sidlsg_flush_reductions(stream) launches all queued reductions as one kernel on that stream.  on = 2: later calls launch their own
// reduction again, what is queued stays queued (the host wrapper brackets exactly ITS calls with 1 / 2, so direct users of the C ABI
// on the same stream are never deferred behind their back).  on = 0: flush and forget the stream.  on = 3: forget the stream WITHOUT launching what is queued (the queue of a backward pass that raised).  Returns the number of reductions flushed (>= 0) or a negative error.
int sidlsg_defer_reductions(void* stream, int on) {
    DeferState& d = dst();
    std::lock_guard<std::mutex> lk(d.mu);
    hipStream_t s = (hipStream_t)stream;
    if (on == 1) { d.q[s].accepting = true; return 0; }
    auto it = d.q.find(s);
    if (it == d.q.end()) return 0;
    if (on == 2) { it->second.accepting = false; return 0; }          // stop queuing, keep what is queued
    if (on == 3) { const int n = (int)it->second.jobs.size(); d.q.erase(it); return n; }      // DISCARD what is queued (a backward pass that died): nothing is launched
    const int n = flush_pairs_locked(s, it->second.jobs);
    d.q.erase(it);
    return sidlsg_last_error() == SIDLSG_OK ? n : -1;
}
int sidlsg_flush_reductions(void* stream) {
    DeferState& d = dst();
    std::lock_guard<std::mutex> lk(d.mu);
    auto it = d.q.find((hipStream_t)stream);
    if (it == d.q.end() || it->second.jobs.empty()) return 0;
    const int n = flush_pairs_locked((hipStream_t)stream, it->second.jobs);
    return sidlsg_last_error() == SIDLSG_OK ? n : -1;
}
int sidlsg_pending_reductions(void* stream) {
    DeferState& d = dst();
    std::lock_guard<std::mutex> lk(d.mu);
    auto it = d.q.find((hipStream_t)stream);
    return it == d.q.end() ? -1 : (int)it->second.jobs.size();
}

}  // extern "C"

extern "C" {

#define SIDLSG_BOTH(name, tmpl, params, args) \
    int name params { return tmpl<bf16> args; }  \
    int name##_f32 params { return tmpl<float> args; }

SIDLSG_BOTH(sidlsg_groupnorm_fwd, groupnorm_fwd_t,
            (const void* x, const float* gamma, const float* beta, void* y, float* stats, float* ws, int B, int HW, int C, int G, float eps, int silu, void* stream),
            (x, gamma, beta, y, stats, ws, B, HW, C, G, eps, silu, stream))
SIDLSG_BOTH(sidlsg_groupnorm_bwd, groupnorm_bwd_t,
            (const void* x, const void* dy, const float* stats, const float* gamma, const float* beta, const void* dres, void* dx, float* dgamma, float* dbeta, float* ws, int B, int HW, int C, int G, int silu, void* stream),
            (x, dy, stats, gamma, beta, dres, dx, dgamma, dbeta, ws, B, HW, C, G, silu, stream))
SIDLSG_BOTH(sidlsg_layernorm_fwd, layernorm_fwd_t,
            (const void* x, const float* gamma, const float* beta, void* y, float* stats, int rows, int C, float eps, void* stream),
            (x, gamma, beta, y, stats, rows, C, eps, stream))
int sidlsg_layernorm_bwd_nblocks(int rows) { return layernorm_bwd_nblocks(rows); }
// e4m3 outputs (bf16 inputs): y8 holds rows * C (B * HW * C) bytes
int sidlsg_layernorm_fwd_fp8(const void* x, const float* gamma, const float* beta, void* y8, float* stats, int rows, int C, float eps, void* stream) {
    return layernorm_fwd_t<bf16, true>(x, gamma, beta, y8, stats, rows, C, eps, stream);
}
int sidlsg_groupnorm_fwd_fp8(const void* x, const float* gamma, const float* beta, void* y8, float* stats, float* ws, int B, int HW, int C, int G,
                             float eps, int silu, void* stream) {
    return groupnorm_fwd_t<bf16, true>(x, gamma, beta, y8, stats, ws, B, HW, C, G, eps, silu, stream);
}
// Grouped variants (bf16; two frozen networks on one stacked batch): samples / rows of the first half are normalised with
// (gamma, beta), those of the second half with (gamma1, beta1).  Backward: data gradient only.
int sidlsg_groupnorm_fwd_g2(const void* x, const float* gamma, const float* beta, const float* gamma1, const float* beta1, void* y, float* stats,
                            float* ws, int B, int HW, int C, int G, float eps, int silu, void* stream) {
    if (!gamma1) return SIDLSG_EINVAL;
    return groupnorm_fwd_t<bf16>(x, gamma, beta, y, stats, ws, B, HW, C, G, eps, silu, stream, gamma1, beta1);
}
int sidlsg_groupnorm_bwd_g2(const void* x, const void* dy, const float* stats, const float* gamma, const float* beta, const float* gamma1,
                            const float* beta1, const void* dres, void* dx, float* ws, int B, int HW, int C, int G, int silu, void* stream) {
    if (!gamma1) return SIDLSG_EINVAL;
    return groupnorm_bwd_t<bf16>(x, dy, stats, gamma, beta, dres, dx, nullptr, nullptr, ws, B, HW, C, G, silu, stream, gamma1, beta1);
}
int sidlsg_layernorm_fwd_g2(const void* x, const float* gamma, const float* beta, const float* gamma1, const float* beta1, void* y, float* stats,
                            int rows, int C, float eps, void* stream) {
    if (!gamma1) return SIDLSG_EINVAL;
    return layernorm_fwd_t<bf16>(x, gamma, beta, y, stats, rows, C, eps, stream, gamma1, beta1);
}
int sidlsg_layernorm_bwd_g2(const void* x, const void* dy, const float* stats, const float* gamma, const float* gamma1, const void* dres, void* dx,
                            int rows, int C, void* stream) {
    if (!gamma1) return SIDLSG_EINVAL;
    return layernorm_bwd_t<bf16>(x, dy, stats, gamma, dres, dx, nullptr, nullptr, nullptr, rows, C, stream, gamma1);
}
SIDLSG_BOTH(sidlsg_layernorm_bwd, layernorm_bwd_t,
            (const void* x, const void* dy, const float* stats, const float* gamma, const void* dres, void* dx, float* dgamma, float* dbeta, float* ws, int rows, int C, void* stream),
            (x, dy, stats, gamma, dres, dx, dgamma, dbeta, ws, rows, C, stream))

}  // extern "C"
