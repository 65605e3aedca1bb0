// Deferred reductions (round 5): the tiny "fold the partial sums" launches that end a normalisation backward (dgamma / dbeta from per-block
// partials) and a pixel-split weight gradient (dW from per-split slabs) are QUEUED on the host instead of launched, and everything queued
// for a stream runs as ONE launch per job kind when the host asks for it (sidlsg_flush_reductions): a trainable backward pass holds ~80 of
// the former and ~170 of the latter, each a 4-16 us launch of a few blocks' worth of work with a dependent-launch boundary behind it.
// The job table travels as a by-value kernel argument (no device table, no copy: graph-capturable like every launch of this library).
//   pairs: out0[i] += sum_p part[p][i][0], out1[i] += sum_p part[p][i][1]   (norm.hip; same block decomposition and arithmetic as
//          colsum_reduce2_kernel).  The caller keeps `part` alive and untouched until the flush.
//   slabs: dW[i] (+)= sum_s ws[s][i]   (gemm.hip: wgrad_reduce_kernel's arithmetic, same summation order -> bit-identical).  The slabs
//          live in the stream's split workspace, which becomes a bump ARENA while deferral is accepting: sidlsg_defer_slab_alloc hands
//          out consecutive pieces and flushes the queue itself when the arena is full; anything else that wants the workspace of that
//          stream (split-K GEMMs) calls sidlsg_defer_release_workspace first.
// Deferral is per stream and per CALL: the host wrapper switches it on right before its own launch and back to "keep the queue, accept
// nothing" right after (sidlsg_defer_reductions 1 / 2), so direct users of the C ABI are never deferred behind their back.
#include "common.h"
#include <algorithm>
#include <map>
#include <mutex>
#include <vector>

namespace {
struct PairJob { const float* part; float* out0; float* out1; unsigned n, P, pstride, blk0, gx, gy; };      // 48 bytes
constexpr int PAIR_JOBS = 72;                                                                              // 3456 bytes of kernarg
struct PairBatch { int njobs, pad; PairJob j[PAIR_JOBS]; };
struct SlabJob { const float* ws; float* dW; unsigned n4, blk0; unsigned short splits, assign; unsigned pad; };   // 32 bytes
constexpr int SLAB_JOBS = 112;                                                                             // 3584 bytes of kernarg
struct SlabBatch { int njobs, pad; SlabJob j[SLAB_JOBS]; };
}  // namespace

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

// dW[i] (+)= sum_s ws[s * n + i], four floats per thread, slabs summed from 0 in slab order (wgrad_reduce_kernel's order)
__global__ __launch_bounds__(256) void wgrad_reduce_batched_kernel(SlabBatch b) {
    int lo = 0, hi = b.njobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (b.j[mid].blk0 <= blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const SlabJob q = b.j[lo];
    const unsigned i4 = (blockIdx.x - q.blk0) * 256 + threadIdx.x;
    if (i4 >= q.n4) return;
    const size_t i = (size_t)i4 * 4, n = (size_t)q.n4 * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (!q.assign) v = *reinterpret_cast<const f32x4*>(q.dW + i);
    int sidx = 0;
    const int splits = q.splits;
    for (; sidx + 3 < splits; sidx += 4) {          // four independent slab loads in flight, fixed summation order
        const f32x4 a = *reinterpret_cast<const f32x4*>(q.ws + (size_t)sidx * n + i);
        const f32x4 c = *reinterpret_cast<const f32x4*>(q.ws + (size_t)(sidx + 1) * n + i);
        const f32x4 d = *reinterpret_cast<const f32x4*>(q.ws + (size_t)(sidx + 2) * n + i);
        const f32x4 e = *reinterpret_cast<const f32x4*>(q.ws + (size_t)(sidx + 3) * n + i);
        v += a; v += c; v += d; v += e;
    }
    for (; sidx < splits; sidx++) v += *reinterpret_cast<const f32x4*>(q.ws + (size_t)sidx * n + i);
    *reinterpret_cast<f32x4*>(q.dW + i) = v;
}

namespace {
struct Q {
    bool accepting = false;
    std::vector<PairJob> pairs;
    std::vector<SlabJob> slabs;
    float* arena = nullptr;          // slab arena = the stream's split workspace while slab jobs are pending
    long long arena_bytes = 0, used = 0;
};
struct DeferState {
    std::mutex mu;
    std::map<hipStream_t, Q> q;      // streams that have used deferral
};
DeferState& dst() { static DeferState s; return s; }

int flush_pairs(hipStream_t s, std::vector<PairJob>& v) {
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
int flush_slabs(hipStream_t s, Q& q) {
    int done = 0;
    std::vector<SlabJob>& v = q.slabs;
    for (size_t o = 0; o < v.size(); o += SLAB_JOBS) {
        SlabBatch b{};
        b.njobs = (int)std::min<size_t>(SLAB_JOBS, v.size() - o);
        unsigned blocks = 0;
        for (int k = 0; k < b.njobs; k++) {
            b.j[k] = v[o + k];
            b.j[k].blk0 = blocks;
            blocks += (b.j[k].n4 + 255) / 256;
        }
        hipLaunchKernelGGL(wgrad_reduce_batched_kernel, dim3(blocks), dim3(256), 0, s, b);
        done += b.njobs;
    }
    v.clear();
    q.used = 0;          // later launches on this stream are ordered behind the reduction: the arena is free again
    return done;
}
}  // namespace

bool sidlsg_defer_pairs(hipStream_t s, const float* part, float* out0, float* out1, int P, size_t pstride, int n, unsigned gx, unsigned gy) {
    DeferState& d = dst();
    std::lock_guard<std::mutex> lk(d.mu);
    auto it = d.q.find(s);
    if (it == d.q.end() || !it->second.accepting) return false;
    it->second.pairs.push_back(PairJob{part, out0, out1, (unsigned)n, (unsigned)P, (unsigned)pstride, 0u, gx, gy});
    return true;
}

// A piece of `bytes` bytes of the stream's split workspace [base, base + total) for the slabs of a deferred weight gradient, or
// null when the stream is not deferring (the caller then uses the workspace from its start and reduces at once, as ever).
float* sidlsg_defer_slab_alloc(hipStream_t s, float* base, long long total, long long bytes) {
    DeferState& d = dst();
    std::lock_guard<std::mutex> lk(d.mu);
    auto it = d.q.find(s);
    if (it == d.q.end() || !it->second.accepting || !base || bytes > total) return nullptr;
    Q& q = it->second;
    if (q.arena != base || q.arena_bytes != total) {          // another workspace: nothing of the old one may stay pending
        flush_slabs(s, q);
        q.arena = base; q.arena_bytes = total; q.used = 0;
    }
    bytes = (bytes + 255) / 256 * 256;
    if (q.used + bytes > total) flush_slabs(s, q);
    float* p = reinterpret_cast<float*>(reinterpret_cast<char*>(base) + q.used);
    q.used += bytes;
    return p;
}
bool sidlsg_defer_slabs(hipStream_t s, const float* ws, float* dW, size_t n, int splits, int assign) {
    DeferState& d = dst();
    std::lock_guard<std::mutex> lk(d.mu);
    auto it = d.q.find(s);
    if (it == d.q.end() || !it->second.accepting || (n & 3) || n / 4 > 0xffffffffull || splits > 65535) return false;
    it->second.slabs.push_back(SlabJob{ws, dW, (unsigned)(n / 4), 0u, (unsigned short)splits, (unsigned short)(assign ? 1 : 0), 0u});
    return true;
}
// Somebody else is about to write the stream's split workspace (a split-K GEMM): pending slab reductions first.
void sidlsg_defer_release_workspace(hipStream_t s) {
    DeferState& d = dst();
    std::lock_guard<std::mutex> lk(d.mu);
    auto it = d.q.find(s);
    if (it != d.q.end() && !it->second.slabs.empty()) flush_slabs(s, it->second);
}

extern "C" {

// on = 1: from now on the norm-backward and weight-gradient entry points called with `stream` queue their final reduction instead of
// launching it; 2: later calls launch their own reduction again, what is queued stays queued; 0: flush and forget the stream.
// Returns the number of reductions flushed (>= 0) or a negative error.
int sidlsg_defer_reductions(void* stream, int on) {
    DeferState& d = dst();
    std::lock_guard<std::mutex> lk(d.mu);
    hipStream_t s = (hipStream_t)stream;
    if (on == 1) { d.q[s].accepting = true; return 0; }
    auto it = d.q.find(s);
    if (it == d.q.end()) return 0;
    if (on == 2) { it->second.accepting = false; return 0; }
    const int n = flush_pairs(s, it->second.pairs) + flush_slabs(s, it->second);
    d.q.erase(it);
    return sidlsg_last_error() == SIDLSG_OK ? n : -1;
}
int sidlsg_flush_reductions(void* stream) {
    DeferState& d = dst();
    std::lock_guard<std::mutex> lk(d.mu);
    auto it = d.q.find((hipStream_t)stream);
    if (it == d.q.end() || (it->second.pairs.empty() && it->second.slabs.empty())) return 0;
    const int n = flush_pairs((hipStream_t)stream, it->second.pairs) + flush_slabs((hipStream_t)stream, it->second);
    return sidlsg_last_error() == SIDLSG_OK ? n : -1;
}
int sidlsg_pending_reductions(void* stream) {
    DeferState& d = dst();
    std::lock_guard<std::mutex> lk(d.mu);
    auto it = d.q.find((hipStream_t)stream);
    return it == d.q.end() ? -1 : (int)(it->second.pairs.size() + it->second.slabs.size());
}

}  // extern "C"
