// In-library kernel timing: per-DISPATCH start / stop timestamps of sampled launches, grouped by kernel family.
//
// bench.py must report each family's achieved rate from kernel durations that agree with `rocprofv3 --kernel-trace` of the same
// command.  HIP events recorded around a launch do not: each record is its own barrier packet, and on ~50 us kernels under three
// concurrent streams the pair measured 1.5x the kernel's own duration (round 3: roofline.frac 0.12 by events vs 0.19 from the
// profile).  hipExtLaunchKernelGGL attaches the two events to the kernel's OWN dispatch packet, so their difference is the
// begin -> end interval the profiler reports, without extra packets in the stream.
//
// Protocol (host, one device):  sidlsg_trace_enable(n) -> [run] -> device synchronise -> sidlsg_trace_read(family, out) ...
// A C-ABI entry point opens a TraceScope(family, work); every kernel it launches through SIDLSG_LAUNCH while the scope is
// sampled gets an event pair from the pool.  Sampling is per CALL (all kernels of a sampled call are timed, e.g. a split-K GEMM
// and its finish kernel), every `stride`-th call of a family.  Disabled (the default): one thread-local pointer test per launch.
#include "common.h"
#include <atomic>
#include <mutex>
#include <vector>

namespace {
// A record keeps the pool indices of ITS event pairs: two host threads with sampled scopes open at once (the forward on the main
// thread, autograd's worker, the input-prefetch thread) interleave their draws from the pool, so "first + 2 k" would hand one
// call's kernels to another family.  8 kernels per call is the deepest entry point (split conv weight gradient + reductions).
constexpr int REC_MAX = 8;
struct Rec { int family; double work, bytes; int count; int ev[REC_MAX]; };
struct State {
    std::mutex mu;
    std::atomic<bool> on{false};       // read outside the mutex on every launch
    std::atomic<unsigned> gen{0};      // bumped by every (re)start: invalidates the thread-local record indices of the old run
    std::vector<hipEvent_t> pool;      // 2 events per kernel
    size_t used = 0;
    std::vector<Rec> recs;
    long long calls[SIDLSG_TRACE_FAMILIES] = {};
    int stride[SIDLSG_TRACE_FAMILIES];
    State() { for (int& s : stride) s = 1; }
};
State& st() { static State s; return s; }
thread_local int t_cur = -1;          // index of the open sampled record of this thread ...
thread_local unsigned t_gen = 0;      // ... valid for this generation of State::recs only
}  // namespace

bool sidlsg_trace_scope_begin(int family, double work, double bytes) {
    State& s = st();
    if (!s.on.load(std::memory_order_relaxed) || family < 0 || family >= SIDLSG_TRACE_FAMILIES) return false;
    std::lock_guard<std::mutex> lk(s.mu);
    if (t_cur >= 0 && t_gen == s.gen.load()) return false;      // nested scope of this thread: the outer one owns the kernels
    t_cur = -1;
    if (!s.on.load()) return false;
    const long long c = s.calls[family]++;
    if (c % s.stride[family]) return false;
    if (s.used + 16 > s.pool.size()) return false;      // pool exhausted: stop sampling, keep counting
    s.recs.push_back({family, work, bytes, 0, {}});
    t_cur = (int)s.recs.size() - 1;
    t_gen = s.gen.load();
    return true;
}
void sidlsg_trace_scope_end() { t_cur = -1; }
bool sidlsg_trace_events(hipEvent_t* e0, hipEvent_t* e1) {
    if (t_cur < 0) return false;
    State& s = st();
    std::lock_guard<std::mutex> lk(s.mu);
    if (t_gen != s.gen.load() || (size_t)t_cur >= s.recs.size()) { t_cur = -1; return false; }      // tracing was restarted under this scope
    Rec& r = s.recs[t_cur];
    if (s.used + 2 > s.pool.size() || r.count >= REC_MAX) return false;
    *e0 = s.pool[s.used];
    *e1 = s.pool[s.used + 1];
    r.ev[r.count++] = (int)s.used;
    s.used += 2;
    return true;
}

extern "C" {

// max_kernels > 0: (re)start tracing with room for that many timed kernels; 0: stop and release the events.
int sidlsg_trace_enable(int max_kernels) {
    State& s = st();
    std::lock_guard<std::mutex> lk(s.mu);
    s.on = false;
    s.gen++;
    s.used = 0;
    s.recs.clear();
    for (long long& c : s.calls) c = 0;
    if (max_kernels <= 0) {
        for (hipEvent_t e : s.pool) (void)hipEventDestroy(e);
        s.pool.clear();
        return SIDLSG_OK;
    }
    while (s.pool.size() < (size_t)max_kernels * 2) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return SIDLSG_EINVAL;
        s.pool.push_back(e);
    }
    s.on = true;
    return SIDLSG_OK;
}

// keep collecting nothing new (the records stay readable): used around code that must not be sampled
int sidlsg_trace_pause(int paused) {
    State& s = st();
    std::lock_guard<std::mutex> lk(s.mu);
    s.on = !paused && !s.pool.empty();
    return SIDLSG_OK;
}

int sidlsg_trace_set_stride(int family, int stride) {
    if (family < 0 || family >= SIDLSG_TRACE_FAMILIES || stride < 1) return SIDLSG_EINVAL;
    State& s = st();
    std::lock_guard<std::mutex> lk(s.mu);
    s.stride[family] = stride;
    return SIDLSG_OK;
}

// out[0] = summed kernel time of the sampled calls (ms), out[1] = their summed work, out[2] = sampled calls, out[3] = all calls
// of the family since enable, out[4] = timed kernels, out[5] = summed algorithmic bytes of the sampled calls (operands read once +
// outputs written once), out[6] = the sampled calls' summed ROOFLINE time in ms: per call max(flop / peak_flops, bytes / peak_bw)
// with the peaks the caller passes in out[7] (flop/s) and out[8] (bytes/s) -- a K = 320 GEMM is HBM-bound, a K = 5760 conv is
// MFMA-bound, and each call is graded against its own bound.  Call after the device has been synchronised; out: 9 doubles.
int sidlsg_trace_read(int family, double* out) {
    if (family < 0 || family >= SIDLSG_TRACE_FAMILIES || !out) return SIDLSG_EINVAL;
    State& s = st();
    std::lock_guard<std::mutex> lk(s.mu);
    double ms = 0, work = 0, bytes = 0, bound = 0;
    const double pf = out[7] > 0 ? out[7] : 2.5e15, pb = out[8] > 0 ? out[8] : 8e12;
    long long sampled = 0, kernels = 0;
    for (const Rec& r : s.recs) {
        if (r.family != family) continue;
        bool ok = r.count > 0;
        double t = 0;
        for (int k = 0; k < r.count; k++) {
            float f = 0.f;
            if (hipEventElapsedTime(&f, s.pool[r.ev[k]], s.pool[r.ev[k] + 1]) != hipSuccess) { ok = false; break; }
            t += f;
        }
        if (!ok) continue;
        ms += t; work += r.work; bytes += r.bytes; sampled++; kernels += r.count;
        bound += 1e3 * ((r.work / pf > r.bytes / pb) ? r.work / pf : r.bytes / pb);
    }
    out[0] = ms; out[1] = work; out[2] = (double)sampled; out[3] = (double)s.calls[family]; out[4] = (double)kernels;
    out[5] = bytes; out[6] = bound;
    return SIDLSG_OK;
}

}  // extern "C"
