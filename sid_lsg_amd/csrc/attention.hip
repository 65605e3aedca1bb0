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
// Head dims that are not multiples of 32 are zero-padded to the next K=32 slice in the QK^T / dO V^T contractions
// (d=40 -> 64, d=80 -> 96; see Frag) and to the next multiple of 16 in the output rows (d=40 -> 48); nothing is
// padded in HBM.  At d=40 the loop is VALU(softmax)-bound, so the file is compiled with MFMA results in VGPRs
// (-amdgpu-mfma-vgpr-form) and -fno-honor-nans, the forward gets its softmax denominator from a ones column in V's
// padding, and the ragged-tile masking is peeled out of the main loop (391 -> 148 VALU instructions per 64-key tile).
// K/V (or Q/dO) tiles are prefetched global->registers one tile ahead of the MFMAs and committed to the other LDS
// buffer after the compute phase (one barrier per tile).  Loads are branch-free buffer loads (rows past the sequence
// end read zeros).
//
// Backward = two kernels without atomics:
//   attn_q<MODE 1> : same loop as forward (per query block, over key tiles): dQ^T = K^T dS^T; computes delta =
//                    rowsum(dO*O) from its fragments in the prologue and publishes it for the second kernel
//   attn_dkdv      : per key block, over query tiles, in the non-transposed orientation S[query][key] so that P / dS
//                    are B operands of the contractions over queries.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

struct AttnParams {
    const bf16 *Q, *K, *V, *dO;
    bf16 *O, *dQ, *dK, *dV;
    float* LSE;          // [B][H][Nq], log2 domain: m + log2(l)
    const float* delta;  // [B][H][Nq] = rowsum(dO * O): written by the dQ kernel (delta_out), read by the dK/dV kernel
    float* delta_out;
    int B, H, Nq, Nk, D;
    int ldq, ldk, ldv, ldo;        // token strides (elements)
    long long bsq, bsk, bsv, bso;  // batch strides (elements)
    float scale2;                  // d^-0.5 * log2(e)
    float scale;                   // d^-0.5
    int xcd;                       // 1: blocks renumbered so that all blocks of one (batch, head) run on ONE XCD (attn_block_coords)
    int rot;                       // 1: every block starts its walk over the streamed tiles (keys: forward / dQ; queries: dK/dV) at its own
                                   //    offset and wraps around, so the co-resident blocks of a head do not read the same tile at the same time
};

// Hardware workgroup L (x fastest) runs on XCD L % 8, so the query / key blocks of one (batch, head) -- neighbours in x -- were spread
// over all 8 XCDs and each XCD's L2 fetched that head's whole K, V (forward, dQ pass) or Q, dO (dK/dV pass) for itself: the HBM-side traffic
// of the N = 4096 self-attention was 3.7x (forward) / 4.9x (backward) the algorithmic bytes (profiles/r04_traffic_attn.json).  With the
// XCD-contiguous renumbering of the GEMM kernels an XCD works through whole heads: the streamed operand is fetched once per head.
DEVFN void attn_block_coords(const AttnParams& p, int& bx, int& h, int& b) {
    bx = blockIdx.x; h = blockIdx.y; b = blockIdx.z;
    if (!p.xcd) return;
    const unsigned nx = gridDim.x, ny = gridDim.y, total = nx * ny * gridDim.z;
    unsigned L = blockIdx.x + nx * (blockIdx.y + ny * blockIdx.z);
    const unsigned q = total >> 3, r = total & 7, xcd = L & 7, idx = L >> 3;
    L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    bx = (int)(L % nx);
    const unsigned t = L / nx;
    h = (int)(t % ny); b = (int)(t / ny);
}

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
constexpr unsigned A_OOB = 0x80000000u;

DEVFN __amdgpu_buffer_rsrc_t mk_rsrc(const bf16* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p), 0, 0x7FFFFFFF, 0x00020000);
}
// descriptor over rows 0..nrows-1 of a [rows][ld] view whose rows are D elements wide: a 16-byte chunk of a row >= nrows
// starts past num_records and reads zeros
DEVFN __amdgpu_buffer_rsrc_t mk_rsrc_rows(const bf16* p, int nrows, int ld, int D) {
    const long long bytes = nrows > 0 ? ((long long)(nrows - 1) * ld + D) * 2 : 0;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p), 0, (int)bytes, 0x00020000);
}
DEVFN bf16x8 bld8(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}

// One MFMA operand row of a head: ceil(DP/32) K=32 slices.  Head dims that are not multiples of 32 (d=40 -> DP=48,
// d=80) zero-pad the last slice instead of using a trailing 16x16x16 MFMA: the K=16 instruction costs the same 4
// passes as the K=32 one on gfx950, and chaining it behind a K=32 result needed a separate accumulator + VALU adds
// (a 16x16x32 result forwarded into the SrcC of a 16x16x16 MFMA gave wrong sums, see git history / DESIGN.md).
// Every contraction pairs one operand read from global memory (exact zeros past D) with one read from an LDS tile
// whose row is over-read by up to N32*32 - DP columns: pad columns are zeroed once per block and the over-read of
// the next row meets zeros on the other side, so only finiteness of the LDS contents matters.
template <int DP>
struct Frag {
    static constexpr int N32 = (DP + 31) / 32;
    bf16x8 w[N32];
};

template <int DP>
DEVFN void frag_from_lds(Frag<DP>& f, const bf16* row, int lg) {
#pragma unroll
    for (int s = 0; s < Frag<DP>::N32; s++) f.w[s] = *reinterpret_cast<const bf16x8*>(row + s * 32 + lg * 8);
}
template <int DP>
DEVFN void frag_from_global(Frag<DP>& f, const bf16* row, int lg, int D, bool ok) {
#pragma unroll
    for (int s = 0; s < Frag<DP>::N32; s++) {
        const int d0 = s * 32 + lg * 8;
        f.w[s] = (ok && d0 + 8 <= D) ? ld8(row + d0) : zero8();
    }
}
template <int DP>
DEVFN f32x4 mma_d(f32x4 acc, const Frag<DP>& a, const Frag<DP>& b) {
#pragma unroll
    for (int s = 0; s < Frag<DP>::N32; s++) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.w[s], b.w[s], acc, 0, 0, 0);
    return acc;
}
// LDS tile of ROWS rows with a row stride of LD = the smallest odd multiple of 16 elements >= DP (32 B x odd): with that
// stride both the ds_read_b128 fragment reads (16-lane groups of the b128 access) and the ds_read_b64_tr_b16 transpose reads
// (32-lane halves) touch every bank slot exactly once -- brute-forced over the access patterns of this file; the former
// DP + 8 stride was 2-way conflicted on both (SQ_LDS_BANK_CONFLICT = 45 % of SQ_LDS_IDX_ACTIVE, profiles/r02_attn_pmc.json).
// Columns DP..LD (when LD > DP) are never read.  The over-read of the K=32 slices (DP = 16, 48, 80: LD == DP) runs into
// the NEXT row's first 16 columns (finite data, multiplied by the global operand's exact zeros) and, for the last row, into
// 16 elements of slack that are zeroed once per block here.
template <int DP>
constexpr int tile_ld() { return ((DP + 15) / 16) % 2 ? (DP + 15) / 16 * 16 : (DP + 15) / 16 * 16 + 16; }
template <int DP, int ROWS>
DEVFN void lds_tile_init(bf16* tile) {
    constexpr int LD = tile_ld<DP>();
    if (threadIdx.x < 2) st8(tile + ROWS * LD + threadIdx.x * 8, zero8());
}
template <int DP, int ROWS>
constexpr int lds_tile_elems() { return ROWS * tile_ld<DP>() + 16; }

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

// register-staged tile: ROWS x DP (zero padded past D / past nrows), 256 threads.  Addressing: one 32-bit byte offset per
// 16-byte chunk, computed ONCE per block (init); the tile position is the wave-uniform soffset of the buffer load, and rows
// past the sequence end are cut off by the descriptor's num_records (mk_rsrc_rows) -- no per-tile address arithmetic or
// bounds tests in the VALU-bound loop (they were ~35 VALU instructions per key tile).
template <int DP, int ROWS>
struct TileRegs {
    static constexpr int C8 = DP / 8;
    static constexpr int TCH = ROWS * C8;
    static constexpr int PER = (TCH + 255) / 256;
    bf16x8 v[PER];
    unsigned voff[PER];
    DEVFN void init(int ld, int D) {
#pragma unroll
        for (int j = 0; j < PER; j++) {
            const int idx = threadIdx.x + 256 * j;
            const int r = idx / C8, c = (idx - r * C8) * 8;
            voff[j] = (idx < TCH && c < D) ? (unsigned)((r * ld + c) * 2) : A_OOB;
        }
    }
    // row0 * ld * 2 must fit 31 bits (checked by the launcher)
    DEVFN void load(__amdgpu_buffer_rsrc_t rs, int ld, int row0) {
        const unsigned soff = (unsigned)row0 * (unsigned)ld * 2u;
#pragma unroll
        for (int j = 0; j < PER; j++) v[j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, voff[j], soff, 0));
    }
    // ones_col >= 0: element [r][ones_col] is written as 1.0 (a zero-pad column of the tile): the MFMA that contracts
    // the tile against the probabilities then yields their row sum in that output row for free.
    DEVFN void store(bf16* dst, int LD, int ones_col = -1) const {
#pragma unroll
        for (int j = 0; j < PER; j++) {
            const int idx = threadIdx.x + 256 * j;
            const int r = idx / C8, c = (idx - r * C8) * 8;
            bf16x8 x = v[j];
            if (c == ones_col) x[0] = f2bf(1.0f);
            if (idx < TCH) st8(dst + r * LD + c, x);
        }
    }
};

// The same tile staged by LDS-DMA (buffer_load_dwordx4 ... lds): no staging registers, no ds_write pass.  A wave's DMA
// instruction moves 64 consecutive 16-byte chunks of the (row-major, stride LD) LDS image; chunk q = (row q / (LD/8), column
// chunk q % (LD/8)).  Lanes whose column chunk lies past D are masked off: the pad chunks are written ONCE per block
// (tile_pad_init: zeros, or the forward's ones column) and survive every refill.  Rows past the sequence end read zeros
// through the descriptor's range check.  `wait()` = this wave's DMAs have landed (the caller's barrier publishes them).
#ifndef SIDLSG_ATTN_DMA
#define SIDLSG_ATTN_DMA 1      // A/B switch (0: register staging everywhere)
#endif
typedef __attribute__((address_space(3))) void* at_lptr_t;
template <int DP, int ROWS>
struct TileDma {
    static constexpr int LD = tile_ld<DP>();
    static constexpr int NCH = LD / 8;
    static constexpr int TCH = ROWS * NCH;
    static constexpr int PER = (TCH + 255) / 256;
    static_assert(TCH % 64 == 0, "whole waves");
    unsigned voff[PER];
    bool on[PER];
    int wave;
    DEVFN void init(int ld, int D) {
        wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
        for (int j = 0; j < PER; j++) {
            const int idx = threadIdx.x + 256 * j;
            const int r = idx / NCH, c = (idx - r * NCH) * 8;
            on[j] = idx < TCH && c < D;
            voff[j] = (unsigned)((r * ld + c) * 2);
        }
    }
    DEVFN void load(__amdgpu_buffer_rsrc_t rs, int ld, int row0, bf16* dst) {
        const unsigned soff = (unsigned)row0 * (unsigned)ld * 2u;
#pragma unroll
        for (int j = 0; j < PER; j++) {
            if (256 * j + 64 * wave >= TCH) continue;                   // wave-uniform
            if (on[j]) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (at_lptr_t)(dst + (256 * j + 64 * wave) * 8), 16, voff[j], soff, 0, 0);
        }
    }
    DEVFN void wait() const { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    // (ISA note: the compiler's waitcnt pass puts its own s_waitcnt vmcnt(0) in front of the first LDS read that may alias an
    // LDS-DMA in flight -- for ds_read_b64_tr_b16 always, for plain reads of the same __shared__ variable too -- so the
    // prefetch of the next tile is waited for before the current tile's first fragment reads.  Issuing the DMA as inline
    // assembly (untracked) and forcing the pre-loop global loads to complete in the compiler's bookkeeping removed those
    // waits from the loop, and measured +-1 % on forward and backward at d = 40 / 64: with 3-4 waves per SIMD the other
    // waves cover that wait.  Not kept.)
};
// pad chunks (columns D .. LD) of a tile buffer: zeros, with 1.0 at column `ones_col` when >= 0 (see TileRegs::store)
template <int DP, int ROWS>
DEVFN void tile_pad_init(bf16* tile, int D, int ones_col) {
    constexpr int LD = tile_ld<DP>();
    const int npad = (LD - D) / 8;
    for (int i = threadIdx.x; i < ROWS * npad; i += 256) {
        const int r = i / npad, c = D + (i - r * npad) * 8;
        bf16x8 x = zero8();
        if (c == ones_col) x[0] = f2bf(1.0f);
        st8(tile + r * LD + c, x);
    }
}

// One interface over both staging schemes: load(..., dst) starts fetching a tile that will live in LDS buffer `dst`,
// commit(dst, ...) completes it (register staging: the ds_write pass; DMA: wait for this wave's transfers).
template <int DP, int ROWS, bool DMA>
struct Tile;
template <int DP, int ROWS>
struct Tile<DP, ROWS, false> {
    TileRegs<DP, ROWS> t;
    DEVFN void init(int ld, int D, bf16* b0, bf16* b1, int ones_col) { (void)b0; (void)b1; (void)ones_col; t.init(ld, D); }
    DEVFN void load(__amdgpu_buffer_rsrc_t rs, int ld, int row0, bf16* dst) { (void)dst; t.load(rs, ld, row0); }
    DEVFN void commit(bf16* dst, int LD, int ones_col = -1) { t.store(dst, LD, ones_col); }
};
template <int DP, int ROWS>
struct Tile<DP, ROWS, true> {
    TileDma<DP, ROWS> t;
    DEVFN void init(int ld, int D, bf16* b0, bf16* b1, int ones_col) {
        t.init(ld, D);
        tile_pad_init<DP, ROWS>(b0, D, ones_col);
        tile_pad_init<DP, ROWS>(b1, D, ones_col);
    }
    DEVFN void load(__amdgpu_buffer_rsrc_t rs, int ld, int row0, bf16* dst) { t.load(rs, ld, row0, dst); }
    DEVFN void commit(bf16* dst, int LD, int ones_col = -1) { (void)dst; (void)LD; (void)ones_col; t.wait(); }
};
// Measured (MI355X, in-session A/B, N = 4096, B = 16): DMA staging speeds the backward kernels up at every head size (d = 40:
// 1685 -> 1578 us, d = 64: 1282 -> 1244 us, d = 80 at N = 1024: 219 -> 199 us) and the d = 40 forward (483 -> 466 us), but slows
// the d = 64 / 80 forward down (390 -> 435 us, 65 -> 70 us: padded rows, masked lanes) -> forward: DP == 48 only.
template <int DP, int MODE>
constexpr bool attn_q_dma() { return SIDLSG_ATTN_DMA && (MODE == 1 || DP == 48); }

// max of three: pattern-matched to v_max3_f32.  attention.hip is compiled with -fno-honor-nans so that llvm.maxnum
// does not put a canonicalising v_max x,x on every MFMA output.  (Do NOT use inline asm on MFMA results: the
// compiler cannot see the operands of an asm statement when it inserts the MFMA->VALU wait states, and short
// MFMA chains (d <= 48) then read stale registers.)
DEVFN float fmax3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
DEVFN float fmax2(float a, float b) { return fmaxf(a, b); }

constexpr int AT_KT = 64;   // keys per LDS tile

// s_setprio around the MFMA clusters (A/B knob, bit mask): 1 = QK^T cluster of attn_q_kernel, 2 = its second contraction (P.V / dS.K,
// and the dP cluster of the dQ pass), 4 = the dV / dK cluster of attn_dkdv_kernel, 8 = its S / dP cluster.  With several
// independent waves per SIMD in different phases the arbiter then prefers the wave that feeds the matrix pipe.
#ifndef SIDLSG_ATTN_PRIO
#define SIDLSG_ATTN_PRIO 0
#endif
// (s_setprio alone does not stay where it is written: MFMAs are register-only and the scheduler moves them across it -- seen in the
// ISA: both flips of the QK^T cluster ended up in front of its first MFMA.  The two sched_barriers let VALU / VMEM / LDS
// instructions cross (mask 0x3F2) and pin only the MFMAs and scalar instructions relative to the flip.)
#define ATTN_PRIO(bit, v) do { if constexpr ((SIDLSG_ATTN_PRIO) & (bit)) { __builtin_amdgcn_sched_barrier(0x3F2); __builtin_amdgcn_s_setprio(v); __builtin_amdgcn_sched_barrier(0x3F2); } } while (0)

// MODE 0: forward (O, LSE).  MODE 1: dQ.   QT = 16-query tiles per wave (block = 4 waves * QT * 16 queries)
// ONES (forward, D == DP - 8 only): V's first pad column holds 1.0, so O^T row D accumulates the softmax
// denominator inside the P.V MFMAs (and is rescaled with O); the 16 VALU adds per query tile disappear.
// K/V tiles are double buffered in LDS: one barrier per key tile.
// (the d = 40 forward is VALU-bound and lives on 4 resident waves per SIMD: cap its registers at 128; the d = 40 dQ kernel
// declared for 2 blocks per SIMD, i.e. up to 256 registers, ran the backward 3 % faster than uncapped (190) or capped at 168)
#ifndef SIDLSG_DQ_OCC
#define SIDLSG_DQ_OCC 2
#endif
// (A software-pipelined variant of the PS forward -- QK^T of tile j+1 beside the exponentials of tile j in one basic block --
// was built and measured neutral, commit 2bea052: the MFMA + VALU core alone runs the
// d = 40 forward in 333 us, LDS fragment reads add ~100 us and tile staging + barrier ~110 us: the loop is bound by those
// stalls, not by MFMA/VALU overlap.)
// Ablation builds of the final PS forward (-DSIDLSG_EXP_ATTN_NOSTAGE / _NOLDS / _NOEXP; N = 4096, d = 40, B = 16, MI355X, after a
// long warm-up): all 457 us; no exponentials 398; no K/V tile staging + barrier 376; fragments from registers instead of LDS 311;
// neither staging nor LDS reads 233; none of the three 197 (= the MFMA + remaining VALU floor).  64 instead of 32 queries per
// wave (half the fragment reads per MFMA, 2 instead of 4 waves per SIMD, 256 VGPRs) measured 465 vs 479 us in one session: the
// cost of the fragment reads is their latency in front of the MFMAs, which occupancy hides and the larger tile does not.
// PS ("pre-scaled"): Q arrives already multiplied by D^-0.5 * log2(e) (the caller folds the factor into the q rows of the
// projection weight, see sidlsg_attn_fwd_ps), so the QK^T MFMA yields the scores in log2 units and its C operand is seeded
// with -m (forward: the running maximum; dQ pass: -LSE): the accumulators come out as s - m and go straight into
// v_exp_f32 -- the 32 v_fma (scale, subtract) per 64-key x 32-query wave tile disappear from the VALU-bound loop.  The dQ
// pass seeds the dP = V dO^T MFMA with -delta the same way.  The first key tile of the forward is peeled (`first`): its
// accumulators start from 0 and the running maximum is SET from it (it may be negative).
template <int DP, int QT, int MODE, bool ONES, bool PS>
__global__ __launch_bounds__(256, (DP <= 48 && MODE == 0) ? 4 : ((DP <= 48 && MODE == 1) ? SIDLSG_DQ_OCC : 1)) void attn_q_kernel(AttnParams p) {
    static_assert(!ONES || (MODE == 0 && DP % 16 == 0), "ones column: forward only");
    constexpr int LD = tile_ld<DP>();
    constexpr int DT = DP / 16;
    constexpr int N32 = Frag<DP>::N32;
    constexpr int TE = lds_tile_elems<DP, AT_KT>();
    __shared__ __attribute__((aligned(16))) bf16 Ks[2][TE];
    __shared__ __attribute__((aligned(16))) bf16 Vs[2][TE];
    int bx_, h_, b_;
    attn_block_coords(p, bx_, h_, b_);
    const int b = b_, h = h_;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    const int q0 = (bx_ * 4 + wave) * (QT * 16);
    const bf16* Qb = p.Q + b * p.bsq + (long long)h * p.D;
    const __amdgpu_buffer_rsrc_t rk = mk_rsrc_rows(p.K + b * p.bsk + (long long)h * p.D, p.Nk, p.ldk, p.D);
    const __amdgpu_buffer_rsrc_t rv = mk_rsrc_rows(p.V + b * p.bsv + (long long)h * p.D, p.Nk, p.ldv, p.D);
    const int ones_col = ONES ? p.D : -1;
#pragma unroll
    for (int i = 0; i < 2; i++) { lds_tile_init<DP, AT_KT>(Ks[i]); lds_tile_init<DP, AT_KT>(Vs[i]); }

    Frag<DP> fq[QT], fdo[MODE == 1 ? QT : 1];
    float lse[QT], dl[QT];
#pragma unroll
    for (int qt = 0; qt < QT; qt++) {
        const int q = q0 + qt * 16 + li;
        const bool ok = q < p.Nq;
        frag_from_global<DP>(fq[qt], Qb + (long long)(ok ? q : 0) * p.ldq, lg, p.D, ok);
        lse[qt] = dl[qt] = 0.f;
        if (MODE == 1) {
            const bf16* dOb = p.dO + b * p.bso + (long long)h * p.D;
            frag_from_global<DP>(fdo[qt], dOb + (long long)(ok ? q : 0) * p.ldo, lg, p.D, ok);
            lse[qt] = ok ? p.LSE[((long long)b * p.H + h) * p.Nq + q] : 0.f;
            // delta = rowsum(dO * O), computed here from the fragments (lane holds 8 * N32 of the row's d values; the 4
            // lane groups of a row are summed with two shuffles) and published for the dK/dV kernel: no separate pass
            Frag<DP> fo;
            const bf16* Ob = p.O + b * p.bso + (long long)h * p.D;
            frag_from_global<DP>(fo, Ob + (long long)(ok ? q : 0) * p.ldo, lg, p.D, ok);
            float dsum = 0.f;
#pragma unroll
            for (int s2 = 0; s2 < N32; s2++)
#pragma unroll
                for (int e = 0; e < 8; e++) dsum += bf2f(fdo[qt].w[s2][e]) * bf2f(fo.w[s2][e]);
            dsum += __shfl_xor(dsum, 16, 64);
            dsum += __shfl_xor(dsum, 32, 64);
            dl[qt] = dsum;
            if (ok && lg == 0) p.delta_out[((long long)b * p.H + h) * p.Nq + q] = dsum;
        }
    }
    f32x4 o[DT][QT];
#pragma unroll
    for (int i = 0; i < DT; i++)
#pragma unroll
        for (int qt = 0; qt < QT; qt++) o[i][qt] = (f32x4){0, 0, 0, 0};
    float m[QT], l[QT];
    f32x4 seed_s[QT], seed_dp[QT];      // PS: C operands of the first QK^T / dP MFMA of a chain
#pragma unroll
    for (int qt = 0; qt < QT; qt++) {
        m[qt] = -INFINITY; l[qt] = 0.f;
        const float a = MODE == 1 ? -lse[qt] : 0.f, c = -dl[qt];
        seed_s[qt] = (f32x4){a, a, a, a};
        seed_dp[qt] = (f32x4){c, c, c, c};
    }

    Tile<DP, AT_KT, attn_q_dma<DP, MODE>()> tk, tv;
    tk.init(p.ldk, p.D, Ks[0], Ks[1], -1);
    tv.init(p.ldv, p.D, Vs[0], Vs[1], ones_col);
    // Rotated walk (p.rot; whole tiles only): logical tile position k0 -> keys [wrap(k0 + kbase), +AT_KT).  Softmax statistics and the
    // sums over keys do not depend on the order of the tiles; with the XCD-contiguous numbering the ~32 blocks of a head are resident
    // together and would otherwise request the same K / V tile within the same microsecond, all the way through the kernel.
    const bool rotate = p.rot && p.Nk % AT_KT == 0 && p.Nk >= 2 * AT_KT;
    const int kbase = rotate ? (int)(((long long)bx_ * (p.Nk / AT_KT)) / gridDim.x) * AT_KT : 0;
    auto wrap = [&](int k) { return k >= p.Nk && rotate ? k - p.Nk : k; };
    tk.load(rk, p.ldk, kbase, Ks[0]);
    tv.load(rv, p.ldv, kbase, Vs[0]);
    tk.commit(Ks[0], LD);
    tv.commit(Vs[0], LD, ones_col);
    __syncthreads();
    int buf = 0;
    // The tile body is instantiated several times: full tiles (no masking code at all; the compiler otherwise if-converts
    // the ragged-tile test into ~90 predicated VALU ops per tile in a VALU-bound loop) and the ragged last tile.
    // Forward: `has_next` is a compile-time flag like `ragged` -- a run-time `if (more)` around the prefetch / commit makes
    // the waitcnt pass merge two paths (same finding as in gemm_v3_kernel; A/B on one MI355X: forward +1.4 % at d=40, +3 %
    // at d=64/80).  The dQ pass measured 1-8 % SLOWER that way (register allocation), so it keeps the run-time test.
    auto tile = [&](const int k0, auto ragged, auto has_next, auto first) {
        const bool more = MODE == 0 ? decltype(has_next)::value : (k0 + AT_KT < p.Nk);
        constexpr bool FIRST = decltype(first)::value;
        // next tile -> registers, or by DMA straight into the other buffer (last read before the previous barrier)
#ifndef SIDLSG_EXP_ATTN_NOSTAGE
        if (more) { const int kn = wrap(k0 + AT_KT + kbase); tk.load(rk, p.ldk, kn, Ks[buf ^ 1]); tv.load(rv, p.ldv, kn, Vs[buf ^ 1]); }
#endif
        const bf16* Kt = Ks[buf];
        const bf16* Vt = Vs[buf];
        f32x4 s[4][QT];
        ATTN_PRIO(1, 1);
#pragma unroll
        for (int kt = 0; kt < 4; kt++) {
            Frag<DP> fk;
#ifdef SIDLSG_EXP_ATTN_NOLDS
            fk = fq[kt % QT];
#else
            frag_from_lds<DP>(fk, Kt + (kt * 16 + li) * LD, lg);
#endif
            // slice-major order: the two MFMAs of one accumulator's chain are QT instructions apart (back to back they
            // serialise on the 16x16x32 result latency)
#pragma unroll
            for (int sl = 0; sl < N32; sl++)
#pragma unroll
                for (int qt = 0; qt < QT; qt++) {
                    const f32x4 c0 = (PS && !(MODE == 0 && FIRST)) ? seed_s[qt] : (f32x4){0, 0, 0, 0};
                    s[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk.w[sl], fq[qt].w[sl], sl == 0 ? c0 : s[kt][qt], 0, 0, 0);
                }
        }
        ATTN_PRIO(1, 0);
        // scores -> probabilities (keys of this lane: k0 + kt*16 + lg*4 + r).  The softmax is the VALU-bound part
        // at d=40: raw v_exp_f32, scale folded into one FMA (or into Q: PS), masking only on the ragged last tile, and the
        // running-max rescale of O deferred until the max grows by > 2^8 (LSE stays exact: m + log2(l)).
        if constexpr (decltype(ragged)::value) {
#pragma unroll
            for (int kt = 0; kt < 4; kt++)
#pragma unroll
                for (int r = 0; r < 4; r++)
                    if (k0 + kt * 16 + lg * 4 + r >= p.Nk) {
#pragma unroll
                        for (int qt = 0; qt < QT; qt++) s[kt][qt][r] = -INFINITY;
                    }
        }
#pragma unroll
        for (int qt = 0; qt < QT; qt++) {
            if (MODE == 0) {
                float mx = fmax3(s[0][qt][0], s[0][qt][1], s[0][qt][2]);
                mx = fmax3(mx, s[0][qt][3], s[1][qt][0]);
                mx = fmax3(mx, s[1][qt][1], s[1][qt][2]);
                mx = fmax3(mx, s[1][qt][3], s[2][qt][0]);
                mx = fmax3(mx, s[2][qt][1], s[2][qt][2]);
                mx = fmax3(mx, s[2][qt][3], s[3][qt][0]);
                mx = fmax3(mx, s[3][qt][1], s[3][qt][2]);
                mx = fmax2(mx, s[3][qt][3]);
                if constexpr (PS) {
                    // the accumulators hold s - m (first tile: s); mx is this lane group's maximum of them
                    if (FIRST || __any(mx > 8.0f)) {
                        mx = fmax2(mx, __shfl_xor(mx, 16, 64));
                        mx = fmax2(mx, __shfl_xor(mx, 32, 64));
                        // first tile: m := row maximum (any sign).  later: m grows by delta = max(0, row maximum of s - m)
                        const float delta = FIRST ? mx : fmax2(mx, 0.f);
                        if (!FIRST) {
                            const float alpha = __builtin_amdgcn_exp2f(-delta);
                            l[qt] *= alpha;
#pragma unroll
                            for (int i = 0; i < DT; i++) o[i][qt] *= alpha;
                        }
                        m[qt] = FIRST ? delta : m[qt] + delta;
                        const float nm = -m[qt];
                        seed_s[qt] = (f32x4){nm, nm, nm, nm};
#pragma unroll
                        for (int kt = 0; kt < 4; kt++)
#pragma unroll
                            for (int r = 0; r < 4; r++) s[kt][qt][r] -= delta;
                    }
                    float sum = 0.f;
#pragma unroll
                    for (int kt = 0; kt < 4; kt++)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
#ifdef SIDLSG_EXP_ATTN_NOEXP
                            const float e = s[kt][qt][r];
#else
                            const float e = __builtin_amdgcn_exp2f(s[kt][qt][r]);
#endif
                            s[kt][qt][r] = e;
                            if (!ONES) sum += e;
                        }
                    if (!ONES) l[qt] += sum;
                } else {
                    mx *= p.scale2;                                    // scale2 > 0: max commutes with the scaling
                    // wave-uniform test on the lanes' LOCAL maxima (this lane group's 16 keys): the common path needs no
                    // cross-lane reduction (2 ds_bpermute round trips per query tile); only when the running max must move is
                    // the row maximum reduced over the 4 lane groups and everything held at the old max rescaled
                    if (__any(mx > m[qt] + 8.0f)) {
                        mx = fmax2(mx, __shfl_xor(mx, 16, 64));
                        mx = fmax2(mx, __shfl_xor(mx, 32, 64));
                        const float mn = fmax2(m[qt], mx);
                        const float alpha = __builtin_amdgcn_exp2f(m[qt] - mn);   // m = -inf on the first tile -> 0
                        l[qt] *= alpha;
                        m[qt] = mn;
#pragma unroll
                        for (int i = 0; i < DT; i++) o[i][qt] *= alpha;
                    }
                    const float nm = -m[qt];
                    float sum = 0.f;
#pragma unroll
                    for (int kt = 0; kt < 4; kt++)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const float e = __builtin_amdgcn_exp2f(fmaf(s[kt][qt][r], p.scale2, nm));
                            s[kt][qt][r] = e;
                            if (!ONES) sum += e;
                        }
                    if (!ONES) l[qt] += sum;
                }
            } else {
                const float nl = -lse[qt];
#pragma unroll
                for (int kt = 0; kt < 4; kt++)
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        s[kt][qt][r] = __builtin_amdgcn_exp2f(PS ? s[kt][qt][r] : fmaf(s[kt][qt][r], p.scale2, nl));
            }
        }
        if (MODE == 1) {
            // dP^T = V dO^T ; dS^T = P^T * (dP^T - delta)   (overwrites s; the constant factor is applied in the epilogue)
            ATTN_PRIO(2, 1);
#pragma unroll
            for (int kt = 0; kt < 4; kt++) {
                Frag<DP> fv;
                frag_from_lds<DP>(fv, Vt + (kt * 16 + li) * LD, lg);
                f32x4 dp[QT];
#pragma unroll
                for (int sl = 0; sl < N32; sl++)
#pragma unroll
                    for (int qt = 0; qt < QT; qt++)
                        dp[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv.w[sl], fdo[qt].w[sl],
                                                                        sl == 0 ? (PS ? seed_dp[qt] : (f32x4){0, 0, 0, 0}) : dp[qt], 0, 0, 0);
#pragma unroll
                for (int qt = 0; qt < QT; qt++)
#pragma unroll
                    for (int r = 0; r < 4; r++) s[kt][qt][r] *= PS ? dp[qt][r] : (dp[qt][r] - dl[qt]);
            }
        }
        // second contraction over keys: forward uses V, dQ uses K
        const bf16* T2 = MODE == 0 ? Vt : Kt;
        ATTN_PRIO(2, 1);
#pragma unroll
        for (int kb = 0; kb < 2; kb++) {
            bf16x8 pb[QT];
#pragma unroll
            for (int qt = 0; qt < QT; qt++) pb[qt] = pack_p(s[2 * kb][qt], s[2 * kb + 1][qt]);
#pragma unroll
            for (int dt = 0; dt < DT; dt++) {
#ifdef SIDLSG_EXP_ATTN_NOLDS
                const bf16x8 fa = fq[dt % QT].w[kb % N32];
#else
                const bf16x8 fa = tr_frag32(T2, LD, kb * 32, dt * 16, li, lg);
#endif
#pragma unroll
                for (int qt = 0; qt < QT; qt++) o[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, pb[qt], o[dt][qt], 0, 0, 0);
            }
        }
        ATTN_PRIO(2, 0);
#ifndef SIDLSG_EXP_ATTN_NOSTAGE
        if (more) {                      // the other buffer was last read before the previous barrier
            tk.commit(Ks[buf ^ 1], LD);
            tv.commit(Vs[buf ^ 1], LD, ones_col);
            __syncthreads();
            buf ^= 1;
        }
#endif
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    int k0 = 0;
    if constexpr (PS && MODE == 0) {     // peeled first tile (sets the running maximum)
        if (p.Nk >= 2 * AT_KT) tile(0, F_{}, T_{}, T_{});
        else if (p.Nk > AT_KT) tile(0, F_{}, T_{}, T_{});
        else if (p.Nk == AT_KT) tile(0, F_{}, F_{}, T_{});
        else tile(0, T_{}, F_{}, T_{});
        k0 = AT_KT;
    }
    for (; k0 + 2 * AT_KT <= p.Nk; k0 += AT_KT) tile(k0, F_{}, T_{}, F_{});       // full tile, a full tile follows
    if (k0 + AT_KT <= p.Nk) {                                                   // last full tile
        if (k0 + AT_KT < p.Nk) tile(k0, F_{}, T_{}, F_{}); else tile(k0, F_{}, F_{}, F_{});
        k0 += AT_KT;
    }
    if (k0 < p.Nk) tile(k0, T_{}, F_{}, F_{});                                   // ragged tail
    // epilogue: lane holds, for query li of tile qt, d = dt*16 + lg*4 + r
#pragma unroll
    for (int qt = 0; qt < QT; qt++) {
        const int q = q0 + qt * 16 + li;
        // dQ = d^-1/2 * sum_k dS K: the gradient with respect to the UNSCALED queries, also when Q arrived pre-scaled
        float inv = MODE == 1 ? p.scale : 1.f;
        if (MODE == 0) {
            float lt;
            if (ONES) {
                lt = __shfl(o[DT - 1][qt][0], li + 32, 64);        // row D = (DT-1)*16 + 8: lane group 2, r = 0
            } else {
                lt = l[qt];
                lt += __shfl_xor(lt, 16, 64);
                lt += __shfl_xor(lt, 32, 64);
            }
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

// dK, dV: block = 4 waves * KT * 16 keys (wave owns KT key tiles), loop over 64-query tiles.
constexpr int AK_QT = 64;
// PS: Q pre-scaled (see attn_q_kernel): lse_s / dl_s hold -LSE / -delta and seed the S and dP accumulators, so the
// probability is exp2(acc) and dS = P * acc' with no further arithmetic.
#ifndef SIDLSG_DKDV_SUB
#define SIDLSG_DKDV_SUB 1      // 64-query sub-tiles per LDS stage (one barrier per stage); A/B knob
#endif
template <int DP, int KT, bool PS>
__global__ __launch_bounds__(256) void attn_dkdv_kernel(AttnParams p) {
    constexpr int LD = tile_ld<DP>();
    constexpr int DT = DP / 16;
    constexpr int SUB = DP <= 96 ? SIDLSG_DKDV_SUB : 1, ST = SUB * AK_QT;      // (the 160-wide heads would not fit the LDS with two sub-tiles)
    // Q / dO tiles double-buffered (2 x 2 x 7 KiB at d = 40): the next tile is written while other waves still read the
    // current one -> ONE barrier per 64-query stage
    constexpr int TE = lds_tile_elems<DP, ST>();
    __shared__ __attribute__((aligned(16))) bf16 Qs2[2][TE];
    __shared__ __attribute__((aligned(16))) bf16 dOs2[2][TE];
    __shared__ __attribute__((aligned(16))) float lse_s[2][ST], dl_s[2][ST];
#pragma unroll
    for (int i = 0; i < 2; i++) { lds_tile_init<DP, ST>(Qs2[i]); lds_tile_init<DP, ST>(dOs2[i]); }
    int bx_, h_, b_;
    attn_block_coords(p, bx_, h_, b_);
    const int b = b_, h = h_;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    const int key0 = (bx_ * 4 + wave) * (KT * 16);
    const __amdgpu_buffer_rsrc_t rq = mk_rsrc_rows(p.Q + b * p.bsq + (long long)h * p.D, p.Nq, p.ldq, p.D);
    const __amdgpu_buffer_rsrc_t rdo = mk_rsrc_rows(p.dO + b * p.bso + (long long)h * p.D, p.Nq, p.ldo, p.D);
    const float* LSEb = p.LSE + ((long long)b * p.H + h) * p.Nq;
    const float* DLb = p.delta + ((long long)b * p.H + h) * p.Nq;
    Frag<DP> fk[KT], fv[KT];  // B operands: (k = d, col = key)
    bool kok[KT];
#pragma unroll
    for (int kt = 0; kt < KT; kt++) {
        const int key = key0 + kt * 16 + li;
        kok[kt] = key < p.Nk;
        frag_from_global<DP>(fk[kt], p.K + b * p.bsk + (long long)(kok[kt] ? key : 0) * p.ldk + (long long)h * p.D, lg, p.D, kok[kt]);
        frag_from_global<DP>(fv[kt], p.V + b * p.bsv + (long long)(kok[kt] ? key : 0) * p.ldv + (long long)h * p.D, lg, p.D, kok[kt]);
    }
    f32x4 dk[KT][DT], dv[KT][DT];
#pragma unroll
    for (int kt = 0; kt < KT; kt++)
#pragma unroll
        for (int i = 0; i < DT; i++) { dk[kt][i] = (f32x4){0, 0, 0, 0}; dv[kt][i] = (f32x4){0, 0, 0, 0}; }

    Tile<DP, ST, SIDLSG_ATTN_DMA != 0> tq, tdo;
    tq.init(p.ldq, p.D, Qs2[0], Qs2[1], -1);
    tdo.init(p.ldo, p.D, dOs2[0], dOs2[1], -1);
    float lse_r = 0.f, dl_r = 0.f;
    int pb_ = 0;
    auto prefetch = [&](int q0, int into) {
        tq.load(rq, p.ldq, q0, Qs2[into]);
        tdo.load(rdo, p.ldo, q0, dOs2[into]);
        if (threadIdx.x < ST) {
            const int q = q0 + threadIdx.x;
            lse_r = q < p.Nq ? LSEb[q] : INFINITY;   // padded query rows contribute p = exp2(-inf) = 0
            dl_r = q < p.Nq ? DLb[q] : 0.f;
            if (PS) { lse_r = -lse_r; dl_r = -dl_r; }
        }
    };
    // rotated walk over the query tiles (see attn_q_kernel): dK / dV are sums over queries
    const bool rotate = p.rot && p.Nq % ST == 0 && p.Nq >= 2 * ST;
    const int qbase = rotate ? (int)(((long long)bx_ * (p.Nq / ST)) / gridDim.x) * ST : 0;
    prefetch(qbase, 0);
    tq.commit(Qs2[0], LD); tdo.commit(dOs2[0], LD);
    if (threadIdx.x < ST) { lse_s[0][threadIdx.x] = lse_r; dl_s[0][threadIdx.x] = dl_r; }
    __syncthreads();
    auto qtile = [&](const int q0, auto has_next) {       // (run-time `more`: the compile-time split measured slower here)
        const bool more = q0 + ST < p.Nq;
        if (more) { int qn = q0 + ST + qbase; if (qn >= p.Nq && rotate) qn -= p.Nq; prefetch(qn, pb_ ^ 1); }
#pragma unroll
        for (int sub = 0; sub < SUB; sub++) {
        if (sub && q0 + sub * AK_QT >= p.Nq) break;
        const bf16* Qs = Qs2[pb_] + sub * AK_QT * LD;
        const bf16* dOs = dOs2[pb_] + sub * AK_QT * LD;
        const float* lse_c = lse_s[pb_] + sub * AK_QT;
        const float* dl_c = dl_s[pb_] + sub * AK_QT;
        f32x4 pp[KT][4], ds[KT][4];
        ATTN_PRIO(8, 1);
#pragma unroll
        for (int qt = 0; qt < 4; qt++) {
            Frag<DP> fq, fdo;  // A operands: (row = query, k = d)
            frag_from_lds<DP>(fq, Qs + (qt * 16 + li) * LD, lg);
            frag_from_lds<DP>(fdo, dOs + (qt * 16 + li) * LD, lg);
#pragma unroll
            for (int kt = 0; kt < KT; kt++) {
                if constexpr (PS) {
                    const f32x4 nl = *reinterpret_cast<const f32x4*>(&lse_c[qt * 16 + lg * 4]);
                    const f32x4 nd = *reinterpret_cast<const f32x4*>(&dl_c[qt * 16 + lg * 4]);
                    const f32x4 s = mma_d<DP>(nl, fq, fk[kt]);       // S[q][key] - LSE[q]: lane col=key li, rows q = lg*4+r
                    const f32x4 dp = mma_d<DP>(nd, fdo, fv[kt]);     // dP - delta
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const float pr = __builtin_amdgcn_exp2f(s[r]);     // padded queries: -LSE = -inf -> 0
                        pp[kt][qt][r] = pr;
                        ds[kt][qt][r] = pr * dp[r];
                    }
                } else {
                const f32x4 s = mma_d<DP>((f32x4){0, 0, 0, 0}, fq, fk[kt]);    // S[q][key]: lane col=key li, rows q = lg*4+r
                const f32x4 dp = mma_d<DP>((f32x4){0, 0, 0, 0}, fdo, fv[kt]);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int ql = qt * 16 + lg * 4 + r;
                    // padded queries (q >= Nq) carry lse = +inf (set at prefetch) -> exp2(-inf) = 0; padded keys are
                    // never stored.  The d^-1/2 factor of dS is applied to dK in the epilogue.
                    const float pr = __builtin_amdgcn_exp2f(fmaf(s[r], p.scale2, -lse_c[ql]));
                    pp[kt][qt][r] = pr;
                    ds[kt][qt][r] = pr * (dp[r] - dl_c[ql]);
                }
                }
            }
        }
        ATTN_PRIO(8, 0);
        ATTN_PRIO(4, 1);
#pragma unroll
        for (int half = 0; half < 2; half++)
#pragma unroll
            for (int dt = 0; dt < DT; dt++) {
                const bf16x8 fa = tr_frag32(dOs, LD, 32 * half, dt * 16, li, lg);   // dO^T (rows d, k = queries)
                const bf16x8 fb = tr_frag32(Qs, LD, 32 * half, dt * 16, li, lg);    // Q^T
#pragma unroll
                for (int kt = 0; kt < KT; kt++) {
                    const bf16x8 pbv = pack_p(pp[kt][2 * half], pp[kt][2 * half + 1]);
                    const bf16x8 dsb = pack_p(ds[kt][2 * half], ds[kt][2 * half + 1]);
                    dv[kt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, pbv, dv[kt][dt], 0, 0, 0);
                    dk[kt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, dsb, dk[kt][dt], 0, 0, 0);
                }
            }
        ATTN_PRIO(4, 0);
        }
        if (more) {          // the other buffer was last read in the previous stage, i.e. before the previous barrier
            pb_ ^= 1;
            tq.commit(Qs2[pb_], LD); tdo.commit(dOs2[pb_], LD);
            if (threadIdx.x < ST) { lse_s[pb_][threadIdx.x] = lse_r; dl_s[pb_][threadIdx.x] = dl_r; }
            __syncthreads();
        }
    };
    {
        int q0 = 0;
        for (; q0 < p.Nq; q0 += ST) qtile(q0, std::true_type{});
    }
#pragma unroll
    for (int kt = 0; kt < KT; kt++) {
        if (!kok[kt]) continue;
        const int key = key0 + kt * 16 + li;
        bf16* dKp = p.dK + b * p.bsk + (long long)key * p.ldk + (long long)h * p.D;
        bf16* dVp = p.dV + b * p.bsv + (long long)key * p.ldv + (long long)h * p.D;
#pragma unroll
        for (int dt = 0; dt < DT; dt++) {
            const int d = dt * 16 + lg * 4;
            if (d + 4 <= p.D) {
                const float kf = PS ? 0.6931471805599453f : p.scale;     // PS: dK = ln2 * dS^T Q' (Q' = the pre-scaled queries)
                bf16x4 a = {f2bf(dk[kt][dt][0] * kf), f2bf(dk[kt][dt][1] * kf), f2bf(dk[kt][dt][2] * kf), f2bf(dk[kt][dt][3] * kf)};
                bf16x4 c = {f2bf(dv[kt][dt][0]), f2bf(dv[kt][dt][1]), f2bf(dv[kt][dt][2]), f2bf(dv[kt][dt][3])};
                *reinterpret_cast<bf16x4*>(dKp + d) = a;
                *reinterpret_cast<bf16x4*>(dVp + d) = c;
            }
        }
    }
}

// Explicit instantiations of every kernel specialisation the dispatcher can launch: hipcc 7.2 has left individual host stubs
// (__device_stub__...) of implicitly instantiated kernel templates out of the object in some builds of this file (undefined
// symbol at dlopen); a definition request pins them all.
#define ATTN_INST(DP, PS)                                                            \
    template __global__ void attn_q_kernel<DP, 2, 0, true, PS>(AttnParams);           \
    template __global__ void attn_q_kernel<DP, 2, 0, false, PS>(AttnParams);          \
    template __global__ void attn_q_kernel<DP, 2, 1, false, PS>(AttnParams);          \
    template __global__ void attn_dkdv_kernel<DP, 1, PS>(AttnParams);
#define ATTN_INST2(DP) ATTN_INST(DP, false) ATTN_INST(DP, true)
ATTN_INST2(16) ATTN_INST2(32) ATTN_INST2(48) ATTN_INST2(64) ATTN_INST2(80) ATTN_INST2(96) ATTN_INST2(128) ATTN_INST2(160)
template __global__ void attn_dkdv_kernel<48, 2, false>(AttnParams);
template __global__ void attn_dkdv_kernel<48, 2, true>(AttnParams);
#undef ATTN_INST
#undef ATTN_INST2

// (the forward / dQ launches are not templated on KT: with two KT instantiations per head size hipcc 7.2's host pass rejected
// the SECOND use of the same attn_q_kernel specialisation -- "no matching function", substitution failure without a reason)
template <int DP, int QT, bool PS>
static int launch_attn_q(const AttnParams& p, int mode, hipStream_t s) {
    const int qb = 4 * QT * 16;
    if (mode == 0) {
        if (p.D == DP - 8) SIDLSG_LAUNCH((attn_q_kernel<DP, QT, 0, true, PS>), dim3((p.Nq + qb - 1) / qb, p.H, p.B), dim3(256), 0, s, p);
        else SIDLSG_LAUNCH((attn_q_kernel<DP, QT, 0, false, PS>), dim3((p.Nq + qb - 1) / qb, p.H, p.B), dim3(256), 0, s, p);
    } else SIDLSG_LAUNCH((attn_q_kernel<DP, QT, 1, false, PS>), dim3((p.Nq + qb - 1) / qb, p.H, p.B), dim3(256), 0, s, p);
    return sidlsg_last_error();
}
template <int DP, int KT, bool PS>
static int launch_attn_dkdv(const AttnParams& p, hipStream_t s) {
    const int kb = 4 * KT * 16;
    SIDLSG_LAUNCH((attn_dkdv_kernel<DP, KT, PS>), dim3((p.Nk + kb - 1) / kb, p.H, p.B), dim3(256), 0, s, p);
    return sidlsg_last_error();
}
template <int DP, int QT, int KT, bool PS>
static int launch_attn(const AttnParams& p, int mode, hipStream_t s) {
    return mode == 2 ? launch_attn_dkdv<DP, KT, PS>(p, s) : launch_attn_q<DP, QT, PS>(p, mode, s);
}
// Which passes renumber.  Round 4 (rocprofv3 per-kernel averages with / without, MI355X, B = 16): forward 314 -> 312 us (N 4096, d 40), 46.2 -> 44.0
// (N 1024, d 80), 19.0 -> 17.0 (N 256, d 160); dQ pass 816 -> 827 / 91.4 -> 86.7 / 26.6 -> 25.0; dK/dV pass 1110 -> 1174 / 119 -> 115 / 36.9 -> 30.5:
// the backward passes of the 4096-token layers LOST -- all resident blocks of an XCD then walk the same Q / dO rows in lockstep -- so they
// kept the round-robin order and with it 4.1x the algorithmic HBM-side bytes (every XCD fetches every head).  Round 5: the ROTATED walk
// (AttnParams::rot: block i of a head starts at tile i * ntiles / nblocks and wraps) takes the lockstep away: N 4096 d 40 dK/dV pass
// 1042 (round robin) / 1123 (contiguous) / 1039 (contiguous + rotated) us, dQ 784 / 782 / 785, forward 523 / 511 / 507; d 64: 757 / 756 / 757,
// 535 / 540 / 515-521, 421 / 415 / 424 (tools/ab/attn_rot.py under rocprofv3, gpurun_out r5c3) -- kernel time and step time (211-214 ms either
// way) are unchanged, the long backward passes now fetch a head's Q / dO once per XCD instead of once per XCD AND per 4-block group.
// HBM-side bytes per call of the micro-benchmark with everything renumbered: forward 364 -> 107 MB (3.66x -> 1.08x the algorithmic bytes),
// backward 1134 -> 436 MB (4.9x -> 1.9x).
// SIDLSG_ATTN_XCD: bit mask of the passes that may renumber (1 forward, 2 dQ, 4 dK/dV; 8: also the long backward passes); default 15.
static int attn_xcd_on(int mode, const AttnParams& p) {
    static const int mask = getenv("SIDLSG_ATTN_XCD") ? atoi(getenv("SIDLSG_ATTN_XCD")) : 15;
    if (!((mask >> mode) & 1)) return 0;
    return mode == 0 || (mask & 8) || (p.Nq <= 1024 && p.Nk <= 1024);
}
// SIDLSG_ATTN_ROT: bit mask of the passes whose blocks start their tile walk at per-block offsets (1 forward, 2 dQ, 4 dK/dV; default 6: the
// backward passes.  The forward never had the lockstep penalty; rotated it is 1-3 % faster alone but fetches 1.57x instead of 1.08x its
// algorithmic bytes -- profiles/r05_traffic_attn*.json -- and bytes are what the step is short of);
// only together with the XCD-contiguous numbering (without it an XCD holds 4 blocks of each of 16 heads: nothing walks in lockstep)
static int attn_rot_on(int mode, const AttnParams& p) {
    static const int mask = getenv("SIDLSG_ATTN_ROT") ? atoi(getenv("SIDLSG_ATTN_ROT")) : 6;
    return p.xcd && ((mask >> mode) & 1);
}
template <bool PS>
static int dispatch_attn(const AttnParams& p_in, int mode, hipStream_t s) {
    AttnParams p = p_in;
    p.xcd = attn_xcd_on(mode, p);
    p.rot = attn_rot_on(mode, p);
    if (p.D % 8 || p.D <= 0 || p.D > 160) return SIDLSG_EINVAL;
    const int dp = (p.D + 15) / 16 * 16;
    // 32 queries per wave (forward / dQ) and 16 keys per wave (dK/dV) keep 2-3 blocks per CU resident; 64 queries per
    // wave at one wave per SIMD measured up to 3x slower at d=40 (the softmax VALU work only overlaps MFMA across waves).
    // dK/dV with 32 keys per wave halves the LDS fragment traffic per MFMA: measured (N=4096, B=16) 3075 -> 2661 us for
    // the whole d=40 backward, neutral at d=64 -> only DP=48 uses it (SIDLSG_ATTN_KT2=0 switches it off for A/B runs).
    static const bool allow_kt2 = !(getenv("SIDLSG_ATTN_KT2") && atoi(getenv("SIDLSG_ATTN_KT2")) == 0);
    const bool kt2 = allow_kt2 && mode == 2 && p.Nk >= 1024;
    switch (dp) {
        case 16: return launch_attn<16, 2, 1, PS>(p, mode, s);
        case 32: return launch_attn<32, 2, 1, PS>(p, mode, s);
        case 48: return kt2 ? launch_attn<48, 2, 2, PS>(p, mode, s) : launch_attn<48, 2, 1, PS>(p, mode, s);
        case 64: return launch_attn<64, 2, 1, PS>(p, mode, s);
        case 80: return launch_attn<80, 2, 1, PS>(p, mode, s);
        case 96: return launch_attn<96, 2, 1, PS>(p, mode, s);
        case 128: return launch_attn<128, 2, 1, PS>(p, mode, s);
        case 160: return launch_attn<160, 2, 1, PS>(p, mode, s);
    }
    return SIDLSG_EINVAL;
}

// buffer offsets are 32-bit: one batch element's rows must span < 2 GiB
static bool attn_spans_ok(int Nq, int Nk, int ldq, int ldk, int ldv, int ldo) {
    const long long lim = 0x7FFFFFFFLL / 2;
    return (long long)Nq * ldq < lim && (long long)Nk * ldk < lim && (long long)Nk * ldv < lim && (long long)Nq * ldo < lim;
}
static int attn_fwd_impl(bool ps, const void* Q, const void* K, const void* V, void* O, float* LSE, int B, int H, int Nq, int Nk, int D,
                         int ldq, int ldk, int ldv, int ldo, long long bsq, long long bsk, long long bsv, long long bso, void* stream) {
    if (((ldq | ldk | ldv | ldo) & 3) || !attn_spans_ok(Nq, Nk, ldq, ldk, ldv, ldo)) return SIDLSG_EINVAL;
    AttnParams p{};
    p.Q = (const bf16*)Q; p.K = (const bf16*)K; p.V = (const bf16*)V; p.O = (bf16*)O; p.LSE = LSE;
    p.B = B; p.H = H; p.Nq = Nq; p.Nk = Nk; p.D = D; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
    p.bsq = bsq; p.bsk = bsk; p.bsv = bsv; p.bso = bso;
    p.scale = 1.0f / sqrtf((float)D); p.scale2 = p.scale * 1.4426950408889634f;
    return ps ? dispatch_attn<true>(p, 0, (hipStream_t)stream) : dispatch_attn<false>(p, 0, (hipStream_t)stream);
}
static int attn_bwd_impl(bool ps, const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* LSE, void* dQ,
                         void* dK, void* dV, float* delta, int B, int H, int Nq, int Nk, int D, int ldq, int ldk, int ldv, int ldo,
                         long long bsq, long long bsk, long long bsv, long long bso, void* stream) {
    if (((ldq | ldk | ldv | ldo) & 3) || !attn_spans_ok(Nq, Nk, ldq, ldk, ldv, ldo)) return SIDLSG_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    AttnParams p{};
    p.Q = (const bf16*)Q; p.K = (const bf16*)K; p.V = (const bf16*)V; p.dO = (const bf16*)dO;
    p.dQ = (bf16*)dQ; p.dK = (bf16*)dK; p.dV = (bf16*)dV; p.LSE = const_cast<float*>(LSE); p.delta = delta;
    p.B = B; p.H = H; p.Nq = Nq; p.Nk = Nk; p.D = D; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
    p.bsq = bsq; p.bsk = bsk; p.bsv = bsv; p.bso = bso;
    p.scale = 1.0f / sqrtf((float)D); p.scale2 = p.scale * 1.4426950408889634f;
    p.O = const_cast<bf16*>((const bf16*)O); p.delta_out = delta;
    if (int e = ps ? dispatch_attn<true>(p, 1, s) : dispatch_attn<false>(p, 1, s)) return e;
    // dK == dV == null: the caller needs the query gradient only (cross-attention of a FROZEN network: K / V come from the
    // text states through frozen weights, nothing upstream wants their gradient) -- the dK/dV pass over 77 keys has only
    // 2 blocks per (batch, head) and costs 25-100 us per layer for nothing
    if (!dK && !dV) return SIDLSG_OK;
    if (!dK || !dV) return SIDLSG_EINVAL;
    return ps ? dispatch_attn<true>(p, 2, s) : dispatch_attn<false>(p, 2, s);
}

extern "C" {

// O[b][q][h*D..] = softmax(Q K^T * D^-0.5) V.   Q/K/V/O token strides ld*, batch strides bs* (elements).
// LSE: [B][H][Nq] fp32 (log2 domain), may be null when no backward is needed.
int sidlsg_attn_fwd(const void* Q, const void* K, const void* V, void* O, float* LSE, int B, int H, int Nq, int Nk, int D,
                    int ldq, int ldk, int ldv, int ldo, long long bsq, long long bsk, long long bsv, long long bso,
                    void* stream) {
    SidlsgTraceScope ts(SIDLSG_FAM_ATTN_FWD, 4.0 * B * H * Nq * (double)Nk * D, 2.0 * B * H * D * (2.0 * Nq + 2.0 * Nk));
    return attn_fwd_impl(false, Q, K, V, O, LSE, B, H, Nq, Nk, D, ldq, ldk, ldv, ldo, bsq, bsk, bsv, bso, stream);
}

// dQ, dK, dV given dO (same layout as O).  delta: workspace [B][H][Nq] fp32.
int sidlsg_attn_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* LSE, void* dQ,
                    void* dK, void* dV, float* delta, int B, int H, int Nq, int Nk, int D, int ldq, int ldk, int ldv, int ldo,
                    long long bsq, long long bsk, long long bsv, long long bso, void* stream) {
    SidlsgTraceScope ts(SIDLSG_FAM_ATTN_BWD, ((dK && dV) ? 10.0 : 6.0) * B * H * Nq * (double)Nk * D, 2.0 * B * H * D * (4.0 * Nq + ((dK && dV) ? 4.0 : 2.0) * Nk));
    return attn_bwd_impl(false, Q, K, V, O, dO, LSE, dQ, dK, dV, delta, B, H, Nq, Nk, D, ldq, ldk, ldv, ldo, bsq, bsk, bsv, bso, stream);
}

// The same with PRE-SCALED queries: Q holds q * (D^-0.5 * log2 e) -- the caller folds the factor into the q rows of the
// projection weight (sidlsg_scale_cast_ranges), which costs no extra rounding: O = softmax2(Q K^T) V with softmax2 in base 2.
// The backward returns the same tensors as sidlsg_attn_bwd: gradients with respect to the UNSCALED queries (dQ = d^-1/2 dS K),
// keys (dK = d^-1/2 dS^T q = ln2 dS^T Q) and values.
int sidlsg_attn_fwd_ps(const void* Q, const void* K, const void* V, void* O, float* LSE, int B, int H, int Nq, int Nk, int D,
                       int ldq, int ldk, int ldv, int ldo, long long bsq, long long bsk, long long bsv, long long bso,
                       void* stream) {
    SidlsgTraceScope ts(SIDLSG_FAM_ATTN_FWD, 4.0 * B * H * Nq * (double)Nk * D, 2.0 * B * H * D * (2.0 * Nq + 2.0 * Nk));
    return attn_fwd_impl(true, Q, K, V, O, LSE, B, H, Nq, Nk, D, ldq, ldk, ldv, ldo, bsq, bsk, bsv, bso, stream);
}
int sidlsg_attn_bwd_ps(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* LSE, void* dQ,
                       void* dK, void* dV, float* delta, int B, int H, int Nq, int Nk, int D, int ldq, int ldk, int ldv, int ldo,
                       long long bsq, long long bsk, long long bsv, long long bso, void* stream) {
    SidlsgTraceScope ts(SIDLSG_FAM_ATTN_BWD, ((dK && dV) ? 10.0 : 6.0) * B * H * Nq * (double)Nk * D, 2.0 * B * H * D * (4.0 * Nq + ((dK && dV) ? 4.0 : 2.0) * Nk));
    return attn_bwd_impl(true, Q, K, V, O, dO, LSE, dQ, dK, dV, delta, B, H, Nq, Nk, D, ldq, ldk, ldv, ldo, bsq, bsk, bsv, bso, stream);
}

}  // extern "C"
