// ---------------------------------------------------------------------------------------------------------------------
// "p8": bf16 GEMM / implicit-GEMM conv3x3 on 256-row output tiles, 512 threads = 8 waves in TWO STAGGERED GROUPS.
// (included by gemm.hip behind gemm_v3_kernel: shares GemmParams, the loader conventions, the epilogue and the split-K slabs)
//
// Why another structure.  gemm_v3_kernel (128 x 160 tile, 4 waves, 2 blocks per CU) stages 230 bytes per MFMA through the
// L2 -> LDS path and issues 0.45 ds_read_b128 per MFMA; tools/ubench measured that path at 55 % of its cap inside the loop and
// the fragment reads alone at -20 % of the MFMA stream.  Both ratios are properties of the tile, not of the schedule:
//
//   CFG 0 "S": 256 x 160 tile, waves 4 (M) x 2 (N), wave tile  64 x 80 -> 166 B / MFMA staged, 0.45  reads / MFMA, 3 K-tile LDS ring (156 KiB)
//   CFG 1 "W": 256 x 320 tile, waves 2 (M) x 4 (N), wave tile 128 x 80 -> 115 B / MFMA staged, 0.325 reads / MFMA, 2 K-tile LDS ring (144 KiB)
//
// One block per CU, two waves per SIMD.  Schedule (the 8-phase idea of the CDNA4 guide, section "The 256^2 8-phase template",
// rebuilt for these tiles): a K-tile is PH phases of 20 MFMAs per wave (S: kk = 0, 1; W: (kk, 64-row half) = 4); each phase
// is   LOAD { ds_read the phase's fragments ; issue this phase's share of a later K-tile's DMA ; lgkmcnt(0) ; counted vmcnt }
//      s_barrier   MMA { s_setprio 1 ; 20 MFMA ; s_setprio 0 }   s_barrier
// and waves 4-7 run ONE barrier behind waves 0-3, so in every barrier interval one wave of each SIMD issues MFMAs while
// the other one reads fragments and issues DMA.  The DMA of a K-tile is in flight across 8-16 barriers; nothing in the loop
// waits for vmcnt(0) (the last tile(s) of a block excepted).
//
// Hazards (interval k = the code between barrier k and k + 1; group 0 LOADs phase i in interval 2 i, group 1 in 2 i + 1):
//  RAW  a DMA'd region is read only after (its issuing wave's counted vmcnt) -> (a barrier) -> (for the other group one more
//       barrier); every wait below sits in the LOAD of the phase BEFORE the first phase that reads the region.
//  WAR  a region is re-filled only by DMA issued in a LOAD that follows, by at least one barrier, the lgkmcnt(0) of the last
//       LOAD (of BOTH groups) that read it.  S: a buffer is read in both phases of its tile -> re-filled during the tile
//       after it (ring of 3).  W: rows 0-127 of A ("A0") and the W rows are last read in phase 2, rows 128-255 ("A1") in
//       phase 3 -> the next tile's loads into the same buffer start in the following tile's phase 0 (ring of 2).
// Every wave executes the same number of barriers on every path (the stagger barrier of group 1 in front of the loop is
// matched by one of group 0 behind it).
//
// Same operand conventions as v3 (direct-to-LDS buffer loads with one 32-bit offset per row, source-side swizzle, conv taps
// through the scalar offset + halo masks, "transposed" MFMA so that a lane owns 8 consecutive output channels), same K order
// (conv: channel chunk outer, tap inner) and the same MFMA operand placement: an output element is accumulated in exactly
// the order of gemm_v3_kernel -- without split-K the two kernels are bit-identical.
// Takes: N % (80 WN) == 0, K % 64 == 0, dense rows or conv with Cin % 64 == 0 and no fused upsampling; the GEGLU fusions through their own two
// instantiations of the 256 x 320 configuration (p8_epilogue_geglu / _bwd below).
template <int CFG>
struct P8 {
    static constexpr int WN = CFG ? 4 : 2;
    static constexpr int MH = CFG ? 2 : 1;            // 64-row halves of a wave's rows
    static constexpr int MT = 4 * MH, NT = 5;
    static constexpr int BM = 256, BN = WN * 80;
    static constexpr int R = CFG ? 2 : 3;             // LDS ring depth in K-tiles
    static constexpr int STAGE = (BM + BN) * BK;      // elements per K-tile buffer
    static constexpr int NTH = 512;
    static constexpr int LDS_BYTES = R * STAGE * 2;
    static constexpr int NWJ = (BN / 8 + 7) / 8;      // W-tile DMA instructions per wave (W: 5; S: 3, the third one on waves 0-3 only)
};

DEVFN int wsw80(int r) {        // W-tile swizzle of a row, relative to its wave's 80-row band (see wsw)
    const int rr = r % 80, q = rr >> 1;
    return rr < 64 ? ((q & 1) | (((q >> 2) & 3) << 1)) : (q & 7);
}

#ifndef SIDLSG_P8_PRIO
#define SIDLSG_P8_PRIO 1
#endif
#ifndef SIDLSG_P8_ASM_DMA
#define SIDLSG_P8_ASM_DMA 0
#endif
#define P8_FENCE() do { __builtin_amdgcn_sched_barrier(0); asm volatile("" ::: "memory"); } while (0)
#define P8_BAR() do { P8_FENCE(); __builtin_amdgcn_s_barrier(); P8_FENCE(); } while (0)
#define P8_LGKM0() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define P8_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

// Epilogue.  The shared fragment-layout epilogue (gemm_epilogue) keeps, per wave, bias + row-vector + residual operands for its whole
// tile in registers next to the accumulators: with the 128 x 80 wave tile that is > 256 registers (it spilled, and its 20 k lines of
// code cost ~90 us per tile in the first measurement -- more than the K loop of a 320 -> 320 conv).  Here the accumulators go through
// LDS as fp32, 32 rows per wave and pass (MT / 2 passes, 80 KiB image), and the tile is finished ROW-MAJOR: a thread owns 8
// consecutive channels of a row, loads the residual with one coalesced 16-byte load (the fragment layout fetched 64-byte pieces of
// 16 rows per instruction) and stores 16 bytes.
// gfx950 retires loads AND stores through one in-order counter, so a load issued behind a pass's stores cannot be waited for before
// those stores are acknowledged by the memory system: with bias / row-vector loads inside the passes every pass paid a store round
// trip (one-K-tile launch: 25 us for a 256 x 320 tile against an 8 us HBM floor).  Hence: bias and the row vectors of the (at most
// P8_RV_IMGS) images the tile touches are copied to LDS once, before the first store, and the residual chunks of pass p + 1 are
// requested BEFORE the stores of pass p -- the wait for them then leaves those stores (and the next request) in flight.
// Arithmetic and its order are those of epilogue_rows8: x = acc * alpha + bias; x += rowvec; x += res; SiLU; round.
// The hot feature sets (bf16 output, no activation: bias only / + residual / + row vector) are straight-line instantiations with
// branch-free range handling (rows >= M get an out-of-range buffer offset: loads return zeros, stores are dropped) -- with per-chunk
// branches the compiler merges the paths with s_waitcnt vmcnt(0) and the counted waits are gone; FEAT 3 reads the flags at run time.
constexpr int P8_RV_IMGS = 5;
template <int CFG, int FEAT>      // FEAT 0: bias only, 1: + residual, 2: + row vector, 3: generic
DEVFN void p8_epilogue_t(const GemmParams& p, f32x4 (&acc)[5][4 * (CFG ? 2 : 1)], float* img, int m0, int n0, int wr, int wn0, int li, int lg, int tid) {
    constexpr int BN = CFG ? 320 : 160, MT = CFG ? 8 : 4;
    constexpr int LDI = BN + 4;                  // fp32 image row stride: + 16 bytes, the 8 rows of a b128 write phase fall on 8 distinct slots
    constexpr int ROWS = CFG ? 64 : 128;         // image rows per pass
    constexpr int CPR = BN / 8;                  // 8-channel chunks per row
    constexpr int NCH = ROWS * CPR / 512;        // chunks per thread and pass (5)
    constexpr int NPASS = MT / 2;
    constexpr bool GENERIC = FEAT == 3;
    float* sbias = img + ROWS * LDI;             // [BN]
    float* srv = sbias + BN;                     // [P8_RV_IMGS][BN]
    const bool has_res = GENERIC ? p.res != nullptr : FEAT == 1;
    const bool has_rv = GENERIC ? p.rowvec != nullptr : FEAT == 2;
    const int flags = GENERIC ? p.flags : 0;
    const float* __restrict__ bias = p.bias;
    const float* __restrict__ rowvec = p.rowvec;
    const int esz = (flags & F_OUT_F32) ? 4 : 2;
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (int)(((long long)(p.M - 1) * p.ldc + p.N) * esz), 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(has_res ? p.res : reinterpret_cast<const bf16*>(p.C)), 0,
                                                                        has_res ? (int)(((long long)(p.M - 1) * p.ldres + p.N) * 2) : 0, 0x00020000);
    const int mlast = min(m0 + 255, p.M - 1);
    const int img0 = has_rv ? m0 / p.rows_per_batch : 0;
    const int nimg = has_rv ? mlast / p.rows_per_batch - img0 + 1 : 0;
    const bool rv_lds = has_rv && (!GENERIC || nimg <= P8_RV_IMGS);
    for (int c = tid; c < BN / 4; c += 512) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(sbias + c * 4) = bias ? *reinterpret_cast<const f32x4*>(bias + n0 + c * 4) : z;
    }
    if (rv_lds)
        for (int c = tid; c < nimg * (BN / 4); c += 512) {
            const int k = c / (BN / 4), cc = c - k * (BN / 4);
            *reinterpret_cast<f32x4*>(srv + k * BN + cc * 4) = *reinterpret_cast<const f32x4*>(rowvec + (size_t)(img0 + k) * p.ldrv + n0 + cc * 4);
        }
    int ccol[NCH], clr[NCH];
#pragma unroll
    for (int i = 0; i < NCH; i++) {
        const int c = tid + i * 512;
        clr[i] = c / CPR;
        ccol[i] = (c - clr[i] * CPR) * 8;
    }
    auto chunk_row = [&](int pass, int i) {
        const int lr = clr[i];
        return m0 + (CFG ? (pass >> 1) * 128 + (lr >> 5) * 64 + (pass & 1) * 32 + (lr & 31) : (lr >> 5) * 64 + pass * 32 + (lr & 31));
    };
    bf16x8 rs[2][NCH];
    auto request_res = [&](int pass, bf16x8 (&dst)[NCH]) {
#pragma unroll
        for (int i = 0; i < NCH; i++) {
            const int m = chunk_row(pass, i);
            const unsigned off = m < p.M ? ((unsigned)m * (unsigned)p.ldres + n0 + ccol[i]) * 2u : OOB;
            dst[i] = has_res ? buf_ld8(rr, off) : zero8();
        }
    };
    request_res(0, rs[0]);
#pragma unroll
    for (int pass = 0; pass < NPASS; pass++) {
        if (pass) __syncthreads();               // the previous image has been consumed
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int mi = 2 * pass + h;
            float* row = img + (wr * 32 + h * 16 + li) * LDI + wn0;
#pragma unroll
            for (int ni = 0; ni < 5; ni++)
                *reinterpret_cast<f32x4*>(row + (ni < 4 ? 32 * (ni >> 1) + lg * 8 + (ni & 1) * 4 : 64 + lg * 4)) = acc[ni][mi];
        }
        __syncthreads();                         // (pass 0: also publishes sbias / srv)
        if (pass + 1 < NPASS) request_res(pass + 1, rs[(pass + 1) & 1]);
#pragma unroll
        for (int i = 0; i < NCH; i++) {
            const int m = chunk_row(pass, i), col = ccol[i];
            const bool ok = m < p.M;
            const float* ip = img + clr[i] * LDI + col;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(ip), v1 = *reinterpret_cast<const f32x4*>(ip + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(sbias + col), b1 = *reinterpret_cast<const f32x4*>(sbias + col + 4);
            f32x4 r0 = {0.f, 0.f, 0.f, 0.f}, r1 = r0;
            if (has_rv) {
                const int im = (ok ? m : m0) / p.rows_per_batch;
                if (rv_lds) {
                    const float* rv = srv + (im - img0) * BN + col;
                    r0 = *reinterpret_cast<const f32x4*>(rv); r1 = *reinterpret_cast<const f32x4*>(rv + 4);
                } else {
                    const float* rv = rowvec + (size_t)im * p.ldrv + n0 + col;
                    r0 = *reinterpret_cast<const f32x4*>(rv); r1 = *reinterpret_cast<const f32x4*>(rv + 4);
                }
            }
            const bf16x8 rsv = rs[pass & 1][i];
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                float t = (e < 4 ? v0[e] : v1[e - 4]) * p.alpha + (e < 4 ? b0[e] : b1[e - 4]);
                if (has_rv) t += e < 4 ? r0[e] : r1[e - 4];
                if (has_res) t += bf2f(rsv[e]);
                if (GENERIC && (flags & F_SILU)) t = silu_f(t);
                x[e] = t;
            }
            if (GENERIC && (flags & F_OUT_F32)) {
                if (!ok) continue;
                float* c = reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n0 + col;
                if (flags & F_ACCUM) {
                    const f32x4 c0 = *reinterpret_cast<const f32x4*>(c), c1 = *reinterpret_cast<const f32x4*>(c + 4);
#pragma unroll
                    for (int e = 0; e < 8; e++) x[e] += e < 4 ? c0[e] : c1[e - 4];
                }
                *reinterpret_cast<f32x4*>(c) = (f32x4){x[0], x[1], x[2], x[3]};
                *reinterpret_cast<f32x4*>(c + 4) = (f32x4){x[4], x[5], x[6], x[7]};
            } else {
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; e++) o[e] = f2bf(x[e]);
                const unsigned off = ok ? ((unsigned)m * (unsigned)p.ldc + n0 + col) * 2u : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rc, off, 0, 0);
            }
        }
    }
}

template <int CFG>
DEVFN void p8_epilogue(const GemmParams& p, f32x4 (&acc)[5][4 * (CFG ? 2 : 1)], float* img, int m0, int n0, int wr, int wn0, int li, int lg, int tid) {
    const bool plain = !(p.flags & (F_SILU | F_OUT_F32 | F_ACCUM));
    const int nimg = p.rowvec ? min(m0 + 255, p.M - 1) / p.rows_per_batch - m0 / p.rows_per_batch + 1 : 0;
    if (plain && !p.rowvec && !p.res) p8_epilogue_t<CFG, 0>(p, acc, img, m0, n0, wr, wn0, li, lg, tid);
    else if (plain && !p.rowvec) p8_epilogue_t<CFG, 1>(p, acc, img, m0, n0, wr, wn0, li, lg, tid);
    else if (plain && !p.res && nimg <= P8_RV_IMGS) p8_epilogue_t<CFG, 2>(p, acc, img, m0, n0, wr, wn0, li, lg, tid);
    else p8_epilogue_t<CFG, 3>(p, acc, img, m0, n0, wr, wn0, li, lg, tid);
}

// ---- GEGLU epilogues of the 256 x 320 configuration (round 6, late): the transformer FeedForward of the grouped frozen pass and of the batch-16 passes are
// the long dense launches of an iteration, and the v3-only fusions (sidlsg_gemm_geglu_bf16 / _bwd_bf16) kept them off these kernels.  Same pass structure as
// p8_epilogue_t (accumulators -> fp32 LDS image, 64 rows per pass, row-major finish, every access branch-free through buffer descriptors); same arithmetic and
// rounding points as the v3 epilogues (gemm.hip): h = bf16(acc * alpha + bias), y = bf16(a * gelu(g)) from the ROUNDED a, g; dy = bf16(acc * alpha).
// Forward: the tile's 320 columns are 160 "a" features [160 nt, +160) and the SAME 160 features of the gate half (W rows F + ...: see the loader), so a row of the
// image holds both operands of its 160 outputs: a thread owns 8 features of a row with BOTH operands (the first version walked all 40 chunks of a row and
// evaluated the GELU in the gate-half lanes as well: +17-20 % on the K = 320 forward) and stores y, h[:, f] and h[:, F + f] (a descriptor of zero records when the
// caller keeps no h: those stores are dropped).
DEVFN void p8_epilogue_geglu(const GemmParams& p, f32x4 (&acc)[5][8], float* img, int m0, int nt, int wr, int wn0, int li, int lg, int tid) {
    constexpr int BN = 320, HB = 160, LDI = BN + 4, ROWS = 64, NPASS = 4;
    constexpr int PPR = HB / 8;                          // (a, gate) chunk PAIRS per image row: a thread owns 8 features of a row, both operands
    constexpr int NPAIR = ROWS * PPR;                    // 1280 per pass = 2.5 per thread: three branch-free rounds, the last one half masked
    constexpr int NIT = (NPAIR + 511) / 512;
    float* sbias = img + ROWS * LDI;
    const int F = p.geglu;
    const float* __restrict__ bias = p.bias;
    bf16* H = reinterpret_cast<bf16*>(p.C);
    const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(H ? H : p.Y, 0, H ? (int)(((long long)(p.M - 1) * p.ldc + p.N) * 2) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(p.Y, 0, (int)(((long long)(p.M - 1) * p.ldy + F) * 2), 0x00020000);
    for (int c = tid; c < BN / 4; c += 512) {
        const int col = c * 4;
        const int gcol = col < HB ? nt * HB + col : F + nt * HB + (col - HB);
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(sbias + col) = bias ? *reinterpret_cast<const f32x4*>(bias + gcol) : z;
    }
    int plr[NIT], pca[NIT];
    bool pok[NIT];
#pragma unroll
    for (int i = 0; i < NIT; i++) {
        const int c = tid + i * 512;
        pok[i] = c < NPAIR;
        const int cc = pok[i] ? c : NPAIR - 1;           // (masked lanes recompute the last pair; their stores are dropped)
        plr[i] = cc / PPR;
        pca[i] = (cc - plr[i] * PPR) * 8;
    }
#pragma unroll
    for (int pass = 0; pass < NPASS; pass++) {
        if (pass) __syncthreads();
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int mi = 2 * pass + h;
            float* row = img + (wr * 32 + h * 16 + li) * LDI + wn0;
#pragma unroll
            for (int ni = 0; ni < 5; ni++)
                *reinterpret_cast<f32x4*>(row + (ni < 4 ? 32 * (ni >> 1) + lg * 8 + (ni & 1) * 4 : 64 + lg * 4)) = acc[ni][mi];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NIT; i++) {
            const int lr = plr[i], ca = pca[i];
            const int m = m0 + (pass >> 1) * 128 + (lr >> 5) * 64 + (pass & 1) * 32 + (lr & 31);
            const bool ok = pok[i] && m < p.M;
            const float* ip = img + lr * LDI + ca;
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(ip), a1 = *reinterpret_cast<const f32x4*>(ip + 4);
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(ip + HB), g1 = *reinterpret_cast<const f32x4*>(ip + HB + 4);
            const f32x4 ba0 = *reinterpret_cast<const f32x4*>(sbias + ca), ba1 = *reinterpret_cast<const f32x4*>(sbias + ca + 4);
            const f32x4 bg0 = *reinterpret_cast<const f32x4*>(sbias + HB + ca), bg1 = *reinterpret_cast<const f32x4*>(sbias + HB + ca + 4);
            bf16x8 av, gv, yv;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                av[e] = f2bf((e < 4 ? a0[e] : a1[e - 4]) * p.alpha + (e < 4 ? ba0[e] : ba1[e - 4]));
                gv[e] = f2bf((e < 4 ? g0[e] : g1[e - 4]) * p.alpha + (e < 4 ? bg0[e] : bg1[e - 4]));
                yv[e] = f2bf(bf2f(av[e]) * gelu_t<bf16>(bf2f(gv[e])));
            }
            const int f = nt * HB + ca;
            const unsigned oy = ok ? ((unsigned)m * (unsigned)p.ldy + f) * 2u : OOB;
            const unsigned oh = ok ? ((unsigned)m * (unsigned)p.ldc + f) * 2u : OOB;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, yv), ry, oy, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, av), rh, oh, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, gv), rh, oh + (unsigned)F * 2u, 0, 0);      // (out of range stays out of range)
        }
    }
}
// Backward: the tile is dy[:, n0 .. n0 + 320) = dOut W2 (never stored); the epilogue reads h[:, n] / h[:, F + n] (requested one pass ahead, before the previous
// pass's stores: one in-order memory counter) and writes dH[:, n] = dy * gelu(g), dH[:, F + n] = dy * a * gelu'(g).
DEVFN void p8_epilogue_geglu_bwd(const GemmParams& p, f32x4 (&acc)[5][8], float* img, int m0, int n0, int wr, int wn0, int li, int lg, int tid) {
    constexpr int BN = 320, LDI = BN + 4, ROWS = 64, CPR = BN / 8, NCH = ROWS * CPR / 512, NPASS = 4;
    const int F = p.geglu_bwd;
    const int bytes = (int)(((long long)(p.M - 1) * p.ldy + 2 * F) * 2);
    const __amdgpu_buffer_rsrc_t rhin = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.Hin), 0, bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rdh = __builtin_amdgcn_make_buffer_rsrc(p.Y, 0, bytes, 0x00020000);
    int ccol[NCH], clr[NCH];
#pragma unroll
    for (int i = 0; i < NCH; i++) {
        const int c = tid + i * 512;
        clr[i] = c / CPR;
        ccol[i] = (c - clr[i] * CPR) * 8;
    }
    auto chunk_off = [&](int pass, int i) -> unsigned {
        const int lr = clr[i];
        const int m = m0 + (pass >> 1) * 128 + (lr >> 5) * 64 + (pass & 1) * 32 + (lr & 31);
        return m < p.M ? ((unsigned)m * (unsigned)p.ldy + n0 + ccol[i]) * 2u : OOB;
    };
    bf16x8 ha[2][NCH], hg[2][NCH];
    auto request = [&](int pass, bf16x8 (&a)[NCH], bf16x8 (&g)[NCH]) {
#pragma unroll
        for (int i = 0; i < NCH; i++) {
            const unsigned off = chunk_off(pass, i);
            a[i] = buf_ld8(rhin, off);
            g[i] = buf_ld8(rhin, off + (unsigned)F * 2u);      // (an out-of-range offset stays out of range)
        }
    };
    request(0, ha[0], hg[0]);
#pragma unroll
    for (int pass = 0; pass < NPASS; pass++) {
        if (pass) __syncthreads();
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int mi = 2 * pass + h;
            float* row = img + (wr * 32 + h * 16 + li) * LDI + wn0;
#pragma unroll
            for (int ni = 0; ni < 5; ni++)
                *reinterpret_cast<f32x4*>(row + (ni < 4 ? 32 * (ni >> 1) + lg * 8 + (ni & 1) * 4 : 64 + lg * 4)) = acc[ni][mi];
        }
        __syncthreads();
        if (pass + 1 < NPASS) request(pass + 1, ha[(pass + 1) & 1], hg[(pass + 1) & 1]);
#pragma unroll
        for (int i = 0; i < NCH; i++) {
            const float* ip = img + clr[i] * LDI + ccol[i];
            const bf16x8 av = ha[pass & 1][i], gv = hg[pass & 1][i];
            bf16x8 da, dg;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                float gl, gd;
                gelu_pair_t<bf16>(bf2f(gv[e]), gl, gd);
                const float d = bf2f(f2bf(ip[e] * p.alpha));
                da[e] = f2bf(d * gl);
                dg[e] = f2bf(d * bf2f(av[e]) * gd);
            }
            const unsigned off = chunk_off(pass, i);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, da), rdh, off, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, dg), rdh, off + (unsigned)F * 2u, 0, 0);
        }
    }
}

template <int MODE, int CFG, int EPI = 0>   // MODE 0 dense rows, 1 conv3x3 (Cin % 64 == 0, no upsampling); EPI 1 / 2: GEGLU forward / backward epilogue (dense, CFG 1)
DEVFN void gemm_p8_body(GemmParams& p) {
    using G = P8<CFG>;
    constexpr int BM = G::BM, BN = G::BN, MT = G::MT, NT = G::NT, R = G::R, STAGE = G::STAGE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* ring = reinterpret_cast<bf16*>(smem);

    // work item -> (tile, split): the logical order and XCD remap of gemm_v3_kernel
    const int tiles_n = p.N / BN;
    const int tiles_m = m_tiles_rt(p, BM);
    const int nk_all = p.K / BK;
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
        const int gm = min(p.group_m, tiles_m - first_m);
        const int in_g = t - g * per_group;
        mt = first_m + in_g % gm;
        nt = in_g / gm;
    }
    const int m0 = group_select<BM>(p, mt);
    const int n0 = nt * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;                                  // stagger group: waves w and w + 4 share a SIMD
    const int wr = CFG ? (wave >> 2) : (wave >> 1), wc = CFG ? (wave & 3) : (wave & 1);
    const int wm0 = wr * 64, wn0 = wc * 80;                     // W: the wave's rows are wm0 + [0, 64) and 128 + wm0 + [0, 64)
    const int li = lane & 15, lg = lane >> 4;
    const int lrow = lane >> 3, lslot = lane & 7;

    // ---- loader state: one 32-bit offset per 8-row DMA instruction (see gemm_v3_kernel)
    const unsigned tap_rs = (unsigned)p.Wd * (unsigned)p.lda * 2u, tap_ps = (unsigned)p.lda * 2u;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16*>(p.A) - (MODE == 1 ? (size_t)(tap_rs + tap_ps) / 2 : 0), 0, (int)(p.a_bytes + (MODE == 1 ? tap_rs + tap_ps : 0u)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.W), 0, (int)p.w_bytes, 0x00020000);
    auto arow = [&](int j) { return CFG ? (j >> 1) * 128 + wave * 16 + (j & 1) * 8 : wave * 32 + j * 8; };      // first tile row of A instruction j
    unsigned aoff[4], tmask[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int r = arow(j) + lrow;
        const int kcs = lslot ^ ((r >> 1) & 7);
        const int m = m0 + r;
        const bool ok = m < p.M;
        if (MODE == 0) {
            aoff[j] = ok ? ((unsigned)m * (unsigned)p.lda + kcs * 8) * 2u : OOB;
            tmask[j] = 0;
        } else {
            const int mm = ok ? m : 0;
            const int hw = p.Ho * p.Wo;
            const int b = mm / hw, rem = mm - b * hw;
            const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
            const int hi0 = ho * p.stride, wi0 = wo * p.stride;          // centre tap
            aoff[j] = ((unsigned)(b * p.H * p.Wd) * (unsigned)p.lda + kcs * 8) * 2u + (unsigned)(hi0 * p.Wd + wi0) * tap_ps;
            unsigned mk = 0;
#pragma unroll
            for (int tp = 0; tp < 9; tp++) {
                const int hi = hi0 - 1 + tp / 3, wi = wi0 - 1 + tp % 3;
                if (ok && hi >= 0 && hi < p.H && wi >= 0 && wi < p.Wd) mk |= 1u << tp;
            }
            tmask[j] = mk;
        }
    }
    unsigned boff[G::NWJ];
#pragma unroll
    for (int j = 0; j < G::NWJ; j++) {
        const int r = (j * 8 + wave) * 8 + lrow;
        const int kcs = lslot ^ wsw80(r);
        // (EPI 1, fused GEGLU: the tile's W rows are 160 "a" rows [160 nt, +160) and the same 160 rows of the gate half, F rows further down)
        const int n = EPI == 1 ? (r < 160 ? nt * 160 + r : p.geglu + nt * 160 + (r - 160)) : n0 + r;
        boff[j] = (r < BN && n < p.N) ? ((unsigned)n * (unsigned)p.K + kcs * 8) * 2u : OOB;
    }
    const int kt_begin = p.kt_per_split ? split * p.kt_per_split : 0;
    const int nk = p.kt_per_split ? min(nk_all, kt_begin + p.kt_per_split) : nk_all;

    int i_asoff = 0, i_k0w = 0;
    unsigned i_bit = 0;
    auto set_issue = [&](int t) {            // scalar state of the K-tile whose DMA is being issued
        if (MODE == 1) {
            const int cc = t / 9;
            const int tap = t - cc * 9;
            const int dh = tap / 3, dw = tap - dh * 3;
            i_asoff = cc * BK * 2 + dh * (int)tap_rs + dw * (int)tap_ps;
            i_k0w = (tap * p.Cin + cc * BK) * 2;
            i_bit = 1u << tap;
        } else {
            i_asoff = i_k0w = t * BK * 2;
        }
    };
    auto glds = [&](__amdgpu_buffer_rsrc_t rs, bf16* dst, unsigned voff, int soff) {
#if SIDLSG_P8_ASM_DMA
        dma16_asm(rs, dst, voff, (unsigned)soff);
#else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)dst, 16, voff, soff, 0, 0);
#endif
    };
    auto dma_a = [&](int j, int buf) {
        const unsigned o = MODE == 1 ? ((tmask[j] & i_bit) ? aoff[j] : OOB) : aoff[j];
        glds(ra, ring + buf * STAGE + arow(j) * BK, o, i_asoff);
    };
    auto dma_w = [&](int j, int buf) { glds(rw, ring + buf * STAGE + BM * BK + (j * 8 + wave) * 8 * BK, boff[j], i_k0w); };

    // ---- fragment addresses (bytes inside a K-tile buffer, kk = 0; kk = 1 is the same address ^ 64)
    const int a_rd = (wm0 + li) * 128 + (((lg ^ (li >> 1)) & 7) << 4);
    int w_rd[NT];
#pragma unroll
    for (int ni = 0; ni < NT; ni++) {
        const bool paired = (ni | 1) < NT;
        const int r = wn0 + (paired ? 32 * (ni >> 1) + (li >> 2) * 8 + (ni & 1) * 4 + (li & 3) : 16 * ni + li);
        w_rd[ni] = BM * 128 + r * 128 + (((lg ^ wsw80(r)) & 7) << 4);
    }
    f32x4 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; i++)
#pragma unroll
        for (int j = 0; j < MT; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 fa[4], fw[NT];
    auto rd_a = [&](const char* bb, int mh, int kk) {
#pragma unroll
        for (int mi = 0; mi < 4; mi++) fa[mi] = *reinterpret_cast<const bf16x8*>(bb + ((a_rd ^ (kk * 64)) + mh * (128 * 128) + mi * (16 * 128)));
    };
    auto rd_w = [&](const char* bb, int kk) {
#pragma unroll
        for (int ni = 0; ni < NT; ni++) fw[ni] = *reinterpret_cast<const bf16x8*>(bb + (w_rd[ni] ^ (kk * 64)));
    };
    auto mma = [&](auto mh_c) {
        constexpr int mh = decltype(mh_c)::value;
        if (SIDLSG_P8_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ni = 0; ni < NT; ni++)
#pragma unroll
            for (int mi = 0; mi < 4; mi++)
                acc[ni][mh * 4 + mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[ni], fa[mi], acc[ni][mh * 4 + mi], 0, 0, 0);
        if (SIDLSG_P8_PRIO) __builtin_amdgcn_s_setprio(0);
    };
    using c0 = std::integral_constant<int, 0>;
    using c1 = std::integral_constant<int, 1>;

    // ---- prologue: the first K-tile (S: the first two), the first one drained
    set_issue(kt_begin);
#pragma unroll
    for (int j = 0; j < 4; j++) dma_a(j, 0);
#pragma unroll
    for (int j = 0; j < G::NWJ; j++)
        if (CFG || j < 2 || wave < 4) dma_w(j, 0);
    if (CFG == 0 && kt_begin + 1 < nk) {
        set_issue(kt_begin + 1);
#pragma unroll
        for (int j = 0; j < 4; j++) dma_a(j, 1);
#pragma unroll
        for (int j = 0; j < G::NWJ; j++)
            if (j < 2 || wave < 4) dma_w(j, 1);
        if (wave < 4) P8_VMCNT(7); else P8_VMCNT(6);
    } else {
        P8_VMCNT(0);
    }
    P8_BAR();
    if (grp) P8_BAR();                      // stagger: group 1 runs one barrier behind group 0

    // One K-tile.  ISSUE: DMA of tile t + R - 1 rides in this tile's LOADs.  S only: WAIT 1 = leave this tile's own DMA in
    // flight (steady state), 2 = drain (no DMA issued: the next tile must be complete), 0 = nothing to wait for (last tile).
    auto tile = [&](const int t, const int buf, auto issue_c, auto wait_c) {
        constexpr bool ISSUE = decltype(issue_c)::value;
        constexpr int WAIT = decltype(wait_c)::value;
        const char* bb = reinterpret_cast<const char*>(ring) + buf * (STAGE * 2);
        const int ib = buf == 0 ? R - 1 : buf - 1;             // ring slot of tile t + R - 1 = the slot of tile t - 1
        if (ISSUE) set_issue(t + R - 1);
        if constexpr (CFG == 1) {
            // phase 0: (kk 0, rows 0-63 of the wave)
            rd_w(bb, 0); rd_a(bb, 0, 0);
            if (ISSUE) { dma_a(0, ib); dma_a(1, ib); }
            P8_LGKM0();
            if (ISSUE) P8_VMCNT(2); else P8_VMCNT(0);          // A1 of THIS tile (issued in the previous tile's phase 3) has landed
            P8_BAR(); mma(c0{}); P8_BAR();
            // phase 1: (kk 0, rows 128-191)
            rd_a(bb, 1, 0);
            if (ISSUE) { dma_w(0, ib); dma_w(1, ib); dma_w(2, ib); }
            P8_LGKM0();
            P8_BAR(); mma(c1{}); P8_BAR();
            // phase 2: (kk 1, rows 0-63): last reads of A0 and W of this buffer
            rd_w(bb, 1); rd_a(bb, 0, 1);
            if (ISSUE) { dma_w(3, ib); dma_w(4, ib); }
            P8_LGKM0();
            P8_BAR(); mma(c0{}); P8_BAR();
            // phase 3: (kk 1, rows 128-191): last reads of A1
            rd_a(bb, 1, 1);
            if (ISSUE) { dma_a(2, ib); dma_a(3, ib); }
            P8_LGKM0();
            if (ISSUE) P8_VMCNT(2);                             // A0 + W of the next tile have landed; its A1 stays in flight
            P8_BAR(); mma(c1{}); P8_BAR();
        } else {
            rd_w(bb, 0); rd_a(bb, 0, 0);
            if (ISSUE) {
#pragma unroll
                for (int j = 0; j < 4; j++) dma_a(j, ib);
            }
            P8_LGKM0();
            P8_BAR(); mma(c0{}); P8_BAR();
            rd_w(bb, 1); rd_a(bb, 0, 1);
            if (ISSUE) {
                dma_w(0, ib); dma_w(1, ib);
                if (wave < 4) dma_w(2, ib);
            }
            P8_LGKM0();
            if (WAIT == 1) { if (wave < 4) P8_VMCNT(7); else P8_VMCNT(6); }      // tile t + 1 (issued one tile ago) has landed
            if (WAIT == 2) P8_VMCNT(0);
            P8_BAR(); mma(c0{}); P8_BAR();
        }
    };
    {
        int kt = kt_begin, buf = 0;
        auto next = [&]() { kt++; buf = buf + 1 == R ? 0 : buf + 1; };
        if constexpr (CFG == 1) {
            for (; kt + 1 < nk; next()) tile(kt, buf, std::true_type{}, c0{});
            tile(kt, buf, std::false_type{}, c0{});
        } else {
            for (; kt + 2 < nk; next()) tile(kt, buf, std::true_type{}, c1{});
            if (kt + 1 < nk) { tile(kt, buf, std::false_type{}, std::integral_constant<int, 2>{}); next(); }
            tile(kt, buf, std::false_type{}, c0{});
        }
    }
    if (!grp) P8_BAR();                     // group 0 waits for group 1's last MMA: every wave has now passed the same number of barriers

    auto out_row = [&](int mi) { return m0 + (mi >> 2) * 128 + wm0 + (mi & 3) * 16 + li; };
    if (p.kt_per_split) {
#pragma unroll
        for (int mi = 0; mi < MT; mi++) {
            const int m = out_row(mi);
            if (m >= p.M) continue;
#pragma unroll
            for (int ni = 0; ni < NT; ni++) {
                const bool paired = (ni | 1) < NT;
                const int nb = n0 + wn0 + (paired ? 32 * (ni >> 1) + lg * 8 + (ni & 1) * 4 : 16 * ni + lg * 4);
                float* dst = p.ws + ((size_t)split * p.Mtot + m) * p.N + nb;
                *reinterpret_cast<f32x4*>(dst) = acc[ni][mi];
            }
        }
        return;
    }
    if constexpr (EPI == 1) p8_epilogue_geglu(p, acc, reinterpret_cast<float*>(smem), m0, nt, wr, wn0, li, lg, tid);
    else if constexpr (EPI == 2) p8_epilogue_geglu_bwd(p, acc, reinterpret_cast<float*>(smem), m0, n0, wr, wn0, li, lg, tid);
    else p8_epilogue<CFG>(p, acc, reinterpret_cast<float*>(smem), m0, n0, wr, wn0, li, lg, tid);
}

template <int MODE>
__global__ __launch_bounds__(512, 1) void gemm_p8s_kernel(GemmParams p) { gemm_p8_body<MODE, 0>(p); }
template <int MODE>
__global__ __launch_bounds__(512, 1) void gemm_p8w_kernel(GemmParams p) { gemm_p8_body<MODE, 1>(p); }

// (plain kernels: a kernel template with a further non-type parameter got no host stub from hipcc 7.2)
__global__ __launch_bounds__(512, 1) void gemm_p8w_geglu_kernel(GemmParams p) { gemm_p8_body<0, 1, 1>(p); }
__global__ __launch_bounds__(512, 1) void gemm_p8w_geglu_bwd_kernel(GemmParams p) { gemm_p8_body<0, 1, 2>(p); }

// The GEGLU fusions on the 256 x 320 configuration: admission (shapes, alignment, 31-bit descriptors) and the cost model of gemm.hip::p8_rule without
// K splits (the epilogues finish their tile).  SIDLSG_P8_GEGLU=0: the v3 kernels as before (A/B switch).
static bool p8_geglu_takes(const GemmParams& p) {
    static const bool on = !(getenv("SIDLSG_P8_GEGLU") && atoi(getenv("SIDLSG_P8_GEGLU")) == 0);
    const int F = p.geglu ? p.geglu : p.geglu_bwd;
    if (!on || !F || (p.geglu && p.geglu_bwd) || F % 160 || p.N % 320 || p.K % BK || p.kt_per_split || p.wscale || p.res || p.rowvec || p.flags) return false;
    if ((p.ldy & 7) || ((uintptr_t)p.Y & 15) || (p.lda & 7) || ((uintptr_t)p.A & 15)) return false;
    if (p.geglu) {
        if ((p.C && ((p.ldc & 7) || ((uintptr_t)p.C & 15))) || (p.bias && ((uintptr_t)p.bias & 15)) || (p.Mg && p.bias1 && ((uintptr_t)p.bias1 & 15))) return false;
        if (((long long)(p.M - 1) * p.ldc + p.N) * 2 >= 0x7FFFFFFFll || ((long long)(p.M - 1) * p.ldy + F) * 2 >= 0x7FFFFFFFll) return false;
    } else {
        if (p.bias || !p.Hin || ((uintptr_t)p.Hin & 15) || ((long long)(p.M - 1) * p.ldy + 2 * F) * 2 >= 0x7FFFFFFFll) return false;
    }
    const int nk = p.K / BK;
    const long long t3 = (long long)m_tiles_rt(p, 128) * (p.N / 160), t8 = (long long)m_tiles_rt(p, 256) * (p.N / 320);
    const double v3_us = (double)((t3 + 511) / 512) * (11.0 + 1.03 * nk), p8_us = (double)((t8 + 255) / 256) * (16.0 + 1.61 * nk);
    return p8_us < v3_us * 0.97;
}
template <int EPI>
static int launch_gemm_p8_geglu(const GemmParams& p, hipStream_t s) {
    using G = P8<1>;
    const int tiles = m_tiles_rt(p, G::BM) * (p.N / G::BN);
    static bool attr_done = false;
    auto kern = EPI == 1 ? &gemm_p8w_geglu_kernel : &gemm_p8w_geglu_bwd_kernel;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
        attr_done = true;
    }
    GemmParams q = p;
    q.group_m = 4;
    SIDLSG_LAUNCH(kern, dim3(tiles), dim3(G::NTH), (size_t)G::LDS_BYTES, s, q);
    return sidlsg_last_error();
}

// admission: what the p8 kernels take at all (the dispatcher adds the tile-count rule)
template <int MODE>
static bool p8_ok(const GemmParams& p, int cfg) {
    if (MODE == 2) return false;
    const int bn = cfg ? 320 : 160;
    if (p.N % bn || p.K % BK || p.geglu || p.geglu_bwd || p.wscale) return false;
    if (MODE == 1) {
        if (p.ups || p.Cin % BK) return false;
        const unsigned long long tap = ((unsigned long long)p.Wd + 1) * (unsigned)p.lda * 2ull;
        if ((unsigned long long)p.a_bytes + tap >= 0x80000000ull) return false;
    }
    // the row-major epilogue moves 16 bytes (32 for fp32 outputs) per access
    if ((p.ldc & 7) || ((uintptr_t)p.C & 15) || (p.res && ((p.ldres & 7) || ((uintptr_t)p.res & 15)))) return false;
    if (p.rowvec && ((p.ldrv & 3) || ((uintptr_t)p.rowvec & 15) || p.rows_per_batch <= 0)) return false;
    if ((p.bias && ((uintptr_t)p.bias & 15)) || (p.Mg && p.bias1 && ((uintptr_t)p.bias1 & 15))) return false;
    const long long mrows = p.Mtot > p.M ? p.Mtot : p.M;          // (C / res are addressed through 31-bit buffer descriptors)
    if (((mrows - 1) * p.ldc + p.N) * ((p.flags & F_OUT_F32) ? 4 : 2) >= 0x7FFFFFFFll || (p.res && ((mrows - 1) * p.ldres + p.N) * 2 >= 0x7FFFFFFFll)) return false;
    return true;
}

template <int MODE, int CFG>
static int launch_gemm_p8(const GemmParams& p, hipStream_t s) {
    using G = P8<CFG>;
    const int tiles = m_tiles_rt(p, G::BM) * (p.N / G::BN);
    static bool attr_done = false;
    auto kern = CFG ? &gemm_p8w_kernel<MODE> : &gemm_p8s_kernel<MODE>;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
        attr_done = true;
    }
    const int nk = p.K / BK;
    const int splits = p.kt_per_split ? (nk + p.kt_per_split - 1) / p.kt_per_split : 1;
    GemmParams q = p;
    q.group_m = 4;
    SIDLSG_LAUNCH(kern, dim3(tiles * splits), dim3(G::NTH), (size_t)G::LDS_BYTES, s, q);
    if (p.kt_per_split) {
        const size_t n4 = ((size_t)p.M * p.N + 3) / 4;
        SIDLSG_LAUNCH(gemm_finish_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, p, splits);
    }
    return sidlsg_last_error();
}
