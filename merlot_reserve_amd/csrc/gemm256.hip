// Second-generation bf16 MFMA GEMM for gfx950: 256 x {128,96} x 64 tiles, 8 waves (4 x 2), one workgroup per CU,
// operands streamed global -> LDS with LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction), a 3-stage
// ring kept TWO k-tiles ahead behind a counted s_waitcnt vmcnt(6) and ONE raw s_barrier per k-tile (never vmcnt(0)
// inside the loop).  LDS images are lane-linear per 1-KiB piece (an LDS-DMA constraint), so the bank-conflict
// swizzles live on the per-lane SOURCE address and are undone by the matching XOR on the fragment reads:
//   K-contiguous tile  [rows][64 k]  (128-B rows):  chunk ^= (row >> 1) & 7         -> conflict-free ds_read_b128
//   K-strided   tile   [64 k][W]     (W = 256|128): chunk ^= 2*((k&3) + 4*((k>>3)&1)) -> conflict-free ds_read_b64_tr_b16
// Out-of-range rows / columns are fetched with an out-of-range buffer offset, which the hardware zero-fills.
// Same contract and epilogue as gemm.hip (mr_gemm dispatches here for large problems).
#include "gemm256_sched.h"
#include "mr_options.h"

namespace g256 {


template <int BN, bool TA, bool TB>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(const G256Args ga) {
    constexpr int NW = 8;
    using GEO = Geo<BN>;
    constexpr int BKT = GEO::BKT, STAGE_A = GEO::STAGE_A;
    constexpr int WCOLS = BN / 2;       // wave tile = 64 x WCOLS: waves 4 x 2
    constexpr int NJ = WCOLS / 16;                       // 16-col MFMA tiles per wave
    constexpr int BW = GEO::BW, NBP = GEO::NBP, STAGE_B = GEO::STAGE_B, OFF_B = GEO::OFF_B;
    constexpr int NSTAGE_A = GEO::NSTAGE_A, NSTAGE_B = GEO::NSTAGE_B, B_AHEAD = GEO::B_AHEAD;
    constexpr int WAITN = GEO::WAITN;
    // Epilogue operands: fetched EARLY (top of the last k-tile, see PRE_LOAD_B64) by the 128 / 96-wide variants; the 256 /
    // 192-wide ones have no registers to park them in and load them in the epilogue itself, row block by row block.
    constexpr bool EARLY = NJ <= 4;
    constexpr int NPRE = EARLY ? 4 * NJ : 1;
    __shared__ __attribute__((aligned(16))) char smem[GEO::LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);        // provably wave-uniform: LDS-DMA bases stay scalar
    const int wm = wave >> 1, wn = wave & 1;
    const int g = lane >> 4, li = lane & 15;

    // PERSISTENT workgroups: block b handles work items b', b' + G, ... where b' is the XCD-aware permutation of b
    // (blocks sharing an XCD's L2 get neighbouring tiles).  Round r of the grid covers items [rG, rG + G).
    const int G = gridDim.x;
    const int xcd = blockIdx.x & 7, qd = G >> 3, rm = G & 7;
    const int bperm = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (blockIdx.x >> 3);
    const int splits = ga.splits;
#define GET_ITEM(w) make_item<BKT, 8>(ga, (w), BN, bperm)

    // ---- issue cursors (one per operand): run ahead of the compute cursor, across item boundaries ----
    int iwa = bperm, ika = 0, ista = 0, iwb = bperm, ikb = 0, istb = 0;
    Item ia = GET_ITEM(iwa), ib = ia;
    unsigned ao0, ao1, ao2, ao3, bo0, bo1, bo2 = 0, bo3 = 0, a_step = 0, b_step = 0;
    __amdgpu_buffer_rsrc_t ra, rb;      // wave-uniform descriptors of each cursor's problem (zero-fill beyond the extent)
#define SET_OFFSETS_A()                                                                                                 \
    do {                                                                                                                \
        const mr_gemm_args& q_ = ga.p[ia.pi];                                                                           \
        const int64_t lda_ = q_.lda, M_ = q_.M, K_ = q_.K;                                                              \
        const int64_t a_rows = TA ? K_ : M_, a_cols = TA ? M_ : K_;                                                     \
        ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(q_.A), 0, (int)(((a_rows - 1) * lda_ + a_cols) * 2), 0x00020000); \
        a_step = TA ? (unsigned)(BKT * lda_ * 2) : (unsigned)(BKT * 2);                                                  \
        ao0 = piece_src<TA, 256, BKT>(wave * 4 + 0, lane, lda_, ia.m0, M_);                                                  \
        ao1 = piece_src<TA, 256, BKT>(wave * 4 + 1, lane, lda_, ia.m0, M_);                                                  \
        ao2 = piece_src<TA, 256, BKT>(wave * 4 + 2, lane, lda_, ia.m0, M_);                                                  \
        ao3 = piece_src<TA, 256, BKT>(wave * 4 + 3, lane, lda_, ia.m0, M_);                                                  \
    } while (0)
#define SET_OFFSETS_B()                                                                                                 \
    do {                                                                                                                \
        const mr_gemm_args& q_ = ga.p[ib.pi];                                                                           \
        const int64_t ldb_ = q_.ldb, N_ = q_.N, K_ = q_.K;                                                              \
        const int64_t b_rows = TB ? N_ : K_, b_cols = TB ? K_ : N_;                                                     \
        rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(q_.B), 0, (int)(((b_rows - 1) * ldb_ + b_cols) * 2), 0x00020000); \
        b_step = TB ? (unsigned)(BKT * 2) : (unsigned)(BKT * ldb_ * 2);                                                  \
        const int64_t nb_ = (BN < BW && ib.n0 + BN < N_) ? ib.n0 + BN : N_;    /* columns the tile really covers */       \
        bo0 = piece_src<!TB, BW, BKT>(wave * NBP + 0, lane, ldb_, ib.n0, nb_);                                               \
        bo1 = piece_src<!TB, BW, BKT>(wave * NBP + 1, lane, ldb_, ib.n0, nb_);                                               \
        if (NBP == 4) {                                                                                                 \
            bo2 = piece_src<!TB, BW, BKT>(wave * NBP + 2, lane, ldb_, ib.n0, nb_);                                           \
            bo3 = piece_src<!TB, BW, BKT>(wave * NBP + 3, lane, ldb_, ib.n0, nb_);                                           \
        }                                                                                                               \
    } while (0)
    // The wave's 4 A pieces / NBP B pieces of the next k-tile of each cursor's sequence (if any).  They are issued between
    // the MFMA batches of a k-tile instead of stalling the wave right after the barrier.
#define ISSUE_A(issued)                                                                                                 \
    do {                                                                                                                \
        (issued) = ia.valid;                                                                                            \
        if (ia.valid) {                                                                                                 \
            char* st_ = smem + ista * STAGE_A + wave * 4096;                                                            \
            const unsigned sa = MR_DIAG_K(ia.kt0 + ika) * a_step;                                                       \
            MR_DMA(ra, MR_LDS_PTR(void, st_), 16, ao0, sa, 0, 0);                                                       \
            MR_DMA(ra, MR_LDS_PTR(void, st_ + 1024), 16, ao1, sa, 0, 0);                                                \
            MR_DMA(ra, MR_LDS_PTR(void, st_ + 2048), 16, ao2, sa, 0, 0);                                                \
            MR_DMA(ra, MR_LDS_PTR(void, st_ + 3072), 16, ao3, sa, 0, 0);                                                \
            ista = (ista == NSTAGE_A - 1) ? 0 : ista + 1;                                                               \
            if (++ika == ia.nkt) {                                                                                      \
                iwa += G;                                                                                               \
                ika = 0;                                                                                                \
                ia = GET_ITEM(iwa);                                                                                     \
                if (ia.valid) SET_OFFSETS_A();                                                                          \
            }                                                                                                           \
        }                                                                                                               \
    } while (0)
#define ISSUE_B()                                                                                                       \
    do {                                                                                                                \
        if (ib.valid) {                                                                                                 \
            char* sb_ = smem + OFF_B + istb * STAGE_B + wave * (NBP * 1024);                                            \
            const unsigned sb = MR_DIAG_K(ib.kt0 + ikb) * b_step;                                                       \
            MR_DMA(rb, MR_LDS_PTR(void, sb_), 16, bo0, sb, 0, 0);                                                       \
            MR_DMA(rb, MR_LDS_PTR(void, sb_ + 1024), 16, bo1, sb, 0, 0);                                                \
            if (NBP == 4) {                                                                                             \
                MR_DMA(rb, MR_LDS_PTR(void, sb_ + 2048), 16, bo2, sb, 0, 0);                                            \
                MR_DMA(rb, MR_LDS_PTR(void, sb_ + 3072), 16, bo3, sb, 0, 0);                                            \
            }                                                                                                           \
            istb = (istb == NSTAGE_B - 1) ? 0 : istb + 1;                                                               \
            if (++ikb == ib.nkt) {                                                                                      \
                iwb += G;                                                                                               \
                ikb = 0;                                                                                                \
                ib = GET_ITEM(iwb);                                                                                     \
                if (ib.valid) SET_OFFSETS_B();                                                                          \
            }                                                                                                           \
        }                                                                                                               \
    } while (0)
    // B before A: when B runs only one k-tile ahead its pieces are the ones the next step needs, and the A pieces behind
    // them are the ones allowed to stay in flight
#define ISSUE_NEXT(issued) do { ISSUE_B(); ISSUE_A(issued); } while (0)




    if (ia.valid) { SET_OFFSETS_A(); SET_OFFSETS_B(); }
    int cw = bperm, csa = 0, csb = 0;
#ifdef MR_DIAG_STAMPS
    unsigned long long* stamps = static_cast<unsigned long long*>(ga.p[0].workspace);
    int nstamp = 0;
#endif
    Item ci = GET_ITEM(cw);
    if (!ci.valid) return;
    bool issued;
    // all but the youngest WAITN pieces have landed (everything when no A piece was issued this step: the stream is ending)
#define RING_WAIT(issued)                                                                                               \
    do {                                                                                                                \
        if (issued) wait_vmcnt<WAITN>();                                                                                \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                           \
    } while (0)
    // prologue: k-tile 0 of both operands, then the steady state's lead (A: k-tile 1; B: k-tile 1 when it runs two ahead)
    ISSUE_NEXT(issued);
    if (B_AHEAD == 2) ISSUE_NEXT(issued); else ISSUE_A(issued);
    RING_WAIT(issued);
    __builtin_amdgcn_s_barrier();

    while (ci.valid) {
        // accumulators are kept TRANSPOSED (mfma(B-frag, A-frag)): lane holds C[m = .. + li][n = .. + 4g + r], i.e. 4
        // consecutive columns of one row, so the epilogue moves 8-byte packed bf16 instead of single elements
        f32x4 acc[4][NJ];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        // Everything the epilogue reads from global memory (bias, and ONE of: residual / act' factor / "rotary" scales) is
        // fetched at the top of the item's LAST k-tile with asm loads: older than that step's LDS-DMA pieces, so the
        // step's own vmcnt(6) retires them under the MFMAs and the epilogue runs on registers only.
        // Every field of the problem that the prefetch / epilogue needs is copied into registers ONCE per item: the
        // k-loop's asm statements clobber "memory", so anything left in kernarg memory would be re-loaded (s_load +
        // full lgkmcnt wait, ~200 cycles each) at every use.  32-bit index math; 64 bits only to form addresses.
        const mr_gemm_args& pc = ga.p[ci.pi];
        const int eM = (int)pc.M, eN = (int)pc.N;
        const int64_t e_ldc = pc.ldc;
        __bf16* const eC = static_cast<__bf16*>(pc.C);
        __bf16* const eC2 = static_cast<__bf16*>(pc.c2);
        const __bf16* const e_bias = static_cast<const __bf16*>(pc.bias);
        const bool has_res = pc.residual != nullptr, has_aux = pc.aux != nullptr;      // never both (host check)
        const __bf16* const pre_src = static_cast<const __bf16*>(has_res ? pc.residual : pc.aux);
        const int64_t pre_ld = has_res ? pc.ldr : pc.ldaux;
        const float* const e_rot = pc.rot_tab;
        const int e_rot_rows = (int)pc.rot_rows, e_rot_cols = (int)pc.rot_cols;
        const int e_grp = (int)pc.out_grp, e_gstride = (int)pc.out_grp_stride, e_goff = (int)pc.out_grp_off;
        const bool do_act = (pc.act == MR_ACT_GELU1702);
        const bool epi_bf16 = (splits == 1) && (pc.c_dtype == MR_DT_BF16);
        const bool pre_rot = epi_bf16 && pre_src == nullptr && e_rot != nullptr && BN >= 128;   // BN = 96 never gets a rot_tab (host)
        const void* const dummy = pc.A;
        float* const e_cs = static_cast<float*>(pc.colsum);
        const int64_t e_ldcs = pc.ldcs;
        auto out_row = [&](int gm) -> int { return e_grp > 0 ? (gm / e_grp) * e_gstride + e_goff + gm % e_grp : gm; };
        u32x2 pbias[NJ];
        u32x2 pre2[NPRE];       // (i, j) -> 4 bf16 of residual / aux;  or, "rotary": (i, j < 2) -> two halves of 4 fp32 scales
        // The residual / aux tile is as large as the output tile: fetched in one go by every CU at once it is a burst of
        // ~17 MB that the last k-tile has to sit through (the counted vmcnt is in order).  So its four 16-row blocks are
        // requested one per k-tile from the item's first k-tiles on, BEHIND that k-tile's LDS-DMA pieces (the ring's wait
        // then leaves those NJ loads, and the previous k-tile's, in flight as well); whatever an item with few k-tiles
        // has not requested by its last k-tile is requested there.
        int pphase = (EARLY && epi_bf16 && pre_src != nullptr) ? 0 : 4, pre_prev = 0;
        auto pre_rows = [&](int i) {
            const int gm = ci.m0 + wm * 64 + i * 16 + li;
            const __bf16* rowp = pre_src + (int64_t)out_row(gm) * pre_ld;
            // 16 bytes per lane in the layout of the widened stores (lanes of a row read 64 contiguous bytes); the
            // epilogue undoes it with the same two v_permlane16_swap.  8-byte loads touched every line four times.
#pragma unroll
            for (int jp = 0; jp < NJ / 2; ++jp) {
                const int gn = ci.n0 + wn * WCOLS + (2 * jp + (g & 1)) * 16 + (g >> 1) * 8;
                const void* src_ = (gm < eM && gn < eN) ? (const void*)(rowp + gn) : dummy;
                const u32x4 v = *reinterpret_cast<const u32x4*>(src_);
                pre2[(i * NJ + 2 * jp) % NPRE] = u32x2{v[0], v[1]};
                pre2[(i * NJ + 2 * jp + 1) % NPRE] = u32x2{v[2], v[3]};
            }
            if (NJ & 1) {
                const int gn = ci.n0 + wn * WCOLS + (NJ - 1) * 16 + g * 4;
                const void* src_ = (gm < eM && gn < eN) ? (const void*)(rowp + gn) : dummy;
                PRE_LOAD_B64(pre2[(i * NJ + NJ - 1) % NPRE], src_);
            }
        };
        // k-loop with the LAST k-tile peeled: the prefetch registers are written (asm) and consumed in straight-line code,
        // so no loop-carried copy of a register whose load is still in flight can be generated.
        auto kstep = [&](auto pre_row_c) {
            constexpr int PRE_ROW = decltype(pre_row_c)::value;      // 16-row block of the residual / aux tile to request, or -1
            // the stage being refilled was last read one step ago, behind that step's barrier
            const char* As = smem + csa * STAGE_A;
            const char* Bs = smem + OFF_B + csb * STAGE_B;
            // Units of 16 MFMAs: (kk, half) with the wave's B columns taken JH 16-column blocks at a time (all of them when
            // NJ <= 4), so the fragments in flight stay at A(kk) + 2 B halves even for the 128-column waves of BN = 256.
            constexpr int NH = (NJ > 4) ? 2 : 1, JH = NJ / NH;
            bf16x8 af[4];
            constexpr int UNITS = (BKT / 32) * NH;
#pragma unroll
            for (int u = 0; u < UNITS; ++u) {
                const int kk = u / NH, h = u % NH;
                bf16x8 bfr[JH];
                if (h == 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) af[i] = MR_DIAG_FRAG((frag<TA, 256, BKT>(As, wm * 64 + i * 16, kk, lane)));
                }
#pragma unroll
                for (int j = 0; j < JH; ++j) bfr[j] = MR_DIAG_FRAG((frag<!TB, BW, BKT>(Bs, wn * WCOLS + (h * JH + j) * 16, kk, lane)));
                // the DMA issue rides in the shadow of the fragment reads' latency / the previous unit's MFMAs
                if (u == 0) ISSUE_B();
                if (u == UNITS / 2) ISSUE_A(issued);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < JH; ++j)
                        MR_DIAG_MFMA(acc[i][h * JH + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][h * JH + j], 0, 0, 0));
            }
            int pre_now = 0;
            if constexpr (EARLY && PRE_ROW >= 0) {
                if (pphase < 4) {             // = this item has a residual / aux operand
                    pre_rows(PRE_ROW);
                    pphase = PRE_ROW + 1;
                    pre_now = 1;
                }
            }
            // k-tile cseq+1 must have landed before anyone reads it: everything but this step's 6 pieces (this also
            // retires the previous item's epilogue stores, which were issued before them) and the residual / aux loads
            // requested behind this k-tile's and the previous k-tile's pieces
            // (a pre_rows call issues PRE_LOADS loads: one 16-byte load per pair of column blocks + one 8-byte load for an odd
            // block.  The counts must be exact: vmcnt retires in order, and any slack here would let pieces of k-tile cseq+1 --
            // older than all of these -- still be in flight when the barrier opens.  They were 2 x NJ / NJ, i.e. 4 / 2 too many,
            // until round 2: never observed to fail, the pieces are a whole k-tile old by then, but not guaranteed either.)
            constexpr int PRE_LOADS = NJ / 2 + (NJ & 1);
            if (EARLY && issued && pre_now + pre_prev == 2) wait_vmcnt<WAITN + 2 * PRE_LOADS>();
            else if (EARLY && issued && pre_now + pre_prev == 1) wait_vmcnt<WAITN + PRE_LOADS>();
            else RING_WAIT(issued);
            pre_prev = pre_now;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            csa = (csa == NSTAGE_A - 1) ? 0 : csa + 1;
            csb = (csb == NSTAGE_B - 1) ? 0 : csb + 1;
        };
        MR_STAMP(0);
        {   // the first four k-tiles request the residual / aux row blocks (static register indices: peeled, not switched)
            const int nloop = ci.nkt - 1;
            if (nloop > 0) kstep(std::integral_constant<int, 0>{});
            if (nloop > 1) kstep(std::integral_constant<int, 1>{});
            if (nloop > 2) kstep(std::integral_constant<int, 2>{});
            if (nloop > 3) kstep(std::integral_constant<int, 3>{});
            for (int t = 4; t < nloop; ++t) kstep(std::integral_constant<int, -1>{});
        }
        MR_STAMP(1);
        {
            if (EARLY && epi_bf16) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int gn = ci.n0 + wn * WCOLS + j * 16 + g * 4;
                    const void* src_ = (e_bias != nullptr && gn < eN) ? (const void*)(e_bias + gn) : dummy;
                    PRE_LOAD_B64(pbias[j], src_);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int gm = ci.m0 + wm * 64 + i * 16 + li;
                    const bool mok = gm < eM;
                    if (pre_src != nullptr) {
                        if (pphase <= i) pre_rows(i);
                    } else if (pre_rot) {
                        const int rr = (e_rot_rows >= eM) ? gm : gm % e_rot_rows;
                        const float* rowp = e_rot + (int64_t)rr * 32 + g * 4;
#pragma unroll
                        for (int j2 = 0; j2 < 4; ++j2) {  // the wave's columns are whole heads (64); dims < 32 are j & 3 = 0, 1
                            const void* src_ = mok ? (const void*)(rowp + (j2 >> 1) * 16 + (j2 & 1) * 2) : dummy;
                            PRE_LOAD_B64(pre2[(i * 4 + j2) % NPRE], src_);
                        }
                    }
                }
            }
            pphase = 4;
            kstep(std::integral_constant<int, -1>{});
        }

        MR_STAMP(2);
        if (EARLY && epi_bf16) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) reg_fence(pbias[j]);
#pragma unroll
            for (int k = 0; k < NPRE; ++k) reg_fence(pre2[k]);
        }
        const int m0 = ci.m0, n0 = ci.n0;
        const int wrow0 = m0 + wm * 64, wcol0 = n0 + wn * WCOLS;
        if (!epi_bf16) {
            // split-K partials and fp32 outputs (contrastive logits): rare, small; direct loads / scalar stores
            const mr_gemm_args& p = ga.p[ci.pi];
            if (splits > 1) {   // raw fp32 accumulators (N % 4 == 0 checked on the host)
                float* Wp = static_cast<float*>(p.workspace) + (int64_t)ci.split * p.M * p.N;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int64_t m = wrow0 + i * 16 + li;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const int64_t n = wcol0 + j * 16 + g * 4;
                        if (m < p.M && n < p.N) *reinterpret_cast<f32x4*>(Wp + m * p.N + n) = acc[i][j];
                    }
                }
            } else {
                const __bf16* bias = static_cast<const __bf16*>(p.bias);
                float* C = static_cast<float*>(p.C);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int64_t m = wrow0 + i * 16 + li;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const int64_t n = wcol0 + j * 16 + g * 4;
                        f32x4 v = acc[i][j];
                        if (bias != nullptr && n < p.N) {
                            const bf16x4 b4 = *reinterpret_cast<const bf16x4*>(bias + n);
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] += (float)b4[r];
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (m < p.M && n + r < p.N) C[m * p.ldc + n + r] = v[r];
                    }
                }
            }
        } else {
            // ---------------- bf16 epilogue: registers only (operands were prefetched during the last k-tile) ----------------
            // The epilogue is VALU-bound (two waves per SIMD run it with the MFMA pipe idle), so it is specialised at compile
            // time for the five operand combinations the step uses -- plain / bias, "rotary", GELU + gelu' copy, residual,
            // aux -- and chosen per item by one wave-uniform switch; MODE 5 keeps every flag dynamic for anything else.
            auto epilogue = [&](auto mode_c) {
                constexpr int MODE = decltype(mode_c)::value;
                constexpr bool GEN = MODE == 5;
                const bool f_rot = GEN ? pre_rot : MODE == 1, f_act = GEN ? do_act : MODE == 2;
                const bool f_c2 = GEN ? eC2 != nullptr : MODE == 2;
                const bool f_res = GEN ? has_res : MODE == 3, f_aux = GEN ? has_aux : MODE == 4;
                if (!EARLY && e_bias != nullptr) {
    #pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const int gn = wcol0 + j * 16 + g * 4;
                        pbias[j] = *reinterpret_cast<const u32x2*>(gn < eN ? e_bias + gn : e_bias);
                    }
                }
                if (e_bias == nullptr) {            // no bias: add zeros (keeps the per-element code branch-free)
    #pragma unroll
                    for (int j = 0; j < NJ; ++j) pbias[j] = u32x2{0u, 0u};
                }
                // 256 / 192-wide tiles (no registers to park epilogue operands in during the k-loop): the WHOLE residual / aux
                // tile is requested here, before the first row block is finished, in the 16-byte layout of the widened stores
                // (the fragment registers are dead by now) -- one exposed load latency per tile instead of one per 16-row
                // block: 8-byte loads issued and awaited row block by row block cost the aux GEMM 27 %, the residual ones
                // 10-22 %.  Likewise the "rotary" scales, one row block ahead.  (Specialised modes only: the generic mode keeps
                // every flag dynamic and would hold all of these live at once.)
                // column sums of the stored tile (the bias gradient: see mr_gemm_args.colsum), aux mode only: that epilogue waits
                // on memory, the extra vector work hides under it
                constexpr bool CS = MODE == 4 && NJ < 8;     // (the host takes 192-wide tiles when column sums are asked for: beside 128 accumulators they spill)
                const bool f_cs = CS && e_cs != nullptr;
                f32x4 cs[CS ? NJ : 1];
                if constexpr (CS) {
    #pragma unroll
                    for (int j = 0; j < NJ; ++j) cs[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                constexpr bool LATE_X = !EARLY && (MODE == 3 || MODE == 4);
                constexpr bool LATE_R = !EARLY && MODE == 1 && BN == 192;     // 256-wide: 2 x 8 scale vectors beside 128 accumulators spill
                constexpr int LXD = (NJ == 8) ? 2 : 4;       // row blocks in flight (256-wide: 2, or the tile's 128 accumulators spill)
                u32x4 lx[LATE_X ? LXD : 1][LATE_X ? NJ / 2 : 1];
                auto lx_fetch = [&](int i, u32x4 (&dst)[LATE_X ? NJ / 2 : 1]) {
                    if constexpr (LATE_X) {
                        const int gm = wrow0 + i * 16 + li;
                        const __bf16* rowp = pre_src + (int64_t)out_row(gm) * pre_ld;
    #pragma unroll
                        for (int jp = 0; jp < NJ / 2; ++jp) {
                            const int gn = wcol0 + (2 * jp + (g & 1)) * 16 + (g >> 1) * 8;
                            dst[jp] = *reinterpret_cast<const u32x4*>((gm < eM && gn < eN) ? (const void*)(rowp + gn) : dummy);
                        }
                    }
                };
                if constexpr (LATE_X) {
    #pragma unroll
                    for (int i = 0; i < LXD; ++i) lx_fetch(i, lx[i]);
                }
                f32x4 rotv[LATE_R ? 2 : 1][LATE_R ? NJ : 1];
                auto rot_fetch = [&](int i, f32x4 (&dst)[LATE_R ? NJ : 1]) {
                    if constexpr (LATE_R) {
                        const int gm = wrow0 + i * 16 + li;
                        const int rr = (gm >= eM) ? 0 : (e_rot_rows >= eM) ? gm : gm % e_rot_rows;
    #pragma unroll
                        for (int j = 0; j < NJ; ++j) {
                            const int gn0 = wcol0 + j * 16 + g * 4;      // the lane's 4 columns lie inside one head's first or second 32 dims
                            const bool on = (gn0 & 63) < 32 && gn0 < e_rot_cols;
                            const f32x4 t = *reinterpret_cast<const f32x4*>(on ? (const void*)(e_rot + (int64_t)rr * 32 + (gn0 & 63)) : (const void*)e_rot);
                            dst[j] = on ? t : f32x4{1.f, 1.f, 1.f, 1.f};
                        }
                    }
                };
                if constexpr (LATE_R) rot_fetch(0, rotv[0]);
                auto finish_pre = [&](int i, int j) -> f32x4 {
                    f32x4 v = acc[i][j];
                    if (!EARLY) {
                        {
                            const bf16x4 b4 = __builtin_bit_cast(bf16x4, pbias[j]);
    #pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] += (float)b4[r];
                        }
                        // "rotary" scale for ANY tile width (a 192-wide tile gives a wave 1.5 heads): the lane's 4 columns start
                        // at a multiple of 4, so they lie inside one head's first or second 32 dims
                        if constexpr (LATE_R) {
                            v *= rotv[i & 1][j];
                        } else {
                            const int gn0 = wcol0 + j * 16 + g * 4;
                            if (f_rot && (gn0 & 63) < 32 && gn0 < e_rot_cols) {
                                const int gm = wrow0 + i * 16 + li;
                                const int rr = (gm >= eM) ? 0 : (e_rot_rows >= eM) ? gm : gm % e_rot_rows;
                                v *= *reinterpret_cast<const f32x4*>(e_rot + (int64_t)rr * 32 + (gn0 & 63));
                            }
                        }
                        return v;
                    }
                    {
                        const bf16x4 b4 = __builtin_bit_cast(bf16x4, pbias[j]);
    #pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += (float)b4[r];
                    }
                    // "rotary" scale: host guarantees BN >= 128, so the wave's columns are whole heads and (n & 63) = 16 (j & 3) + 4 g;
                    // the table depends on the row and the dim only, so one prefetched set serves both heads of a 128-wide wave
                    if (f_rot && (j & 3) < 2 && wcol0 + (j >> 2) * 64 < e_rot_cols) {
                        const u32x2 lo = pre2[(i * 4 + (j & 1) * 2) % NPRE], hi = pre2[(i * 4 + (j & 1) * 2 + 1) % NPRE];
                        v *= __builtin_bit_cast(f32x4, u32x4{lo[0], lo[1], hi[0], hi[1]});
                    }
                    return v;
                };
                // Stores are widened to 16 B: lanes l and l+16 hold columns 4g..4g+3 and 4g+4.. of the same row, so one
                // v_permlane16_swap per dword between the registers of two adjacent 16-column blocks (rows 1,3 of the first
                // <-> rows 0,2 of the second) leaves every lane with 8 contiguous columns: lane rows g = 0/2 own block 2jp
                // (columns 0-7 / 8-15), g = 1/3 own block 2jp+1.  A wave instruction then writes 64 contiguous bytes per
                // row instead of 32 (the 8-byte form was store-issue bound).
                auto store_pair = [&](__bf16* rowp, bool mok, int jp, bf16x4 va, bf16x4 vb) {
                    u32x2 ua = __builtin_bit_cast(u32x2, va), ub = __builtin_bit_cast(u32x2, vb);
                    const auto s0 = __builtin_amdgcn_permlane16_swap(ua[0], ub[0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap(ua[1], ub[1], false, false);
                    const int col = wcol0 + (2 * jp + (g & 1)) * 16 + (g >> 1) * 8;
                    if (MR_DIAG_ST(mok && col < eN)) *reinterpret_cast<u32x4*>(rowp + col) = u32x4{s0[0], s1[0], s0[1], s1[1]};
                };
    #pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int gm = wrow0 + i * 16 + li;
                    const bool mok = gm < eM;
                    const int64_t roff = (int64_t)out_row(gm) * e_ldc;
                    __bf16* const crow = eC + roff;
                    __bf16* const c2row = eC2 + roff;
                    bf16x4 oc[NJ], od[NJ], xs[NJ];
                    if constexpr (LATE_R) { if (i < 3) rot_fetch(i + 1, rotv[(i + 1) & 1]); }
                    if constexpr (LATE_X) {       // undo the 16-byte load layout (as the EARLY path below does)
    #pragma unroll
                        for (int jp = 0; jp < NJ / 2; ++jp) {
                            const u32x4 v4 = lx[i % LXD][jp];
                            const auto s0 = __builtin_amdgcn_permlane16_swap(v4[0], v4[2], false, false);
                            const auto s1 = __builtin_amdgcn_permlane16_swap(v4[1], v4[3], false, false);
                            xs[2 * jp] = __builtin_bit_cast(bf16x4, u32x2{s0[0], s1[0]});
                            xs[2 * jp + 1] = __builtin_bit_cast(bf16x4, u32x2{s0[1], s1[1]});
                        }
                        if (i + LXD < 4) lx_fetch(i + LXD, lx[i % LXD]);
                    }
                    if (EARLY && (f_res || f_aux)) {
    #pragma unroll
                        for (int jp = 0; jp < NJ / 2; ++jp) {
                            const u32x2 lo = pre2[(i * NJ + 2 * jp) % NPRE], hi = pre2[(i * NJ + 2 * jp + 1) % NPRE];
                            const auto s0 = __builtin_amdgcn_permlane16_swap(lo[0], hi[0], false, false);
                            const auto s1 = __builtin_amdgcn_permlane16_swap(lo[1], hi[1], false, false);
                            xs[2 * jp] = __builtin_bit_cast(bf16x4, u32x2{s0[0], s1[0]});
                            xs[2 * jp + 1] = __builtin_bit_cast(bf16x4, u32x2{s0[1], s1[1]});
                        }
                        if (NJ & 1) xs[NJ - 1] = __builtin_bit_cast(bf16x4, pre2[(i * NJ + NJ - 1) % NPRE]);
                    }
    #pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        f32x4 v = finish_pre(i, j);
                        bf16x4 o;
    #pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = (__bf16)v[r];
                        od[j] = o;
                        if (f_act) {
    #pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float sg = sigmoid1702(v[r]);
                                o[r] = (__bf16)(v[r] * sg);
                                od[j][r] = (__bf16)(sg + 1.702f * v[r] * sg * (1.0f - sg));
                            }
                        }
                        if (f_res || f_aux) {
                            bf16x4 xx;
                            if (EARLY || LATE_X) xx = xs[j];
                            else {
                                const int gn = wcol0 + j * 16 + g * 4;
                                xx = (mok && gn < eN) ? *reinterpret_cast<const bf16x4*>(pre_src + (int64_t)out_row(gm) * pre_ld + gn) : bf16x4{};
                            }
    #pragma unroll
                            for (int r = 0; r < 4; ++r)      // aux = act'(pre-activation) saved by the forward GEMM's c2
                                o[r] = f_res ? (__bf16)((float)o[r] + (float)xx[r]) : (__bf16)((float)o[r] * (float)xx[r]);
                        }
                        oc[j] = o;
                        if constexpr (CS) {
                            if (f_cs && mok) {
    #pragma unroll
                                for (int r = 0; r < 4; ++r) cs[j][r] += (float)o[r];
                            }
                        }
                    }
    #pragma unroll
                    for (int jp = 0; jp < NJ / 2; ++jp) {
                        if (f_c2) store_pair(c2row, mok, jp, od[2 * jp], od[2 * jp + 1]);
                        store_pair(crow, mok, jp, oc[2 * jp], oc[2 * jp + 1]);
                    }
                    if (NJ & 1) {                                          // BN = 96: the odd block keeps 8-byte stores
                        const int col = wcol0 + (NJ - 1) * 16 + g * 4;
                        if (mok && col < eN) {
                            if (f_c2) *reinterpret_cast<bf16x4*>(c2row + col) = od[NJ - 1];
                            *reinterpret_cast<bf16x4*>(crow + col) = oc[NJ - 1];
                        }
                    }
                }
                if constexpr (CS) {
                    if (f_cs) {       // the wave's 64 rows: 4 row blocks summed in registers, the 16 rows of a block across the DPP row
                        float* const prow = e_cs + (int64_t)((m0 / BM) * 4 + wm) * e_ldcs;
    #pragma unroll
                        for (int j = 0; j < NJ; ++j) {
    #pragma unroll
                            for (int r = 0; r < 4; ++r) cs[j][r] = row16_sum(cs[j][r]);
                            const int col = wcol0 + j * 16 + g * 4;
                            if (li == 0 && col < eN) *reinterpret_cast<f32x4*>(prow + col) = cs[j];
                        }
                    }
                }
            };
            int mode = 5;
            if (eC2 == nullptr && !do_act) {
                if (pre_src == nullptr) mode = pre_rot ? 1 : (e_rot == nullptr ? 0 : 5);
                else mode = has_res ? 3 : 4;
            } else if (eC2 != nullptr && do_act && pre_src == nullptr && e_rot == nullptr) {
                mode = 2;
            }
            switch (mode) {
                case 0: epilogue(std::integral_constant<int, 0>{}); break;
                case 1: epilogue(std::integral_constant<int, 1>{}); break;
                case 2: epilogue(std::integral_constant<int, 2>{}); break;
                case 3: epilogue(std::integral_constant<int, 3>{}); break;
                case 4: epilogue(std::integral_constant<int, 4>{}); break;
                default: epilogue(std::integral_constant<int, 5>{}); break;
            }
        }
        MR_STAMP(3);
#ifdef MR_DIAG_STAMPS
        ++nstamp;
#endif
        cw += G;
        ci = GET_ITEM(cw);
    }
}

template <int BN>
static void launch(const G256Args& ga, dim3 grid, hipStream_t s) {
    dim3 block(512);
    const bool ta = ga.p[0].transA, tb = ga.p[0].transB;
    if (!ta && !tb) hipLaunchKernelGGL((gemm256_kernel<BN, false, false>), grid, block, 0, s, ga);
    else if (!ta && tb) hipLaunchKernelGGL((gemm256_kernel<BN, false, true>), grid, block, 0, s, ga);
    else if (ta && !tb) hipLaunchKernelGGL((gemm256_kernel<BN, true, false>), grid, block, 0, s, ga);
    else hipLaunchKernelGGL((gemm256_kernel<BN, true, true>), grid, block, 0, s, ga);
}

}  // namespace g256

constexpr int64_t NUM_CU = 256;    // MI355X

// Returns true when the problem suits the 256-row kernel (then *splits / tiling are filled in by mr_gemm256_launch).
bool mr_gemm256_eligible(const mr_gemm_args* a) {
    if (a->M < 512 || a->N < 96) return false;
    if (a->residual && a->aux) return false;              // the epilogue prefetches ONE bf16 operand
    if (!a->transA && a->K % 64 != 0) return false;       // K-contiguous operands are fetched in whole 128-B rows
    if (a->transB && a->K % 64 != 0) return false;
    const int64_t a_rows = a->transA ? a->K : a->M, b_rows = a->transB ? a->N : a->K;
    if (a_rows * a->lda * 2 >= (1LL << 31) || b_rows * a->ldb * 2 >= (1LL << 31)) return false;   // 32-bit buffer offsets
    return true;
}

extern "C" int64_t mr_gemm_colsum_rows(int64_t M) { return 4 * ((M + g256::BM - 1) / g256::BM); }
extern "C" int32_t mr_gemm_colsum_supported(const mr_gemm_args* a) {
    return a && mr_gemm256_eligible(a) && a->c_dtype == MR_DT_BF16 && a->aux && !a->residual && !a->c2 && a->act == MR_ACT_NONE &&
           !a->rot_tab && a->out_grp == 0 && a->N % 4 == 0;
}

int mr_gemm256_launch(const mr_gemm_args* a, hipStream_t s, void (*reduce)(const mr_gemm_args*, int64_t, hipStream_t)) {
    const int64_t tm = (a->M + g256::BM - 1) / g256::BM;
    // tile width: the one that wastes the fewest CU-rounds (256 workgroups per round, one per CU)
    int bn = 128;
    static int force_bn = -1, grid_mode = -1, c256_cost = -1, c192_cost = -1;
    if (force_bn < 0) force_bn = mr_env_int("MR_G256_BN", 0);
    if (grid_mode < 0) grid_mode = mr_env_int("MR_G256_GRID", 0);
    if (c256_cost < 0) c256_cost = mr_env_int("MR_G256_C256", 180);
    if (c192_cost < 0) c192_cost = mr_env_int("MR_G256_C192", 150);
    const bool can256 = a->N >= 256, can192 = a->N >= 192;
    {
        const int64_t t128 = tm * ((a->N + 127) / 128), t96 = tm * ((a->N + 95) / 96), t256 = tm * ((a->N + 255) / 256);
        // measured: a 96-wide tile costs ~0.91 of a 128-wide one (the A side and the LDS-DMA issue do not shrink); a
        // 256-wide one ~1.9 (twice the FLOPs for 4/3 of the LDS-DMA bytes, but its 2-stage ring drains behind every k-tile)
        const int64_t c128 = ((t128 + 255) / 256) * 100, c96 = ((t96 + 255) / 256) * 91, c256 = ((t256 + 255) / 256) * c256_cost;
        int64_t best = c128;
        if (c96 < best && !a->rot_tab) { bn = 96; best = c96; }       // the prefetched "rotary" scales assume whole heads per wave
        if (can256 && c256 < best) { bn = 256; best = c256; }
        const int64_t t192 = tm * ((a->N + 191) / 192), c192 = ((t192 + 255) / 256) * c192_cost;
        if (can192 && c192 < best) { bn = 192; best = c192; }
    }
    {
        const int f = g_mr_opt_tile_n ? g_mr_opt_tile_n : force_bn;
        if (f == 96 || f == 128 || (f == 256 && can256) || (f == 192 && can192)) bn = f;
    }
    if (a->rot_tab && bn == 96) bn = 128;
    if (a->colsum && bn == 256 && can192) bn = 192;     // the column-sum accumulators beside a 256-wide tile's 128 accumulators spill (measured: +22 us)
    const int64_t tn = (a->N + bn - 1) / bn;
    const int64_t bk = g256::BK, full_grid = NUM_CU;
    const int64_t nk = (a->K + bk - 1) / bk;
    int64_t splits = 1;
    const bool plain = !a->rot_tab && !a->c2 && a->act == MR_ACT_NONE && !a->residual && !a->aux && a->out_grp == 0;
    if (a->workspace && plain && a->N % 4 == 0 && tm * tn < 192 && nk >= 16) {
        splits = (512 + tm * tn - 1) / (tm * tn);
        if (splits > nk / 6) splits = nk / 6;
        if (splits > 64) splits = 64;
        const int64_t fit = a->workspace_bytes / (a->M * a->N * (int64_t)sizeof(float));
        if (splits > fit) splits = fit;
        if (splits < 2) splits = 1;
    }
    int64_t kps = (nk + splits - 1) / splits;
    splits = (nk + kps - 1) / kps;
    const int64_t nwork = tm * tn * splits;
    int64_t gsz = nwork < full_grid ? nwork : full_grid;         // persistent: one workgroup per CU (two with 4 waves)
    if (grid_mode == 1) gsz = nwork;                             // (experiment) one item per workgroup
    else if (grid_mode > 1) gsz = nwork < grid_mode ? nwork : grid_mode;
    dim3 grid((unsigned)gsz);
    g256::G256Args ga;
    memset(&ga, 0, sizeof(ga));
    ga.count = 1; ga.nwork = (int)nwork; ga.splits = (int)splits; ga.kt_per_split = (int)kps;
    static int xmode_env = -1;
    if (xmode_env < 0) xmode_env = mr_env_int("MR_G256_XMODE", 1);
    if (xmode_env && splits == 1 && gsz == full_grid && nwork >= 2 * full_grid) {
        // choose the XCD partition: fewest rounds first (the slowest XCD sets the time), then least traffic out of L2
        const double a_bytes = 2.0 * a->M * a->K, b_bytes = 2.0 * a->N * a->K;
        double best = 1e300;
        for (int px = 1; px <= 8; px *= 2) {
            const int py = 8 / px;
            if (px > tm || py > tn) continue;
            if (mr_opts().xpx > 0 && px != mr_opts().xpx) continue;
            int64_t rounds = 0;
            for (int xi = 0; xi < px; ++xi)
                for (int xj = 0; xj < py; ++xj) {
                    const int64_t hm = (xi + 1) * tm / px - xi * tm / px, hn = (xj + 1) * tn / py - xj * tn / py;
                    const int64_t r = (hm * hn + full_grid / 8 - 1) / (full_grid / 8);
                    if (r > rounds) rounds = r;
                }
            const bool b_fits = b_bytes / py < 2.5e6;         // a cell's B slice stays in the XCD's 4-MiB L2
            const double traffic = a_bytes * py + b_bytes * px * (b_fits ? 1.0 : (double)rounds);
            const double cost = (double)rounds * 1e12 + traffic;
            if (cost < best) { best = cost; ga.px = px; ga.py = py; }
        }
        if (best < 1e300) {
            ga.xmode = 1; ga.tm = (int)tm; ga.tn = (int)tn; ga.xpanel = mr_opts().xpanel;
        }
    }
    ga.tiles_n[0] = (int)tn;
    ga.tile_start[0] = 0;
    for (int k = 1; k <= g256::MAXG; ++k) ga.tile_start[k] = 0x7fffffff;
    ga.p[0] = *a;
    if (bn == 256) g256::launch<256>(ga, grid, s);
    else if (bn == 192) g256::launch<192>(ga, grid, s);
    else if (bn == 128) g256::launch<128>(ga, grid, s);
    else g256::launch<96>(ga, grid, s);
    if (splits > 1) reduce(a, splits, s);
    mr_note_route("g256::gemm256_kernel<%d,%d,%d>%s", bn, (int)a->transA, (int)a->transB, splits > 1 ? " + splitk_reduce" : "");
    return 0;
}

// Grouped launch: count <= 4 problems with identical transA/transB, each eligible for the 256-row kernel, epilogue
// limited to bias; 128-wide tiles, no split-K.  Returns false if the group does not qualify (caller falls back).
bool mr_gemm256_grouped(const mr_gemm_args* list, int count, hipStream_t s) {
    if (count < 1 || count > g256::MAXG) return false;
    for (int k = 0; k < count; ++k) {
        const mr_gemm_args* a = &list[k];
        if (!mr_gemm256_eligible(a) || a->transA != list[0].transA || a->transB != list[0].transB) return false;
        if (a->rot_tab || a->c2 || a->act != MR_ACT_NONE || a->residual || a->aux || a->out_grp != 0) return false;
    }
    // tile width for the whole group: fewest CU-rounds, a 256-wide tile costing ~1.8 of a 128-wide one (the large model's
    // four weight gradients are 384 tiles of 128 = 1.5 rounds -> 2, but 192 tiles of 256 = one round)
    static int c256_cost = -1;
    if (c256_cost < 0) c256_cost = mr_env_int("MR_G256_GROUP_C256", 180);
    int64_t t128 = 0, t256 = 0;
    bool can256 = true;
    for (int k = 0; k < count; ++k) {
        const int64_t tm = (list[k].M + g256::BM - 1) / g256::BM;
        t128 += tm * ((list[k].N + 127) / 128);
        t256 += tm * ((list[k].N + 255) / 256);
        can256 = can256 && list[k].N % 256 == 0;
    }
    int bn = (can256 && ((t256 + NUM_CU - 1) / NUM_CU) * c256_cost < ((t128 + NUM_CU - 1) / NUM_CU) * 100) ? 256 : 128;
    if (g_mr_opt_group_tile_n == 128 || (g_mr_opt_group_tile_n == 256 && can256)) bn = g_mr_opt_group_tile_n;       // a forced width (tests, A/B)
    g256::G256Args ga;
    memset(&ga, 0, sizeof(ga));
    int64_t tiles = 0;
    for (int k = 0; k <= g256::MAXG; ++k) ga.tile_start[k] = 0x7fffffff;
    int64_t nk_max = 0;
    for (int k = 0; k < count; ++k) {
        const mr_gemm_args* a = &list[k];
        const int64_t tm = (a->M + g256::BM - 1) / g256::BM, tn = (a->N + bn - 1) / bn;
        ga.tiles_n[k] = (int)tn;
        ga.tile_start[k] = (int)tiles;
        tiles += tm * tn;
        ga.p[k] = *a;
        const int64_t nk = (a->K + g256::BK - 1) / g256::BK;
        if (nk > nk_max) nk_max = nk;
    }
    ga.count = count; ga.nwork = (int)tiles; ga.splits = 1; ga.kt_per_split = (int)nk_max;
    dim3 grid((unsigned)(tiles < NUM_CU ? tiles : NUM_CU));
    if (bn == 256) g256::launch<256>(ga, grid, s);
    else g256::launch<128>(ga, grid, s);
    mr_note_route("g256::gemm256_kernel<%d,%d,%d> grouped x%d", bn, (int)list[0].transA, (int)list[0].transB, count);
    return true;
}
