// Fifth-generation bf16 MFMA GEMM for gfx950, "NT" operands:  C[M,N] = A[M,K] . B[N,K]^T, both K-contiguous.  Same interface, epilogue
// text (gemm3_epilogue.inc) and tile-list decode as the ping-pong (gemm3.hip) and one-wave-per-SIMD (gemm4.hip) kernels; what
// changes is who shares a CU:
//
// TWO independent workgroups per CU.  A workgroup is FOUR waves (one per SIMD) as 2 (M) x 2 (N) on a 256 x 128 tile: a wave owns
// 128 x 64 = 8 x 4 accumulators of v_mfma_f32_16x16x32_bf16 -- the ping-pong kernel's wave tile -- and at most 256 registers, so
// two workgroups are resident per CU and every SIMD hosts one wave of each.  They share nothing but the CU: each has its own
// barrier, its own 72-KiB LDS ring and its own tile list, so one workgroup's epilogue -- 8 k cycles of GELU + gelu' on the vector
// ALU, the bf16 conversions, the ~3.5 k-cycle drain of its stores (vmcnt retires in order) -- runs UNDER the other's k-loop instead
// of idling the matrix pipe of the whole CU, which is what bounds the K = 768 problems in the one-workgroup-per-CU kernels (a tile's
// fixed cost there is 6.6-9.2 us of a 23-26 us tile: DESIGN.md section 3).  Persistent workgroups with equal tile lists would run in
// lockstep (both in their k-loops, then both in their epilogues), and the phase between two residents is neutrally stable, so it is
// SET at the start: the workgroup in the second wave slot of its SIMDs sleeps for about one k-loop first (a speed knob only --
// nothing depends on which workgroups share a CU).
//
// k-loop: BK = 32 steps (64-byte LDS rows), a 3-stage ring of (256 + 128) x 32 operand images = 24 KiB per stage, filled by LDS-DMA
// three steps ahead: step t multiplies the fragments of stage t from registers (read during step t-1), reads the fragments of stage
// t+1 from LDS and requests stage t+3 into the buffer stage t was read from; it ends with
//     s_waitcnt vmcnt(6) (own pieces of stage t+2 landed: all but the 6 just requested) . lgkmcnt(0) (own reads of stage t+1 done) . s_barrier
// behind which stage t+2 is visible to every wave and stage t+1's buffer is free for the requests of step t+1: safe by construction.
// Per step and wave: 32 MFMAs, 12 ds_read_b128, 6 LDS-DMA requests (4 A + 2 B pieces of 1 KiB), each behind a pair of MFMAs in
// source order pinned by scheduling barriers (as in gemm4.hip).  The requests are unconditional (a cursor without an item requests
// from an empty descriptor: zero fill into a stage nobody reads), the cursor runs across output tiles, and so do the fragment reads.
// TM = 128 (round 4, option "gemm5" = 3 / the few-tile policy): the same kernel on 128 x 128 tiles -- four waves as 2 x 2 of 64 x 64 (4 x 4
// accumulators, 16 MFMAs per step and wave), an EIGHT-stage ring (128 KiB, one workgroup per CU) -- for problems whose 256-row tile grids
// leave most of the chip empty (M = 2308 / 3072 rows, N = 768 / 1024: the VCR ViT, the span tower): twice the tiles, half the k-loop per tile,
// at 1.33 x the operand bytes per FLOP of the 256 x 128 tile (LDS-bound: 48 KiB through the LDS per 256 MFMA cycles; 550 TF/s at best).
// Measured (gemm3_test g5time, us; ping-pong / one-barrier kernels | this geometry): 2308 x 1024 x 4096 46-52 | 35, x 3072 46 | 30, x 1024 20 | 13;
// 3072 x 768 x 3072 38-48 | 30, x 768 17.5 | 11.7.  (A three-stage ring with three workgroups per CU was 10 % slower on the long-K shapes, a
// six-stage ring on 256 x 128 tiles never faster than two workgroups per CU: both removed.)
// 64-byte rows: a 16-row fragment block is one 1-KiB LDS-DMA piece (lane l -> row l >> 2, 16-byte slot l & 3); slot = chunk ^
// ((0 - (row >> 2)) & 3) on the SOURCE address and on the ds_read_b128 address makes the reads conflict-free over the hardware's
// four 16-lane groups (checked by enumeration: every group covers 16 distinct 16-byte columns of the 256-byte bank row).
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include "gemm256_sched.h"
#include "mr_options.h"

namespace g5 {

using namespace g256;

#ifndef MR_G3_AUX_C
#define MR_G3_AUX_C 0
#endif
#ifndef MR_G3_AUX_C2
#define MR_G3_AUX_C2 0
#endif

constexpr int BN5 = 128;
// NW = 4 waves.  (An eight-wave workgroup on 256 x 256 tiles under this k-loop -- the ping-pong kernel's geometry and operand traffic, BK = 32
// steps, no wave-group phases -- was slower than the ping-pong kernel on every shape: commit 502fd27, removed.)
// NS = ring stages.  3: a request has ONE step to land (it is waited for at the end of the step after the one that issued it) -- fine beside a
// second resident workgroup.  The 128-row geometry (one workgroup per CU: its problems have fewer tiles than the chip has CUs) requests NS = 8
// steps ahead and waits for the stage two steps ahead, leaving six steps for the L2 round trip.
template <int NW, int TM = 256, int NS = 3> struct Geo5 {
    static_assert(NW == 4 && (TM == 256 || TM == 128), "geometries");
    static constexpr int NSTG = NS;
    static constexpr int BN = 128;
    static constexpr int MI = TM / 32;                                             // 16-row blocks per wave (two wave rows)
    static constexpr int STG_A = TM * 64, STG_B = BN * 64, STG = STG_A + STG_B, LDS = NSTG * STG;
    static constexpr int PA = (TM / 16) / NW, PB = (BN / 16) / NW, ND = PA + PB;   // 1-KiB pieces per wave and step: A, B, both
    static constexpr int OCC = NS > 3 ? 1 : TM == 128 ? 3 : 2;                     // workgroups per CU
};
static_assert(Geo5<4>::LDS == 72 * 1024 && 2 * Geo5<4>::LDS <= 160 * 1024, "LDS budgets");
static_assert(Geo5<4, 128, 8>::LDS == 128 * 1024, "LDS budget of the 128 x 128 geometry");

__device__ __forceinline__ int swz5(int row) { return (0 - ((row >> 2) & 3)) & 3; }

template <int MODE, int NW, int TM = 256, int NS = 3>
__global__ __launch_bounds__(NW * 64, (Geo5<NW, TM, NS>::OCC)) void gemm5_kernel(const G256Args ga, const int stagger_mode, const int stagger_units) {
    constexpr int NJ = 4, WCOLS = 64;
    using GEO = Geo5<NW, TM, NS>;
    constexpr int NSTG = NS;
    constexpr int STG = GEO::STG, STG_A = GEO::STG_A, PA = GEO::PA, PB = GEO::PB, ND = GEO::ND, BNT = GEO::BN, MI = GEO::MI;
    // store instructions a wave issues per tile (all unconditional, see gemm3_epilogue.inc): the first step behind an epilogue may
    // leave them -- and its own 6 requests -- outstanding while it waits for the stage requested AHEAD of the epilogue
    constexpr int NST = MI * (NJ / 2) * (MODE == 2 ? 2 : 1);
    constexpr int WAIT_RING = (NSTG - 2) * ND;        // requests that may stay in flight at a step's end: the stages three and more steps ahead
    constexpr int WAIT_FIRST = WAIT_RING + NST;
    static_assert(WAIT_FIRST <= 63, "vmcnt is a 6-bit counter");
    __shared__ __attribute__((aligned(16))) char smem[GEO::LDS];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int g = lane >> 4, li = lane & 15;

    const int G = gridDim.x;
    const int xcd = blockIdx.x & 7, qd = G >> 3, rm = G & 7;
    const int bperm = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (blockIdx.x >> 3);
    int m0v, n0v;
    item_pp(ga, bperm, lane, G, BNT, m0v, n0v, TM);
    auto item_m0 = [&](int q) -> int { const int r = __builtin_amdgcn_readlane(m0v, q & 63); return q < 64 ? r : -1; };
    auto item_n0 = [&](int q) -> int { return __builtin_amdgcn_readlane(n0v, q & 63); };
    if (item_m0(0) < 0) return;
    const mr_gemm_args& p0 = ga.p[0];
    const int nks = (int)(p0.K >> 5);
    const unsigned lda2 = (unsigned)p0.lda * 2u, ldb2 = (unsigned)p0.ldb * 2u;
    const char* const Aptr = static_cast<const char*>(p0.A);
    const char* const Bptr = static_cast<const char*>(p0.B);
    const int a_extent = (int)(((p0.M - 1) * p0.lda + p0.K) * 2), b_extent = (int)(((p0.N - 1) * p0.ldb + p0.K) * 2);
    auto rel = [&](int piece, unsigned ld2) -> unsigned {      // byte offset of this lane's chunk of 1-KiB piece `piece` in an item at row 0
        const int row = piece * 16 + (lane >> 2);
        return (unsigned)row * ld2 + (unsigned)(((lane & 3) ^ swz5(row)) * 16);
    };
    // one per-lane offset per operand: a wave's pieces are 16 rows apart, a wave-uniform stride that rides in the scalar offset
    const unsigned pao = rel(wave * PA, lda2), pbo = rel(wave * PB, ldb2);
    const unsigned pstep_a = 16u * lda2, pstep_b = 16u * ldb2;
    // fragment reads: lane (g, li) of a 16-row block reads k-chunk g of row li
    const int fo = li * 64 + ((g ^ swz5(li)) << 4);
    const int fa = wr * (TM * 32) + fo, fb = STG_A + wc * 4096 + fo;

    // The phase between the two workgroups of a CU (see the header): the one whose waves sit in the odd wave slots waits ~ one k-loop.
    if (stagger_mode != 0) {
        bool late;
        if (stagger_mode == 1) late = (__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 1) != 0;          // HW_REG_HW_ID.WAVE_ID bit 0
        else if (stagger_mode == 2) late = (int)blockIdx.x >= (G >> 1);
        else late = ((blockIdx.x >> 3) & 1) != 0;
        if (late)
            for (int n = 0; n < stagger_units; ++n) __builtin_amdgcn_s_sleep(32);        // 2048 cycles each
    }

    // ---- ONE issue cursor for both operands, three steps ahead of the compute cursor, across item boundaries ----
    int ik = 0, qi = 0, ist = 0;
    __amdgpu_buffer_rsrc_t ra_c, rb_c;
    auto set_item = [&](int q_) {
        const int mi = item_m0(q_), ni = item_n0(q_);
        const int aoff = (mi >= 0 ? mi : 0) * (int)lda2, boff = (mi >= 0 ? ni : 0) * (int)ldb2;
        ra_c = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Aptr) + aoff, 0, mi >= 0 ? a_extent - aoff : 0, 0x00020000);
        rb_c = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Bptr) + boff, 0, mi >= 0 ? b_extent - boff : 0, 0x00020000);
    };
    set_item(0);
#define G5_ISSUE_ALL()                                                                                                  \
    do {                                                                                                                \
        char* st_ = smem + ist * STG;                                                                                   \
        const unsigned so_ = (unsigned)ik * 64u;                                                                        \
        _Pragma("unroll") for (int j_ = 0; j_ < PA; ++j_) MR_DMA(ra_c, MR_LDS_PTR(void, st_ + wave * (PA * 1024) + j_ * 1024), 16, pao, so_ + j_ * pstep_a, 0, 0); \
        _Pragma("unroll") for (int j_ = 0; j_ < PB; ++j_) MR_DMA(rb_c, MR_LDS_PTR(void, st_ + STG_A + wave * (PB * 1024) + j_ * 1024), 16, pbo, so_ + j_ * pstep_b, 0, 0); \
    } while (0)
#define G5_ADVANCE()                                                                                                    \
    do {                                                                                                                \
        ist = (ist == NSTG - 1) ? 0 : ist + 1;                                                                          \
        if (++ik == nks) {                                                                                              \
            ik = 0;                                                                                                     \
            set_item(++qi);                                                                                             \
        }                                                                                                               \
    } while (0)
#define G5_SB() __builtin_amdgcn_sched_barrier(0)
#ifdef MR_G5_STAMPS        /* diagnostic build: s_memtime at the start of a tile's k-loop, at its end and at the end of its epilogue -> workspace */
#define G5_STAMP(slot)                                                                                         \
    do {                                                                                                       \
        unsigned long long t_;                                                                                 \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                             \
        if (tid == 0 && qc < 8) static_cast<unsigned long long*>(p0.workspace)[(blockIdx.x * 8 + qc) * 4 + (slot)] = t_; \
    } while (0)
#else
#define G5_STAMP(slot) do {} while (0)
#endif
#ifdef MR_G5_NOREAD        /* timing-only diagnostic builds (wrong results): scripts/build_g5_variants.sh */
#define G5_RD(dst, ptr) asm volatile("" : "+v"(dst))
#else
#define G5_RD(dst, ptr) (dst) = *reinterpret_cast<const bf16x8*>(ptr)
#endif
#ifdef MR_G5_NOBAR
#define G5_BARRIER() do {} while (0)
#else
#define G5_BARRIER() __builtin_amdgcn_s_barrier()
#endif
    int qc = 0, rs = 0;                 // rs: the stage the NEXT fragments are read from
    int cm0 = item_m0(0), cn0 = item_n0(0);
    bool have_stores = false;
    constexpr bool BIAS = MODE <= 2;
    f32x4 binit[NJ];
    auto fetch_bias = [&](int n0_) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) binit[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (BIAS) {
            const __bf16* const bp = static_cast<const __bf16*>(p0.bias);
            if (bp != nullptr) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int gn = n0_ + wc * WCOLS + j * 16 + g * 4;
                    if (gn < (int)p0.N) {
                        const bf16x4 b4 = *reinterpret_cast<const bf16x4*>(bp + gn);
#pragma unroll
                        for (int r = 0; r < 4; ++r) binit[j][r] = (float)b4[r];
                    }
                }
            }
        }
    };
    fetch_bias(cn0);

    // prologue: the first NSTG stages requested; stage 0 has landed when only the later ones are in flight
#pragma unroll
    for (int s_ = 0; s_ < NSTG; ++s_) { G5_ISSUE_ALL(); G5_ADVANCE(); }
    wait_vmcnt<(NSTG - 1) * ND>();
    G5_SB();
    __builtin_amdgcn_s_barrier();
    G5_SB();
    // Fragment registers: the B fragments and the last two A row blocks are double buffered (parity = step & 1: K % 64 == 0, so a
    // tile's first step has parity 0); A row blocks 0-5 are replaced IN PLACE behind their own MFMAs (72 registers instead of 96:
    // beside the 128 accumulators the rotary / GELU / column-sum epilogues spill with 96).  The in-place read of row block 5 is the
    // step's last: it has the 8 MFMAs of row blocks 6 and 7 to land before the step's closing wait.
    // (TM = 128: four row blocks -- 0, 1 in place, 2, 3 double buffered; "a67" = the last two row blocks of either geometry)
    constexpr int MIP = MI - 2;                       // row blocks replaced in place
    bf16x8 a[MIP], a67[2][2], b[NJ][2];
    auto read_frags0 = [&](const char* st) {          // all fragments of a stage into parity 0 (prologue, tile switch)
#pragma unroll
        for (int j = 0; j < NJ; ++j) b[j][0] = *reinterpret_cast<const bf16x8*>(st + fb + j * 1024);
#pragma unroll
        for (int i = 0; i < MIP; ++i) a[i] = *reinterpret_cast<const bf16x8*>(st + fa + i * 1024);
        a67[0][0] = *reinterpret_cast<const bf16x8*>(st + fa + MIP * 1024);
        a67[1][0] = *reinterpret_cast<const bf16x8*>(st + fa + (MIP + 1) * 1024);
    };
    read_frags0(smem);
    wait_vmcnt<WAIT_RING>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    G5_SB();
    __builtin_amdgcn_s_barrier();
    G5_SB();
    rs = 1;

    while (cm0 >= 0) {
        // accumulators TRANSPOSED (mfma(B-frag, A-frag)): the lane holds C[m = .. + li][n = .. + 4 g + r]; bias modes start from the bias
        f32x4 acc[MI][NJ];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = binit[j];

        // one k-step: multiply the fragments of parity CUR, read the next step's into parity 1 - CUR, request three steps ahead
        auto step = [&](auto cur_c, auto first_c, auto last_c) {
            constexpr int CUR = decltype(cur_c)::value, NXT = 1 - CUR;
            constexpr bool FIRST = decltype(first_c)::value;        // the tile's first step: the previous tile's stores may be in flight
            constexpr bool LAST = decltype(last_c)::value;          // the tile's last step reads no fragments: the next tile's first ones are read behind the epilogue (registers)
            const char* const rd = smem + rs * STG;
            char* const st_ = smem + ist * STG;
            const unsigned so_ = (unsigned)ik * 64u;
#pragma unroll
            for (int r = 0; r < MI; ++r) {
                const bf16x8 ar = r < MIP ? a[r < MIP ? r : 0] : a67[r - MIP < 0 ? 0 : r - MIP][CUR];
                acc[r][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[0][CUR], ar, acc[r][0], 0, 0, 0);
                acc[r][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[1][CUR], ar, acc[r][1], 0, 0, 0);
                G5_SB();
                if constexpr (!LAST) {               // the double-buffered fragments first: b0..b3, then (eight row blocks) row blocks 6, 7
                    if (r < 4) G5_RD(b[r][NXT], rd + fb + r * 1024);
                    else if (r < 6) G5_RD(a67[r - 4][NXT], rd + fa + (r + 2) * 1024);
                }
                if (r < PA) MR_DMA(ra_c, MR_LDS_PTR(void, st_ + wave * (PA * 1024) + r * 1024), 16, pao, so_ + r * pstep_a, 0, 0);
                else if (r < ND) MR_DMA(rb_c, MR_LDS_PTR(void, st_ + STG_A + wave * (PB * 1024) + (r - PA) * 1024), 16, pbo, so_ + (r - PA) * pstep_b, 0, 0);
                G5_SB();
                acc[r][2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[2][CUR], ar, acc[r][2], 0, 0, 0);
                acc[r][3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[3][CUR], ar, acc[r][3], 0, 0, 0);
                G5_SB();
                if constexpr (!LAST) {
                    if (r < MIP) G5_RD(a[r < MIP ? r : 0], rd + fa + r * 1024);       // in place: this row block's MFMAs are issued
                    else if (MI == 4) G5_RD(a67[(r - MIP) & 1][NXT], rd + fa + r * 1024);   // four row blocks: the double-buffered pair rides in these slots
                }
                G5_SB();
            }
            if (FIRST && have_stores) wait_vmcnt<WAIT_FIRST>();
            else wait_vmcnt<WAIT_RING>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            G5_SB();
            G5_BARRIER();
            G5_SB();
            if constexpr (!LAST) rs = (rs == NSTG - 1) ? 0 : rs + 1;
            G5_ADVANCE();
        };
        G5_STAMP(0);
        step(std::integral_constant<int, 0>{}, std::true_type{}, std::false_type{});
        for (int t = 1; t + 2 < nks; t += 2) {
            step(std::integral_constant<int, 1>{}, std::false_type{}, std::false_type{});
            step(std::integral_constant<int, 0>{}, std::false_type{}, std::false_type{});
        }
        step(std::integral_constant<int, 1>{}, std::false_type{}, std::true_type{});          // (K % 64 == 0: an even number of steps >= 4)

        G5_STAMP(1);
        // ---------------- epilogue (bf16 output; registers + ordinary loads, no LDS) ----------------
        {
            const int wrow0 = cm0 + wr * (TM / 2), wcol0 = cn0 + wc * WCOLS;
            constexpr int MR_EPI_MI = MI;
#define MR_EPI_ROW_FENCE() do {} while (0)
#define MR_EPI_FULL_LINES 0      // (full-line stores: the trade of halves costs this kernel spilled registers at its allocation limit; gemm3.hip only)
#include "gemm3_epilogue.inc"
#undef MR_EPI_FULL_LINES
#undef MR_EPI_ROW_FENCE
#ifndef MR_G3_NOSTORE
            have_stores = true;
#endif
        }
        G5_STAMP(2);
        ++qc;
        cm0 = item_m0(qc);
        cn0 = item_n0(qc);
        // the next tile's first fragments (stage `rs`: landed and visible since the barrier of the last step but one); the barrier keeps
        // the requests of the next step, which refill this stage's buffer, behind every wave's reads
        if (cm0 >= 0) {
            read_frags0(smem + rs * STG);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            G5_SB();
            __builtin_amdgcn_s_barrier();
            G5_SB();
            rs = (rs == NSTG - 1) ? 0 : rs + 1;
        }
    }
    // the ring's last requests (zero fill) and the stores retire before the LDS is released
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

}  // namespace g5


// Same operand / epilogue contract as the ping-pong kernel; mr_gemm asks this kernel FIRST (mr_gemm5_wanted -> mr_gemm5_takes checks the operand conditions itself).
bool mr_gemm5_takes(const mr_gemm_args* a) {
    if (a->transA || !a->transB || a->c_dtype != MR_DT_BF16 || a->K % 64 != 0 || a->K < 128) return false;
    if (a->M * a->lda * 2 >= (1LL << 31) || a->N * a->ldb * 2 >= (1LL << 31)) return false;
    {
        const int64_t last = a->out_grp > 0 ? ((a->M - 1) / a->out_grp) * a->out_grp_stride + a->out_grp_off + (a->M - 1) % a->out_grp : a->M - 1;
        if ((last * a->ldc + a->N) * 2 >= (1LL << 31)) return false;
    }
    const int n_extra = (a->rot_tab != nullptr) + (a->residual != nullptr) + (a->aux != nullptr) + (a->c2 != nullptr);
    if (n_extra > 1) return false;
    if ((a->c2 != nullptr) != (a->act == MR_ACT_GELU1702)) return false;
    if (a->colsum && !a->aux) return false;
    if (a->bias && (a->residual || a->aux)) return false;
    const int64_t ncu = (g_mr_opt_gemm_cus >= 64 && g_mr_opt_gemm_cus < 256) ? (g_mr_opt_gemm_cus & ~7) : 256;
    if (((a->M + 127) / 128) * ((a->N + 127) / 128) > 64 * 2 * ncu) return false;          // <= 64 items per workgroup (either tile height)
    return true;
}

// Tile height of a problem: 128 x 128 tiles (deep ring, one workgroup per CU) where even the 128-row grid has no more tiles than the chip has
// CUs, or on request (option "gemm5" = 3); 256 x 128 (two workgroups per CU) otherwise.
static bool g5_rows128(const mr_gemm_args* a) {
    const int v = mr_opts().gemm5;
    if (v == 3) return true;
    if (v == 1) return false;
    static int env = -2;
    if (env == -2) env = mr_env_int("MR_G5_ROWS128", -1);
    if (env >= 0) return env != 0;
    return ((a->M + 127) / 128) * ((a->N + 127) / 128) <= 256;
}

int mr_gemm5_launch(const mr_gemm_args* a, hipStream_t s) {
    const int64_t ncu = (g_mr_opt_gemm_cus >= 64 && g_mr_opt_gemm_cus < 256) ? (g_mr_opt_gemm_cus & ~7) : 256;
    const bool r128 = g5_rows128(a);                    // 128 x 128 tiles, deep ring, one workgroup per CU
    const int64_t slots = r128 ? ncu : 2 * ncu;         // workgroups per CU
    const int64_t bn = g5::BN5;
    const int64_t tm = (a->M + (r128 ? 127 : 255)) / (r128 ? 128 : 256), tn = (a->N + bn - 1) / bn, nwork = tm * tn;
    const int64_t gsz = nwork < slots ? nwork : slots;
    g256::G256Args ga;
    memset(&ga, 0, sizeof(ga));
    ga.count = 1; ga.nwork = (int)nwork; ga.splits = 1; ga.kt_per_split = (int)(a->K / 32);
    if (gsz == slots && nwork >= 2 * slots) {
        // XCD partition of the tile grid (see G256Args.xmode): fewest rounds first, then least traffic out of L2
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
                    const int64_t r = (hm * hn + slots / 8 - 1) / (slots / 8);
                    if (r > rounds) rounds = r;
                }
            const bool b_fits = b_bytes / py < 2.5e6;
            const double traffic = a_bytes * py + b_bytes * px * (b_fits ? 1.0 : (double)rounds);
            const double cost = (double)rounds * 1e12 + traffic;
            if (cost < best) { best = cost; ga.px = px; ga.py = py; }
        }
        if (best < 1e300) { ga.xmode = 1; ga.tm = (int)tm; ga.tn = (int)tn; ga.xpanel = mr_opts().xpanel; }
    }
    ga.tiles_n[0] = (int)tn;
    ga.tile_start[0] = 0;
    for (int k = 1; k <= g256::MAXG; ++k) ga.tile_start[k] = 0x7fffffff;
    ga.p[0] = *a;
    int mode = 5;
    if (a->c2) mode = 2;
    else if (a->rot_tab) mode = 1;
    else if (a->residual) mode = 3;
    else if (a->aux) mode = 4;
    else if (a->bias) mode = 0;
    static int st_env = -2, su_env = -2;
    if (st_env == -2) st_env = mr_env_int("MR_G5_STAGGER", -1);
    if (su_env == -2) su_env = mr_env_int("MR_G5_STAGGER_PCT", -1);
    int st_mode = g_mr_opt_gemm5_stagger >= 0 ? g_mr_opt_gemm5_stagger : st_env >= 0 ? st_env : 1;
    if (gsz <= ncu || r128) st_mode = 0;         // one workgroup per CU: nobody to be out of phase with (three per CU: no phase is set)
    // ~ one k-loop alone on the CU: K / 32 steps of 512 MFMA cycles, in units of 2048 cycles (percent knob for experiments)
    const int pct = su_env >= 0 ? su_env : 100;
    const int st_units = (int)((a->K / 32) * 600 * pct / 100 / 2048);
    dim3 grid((unsigned)gsz), block(256);
    {
        static int dbg = -1;
        if (dbg < 0) dbg = mr_env_int("MR_G5_DEBUG", 0);
        if (dbg == 1) {
            dbg = 2;
            int nb = -1;
            hipError_t e_ = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, g5::gemm5_kernel<5, 4>, 256, 0);
            hipFuncAttributes fa_;
            hipError_t e2_ = hipFuncGetAttributes(&fa_, reinterpret_cast<const void*>(g5::gemm5_kernel<5, 4>));
            fprintf(stderr, "[gemm5] occupancy query: %d blocks / CU (err %d); regs %d, static LDS %zu, scratch %zu (err %d); grid %ld, stagger mode %d units %d\n", nb, (int)e_,
                    fa_.numRegs, fa_.sharedSizeBytes, fa_.localSizeBytes, (int)e2_, (long)gsz, st_mode, st_units);
        }
    }
#define G5_LAUNCH(MODE)                                                                                       \
    do {                                                                                                      \
        if (r128) hipLaunchKernelGGL((g5::gemm5_kernel<MODE, 4, 128, 8>), grid, block, 0, s, ga, st_mode, st_units); \
        else hipLaunchKernelGGL((g5::gemm5_kernel<MODE, 4>), grid, block, 0, s, ga, st_mode, st_units);       \
    } while (0)
    switch (mode) {
        case 0: G5_LAUNCH(0); break;
        case 1: G5_LAUNCH(1); break;
        case 2: G5_LAUNCH(2); break;
        case 3: G5_LAUNCH(3); break;
        case 4: G5_LAUNCH(4); break;
        default: G5_LAUNCH(5); break;
    }
#undef G5_LAUNCH
    mr_note_route(r128 ? "g5::gemm5_kernel<%d,4,128,8>" : "g5::gemm5_kernel<%d,4>", mode);
    MR_CHECK_LAUNCH("mr_gemm (gemm5)");
    return MR_OK;
}

// Which problems go to this kernel: mr_set_option("gemm5") 1 = every one it can take (tests, A/B), 0 = none, -1 = the policy below.
bool mr_gemm5_wanted(const mr_gemm_args* a) {
    static int env = -2;
    if (env == -2) env = mr_env_int("MR_GEMM5", -1);
    const int v = g_mr_opt_gemm5 >= 0 ? g_mr_opt_gemm5 : env;
    if (v == 0 || a->colsum != nullptr && !a->aux) return false;
    if (!mr_gemm5_takes(a)) return false;
    if (v == 1 || v == 3) return true;
    // Default policy (measured, scripts/micro/gemm3_test g5time): the few-tile short-K problems -- at most one 256 x 128 tile per CU, K <= 1024:
    // the audio / span towers' 768-wide projections, the span and VCR-ViT QKV -- run 8-13 % faster here than on the one-barrier kernel's
    // 96-wide tiles (16.3 vs 18.0, 20.6 vs 23.4, 15.3 vs 17.4, 24.5 vs 26.9 us); everything with more tiles or a longer K is slower
    // (the 256 x 128 tile pair of a CU writes 1.5 x the operand bytes into LDS: DESIGN.md section 3).
    // Round 4, 128 x 128 tiles: problems with at most one such tile per CU (M = 2308 / 3072, N = 768 / 1024, any K) run 25-40 % faster there than
    // on either of the above (the header's table); with more tiles the 256-row geometries win (audio tower, M = 5952: 36-38 vs 52 us at K = 3072).
    if (a->colsum != nullptr || a->M < 1024 || a->N < 256) return false;
    if (g5_rows128(a)) return true;
    const int64_t tiles = ((a->M + 255) / 256) * ((a->N + 127) / 128);
    return tiles <= 256 && a->K <= 1024;
}
