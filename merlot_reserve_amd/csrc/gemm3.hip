// Third-generation bf16 MFMA GEMM for gfx950, "NT" operands only:  C[M,N] = A[M,K] . B[N,K]^T, both K-contiguous
// (activations x dgrad weights as flax stores them; forward weights through their transposed bf16 working copy).
//
// 256 x {256,192} x 64 tiles, 8 waves as 2 (M) x 4 (N): a wave owns 128 x {64,48} of the tile = 8 x {4,3} accumulators of
// v_mfma_f32_16x16x32_bf16, computed per k-tile in FOUR quadrants of 64 x {32,32|16} (16 | 8 MFMAs each).
//
// Two wave groups, PING-PONG.  Waves 0-3 and 4-7 are the two waves of each SIMD.  Every quadrant is a phase
//     [ LDS fragment reads of the phase | LDS-DMA issue ]  s_barrier  [ MFMAs ]  s_barrier
// and group 1 runs one barrier behind group 0, so that between any two barriers one wave of a SIMD issues MFMAs while its
// partner issues LDS reads and DMA: the matrix pipe never waits for fragment reads of its own wave, which is what bounds
// the one-barrier-per-k-tile kernel of gemm256.hip (k-loop decomposition).
//
// Schedules (template parameter PH, chosen per problem by mr_gemm3_launch):
// PH = 1, the default: ONE phase per k-tile and wave.  Memory section: B (8 | 6 ds_read_b128) + A0 (8) + 8 | 6 DMA pieces; MFMA
// section: A0 x B row block by row block, each block's A1 fragments requested into the registers its last A0 MFMA has just read
// (fragment registers stay at 64), then A1 x B: 64 | 48 MFMAs between two barriers.  The groups split the operands: group 0 issues
// every B piece (k-tile t+1, waited for at the END of its MFMA section: both groups read it after the closing barrier), group 1
// every A piece (k-tile t+2, counted wait in its next memory section).
// PH = 2: per k-tile and wave TWO phases: phase 0 reads B (8 | 6 ds_read_b128) + A0 (8) and multiplies A0 x B (32 | 24 MFMAs),
// phase 1 reads A1 (8) and multiplies A1 x B; A0 / A1 = the wave's two 64-row halves.  (-DMR_G3_PH4: four phases of 16 | 8
// MFMAs, quadrants (A0,B0) (A0,B1) (A1,B1) (A1,B0): twice the barriers per MFMA.)
// Operand rings, filled by LDS-DMA (buffer_load ... lds, 1 KiB per wave instruction, zero fill beyond the extents):
//     A: 3 stages x 32 KiB, two k-tiles ahead (issued in phase 0);
//     B: 2 stages x {32,24} KiB: k-tile t+2 overwrites k-tile t's stage in phase 1, after the B reads of k-tile t
//        (phase 0) have been retired by a counted lgkmcnt AHEAD of that phase's barrier.  Every LDS read that a later
//        DMA overwrites is retired ahead of a barrier the issuing wave passes before it issues: safe by construction.
//   = 160 KiB (BN = 256) | 144 KiB (BN = 192).  One counted s_waitcnt vmcnt per k-tile (never 0 inside a tile's loop), in
// phase 3 ahead of the barrier that precedes the first read of k-tile t+1.  The issue cursors run across output tiles
// (persistent workgroups), so a tile's first two k-tiles land under the previous tile's epilogue.
// The swizzle (chunk ^= (row >> 1) & 7) lives on the DMA's per-lane SOURCE address and is undone by the fragment reads.
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include "gemm256_sched.h"
#include "mr_options.h"

namespace g3 {

using namespace g256;

template <int BN> struct Geo3 {
    static_assert(BN == 256 || BN == 192, "tile widths of the ping-pong kernel");
    static constexpr int WCOLS = BN / 4, NJ = WCOLS / 16, JB0 = 2, JB1 = NJ - 2;
    static constexpr int STAGE_A = 256 * 128, NSTAGE_A = 3;
    static constexpr int STAGE_B = BN * 128, NSTAGE_B = 2;
    static constexpr int OFF_B = NSTAGE_A * STAGE_A;
    static constexpr int LDS_BYTES = OFF_B + NSTAGE_B * STAGE_B;
    static constexpr int NBP_HI = (BN == 256) ? 2 : 1;      // B pieces per wave for rows 128.. of the B tile
    static constexpr int WAITN = 4 + 2 + NBP_HI;            // DMA pieces a wave issues per k-tile
};
static_assert(Geo3<256>::LDS_BYTES == 160 * 1024 && Geo3<192>::LDS_BYTES == 144 * 1024, "LDS budget");

#ifdef MR_G3_NOSYNC_EPI
constexpr bool EPI_SYNC = false;
#else
constexpr bool EPI_SYNC = true;
#endif
// cache policy bits of the epilogue stores (raw_buffer_store aux: 1 = sc0, 2 = nt, 16 = sc1): experiment knobs, default 0
#ifndef MR_G3_AUX_C
#define MR_G3_AUX_C 0
#endif
#ifndef MR_G3_AUX_C2
#define MR_G3_AUX_C2 0
#endif
#ifndef MR_G3_NA0
#define MR_G3_NA0 4
#endif
constexpr int NA0 = MR_G3_NA0;        // A pieces (of a wave's four per k-tile) issued in phase 0
#ifdef MR_G3_PH4
constexpr bool PH4 = true;
#else
constexpr bool PH4 = false;
#endif

// MODE (the epilogue, fixed per launch: one problem per launch):
//   0: bias | 1: bias + "rotary" scales | 2: bias + GELU, gelu' copy to c2 | 3: + residual | 4: x aux (+ column sums)
// PH = phases per k-tile.  2: [B + A0 reads | 4 A pieces] [32 MFMAs] / [A1 reads | 4 B pieces] [32 MFMAs], both groups issue both operands.
// 1: ONE phase of 64 | 48 MFMAs -- half the barriers per MFMA: the memory section reads B + A0 and issues 8 pieces, the A1 fragments
// replace the A0 fragments in place INSIDE the MFMA section (two ds_read_b128 behind each row block's last A0 MFMA, 24+ MFMAs ahead
// of their use), and the groups split the operands: group 0 issues every B piece (one k-tile ahead), group 1 every A piece (two ahead).
template <int BN, int MODE, int PH>
__global__ __launch_bounds__(512, 2) void gemm3_kernel(const G256Args ga) {
    using GEO = Geo3<BN>;
    constexpr int WCOLS = GEO::WCOLS, NJ = GEO::NJ, JB1 = GEO::JB1;
    constexpr int STAGE_A = GEO::STAGE_A, STAGE_B = GEO::STAGE_B, OFF_B = GEO::OFF_B;
    constexpr int NSTAGE_A = GEO::NSTAGE_A, NSTAGE_B = GEO::NSTAGE_B, NBP_HI = GEO::NBP_HI, WAITN = GEO::WAITN;
    // store instructions a wave issues per tile, ALL unconditional (masked lanes store to an out-of-range buffer offset, which the
    // hardware drops): vmcnt retires in order, so the first k-tile behind an epilogue may leave exactly these -- and its own DMA
    // pieces -- in flight while waiting for the k-tile issued AHEAD of the epilogue; the stores then drain under two k-tiles of
    // MFMAs instead of stalling the wave at its first counted wait.  (The column-sum stores of MODE 4 are not counted: fewer
    // outstanding operations than allowed only makes the wait conservative.)
    constexpr int NST = 8 * (NJ / 2 + (NJ & 1)) * (MODE == 2 ? 2 : 1);
    static_assert(WAITN + NST <= 63, "vmcnt is a 6-bit counter");
    __shared__ __attribute__((aligned(16))) char smem[GEO::LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;             // wr = the wave GROUP (and the tile's row half), wc = column quarter
    const int g = lane >> 4, li = lane & 15;

    const int G = gridDim.x;
    const int xcd = blockIdx.x & 7, qd = G >> 3, rm = G & 7;
    const int bperm = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (blockIdx.x >> 3);
    // The workgroup's items (output tiles), decoded ONCE: lane q holds item q (m0 < 0 = none), fetched with v_readlane.  Decoding
    // an item at every cursor switch -- integer divisions and kernel-argument loads inside a phase that all eight waves wait
    // for -- cost ~4 000 cycles per tile (in-kernel stamps).  <= 64 items per workgroup (host check).
    int m0v, n0v;
    item_pp(ga, bperm, lane, G, BN, m0v, n0v);
    auto item_m0 = [&](int q) -> int { return q < 64 ? __builtin_amdgcn_readlane(m0v, q) : -1; };
    auto item_n0 = [&](int q) -> int { return __builtin_amdgcn_readlane(n0v, q & 63); };
    if (item_m0(0) < 0) return;
#ifdef MR_G3_STAGGER          // experiment (scripts/build_g3_variants.sh): odd workgroups start late, so that their epilogues' store bursts fall into the others' k-loops
    if (blockIdx.x & 8) for (int i = 0; i < MR_G3_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
#endif
    const mr_gemm_args& p0 = ga.p[0];
    const int nkt = (int)(p0.K >> 6);
    const unsigned lda2 = (unsigned)p0.lda * 2u, ldb2 = (unsigned)p0.ldb * 2u;
    // operand descriptors: rows >= M (A) / >= N (B) lie beyond num_records and read as zeros -- the per-lane offsets carry the
    // item's first row, so no predicate per piece: offset(item) = offset(row 0 item) + m0 * lda2
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p0.A), 0, (int)(((p0.M - 1) * p0.lda + p0.K) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p0.B), 0, (int)(((p0.N - 1) * p0.ldb + p0.K) * 2), 0x00020000);
    auto rel = [&](int piece, unsigned ld2) -> unsigned {      // byte offset of this lane's chunk of 1-KiB piece `piece` in an item at row 0
        const int row = piece * 8 + (lane >> 3);
        return (unsigned)row * ld2 + (unsigned)(((lane & 7) ^ swz_kc(row)) * 16);
    };
    const unsigned ar0 = rel(wave * 2, lda2), ar1 = rel(wave * 2 + 1, lda2), ar2 = rel(16 + wave * 2, lda2), ar3 = rel(16 + wave * 2 + 1, lda2);
    const unsigned br0 = rel(wave * 2, ldb2), br1 = rel(wave * 2 + 1, ldb2);
    const unsigned br2 = rel(NBP_HI == 2 ? 16 + wave * 2 : 16 + wave, ldb2), br3 = rel(16 + wave * 2 + 1, ldb2);

    // PH == 1: the wave's 8 (6: B of a 192-wide tile) pieces of ITS operand -- group 0: B, group 1: A
    constexpr int PB1 = (BN == 256) ? 8 : 6;
    unsigned po[8];
    if constexpr (PH == 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) po[j] = (wr == 0) ? rel((wave & 3) * PB1 + (j < PB1 ? j : 0), ldb2) : rel((wave & 3) * 8 + j, lda2);
    }
    // ---- issue cursors (one per operand), two k-tiles ahead of the compute cursor, across item boundaries ----
    int qa = 0, ika = 0, ista = 0, qb = 0, ikb = 0, istb = 0;
    bool va = true, vb = true, issued = false;
    unsigned abase = (unsigned)item_m0(0) * lda2, bbase = (unsigned)item_n0(0) * ldb2;
    // the issue slots of a k-tile: A rows 0-127, A rows 128-255 (+ advance), B rows 0-127, B rows 128.. (+ advance)
#define G3_ISSUE_A_PIECE(ar_, off_)                                                                                     \
    do {                                                                                                                \
        if (va) MR_DMA(ra, MR_LDS_PTR(void, smem + ista * STAGE_A + (off_) + wave * 2048), 16, abase + (ar_), (unsigned)ika * 128u, 0, 0); \
    } while (0)
#define G3_ADVANCE_A()                                                                                                  \
    do {                                                                                                                \
        if (va) {                                                                                                       \
            ista = (ista == NSTAGE_A - 1) ? 0 : ista + 1;                                                               \
            if (++ika == nkt) {                                                                                         \
                ika = 0;                                                                                                \
                const int m_ = item_m0(++qa);                                                                           \
                va = m_ >= 0;                                                                                           \
                abase = (unsigned)m_ * lda2;                                                                            \
            }                                                                                                           \
        }                                                                                                               \
    } while (0)
    // the wave's four A pieces of a k-tile (rows 0-127: two, rows 128-255: two); the first NA0 are issued in phase 0, the rest lead
    // phase 1's B pieces, so that the two phases' issue work (LDS reads + DMA) is balanced: 16 + NA0 against 8 + (8 - NA0) instructions
#define G3_ISSUE_A_LO()                                                                                                 \
    do {                                                                                                                \
        issued = va;                                                                                                    \
        G3_ISSUE_A_PIECE(ar0, 0);                                                                                       \
        G3_ISSUE_A_PIECE(ar1, 1024);                                                                                    \
    } while (0)
#define G3_ISSUE_A_HI()                                                                                                 \
    do {                                                                                                                \
        G3_ISSUE_A_PIECE(ar2, 16384);                                                                                   \
        G3_ISSUE_A_PIECE(ar3, 16384 + 1024);                                                                            \
        G3_ADVANCE_A();                                                                                                 \
    } while (0)
#define G3_ISSUE_B_LO()                                                                                                 \
    do {                                                                                                                \
        if (vb) {                                                                                                       \
            char* sb_ = smem + OFF_B + istb * STAGE_B + wave * 2048;                                                    \
            const unsigned sb = (unsigned)ikb * 128u;                                                                   \
            MR_DMA(rb, MR_LDS_PTR(void, sb_), 16, bbase + br0, sb, 0, 0);                                               \
            MR_DMA(rb, MR_LDS_PTR(void, sb_ + 1024), 16, bbase + br1, sb, 0, 0);                                        \
        }                                                                                                               \
    } while (0)
#define G3_ISSUE_B_HI()                                                                                                 \
    do {                                                                                                                \
        if (vb) {                                                                                                       \
            char* sb_ = smem + OFF_B + istb * STAGE_B + 16384 + wave * (NBP_HI * 1024);                                 \
            const unsigned sb = (unsigned)ikb * 128u;                                                                   \
            MR_DMA(rb, MR_LDS_PTR(void, sb_), 16, bbase + br2, sb, 0, 0);                                               \
            if (NBP_HI == 2) MR_DMA(rb, MR_LDS_PTR(void, sb_ + 1024), 16, bbase + br3, sb, 0, 0);                       \
            istb = (istb == NSTAGE_B - 1) ? 0 : istb + 1;                                                               \
            if (++ikb == nkt) {                                                                                         \
                ikb = 0;                                                                                                \
                vb = item_m0(++qb) >= 0;                                                                                \
                bbase = (unsigned)item_n0(qb) * ldb2;                                                                   \
            }                                                                                                           \
        }                                                                                                               \
    } while (0)
#define G3_RING_WAIT()                                                                                                  \
    do {                                                                                                                \
        if (issued) wait_vmcnt<WAITN>();                                                                                \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                           \
    } while (0)
#define G3_BARRIER()                                                                                                    \
    do {                                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        __builtin_amdgcn_s_barrier();                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
    } while (0)
#define G3_LGKM(n)                                                                                                      \
    do {                                                                                                                \
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory");                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
    } while (0)

    // PH == 1: all 8 A pieces of the A cursor's k-tile (group 1) / all PB1 B pieces of the B cursor's k-tile (group 0)
#define G3_ISSUE_A8()                                                                                                   \
    do {                                                                                                                \
        issued = va;                                                                                                    \
        if (va) {                                                                                                       \
            char* st_ = smem + ista * STAGE_A + (wave & 3) * 8192;                                                      \
            const unsigned sa = (unsigned)ika * 128u;                                                                   \
            _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) MR_DMA(ra, MR_LDS_PTR(void, st_ + j_ * 1024), 16, abase + po[j_], sa, 0, 0); \
        }                                                                                                               \
        G3_ADVANCE_A();                                                                                                 \
    } while (0)
#define G3_ISSUE_B8()                                                                                                   \
    do {                                                                                                                \
        issued = vb;                                                                                                    \
        if (vb) {                                                                                                       \
            char* sb_ = smem + OFF_B + istb * STAGE_B + (wave & 3) * (PB1 * 1024);                                      \
            const unsigned sb = (unsigned)ikb * 128u;                                                                   \
            _Pragma("unroll") for (int j_ = 0; j_ < PB1; ++j_) MR_DMA(rb, MR_LDS_PTR(void, sb_ + j_ * 1024), 16, bbase + po[j_], sb, 0, 0); \
            istb = (istb == NSTAGE_B - 1) ? 0 : istb + 1;                                                               \
            if (++ikb == nkt) {                                                                                         \
                ikb = 0;                                                                                                \
                vb = item_m0(++qb) >= 0;                                                                                \
                bbase = (unsigned)item_n0(qb) * ldb2;                                                                   \
            }                                                                                                           \
        }                                                                                                               \
    } while (0)
    if constexpr (PH == 1) {
        // prologue: group 1 issues A of k-tiles 0 and 1 (k-tile 0 landed when only the second one's pieces are in flight), group 0 B of k-tile 0
        if (wr == 1) { G3_ISSUE_A8(); G3_ISSUE_A8(); if (issued) wait_vmcnt<8>(); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        else { G3_ISSUE_B8(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    } else {
    // prologue: k-tiles 0 and 1 of the sequence; k-tile 0 has landed when all but the second one's pieces have
    G3_ISSUE_A_LO(); G3_ISSUE_A_HI(); G3_ISSUE_B_LO(); G3_ISSUE_B_HI();
    G3_ISSUE_A_LO(); G3_ISSUE_A_HI(); G3_ISSUE_B_LO(); G3_ISSUE_B_HI();
    G3_RING_WAIT();
    }
    G3_BARRIER();
    if (wr == 1) G3_BARRIER();          // group 1 runs one barrier behind group 0 from here on

    int qc = 0, csa = 0, csb = 0;
    int cm0 = item_m0(0), cn0 = item_n0(0);
    bool have_stores = false;          // an epilogue's stores may be in flight (never before the workgroup's first tile)
    // The bias is the accumulators' INITIAL value (no add per element in the epilogue).  A tile's bias is fetched and widened in
    // the PREVIOUS tile's epilogue, ahead of its stores (a load consumed behind them would wait for them: vmcnt retires in order).
    constexpr bool BIAS = MODE <= 2;             // residual / aux problems carry no bias (host check)
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
#ifdef MR_G3_STAMPS
    unsigned long long* stamps = static_cast<unsigned long long*>(ga.p[0].workspace);
    int nstamp = 0;
#define G3_STAMP(slot)                                                                              \
    do {                                                                                            \
        unsigned long long t_;                                                                      \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                  \
        if (lane == 0 && (wave & 3) == 0 && nstamp < 8) stamps[((blockIdx.x * 2 + wr) * 8 + nstamp) * 4 + (slot)] = t_;  \
    } while (0)
    // per-k-tile stamps of the first 8 workgroups' wave 0 (second KiB-aligned region of the workspace)
#define G3_KSTAMP(t)                                                                                \
    do {                                                                                            \
        unsigned long long t_;                                                                      \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                  \
        if (lane == 0 && wave == 0 && blockIdx.x < 8 && nstamp < 4 && (t) < 16) stamps[65536 + (blockIdx.x * 4 + nstamp) * 16 + (t)] = t_;  \
    } while (0)
#else
#define G3_STAMP(slot) do {} while (0)
#define G3_KSTAMP(t) do {} while (0)
#endif
    while (cm0 >= 0) {
        // accumulators TRANSPOSED (mfma(B-frag, A-frag)): the lane holds C[m = .. + li][n = .. + 4 g + r]
        f32x4 acc[8][NJ];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = binit[j];

        G3_STAMP(0);
        auto ktile = [&](auto first_c) {
            constexpr bool FIRST = decltype(first_c)::value;       // the tile's first k-tile: the previous tile's stores are in flight
            const char* As = smem + csa * STAGE_A;
            const char* Bs = smem + OFF_B + csb * STAGE_B;
            bf16x8 a[4][2], b[NJ][2];
            auto read_b = [&](int j0, int j1) {
#pragma unroll
                for (int j = j0; j < j1; ++j)
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) b[j][kk] = frag<false, 256, 64>(Bs, wc * WCOLS + j * 16, kk, lane);
            };
            auto read_a = [&](int half) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) a[i][kk] = frag<false, 256, 64>(As, wr * 128 + half * 64 + i * 16, kk, lane);
            };
            auto mfmas = [&](int half, int j0, int j1) {
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = j0; j < j1; ++j)
                            acc[half * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j][kk], a[i][kk], acc[half * 4 + i][j], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
            };
            auto tile_wait = [&]() {        // k-tile t+1 has landed behind the next barrier
                if (FIRST && issued && have_stores) wait_vmcnt<WAITN + NST>();
                else G3_RING_WAIT();
            };
            if constexpr (PH == 1) {
                // ---------------- memory section: B + A0 fragments, this group's 8 pieces ----------------
                read_b(0, NJ);
                __builtin_amdgcn_sched_barrier(0);      // B before A: the counted wait below relies on the issue order
                read_a(0);
                __builtin_amdgcn_sched_barrier(0);
                if (wr == 0) {
                    G3_ISSUE_B8();                       // k-tile t+1 into the stage k-tile t-1 was read from (its reads were retired ahead of a barrier)
                    G3_LGKM(8);
                } else {
                    G3_ISSUE_A8();                       // k-tile t+2
                    G3_LGKM(8);
                    // A of k-tile t+1 (issued one k-tile ago) has landed behind the coming barrier: group 0 reads it right after it
                    if (FIRST && issued && have_stores) wait_vmcnt<8 + NST>();
                    else if (issued) wait_vmcnt<8>();
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                G3_BARRIER();
                G3_LGKM(0);
                // ---------------- MFMA section: A0 x B row block by row block, each block's A1 fragments requested into the registers
                // its last A0 MFMA has just read; then A1 x B ----------------
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                        for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j][kk], a[i][kk], acc[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) a[i][kk] = frag<false, 256, 64>(As, wr * 128 + 64 + i * 16, kk, lane);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                        for (int j = 0; j < NJ; ++j) acc[4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j][kk], a[i][kk], acc[4 + i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                __builtin_amdgcn_s_setprio(0);
                if (wr == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // B of k-tile t+1 has landed behind the closing barrier
                G3_BARRIER();
            } else if constexpr (!PH4) {
                // ---------------- phase 0: A0 x B.  The B reads are retired AHEAD of the barrier (phase 1 restages B) ----------------
                read_b(0, NJ);
                __builtin_amdgcn_sched_barrier(0);      // B before A: the counted wait below relies on the issue order
                read_a(0);
                __builtin_amdgcn_sched_barrier(0);
                G3_ISSUE_A_LO();
                if constexpr (NA0 >= 3) G3_ISSUE_A_PIECE(ar2, 16384);
                if constexpr (NA0 >= 4) { G3_ISSUE_A_PIECE(ar3, 16384 + 1024); G3_ADVANCE_A(); }
                G3_LGKM(8);
                G3_BARRIER();
                G3_LGKM(0);
                mfmas(0, 0, NJ);
                G3_BARRIER();
                // ---------------- phase 1: A1 x B.  The A reads are retired ahead of the barrier (the next phase 0 restages A) ----------------
                read_a(1);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (NA0 < 3) G3_ISSUE_A_PIECE(ar2, 16384);
                if constexpr (NA0 < 4) { G3_ISSUE_A_PIECE(ar3, 16384 + 1024); G3_ADVANCE_A(); }
                G3_ISSUE_B_LO();
                G3_ISSUE_B_HI();
                G3_LGKM(0);
                tile_wait();
                G3_BARRIER();
                mfmas(1, 0, NJ);
                G3_BARRIER();
            } else {
                // ---------------- four phases: (A0,B0) (A0,B1) (A1,B1) (A1,B0) ----------------
                read_b(0, 2);
                read_a(0);
                __builtin_amdgcn_sched_barrier(0);
                G3_ISSUE_A_LO();
                G3_BARRIER();
                G3_LGKM(0);
                mfmas(0, 0, 2);
                G3_BARRIER();
                read_b(2, NJ);
                __builtin_amdgcn_sched_barrier(0);
                G3_ISSUE_A_HI();
                G3_LGKM(0);             // ahead of the barrier: B is restaged from the next phase on
                G3_BARRIER();
                mfmas(0, 2, NJ);
                G3_BARRIER();
                read_a(1);
                __builtin_amdgcn_sched_barrier(0);
                G3_ISSUE_B_LO();
                G3_BARRIER();
                G3_LGKM(0);
                mfmas(1, 2, NJ);
                G3_BARRIER();
                G3_ISSUE_B_HI();
                tile_wait();
                G3_BARRIER();
                mfmas(1, 0, 2);
                G3_BARRIER();
            }
            csa = (csa == NSTAGE_A - 1) ? 0 : csa + 1;
            csb = (csb == NSTAGE_B - 1) ? 0 : csb + 1;
        };
        G3_KSTAMP(0);
        ktile(std::integral_constant<bool, true>{});
        for (int t = 1; t < nkt; ++t) { G3_KSTAMP(t); ktile(std::integral_constant<bool, false>{}); }
        G3_STAMP(1);

        // ---------------- epilogue (bf16 output; registers + ordinary loads, no LDS) ----------------
        // With EPI_SYNC the two groups' epilogues run side by side (group 0 waits for group 1's last MFMAs, group 1 waits behind
        // its epilogue for group 0's, which restores the one-barrier offset); without, they run one after the other.
        if (EPI_SYNC && wr == 0) G3_BARRIER();
        {
            const int wrow0 = cm0 + wr * 128, wcol0 = cn0 + wc * WCOLS;
            constexpr int MR_EPI_MI = 8;
#define MR_EPI_ROW_FENCE() do {} while (0)
#include "gemm3_epilogue.inc"
#undef MR_EPI_ROW_FENCE
#ifndef MR_G3_NOSTORE
            have_stores = true;
#endif
        }
        G3_STAMP(2);
#ifdef MR_G3_STAMPS
        ++nstamp;
#endif
        if (EPI_SYNC && wr == 1) G3_BARRIER();
        ++qc;
        cm0 = item_m0(qc);
        cn0 = item_n0(qc);
    }
    if (wr == 0) G3_BARRIER();          // pairs with group 1's last barrier
#ifdef MR_DIAG_RELEASE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}


// ---------------------------------------------------------------------------------------------------------------------------
// "TN" variant for the weight gradients:  C[M,N] = A^T . B with A stored [K, M] and B stored [K, N] (activations x upstream
// gradients, the contraction index = tokens is the ROW index of both).  Same two-group ping-pong k-loop; the operand tiles are
// [64 k][256] images read with ds_read_b64_tr_b16 (two per MFMA operand).  One 256 x 256 output tile per workgroup, no
// persistence (K = tokens is long: 93-241 k-tiles, the prologue and the plain bf16 epilogue are < 2 % of a tile), up to 20
// problems per launch: the weight gradients of TWO transformer layers fill the chip without split-K (216 tiles for the base
// model) where one layer's four are 108; FOUR layers of the large model are 768 tiles = three full rounds where one layer's 192
// leave a quarter of the CUs idle.
struct TNProb { const void* A; const void* B; void* C; int M, N, lda, ldb, ldc, pad; };
constexpr int TN_MAXG = 20;
struct TNArgs { int count, K; int tile_start[TN_MAXG + 1]; int tiles_n[TN_MAXG]; TNProb p[TN_MAXG]; };

__global__ __launch_bounds__(512, 2) void gemm3_tn_kernel(const TNArgs ta) {
    constexpr int NJ = 4, STAGE = 64 * 512, NSTAGE_A = 3, NSTAGE_B = 2, OFF_B = NSTAGE_A * STAGE, WAITN = 8;
    __shared__ __attribute__((aligned(16))) char smem[(NSTAGE_A + NSTAGE_B) * STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int g = lane >> 4, li = lane & 15;
    const int G = gridDim.x;
    const int xcd = blockIdx.x & 7, qd = G >> 3, rm = G & 7;
    const int tile = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (blockIdx.x >> 3);
    int pi = 0;
#pragma unroll
    for (int k = 1; k < TN_MAXG; ++k) pi += (k < ta.count && tile >= ta.tile_start[k]) ? 1 : 0;
    const TNProb& pr = ta.p[pi];
    const int lt = tile - ta.tile_start[pi], tn = ta.tiles_n[pi];
    const int m0 = (lt / tn) * 256, n0 = (lt % tn) * 256;
    const int M = pr.M, N = pr.N, K = ta.K;
    const int nkt = (K + 63) >> 6;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(pr.A), 0, (int)(((int64_t)(K - 1) * pr.lda + M) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(pr.B), 0, (int)(((int64_t)(K - 1) * pr.ldb + N) * 2), 0x00020000);
    const unsigned a_step = 64u * (unsigned)pr.lda * 2u, b_step = 64u * (unsigned)pr.ldb * 2u;
    unsigned ao[4], bo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        ao[j] = piece_src<true, 256, 64>(wave * 4 + j, lane, pr.lda, m0, M);
        bo[j] = piece_src<true, 256, 64>(wave * 4 + j, lane, pr.ldb, n0, N);
    }
    int ik = 0, ista = 0, istb = 0;
    bool issued = false;
#define TN_ISSUE_A()                                                                                                    \
    do {                                                                                                                \
        issued = ik < nkt;                                                                                              \
        if (issued) {                                                                                                   \
            char* st_ = smem + ista * STAGE + wave * 4096;                                                              \
            const unsigned sa = (unsigned)ik * a_step;                                                                  \
            MR_DMA(ra, MR_LDS_PTR(void, st_), 16, ao[0], sa, 0, 0);                                                     \
            MR_DMA(ra, MR_LDS_PTR(void, st_ + 1024), 16, ao[1], sa, 0, 0);                                              \
            MR_DMA(ra, MR_LDS_PTR(void, st_ + 2048), 16, ao[2], sa, 0, 0);                                              \
            MR_DMA(ra, MR_LDS_PTR(void, st_ + 3072), 16, ao[3], sa, 0, 0);                                              \
            ista = (ista == NSTAGE_A - 1) ? 0 : ista + 1;                                                               \
        }                                                                                                               \
    } while (0)
#define TN_ISSUE_B()                                                                                                    \
    do {                                                                                                                \
        if (ik < nkt) {                                                                                                 \
            char* sb_ = smem + OFF_B + istb * STAGE + wave * 4096;                                                      \
            const unsigned sb = (unsigned)ik * b_step;                                                                  \
            MR_DMA(rb, MR_LDS_PTR(void, sb_), 16, bo[0], sb, 0, 0);                                                     \
            MR_DMA(rb, MR_LDS_PTR(void, sb_ + 1024), 16, bo[1], sb, 0, 0);                                              \
            MR_DMA(rb, MR_LDS_PTR(void, sb_ + 2048), 16, bo[2], sb, 0, 0);                                              \
            MR_DMA(rb, MR_LDS_PTR(void, sb_ + 3072), 16, bo[3], sb, 0, 0);                                              \
            istb = (istb == NSTAGE_B - 1) ? 0 : istb + 1;                                                               \
            ++ik;                                                                                                       \
        }                                                                                                               \
    } while (0)
    TN_ISSUE_A(); TN_ISSUE_B();
    TN_ISSUE_A(); TN_ISSUE_B();
    if (issued) wait_vmcnt<WAITN>(); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    G3_BARRIER();
    if (wr == 1) G3_BARRIER();

    f32x4 acc[8][NJ];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    int csa = 0, csb = 0;
    for (int t = 0; t < nkt; ++t) {
        const char* As = smem + csa * STAGE;
        const char* Bs = smem + OFF_B + csb * STAGE;
        bf16x8 a[4][2], b[NJ][2];
        auto read_a = [&](int half) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) a[i][kk] = frag<true, 256, 64>(As, wr * 128 + half * 64 + i * 16, kk, lane);
        };
        auto mfmas = [&](int half) {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        acc[half * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j][kk], a[i][kk], acc[half * 4 + i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        };
        // phase 0: A0 x B; every read of the phase is retired AHEAD of the barrier (phase 1 restages B; lgkmcnt counts to 15 only)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) b[j][kk] = frag<true, 256, 64>(Bs, wc * 64 + j * 16, kk, lane);
        read_a(0);
        __builtin_amdgcn_sched_barrier(0);
        TN_ISSUE_A();
        G3_LGKM(0);
        G3_BARRIER();
        mfmas(0);
        G3_BARRIER();
        // phase 1: A1 x B
        read_a(1);
        __builtin_amdgcn_sched_barrier(0);
        TN_ISSUE_B();
        G3_LGKM(0);
        if (issued) wait_vmcnt<WAITN>(); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        G3_BARRIER();
        mfmas(1);
        G3_BARRIER();
        csa = (csa == NSTAGE_A - 1) ? 0 : csa + 1;
        csb = (csb == NSTAGE_B - 1) ? 0 : csb + 1;
    }
    if (wr == 0) G3_BARRIER();          // pairs with group 1's last barrier

    // epilogue: bf16 stores, widened to 16 B (see the NT kernel's store_pair)
    const unsigned c_bytes = (unsigned)(((int64_t)(M - 1) * pr.ldc + N) * 2);
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(pr.C, 0, (int)c_bytes, 0x00020000);
    const int wrow0 = m0 + wr * 128, wcol0 = n0 + wc * 64;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int gm = wrow0 + i * 16 + li;
        const unsigned rowoff = (unsigned)gm * (unsigned)pr.ldc * 2u;
#pragma unroll
        for (int jp = 0; jp < NJ / 2; ++jp) {
            bf16x4 va, vb;
#pragma unroll
            for (int r = 0; r < 4; ++r) { va[r] = (__bf16)acc[i][2 * jp][r]; vb[r] = (__bf16)acc[i][2 * jp + 1][r]; }
            u32x2 ua = __builtin_bit_cast(u32x2, va), ub = __builtin_bit_cast(u32x2, vb);
            const auto s0 = __builtin_amdgcn_permlane16_swap(ua[0], ub[0], false, false);
            const auto s1 = __builtin_amdgcn_permlane16_swap(ua[1], ub[1], false, false);
            const int col = wcol0 + (2 * jp + (g & 1)) * 16 + (g >> 1) * 8;
            const unsigned off = (gm < M && col < N) ? rowoff + (unsigned)col * 2u : OOB;
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{s0[0], s1[0], s0[1], s1[1]}, rc, off, 0, 0);
        }
    }
}

}  // namespace g3

constexpr int64_t NUM_CU3 = 256;    // MI355X
bool mr_gemm4_takes(const mr_gemm_args* a);
int mr_gemm4_launch(const mr_gemm_args* a, int bn, const g256::G256Args& ga, int64_t gsz, hipStream_t s);

// The ping-pong kernel takes: NT operands (A [M,K], B [N,K], K % 64 == 0), bf16 output, at least one full round of 256-row tiles'
// worth of work, and one of the epilogue combinations the step uses -- bias; bias + "rotary"; bias + GELU with the gelu' copy;
// residual; aux (with or without column sums).
// persistent workgroups = CUs a launch may fill (mr_set_option "gemm_cus": 256, or fewer -- a multiple of 8 -- while a collective
// kernel holds CUs: a 256-workgroup grid would then run its last workgroups as a second round behind the others)
static int64_t g3_ncu() { return (g_mr_opt_gemm_cus >= 64 && g_mr_opt_gemm_cus < NUM_CU3) ? (g_mr_opt_gemm_cus & ~7) : NUM_CU3; }

bool mr_gemm3_eligible(const mr_gemm_args* a) {
    static int env = -1;
    if (env < 0) env = mr_env_int("MR_GEMM3", 1);
    if (!env || !g_mr_opt_gemm3) return false;
    if (a->transA || !a->transB || a->c_dtype != MR_DT_BF16) return false;
    const bool forced = g_mr_opt_gemm3 == 256 || g_mr_opt_gemm3 == 192;      // tests: any shape the kernel can take
    if (a->K % 64 != 0) return false;
    if (!forced && (a->K < 128 || a->N < 192 || a->M < 1024)) return false;
    if (a->M * a->lda * 2 >= (1LL << 31) || a->N * a->ldb * 2 >= (1LL << 31)) return false;    // 32-bit buffer offsets
    {
        const int64_t last = a->out_grp > 0 ? ((a->M - 1) / a->out_grp) * a->out_grp_stride + a->out_grp_off + (a->M - 1) % a->out_grp : a->M - 1;
        if ((last * a->ldc + a->N) * 2 >= (1LL << 31)) return false;
    }
    const int n_extra = (a->rot_tab != nullptr) + (a->residual != nullptr) + (a->aux != nullptr) + (a->c2 != nullptr);
    if (n_extra > 1) return false;
    if ((a->c2 != nullptr) != (a->act == MR_ACT_GELU1702)) return false;
    if (a->colsum && !a->aux) return false;
    if (a->bias && (a->residual || a->aux)) return false;      // the bias rides in the accumulators of the bias modes only
    // enough tiles to fill the chip (short-K problems with few tiles go to the split-K path of the one-barrier kernel)
    const int64_t tm = (a->M + 255) / 256;
    if (tm * ((a->N + 191) / 192) > 64 * g3_ncu()) return false;     // <= 64 items per workgroup (the kernel keeps them one per lane)
    static int min_tiles = -1;
    if (min_tiles < 0) min_tiles = mr_env_int("MR_G3_MIN_TILES", 128);
    if (!forced && tm * ((a->N + 255) / 256) < min_tiles) return false;
    return true;
}

int mr_gemm3_launch(const mr_gemm_args* a, hipStream_t s) {
    const int64_t tm = (a->M + 255) / 256;
    // tile width: fewest CU-rounds, a 192-wide tile costing 3/4 of a 256-wide one
    const int64_t t256 = tm * ((a->N + 255) / 256), t192 = tm * ((a->N + 191) / 192);
    const int64_t ncu = g3_ncu();
    const int64_t c256 = ((t256 + ncu - 1) / ncu) * 100, c192 = ((t192 + ncu - 1) / ncu) * 78;
    int bn = (c192 < c256) ? 192 : 256;
    if (g_mr_opt_gemm3 == 256 || g_mr_opt_gemm3 == 192) bn = g_mr_opt_gemm3;
    const int64_t tn = (a->N + bn - 1) / bn, nwork = tm * tn;
    const int64_t gsz = nwork < ncu ? nwork : ncu;
    g256::G256Args ga;
    memset(&ga, 0, sizeof(ga));
    ga.count = 1; ga.nwork = (int)nwork; ga.splits = 1; ga.kt_per_split = (int)(a->K / 64);
    if (gsz == ncu && nwork >= 2 * ncu) {
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
                    const int64_t r = (hm * hn + ncu / 8 - 1) / (ncu / 8);
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
    {   // the one-wave-per-SIMD kernel (gemm4.hip) shares this tile plan
        static int g4_env = -1;
        if (g4_env < 0) g4_env = mr_env_int("MR_GEMM4", 1);
        const int g4 = g_mr_opt_gemm4 >= 0 ? g_mr_opt_gemm4 : g4_env;
        // (not the 256-wide bias mode: 256 accumulators + the bias rows overflow the wave's 512 registers -- 183 spilled, reloads inside
        // the k-loop, 299 vs 91 us -- so that combination stays on the ping-pong kernel and gemm4<256,0> is not instantiated)
        if (g4 && mr_gemm4_takes(a) && !(bn == 256 && a->bias && !a->residual)) return mr_gemm4_launch(a, bn, ga, gsz, s);
    }
    dim3 grid((unsigned)gsz), block(512);
    int mode = 0;
    if (a->c2) mode = 2;
    else if (a->rot_tab) mode = 1;
    else if (a->residual) mode = 3;
    else if (a->aux) mode = 4;
    static int ph_env = -1;
    if (ph_env < 0) ph_env = mr_env_int("MR_G3_PH", 0);
    // one phase per k-tile (half the barriers: 2-7 % faster at K <= 4096) except where it measured slower: the 256-wide aux / column-sum
    // epilogue (its prefetch registers + the one-phase schedule's spill: 94 vs 86 us on the fc1 dgrad) and very long k-loops, where B
    // requested one phase ahead by ONE group lands later than the two-phase schedule's (8192^3: 745 vs 729 us)
    const int ph_auto = ((mode == 4 && bn == 256) || a->K >= 8192) ? 2 : 1;
    int ph = g_mr_opt_gemm3_ph ? g_mr_opt_gemm3_ph : ph_env ? ph_env : ph_auto;
    if (mode == 4 && bn == 256) ph = 2;      // <256,4,1> is not built: 20 spilled registers with a reload inside the k-loop's MFMA block
#define G3_LAUNCH(MODE)                                                                               \
    do {                                                                                              \
        if (ph == 1) {                                                                                \
            if (bn == 256) hipLaunchKernelGGL((g3::gemm3_kernel<256, MODE, (MODE == 4 ? 2 : 1)>), grid, block, 0, s, ga);   \
            else hipLaunchKernelGGL((g3::gemm3_kernel<192, MODE, 1>), grid, block, 0, s, ga);             \
        } else {                                                                                      \
            if (bn == 256) hipLaunchKernelGGL((g3::gemm3_kernel<256, MODE, 2>), grid, block, 0, s, ga);   \
            else hipLaunchKernelGGL((g3::gemm3_kernel<192, MODE, 2>), grid, block, 0, s, ga);             \
        }                                                                                             \
    } while (0)
    switch (mode) {
        case 0: G3_LAUNCH(0); break;
        case 1: G3_LAUNCH(1); break;
        case 2: G3_LAUNCH(2); break;
        case 3: G3_LAUNCH(3); break;
        default: G3_LAUNCH(4); break;
    }
#undef G3_LAUNCH
    mr_note_route("g3::gemm3_kernel<%d,%d,%d>", bn, mode, ph);
    return 0;
}

// Grouped weight gradients on the TN ping-pong kernel: <= 8 problems, all transA = 1 / transB = 0 with one K, bf16 outputs, no
// epilogue.  Returns false when the group does not qualify (the caller falls back to the one-barrier kernel's grouped launch).
bool mr_gemm3_tn_grouped(const mr_gemm_args* list, int count, hipStream_t s) {
    static int env = -1;
    if (env < 0) env = mr_env_int("MR_GEMM3_TN", 1);
    if (!env || !g_mr_opt_gemm3 || count < 1 || count > g3::TN_MAXG) return false;
    if (g_mr_opt_group_tile_n != 0) return false;   // an explicitly requested tile width of the one-barrier kernel
    g3::TNArgs ta;
    memset(&ta, 0, sizeof(ta));
    int64_t tiles = 0;
    for (int k = 0; k < count; ++k) {
        const mr_gemm_args* a = &list[k];
        if (!a->transA || a->transB || a->c_dtype != MR_DT_BF16 || a->K != list[0].K) return false;
        if (a->bias || a->rot_tab || a->c2 || a->act != MR_ACT_NONE || a->residual || a->aux || a->out_grp != 0 || a->colsum) return false;
        if (a->M % 8 || a->N % 8 || a->lda % 8 || a->ldb % 8 || a->ldc % 8) return false;
        if (a->K * a->lda * 2 >= (1LL << 31) || a->K * a->ldb * 2 >= (1LL << 31) || a->M * a->ldc * 2 >= (1LL << 31)) return false;
        const int64_t tm = (a->M + 255) / 256, tn = (a->N + 255) / 256;
        ta.tile_start[k] = (int)tiles;
        ta.tiles_n[k] = (int)tn;
        tiles += tm * tn;
        ta.p[k] = g3::TNProb{a->A, a->B, a->C, (int)a->M, (int)a->N, (int)a->lda, (int)a->ldb, (int)a->ldc, 0};
    }
    ta.tile_start[count] = (int)tiles;
    ta.count = count;
    ta.K = (int)list[0].K;
    // worth it when the tiles fill most of the chip (one tile per workgroup, one workgroup per CU) and K is long
    const bool forced = g_mr_opt_gemm3 == 256;
    if (!forced && (tiles < 160 || tiles > 4 * NUM_CU3 || list[0].K < 2048)) return false;
    hipLaunchKernelGGL(g3::gemm3_tn_kernel, dim3((unsigned)tiles), dim3(512), 0, s, ta);
    mr_note_route("g3::gemm3_tn_kernel grouped x%d", count);
    return true;
}
