// Shared by gemm256.hip (one barrier per k-tile) and gemm3.hip (two wave groups, ping-pong): LDS-DMA piece addressing and its
// source-side swizzles, fragment reads, the work-item decode of the persistent grid (XCD-blocked order, split-K, grouped launch).
#pragma once
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include "mr_common.h"

#ifdef MR_DIAG_SAMEK
#define MR_DIAG_K(k) 0u     /* timing-only build: wrong results */
#else
#define MR_DIAG_K(k) (unsigned)(k)
#endif

#if defined(MR_DIAG_NOFRAG) || defined(MR_DIAG_NOMFMA)   /* timing-only builds: no LDS fragment reads */
#define MR_DIAG_FRAG(x) bf16x8{}
#else
#define MR_DIAG_FRAG(x) (x)
#endif
#ifdef MR_DIAG_NOMFMA
#define MR_DIAG_MFMA(x) do {} while (0)
#else
#define MR_DIAG_MFMA(x) x
#endif
#ifdef MR_DIAG_NOLOAD      /* timing-only build (wrong results): no LDS-DMA at all, cursors still advance */
#define MR_DMA(...) do {} while (0)
#else
#define MR_DMA(...) __builtin_amdgcn_raw_ptr_buffer_load_lds(__VA_ARGS__)
#endif
#ifdef MR_DIAG_NOSTORE     /* timing-only build (no output): the bf16 epilogue computes everything and stores nothing */
#define MR_DIAG_ST(c) ((c) && ga.nwork < 0)
#else
#define MR_DIAG_ST(c) (c)
#endif
#ifdef MR_DIAG_STAMPS
#define MR_STAMP(slot)                                                                              \
    do {                                                                                            \
        unsigned long long t_;                                                                      \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                  \
        if (tid == 0 && nstamp < 64) stamps[(blockIdx.x * 64 + nstamp) * 4 + (slot)] = t_;          \
    } while (0)
#else
#define MR_STAMP(slot) do {} while (0)
#endif

namespace g256 {

constexpr int BM = 256, BK = 64;
// B is staged 128 wide (BN = 128 | 96: 48-KiB stages, 3-stage ring, two k-tiles ahead) or 256 wide (BN = 256: 64-KiB
// stages, 2-stage ring, one k-tile ahead).  The k-loop is bound by the LDS-DMA fill rate of a CU (~60-70 GB/s measured
// with the MFMAs compiled out), so the 256 x 256 tile -- 2/3 of the bytes per FLOP -- is the fast one wherever the
// problem has enough tiles for it.
// BN = 192 uses the 256-wide geometry with the B columns beyond 192 left to the buffer descriptor's zero fill (no L2
// traffic): 244 tiles instead of 183 for M = 15424, N = 768 -- one full round of the 256 CUs -- at 3/4 of the MFMAs
// and 7/8 of the bytes of a 256-wide tile.
template <int BN> struct Geo {
    static constexpr int NW = 8, BKT = BK;                 // eight waves, one 512-thread workgroup per CU
    static constexpr int BW = (BN > 128) ? 256 : 128;
    static constexpr int STAGE_A = BM * BKT * 2;           // 32 KiB
    static constexpr int STAGE_B = BW * BKT * 2;
    // Two rings.  A (32 KiB per k-tile) always runs TWO k-tiles ahead of the MFMAs in 3 stages.  B runs two ahead in 3
    // stages when it is 128 wide (144 KiB in all) and ONE ahead in 2 stages when it is 256 wide (96 + 64 = 160 KiB, all of
    // the LDS): the B pieces of k-tile t+1 are issued first in step t and must have landed by its end, the A pieces of
    // k-tile t+2 are issued behind them and may stay in flight across the barrier.  (With both operands one k-tile ahead
    // in 2 stages the whole 64-KiB fill sat between the issue and the end of the same k-tile: 3600 cycles per k-tile
    // against 2060 of MFMA.)
    static constexpr int NSTAGE_A = 3;
    static constexpr int B_AHEAD = (BN > 128) ? 1 : 2;
    static constexpr int NSTAGE_B = B_AHEAD + 1;
    static constexpr int OFF_B = NSTAGE_A * STAGE_A;
    static constexpr int LDS_BYTES = OFF_B + NSTAGE_B * STAGE_B;
    static constexpr int NBP = STAGE_B / 1024 / NW;        // 1-KiB B pieces per wave and k-tile (A: always 4)
    static_assert(STAGE_A / 1024 / NW == 4, "4 A pieces per wave");
    // pieces of one step that may still be in flight behind its barrier: the step's A pieces, plus its B pieces when B
    // also runs two ahead
    static constexpr int WAITN = 4 + (B_AHEAD == 2 ? NBP : 0);
};
static_assert(Geo<256>::LDS_BYTES == 160 * 1024 && Geo<128>::LDS_BYTES == 144 * 1024, "LDS budget");
constexpr unsigned OOB = 0x80000000u;      // >= any operand extent (< 2^31 B, checked on the host); + soffset cannot wrap

typedef __attribute__((ext_vector_type(8))) short s16x8;

__device__ __forceinline__ int swz_kc(int row) { return (row >> 1) & 7; }
__device__ __forceinline__ int swz_ks(int k) { return 2 * ((k & 3) + 4 * ((k >> 3) & 1)); }

// byte offset (into the operand's buffer, for k-tile 0) of the 16-byte chunk that lane `lane` of piece `p` fetches.
// The k-tile advance is a wave-uniform scalar offset (128 B per k-tile for K-contiguous, 64 rows for K-strided), and
// k-rows past K fall beyond the operand's extent, so validity does not depend on the k-tile.

template <bool TR, int W, int BKT = 64>
__device__ __forceinline__ unsigned piece_src(int p, int lane, int64_t ld, int64_t own0, int64_t own_n) {
    if (!TR) {
        const int row = p * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ swz_kc(row);
        const int64_t grow = own0 + row;
        if (grow >= own_n) return OOB;
        return (unsigned)((grow * ld + chunk * 8) * 2);
    } else {
        constexpr int C = W / 8;                 // chunks per k-row
        const int krow = p * (64 / C) + lane / C;
        const int chunk = (lane % C) ^ swz_ks(krow);
        const int64_t gcol = own0 + chunk * 8;
        if (gcol >= own_n) return OOB;
        return (unsigned)((krow * ld + gcol) * 2);
    }
}

// Epilogue operands are fetched EARLY (top of the item's last k-tile) with ordinary loads.  With LDS-DMA in flight hipcc
// turns the first use of an ordinary load into s_waitcnt vmcnt(0); issued a whole k-tile before that first use (in the
// epilogue), the wait finds them -- and the ring's pieces issued during the last k-tile -- already landed.
// (Inline-asm loads retired by the loop's counted vmcnt were tried: the register allocator may split the live range of
// an asm result, i.e. copy the register before the data lands -- wrong values, no fault.  Not used.)
#define PRE_LOAD_B64(dst, ptr) (dst) = *reinterpret_cast<const u32x2*>(ptr)
__device__ __forceinline__ void reg_fence(u32x2& v) { asm volatile("" : "+v"(v)); }

template <bool TR, int W, int BKT = 64>
__device__ __forceinline__ bf16x8 frag(const char* tile, int own0, int kk, int lane) {
    const int g = lane >> 4, i = lane & 15;
    if (!TR) {
        const int row = own0 + i;
        return *reinterpret_cast<const bf16x8*>(tile + row * 128 + (((kk * 4 + g) ^ swz_kc(row)) << 4));
    } else {
        const int q = i >> 2, p = i & 3;
        const int k0 = kk * 32 + g * 8 + q;                 // rows k0 and k0 + 4 share swz bits except (k & 3)
        const int chunk = (own0 >> 3) + (p >> 1);
        // rows k0 and k0 + 4 have the same swizzle (bits 0-1 and bit 3 of k are equal), so one address + an immediate.
        // Inline asm on purpose: hipcc puts s_waitcnt vmcnt(0) in front of the ds_read_tr builtin while LDS-DMA is in
        // flight (it cannot disambiguate it from the DMA's LDS writes), which would drain the prefetch ring every
        // k-tile.  The caller waits with an explicit lgkmcnt(0) + sched_barrier before the MFMAs (guide 5.7 form iii).
        const unsigned a0 = (unsigned)(uintptr_t)MR_LDS_PTR(const char, tile + k0 * (W * 2) + ((chunk ^ swz_ks(k0)) << 4) + (p & 1) * 8);
        s16x4 lo, hi;
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a0));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(a0), "i"(4 * W * 2));
        s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, both);
    }
}

// sum over the 16 lanes of a DPP row (lanes 16 g .. 16 g + 15: the 16 output rows a lane group holds for one column set):
// quad butterflies, then the two mirrors -- four v_add with DPP modifiers, no LDS
__device__ __forceinline__ float row16_sum(float v) {
#define MR_DPP(x, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, 0xF, 0xF, true))
    v += MR_DPP(v, 0xB1);      // quad_perm [1,0,3,2]
    v += MR_DPP(v, 0x4E);      // quad_perm [2,3,0,1]
    v += MR_DPP(v, 0x141);     // row_half_mirror
    v += MR_DPP(v, 0x140);     // row_mirror
#undef MR_DPP
    return v;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// One unit of work: an output tile (and, under split-K, one K range of it).
struct Item {
    int pi, m0, n0, kt0, nkt, split;
    bool valid;
};

constexpr int MAXG = 4;
// One launch = up to MAXG independent problems (same layouts and tile width) sharing the persistent grid: the four
// weight gradients of a transformer layer fill the 256 CUs together, with no split-K traffic.
struct G256Args {
    int count, nwork, splits, kt_per_split;
    // XCD-blocked tile order (xmode = 1; single problem, no split-K, grid = 256): the 8 XCDs own the cells of a px x py
    // partition of the tile grid and walk their cell in panels of XPANEL tile columns, so that the 32 workgroups of an
    // XCD -- which move through K in lockstep -- ask their L2 for few distinct operand strips at a time, and a cell's
    // slice of B stays L2-resident while its A strips stream through.  (With the plain row-major order every XCD swept
    // all of B once per round of 256 tiles: rocprofv3 FETCH_SIZE showed 6x the operand bytes leaving L2 per launch.)
    int xmode, px, py, tm, tn;
    int xpanel;            // tile columns per panel of a cell's walk (0 = XPANEL; option "gemm_xpanel", an experiment knob)
    int tiles_n[MAXG], tile_start[MAXG + 1];
    mr_gemm_args p[MAXG];
};

constexpr int XPANEL = 8;
template <int BKT, int GSH>     // GSH = log2(workgroups of a full grid): 8, or 9 for the two-per-CU kernel
__device__ __forceinline__ Item make_item(const G256Args& ga, int w, int bn, int bperm) {
    Item it;
    if (ga.xmode == 1) {
        constexpr int PX = 1 << (GSH - 3);                                 // workgroups per XCD
        const int r = w >> GSH, bp = w & ((1 << GSH) - 1), x = bp / PX, sl = bp % PX;     // w = bperm + r * 256, bperm = xcd * 32 + slot
        const int xi = x / ga.py, xj = x - xi * ga.py;
        const int m_lo = xi * ga.tm / ga.px, hm = (xi + 1) * ga.tm / ga.px - m_lo;
        const int n_lo = xj * ga.tn / ga.py, hn = (xj + 1) * ga.tn / ga.py - n_lo;
        const int q = r * PX + sl;
        it.valid = q < hm * hn;
        const int xp_ = ga.xpanel > 0 ? ga.xpanel : XPANEL;
        const int gw = hn < xp_ ? hn : xp_;
        const int panel = q / (hm * gw), rem = q - panel * hm * gw;
        const int left = hn - panel * gw, pw = left < gw ? left : gw;          // the last panel may be narrower
        const int m = rem / (pw > 0 ? pw : 1), n = panel * gw + rem - m * pw;
        it.pi = 0;
        it.split = 0;
        it.m0 = (m_lo + m) * BM;
        it.n0 = (n_lo + n) * bn;
        it.kt0 = 0;
        it.nkt = (int)((ga.p[0].K + BKT - 1) / BKT);
        return it;
    }
    it.valid = w < ga.nwork;
    const int tile = (ga.splits == 1) ? w : w / ga.splits;
    it.split = w - tile * ga.splits;
    it.pi = (tile >= ga.tile_start[1]) + (tile >= ga.tile_start[2]) + (tile >= ga.tile_start[3]);
    const int lt = tile - ga.tile_start[it.pi], tn = ga.tiles_n[it.pi];
    it.m0 = (lt / tn) * BM;
    it.n0 = (lt % tn) * bn;
    const int nk_all = (int)((ga.p[it.pi].K + BKT - 1) / BKT);
    it.kt0 = it.split * ga.kt_per_split;
    const int kt1 = (it.kt0 + ga.kt_per_split < nk_all) ? it.kt0 + ga.kt_per_split : nk_all;
    it.nkt = kt1 - it.kt0;
    return it;
}

// The persistent kernels of gemm3.hip / gemm4.hip decode their items once per workgroup (lane q = the workgroup's q-th item) for ANY
// grid size G (a multiple of 8 when the XCD partition is on: G / 8 workgroups per XCD): 256 = the whole chip, 240 = 30 per XCD, which
// leaves two CUs of every XCD to a resident collective kernel (mr_set_option "gemm_cus").  Single problem, no split-K.
__device__ __forceinline__ void item_pp(const G256Args& ga, int bperm, int q, int G, int bn, int& m0, int& n0, int bm = BM) {
    if (ga.xmode == 1) {
        const int px_ = G >> 3;                                             // workgroups per XCD
        const int x = bperm / px_, sl = bperm - x * px_;
        const int xi = x / ga.py, xj = x - xi * ga.py;
        const int m_lo = xi * ga.tm / ga.px, hm = (xi + 1) * ga.tm / ga.px - m_lo;
        const int n_lo = xj * ga.tn / ga.py, hn = (xj + 1) * ga.tn / ga.py - n_lo;
        const int qq = q * px_ + sl;
        const int xp_ = ga.xpanel > 0 ? ga.xpanel : XPANEL;
        const int gw = hn < xp_ ? hn : xp_;
        const int panel = qq / (hm * gw), rem = qq - panel * hm * gw;
        const int left = hn - panel * gw, pw = left < gw ? left : gw;          // the last panel may be narrower
        const int m = rem / (pw > 0 ? pw : 1), n = panel * gw + rem - m * pw;
        m0 = qq < hm * hn ? (m_lo + m) * bm : -1;
        n0 = (n_lo + n) * bn;
    } else {
        const int w = bperm + q * G, tn = ga.tiles_n[0];
        m0 = w < ga.nwork ? (w / tn) * bm : -1;
        n0 = (w % tn) * bn;
    }
}

}  // namespace g256
