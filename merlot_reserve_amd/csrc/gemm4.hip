// Fourth-generation bf16 MFMA GEMM for gfx950, "NT" operands:  C[M,N] = A[M,K] . B[N,K]^T, both K-contiguous.  Same interface, tile
// grid, operand rings and epilogue as the ping-pong kernel (gemm3.hip); what changes is who multiplies:
//
// ONE wave per SIMD.  256 x {256,192} x 64 tiles, FOUR waves as 2 (M) x 2 (N): a wave owns 128 x {128,96} of the tile = 8 x {8,6}
// accumulators of v_mfma_f32_16x16x32_bf16 (256 | 192 registers, in AGPRs -- the wave has the SIMD's whole 512-entry file), i.e. per
// k-tile it reads (128 + 128) rows of fragments for 128 MFMAs where an eight-wave 128 x 64 block reads (128 + 64) for 64: the LDS
// bytes per FLOP that bound the eight-wave k-loops drop by a third (192 -> 128 KiB of ds_read_b128 per
// k-tile and CU).  This is the shape of the vendor library's kernels for these problems (MT256x256x64 / MT256x192x64, 4 waves).
// With nobody else on the SIMD to fill its gaps, the wave's own instruction stream is software-pipelined.  A k-tile is two halves
// of 64 | 48 MFMAs (k = 0..31, 32..63 of the tile):
//     half 0:  MFMAs of k-half 0   beside  the 16 | 14 ds_read_b128 of k-half 1's fragments  and the 8 LDS-DMA requests of A(t+2)
//     s_waitcnt vmcnt(8) (own pieces of k-tile t+1 landed: all but the newest 8) . lgkmcnt(0) . s_barrier
//     half 1:  MFMAs of k-half 1   beside  the reads of k-tile t+1's k-half 0                and the 8 | 6 requests of B(t+2)
// Every read and request is independent of the half's MFMAs and sits behind a PAIR of MFMAs: at most two other instructions
// between two MFMAs, in SOURCE order pinned by scheduling barriers (the compiler's own placement, also under sched_group_barrier,
// bunches the requests with their M0 writes and the scalar bookkeeping: 846 vs 703 us on 8192^3).
// ONE barrier per k-tile (128 | 96 MFMAs): behind it every wave's pieces of k-tile t+1 are visible, and every read of k-tile t's
// stages has been retired (the wait in front of it) -- B(t+2) then refills k-tile t's B stage, A(t+2) the stage k-tile t-1 was
// read from: safe by construction.  Rings as in gemm3: A 3 stages x 32 KiB, B 2 stages x {32,24} KiB; every wave issues 8 A +
// {8,6} B pieces per k-tile UNCONDITIONALLY -- a cursor that has run out of tiles requests from an empty descriptor (zero fill
// into a stage nobody reads) -- which keeps the k-loop free of branches and the counted vmcnt exact.  One issue cursor serves both
// operands (always two k-tiles ahead); it moves at the k-tile's end, where an item switch rebuilds the two descriptors (base = the
// item's first row) in a scalar branch outside the MFMA stream, so a request inside it costs its M0 write and nothing else.
// The cursor runs across output tiles, and so do the fragment reads: the next tile's first fragments are in registers before the
// epilogue starts.  Instantiated for the bias / residual / plain epilogues; rotary scales, GELU + gelu' and aux + column sums need
// more registers than the 256 VGPRs beside the accumulators leave (spill reloads inside the k-loop) and stay with gemm3.
// Measured (sustained, interleaved with gemm3): 15424 x 768 x 3072 59.4 vs 63.7 us, x 2304 46.2 vs 50.1, 15424 x 2304 x 768
// plain 57.0 vs 61.6, 8192^3 706 vs 738 (the vendor library: 683); K = 768, N = 3072 ties (72.5 vs 72.2): there the epilogue,
// which no second wave hides here either, is a third of the tile.
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include "gemm256_sched.h"
#include "mr_options.h"

namespace g4 {

using namespace g256;

#ifndef MR_G3_AUX_C
#define MR_G3_AUX_C 0
#endif
#ifndef MR_G3_AUX_C2
#define MR_G3_AUX_C2 0
#endif

template <int BN> struct Geo4 {
    static_assert(BN == 256 || BN == 192, "tile widths");
    static constexpr int WCOLS = BN / 2, NJ = WCOLS / 16;
    static constexpr int STAGE_A = 256 * 128, NSTAGE_A = 3;
    static constexpr int STAGE_B = BN * 128, NSTAGE_B = 2;
    static constexpr int OFF_B = NSTAGE_A * STAGE_A;
    static constexpr int LDS_BYTES = OFF_B + NSTAGE_B * STAGE_B;
    static constexpr int PB = BN / 32;                       // 1-KiB B pieces per wave and k-tile (A: 8)
};
static_assert(Geo4<256>::LDS_BYTES == 160 * 1024 && Geo4<192>::LDS_BYTES == 144 * 1024, "LDS budget");

template <int BN, int MODE>
__global__ __launch_bounds__(256) void gemm4_kernel(const G256Args ga) {
    using GEO = Geo4<BN>;
    constexpr int WCOLS = GEO::WCOLS, NJ = GEO::NJ, PB = GEO::PB;
    constexpr int STAGE_A = GEO::STAGE_A, STAGE_B = GEO::STAGE_B, OFF_B = GEO::OFF_B;
    // store instructions a wave issues per tile (all unconditional, see gemm3_epilogue.inc); the first k-tile behind an epilogue may
    // leave them -- and the 8 A pieces that are always allowed in flight -- outstanding, up to the 6-bit counter's range
    constexpr int NST = 8 * (NJ / 2) * (MODE == 2 ? 2 : 1);
    constexpr int WAIT_FIRST = (8 + NST <= 63) ? 8 + NST : 63;
    __shared__ __attribute__((aligned(16))) char smem[GEO::LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int g = lane >> 4, li = lane & 15;

    const int G = gridDim.x;
    const int xcd = blockIdx.x & 7, qd = G >> 3, rm = G & 7;
    const int bperm = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (blockIdx.x >> 3);
    // the workgroup's items (output tiles), decoded once: lane q holds item q (m0 < 0 = none); <= 64 items per workgroup (host check)
    int m0v, n0v;
    item_pp(ga, bperm, lane, G, BN, m0v, n0v);
    // (the readlane is unconditional: a convergent operation under a condition would put a branch into the k-loop's issue slot)
    auto item_m0 = [&](int q) -> int { const int r = __builtin_amdgcn_readlane(m0v, q & 63); return q < 64 ? r : -1; };
    auto item_n0 = [&](int q) -> int { return __builtin_amdgcn_readlane(n0v, q & 63); };
    if (item_m0(0) < 0) return;
    const mr_gemm_args& p0 = ga.p[0];
    const int nkt = (int)(p0.K >> 6);
    const unsigned lda2 = (unsigned)p0.lda * 2u, ldb2 = (unsigned)p0.ldb * 2u;
    // Operand descriptors are rebuilt per k-tile for the cursor's item: base = the item's first row, num_records = what is left of the
    // operand behind it (rows >= M / >= N read as zeros; 0 records = a cursor without an item: every piece zero-filled).  The per-lane
    // offsets are then the same for every item and k-tile -- no vector add per DMA inside the MFMA stream (one wave per SIMD: every
    // instruction of the k-loop takes one of this wave's issue slots).
    const char* const Aptr = static_cast<const char*>(p0.A);
    const char* const Bptr = static_cast<const char*>(p0.B);
    const int a_extent = (int)(((p0.M - 1) * p0.lda + p0.K) * 2), b_extent = (int)(((p0.N - 1) * p0.ldb + p0.K) * 2);
    auto rel = [&](int piece, unsigned ld2) -> unsigned {      // byte offset of this lane's chunk of 1-KiB piece `piece` in an item at row 0
        const int row = piece * 8 + (lane >> 3);
        return (unsigned)row * ld2 + (unsigned)(((lane & 7) ^ swz_kc(row)) * 16);
    };
    unsigned pao[8], pbo[8];      // (pbo[PB]: the host pass then drops the kernel stub without a diagnostic)
#pragma unroll
    for (int j = 0; j < 8; ++j) pao[j] = rel(wave * 8 + j, lda2);
#pragma unroll
    for (int j = 0; j < PB; ++j) pbo[j] = rel(wave * PB + j, ldb2);

    // ---- ONE issue cursor for both operands, two k-tiles ahead of the compute cursor, across item boundaries: A(t+2) is requested in
    // half 0 of k-tile t, B(t+2) in half 1, and the cursor moves at the k-tile's end -- the item switch (descriptors rebuilt) is a
    // scalar branch there, outside the MFMA stream; inside it a request costs the M0 write and the DMA instruction, nothing else ----
    int ik = 0, qi = 0, ista = 0, istb = 0;
    __amdgpu_buffer_rsrc_t ra_c, rb_c;
    auto set_item = [&](int q_) {
        const int mi = item_m0(q_), ni = item_n0(q_);
        const int aoff = (mi >= 0 ? mi : 0) * (int)lda2, boff = (mi >= 0 ? ni : 0) * (int)ldb2;
        ra_c = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Aptr) + aoff, 0, mi >= 0 ? a_extent - aoff : 0, 0x00020000);
        rb_c = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Bptr) + boff, 0, mi >= 0 ? b_extent - boff : 0, 0x00020000);
    };
    set_item(0);
#define G4_ISSUE_A()                                                                                                    \
    do {                                                                                                                \
        char* st_ = smem + ista * STAGE_A + wave * 8192;                                                                \
        const unsigned so_ = (unsigned)ik * 128u;                                                                       \
        _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) MR_DMA(ra_c, MR_LDS_PTR(void, st_ + j_ * 1024), 16, pao[j_], so_, 0, 0); \
    } while (0)
#define G4_ISSUE_B()                                                                                                    \
    do {                                                                                                                \
        char* st_ = smem + OFF_B + istb * STAGE_B + wave * (PB * 1024);                                                 \
        const unsigned so_ = (unsigned)ik * 128u;                                                                       \
        _Pragma("unroll") for (int j_ = 0; j_ < PB; ++j_) MR_DMA(rb_c, MR_LDS_PTR(void, st_ + j_ * 1024), 16, pbo[j_], so_, 0, 0); \
    } while (0)
#define G4_ADVANCE()                                                                                                    \
    do {                                                                                                                \
        ista = (ista == 2) ? 0 : ista + 1;                                                                              \
        istb ^= 1;                                                                                                      \
        if (++ik == nkt) {                                                                                              \
            ik = 0;                                                                                                     \
            set_item(++qi);                                                                                             \
        }                                                                                                               \
    } while (0)
#define G4_SB() __builtin_amdgcn_sched_barrier(0)
    int qc = 0, csa = 0, csb = 0;
    int cm0 = item_m0(0), cn0 = item_n0(0);
    bool have_stores = false;          // an epilogue's stores may be in flight (never before the workgroup's first tile)
    // the bias is the accumulators' initial value; a tile's bias is fetched in the previous tile's epilogue, ahead of its stores
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

    // prologue: B(0), A(0), B(1), A(1); k-tile 0 has landed when only the second pair is in flight
    G4_ISSUE_B(); G4_ISSUE_A(); G4_ADVANCE(); G4_ISSUE_B(); G4_ISSUE_A(); G4_ADVANCE();
    wait_vmcnt<PB + 8>();
    G4_SB();
    __builtin_amdgcn_s_barrier();
    G4_SB();
    // The next tile's first fragments are read under the last MFMAs of the current one and stay in registers across the epilogue
    // (64 registers): this kernel is instantiated for the epilogues that leave room for them (bias, residual, plain), the others
    // (rotary scales, GELU + gelu', aux + column sums) stay with the ping-pong kernel, whose two waves per SIMD hide their latencies.
    static_assert(MODE == 0 || MODE == 3 || MODE == 5, "epilogues of the one-wave-per-SIMD kernel");
    bf16x8 a[8][2], b[NJ][2];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i][0] = frag<false, 256, 64>(smem, wr * 128 + i * 16, 0, lane);
#pragma unroll
    for (int j = 0; j < NJ; ++j) b[j][0] = frag<false, 256, 64>(smem + OFF_B, wc * WCOLS + j * 16, 0, lane);

    while (cm0 >= 0) {
        // accumulators TRANSPOSED (mfma(B-frag, A-frag)): the lane holds C[m = .. + li][n = .. + 4 g + r]
        // bias modes: the bias is the accumulators' initial value; the others start from the MFMA's inline-constant zero C operand
        // (no 256 v_accvgpr_write per tile on a SIMD that has nothing else to run)
        f32x4 acc[8][NJ];
        if constexpr (BIAS) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = binit[j];
        }

        auto ktile = [&](auto first_c) {
            constexpr bool FIRST = decltype(first_c)::value;       // the tile's first k-tile: the previous tile's stores may be in flight
            const char* As = smem + csa * STAGE_A;
            const char* Bs = smem + OFF_B + csb * STAGE_B;
            csa = (csa == 2) ? 0 : csa + 1;
            csb ^= 1;
            const char* An = smem + csa * STAGE_A;
            const char* Bn = smem + OFF_B + csb * STAGE_B;
            // One half = 8 rounds (one per 16-row block i) of NJ MFMAs in pairs; behind a pair: one fragment read of the OTHER k-half
            // (two per round, three in rounds 0 and 1 of a 256-wide tile, none in round 7: the data is needed right behind the half)
            // or one LDS-DMA request (round r: piece r).  The order is the SOURCE order, pinned by a scheduling barrier behind
            // every slot: with one wave on the SIMD the matrix pipe idles whenever more than ~3 other instructions sit between two
            // MFMAs, and the compiler's own placement (also with sched_group_barrier) bunches the requests and their M0 writes.
            auto half = [&](auto kk_c, const char* Ar, const char* Br, auto is_a, auto np_c) {
                constexpr int KK = decltype(kk_c)::value, KR = 1 - KK;          // multiply k-half KK, read fragments of k-half KR from Ar / Br
                constexpr bool ISA = decltype(is_a)::value;
                constexpr int NP = decltype(np_c)::value;                        // requests of this half (A: 8, B: PB)
                char* const st_ = ISA ? smem + ista * STAGE_A + wave * 8192 : smem + OFF_B + istb * STAGE_B + wave * (PB * 1024);
                const unsigned so_ = (unsigned)ik * 128u;
                auto rd = [&](int n) {              // the half's n-th fragment read, in the order the next multiply needs them: a0, b0 .. b(NJ-1), a1 .. a7
                    if (n >= 8 + NJ) return;
                    if (n == 0) a[0][KR] = frag<false, 256, 64>(Ar, wr * 128, KR, lane);
                    else if (n <= NJ) b[n - 1][KR] = frag<false, 256, 64>(Br, wc * WCOLS + (n - 1) * 16, KR, lane);
                    else a[n - NJ][KR] = frag<false, 256, 64>(Ar, wr * 128 + (n - NJ) * 16, KR, lane);
                };
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int rd0 = (NJ == 8) ? (r < 2 ? 3 * r : 2 * r + 2) : 2 * r;      // first read of round r
                    const int nrd = (r == 7) ? 0 : (NJ == 8 && r < 2) ? 3 : 2;
#pragma unroll
                    for (int pr = 0; pr < NJ / 2; ++pr) {
                        constexpr bool ZC = FIRST && KK == 0 && !BIAS;          // the tile's very first multiply
                        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
                        acc[r][2 * pr] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[2 * pr][KK], a[r][KK], ZC ? zero4 : acc[r][2 * pr], 0, 0, 0);
                        acc[r][2 * pr + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[2 * pr + 1][KK], a[r][KK], ZC ? zero4 : acc[r][2 * pr + 1], 0, 0, 0);
                        G4_SB();
                        if (pr == 0 && nrd >= 1) rd(rd0);
                        if (pr == 1 && r < NP) {
                            if constexpr (ISA) MR_DMA(ra_c, MR_LDS_PTR(void, st_ + r * 1024), 16, pao[r], so_, 0, 0);
                            else MR_DMA(rb_c, MR_LDS_PTR(void, st_ + r * 1024), 16, pbo[r], so_, 0, 0);
                        }
                        if (pr == 2 && nrd >= 2) rd(rd0 + 1);
                        if (pr == 3 && nrd >= 3) rd(rd0 + 2);
                        G4_SB();
                    }
                }
            };
            // ---------------- half 0: k = 0..31 of the tile; reads k = 32..63; requests A(t+2) into the stage k-tile t-1 was read from ----------------
            half(std::integral_constant<int, 0>{}, As, Bs, std::true_type{}, std::integral_constant<int, 8>{});
            // k-tile t+1 has landed (every group issued before the newest 8 A pieces) and every read of k-tile t's stages is retired
#ifndef MR_G4_NOSYNC        /* timing-only diagnostic build (wrong results): no wait, no barrier */
            if (FIRST && have_stores) wait_vmcnt<WAIT_FIRST>();
            else wait_vmcnt<8>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            G4_SB();
            __builtin_amdgcn_s_barrier();
            G4_SB();
#endif
            // ---------------- half 1: k = 32..63; reads k-tile t+1's k = 0..31; requests B(t+2) into the stage k-tile t was read from ----------------
            half(std::integral_constant<int, 1>{}, An, Bn, std::false_type{}, std::integral_constant<int, PB>{});
            G4_ADVANCE();
        };
        ktile(std::true_type{});
        for (int t = 1; t < nkt; ++t) ktile(std::false_type{});

        // ---------------- epilogue (bf16 output; registers + ordinary loads, no LDS) ----------------
        {
            const int wrow0 = cm0 + wr * 128, wcol0 = cn0 + wc * WCOLS;
            constexpr int MR_EPI_MI = 8;
#define MR_EPI_ROW_FENCE() __builtin_amdgcn_sched_barrier(0)
#define MR_EPI_FULL_LINES 0      // (full-line stores: the trade of halves costs this kernel spilled registers at its allocation limit; gemm3.hip only)
#include "gemm3_epilogue.inc"
#undef MR_EPI_FULL_LINES
#undef MR_EPI_ROW_FENCE
#ifndef MR_G3_NOSTORE
            have_stores = true;
#endif
        }
        ++qc;
        cm0 = item_m0(qc);
        cn0 = item_n0(qc);
    }
    // the ring's last requests (zero fill) and the stores retire before the LDS is released
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

}  // namespace g4


// Same eligibility and tile plan as the ping-pong kernel (mr_gemm3_eligible / the plan below mirrors mr_gemm3_launch): the caller
// has checked mr_gemm3_eligible(a).
// MODE 5 = plain product without bias (dgrads): mode 0's epilogue, and the tile's first multiply takes the MFMA's inline zero.
bool mr_gemm4_takes(const mr_gemm_args* a) { return !a->c2 && !a->rot_tab && !a->aux; }

// Same eligibility and tile plan as the ping-pong kernel: the caller (mr_gemm3_launch) has built `ga` and checked mr_gemm4_takes(a).
int mr_gemm4_launch(const mr_gemm_args* a, int bn, const g256::G256Args& ga, int64_t gsz, hipStream_t s) {
    const int mode = a->residual ? 3 : a->bias ? 0 : 5;
    MR_CHECK_ARG(!(bn == 256 && mode == 0), "mr_gemm (gemm4): the 256-wide bias mode is not built (register budget); route it to the ping-pong kernel");
    dim3 grid((unsigned)gsz), block(256);
#define G4_LAUNCH(MODE)                                                                               \
    do {                                                                                              \
        if (bn == 256) hipLaunchKernelGGL((g4::gemm4_kernel<256, MODE>), grid, block, 0, s, ga);      \
        else hipLaunchKernelGGL((g4::gemm4_kernel<192, MODE>), grid, block, 0, s, ga);                \
    } while (0)
    switch (mode) {
        case 0: hipLaunchKernelGGL((g4::gemm4_kernel<192, 0>), grid, block, 0, s, ga); break;
        case 3: G4_LAUNCH(3); break;
        default: G4_LAUNCH(5); break;
    }
#undef G4_LAUNCH
    mr_note_route("g4::gemm4_kernel<%d,%d>", bn, mode);
    MR_CHECK_LAUNCH("mr_gemm (gemm4)");
    return MR_OK;
}
