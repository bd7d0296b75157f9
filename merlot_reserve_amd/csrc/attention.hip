// Fused multi-head attention, forward and backward, head dim 64, for gfx950 (MI355X).
//
// Replaces apply_attention (mreserve/modeling.py:188-200) + flax dot_product_attention_weights:
//   scores = (q / 8) . k + bias,  bias = 0 where allowed else -1e10 (modeling.py:353-356),  softmax over keys,  o = P v.
// The [S,S] mask of modeling.py:743-756 is never materialised: it is allowed(i,j) = code[i] == code[j] && code[i] >= 0
// with one int32 per position (valid ? video_src : -1).
//
// Layout trick (all three kernels): scores are computed TRANSPOSED, S^T = K . Q^T, with v_mfma_f32_16x16x32_bf16, so
// a lane holds, for ONE query (its accumulator column), 4 consecutive keys per 16-key block.  Those accumulators,
// converted to bf16, ARE the B operand of the next product (O^T = V^T . P^T, dQ^T = K^T . dS^T): the MFMA k index
// is only a summation index, so it is permuted identically on both operands
//     key(g, j) = 32 t + 16 (j >> 2) + 4 g + (j & 3)        (g = lane >> 4, j = 0..7, t = k-step)
// and the other operand (V^T, K^T, dO^T, Q^T) is read with ds_read_b64_tr_b16 from a row-major LDS tile.
// Softmax statistics are then per lane (no shuffles except a 4-lane max/sum) and nothing goes through LDS twice.
//
// Pipeline: a wave owns 16*QB queries (keys in the dK/dV kernel); the K/V (Q/dO) tiles of the inner loop are double
// buffered in LDS and filled by LDS-DMA (buffer_load ... lds, 4 wave instructions per tile pair): the next tile is requested
// straight into the other buffer at the top of the iteration and waited for (vmcnt(0)) at its end -- one barrier per tile, no
// staging registers, no LDS stores, rows past the sequence zero-filled by the buffer descriptor.  (Round 2: the register-staged
// version's global-load issue and LDS stores were ~1 200 of a tile's ~4 300 wave cycles in the forward kernel, and its 16 staging
// registers kept that kernel at two waves per SIMD; with three: joint 79.5 -> 68.1 us, ViT 37.3 -> 31.5 us.)
//
// Per-score vector work of the BACKWARD kernels (~10 vector instructions per score against 1/16 of an MFMA):
//   * the 1/8 of "query / sqrt(depth)" is folded into the Q (dQ) or K (dK/dV) fragments once per workgroup --
//     exact in bf16 -- so the MFMA result IS the reference's score;
//   * p = exp2(fma(s, log2 e, -m log2 e)): one FMA + one exponential per score;
//   * the additive bias is one compare + add + select against a staged per-key pair (code, bias-if-not-allowed);
//   * delta = rowsum(dO * O) is computed by the dQ kernel from the fragments it holds anyway (no separate pass).
// The reference's bias for a disallowed pair is -1e10 (modeling.py:353-356), whose only observable effects in fp32 are
// (a) weight exactly 0 for a disallowed key of a row that has an allowed key and (b) a UNIFORM softmax over all S keys for
// a row with no allowed key (a PAD query: every score rounds to the same -1e10).  Both are kept exactly with the bias
// constant -2^33 in the backward kernels (bias * log2 e is exact; a PAD row is recognised by its LSE < PAD_LSE and given
// P = 1/S); the forward keeps the literal -1e10 (in the exp2 domain), for which every score of a PAD row is the same number.
#include <stdlib.h>
#include <type_traits>
#include "mr_common.h"
#include "mr_options.h"

// diagnostic builds (scripts/build_diag.sh): -DMR_ATTN_OCC=n / -DMR_ATTN_OCC_DQ=n set the waves per SIMD the register allocator targets
// (the forward kernel fits three by itself since its K / V tiles are LDS-DMA staged: 162-168 registers)
#ifndef MR_ATTN_OCC_DQ      /* masked dQ kernel: 184 registers by itself; held to three waves per SIMD it spills a few and is still faster (joint backward 181 -> 175 us) */
#define MR_ATTN_OCC_DQ 3
#endif
#ifndef MR_ATTN_OCC_DQ_UNMASKED   /* unmasked dQ kernel: fits three waves per SIMD since the ragged last tile is peeled */
#define MR_ATTN_OCC_DQ_UNMASKED 3
#endif
#ifndef MR_ATTN_OCC
#define MR_ATTN_OCC 2
#endif
// -DMR_ATTN_STAMPS (diagnostic build): s_memtime stamps of the forward kernel's phases for the first tiles of the first workgroups,
// into a buffer of their own that mr_diag_attn_stamps() copies out; no stamp executes in the product build
#ifdef MR_ATTN_STAMPS
#ifndef MR_ATTN_STAMP_K       /* which kernel stamps: 0 = every instrumented one (the last launch wins), 1 forward, 2 dQ, 3 dK / dV, 4 one-pass backward */
#define MR_ATTN_STAMP_K 0
#endif
__device__ unsigned long long g_attn_stamps[512 * 16 * 8];
#define MR_ASTAMP(slot)                                                                                        \
    do {                                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        unsigned long long t_;                                                                                 \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                             \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        if ((MR_ATTN_STAMP_K == 0 || MR_ATTN_STAMP_K == stamp_id) && lane == 0 && wave == 0 && blockIdx.x < 512 && t < 16) g_attn_stamps[(blockIdx.x * 16 + t) * 8 + (slot)] = t_; \
    } while (0)
// workgroup begin / end (tile row 15 of the workgroup's stamps, slots 0 / 1; the loop-top stamp of tile 0 marks the prologue's end)
#define MR_ASTAMP_WG(slot)                                                                                     \
    do {                                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        unsigned long long t_;                                                                                 \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                             \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        if ((MR_ATTN_STAMP_K == 0 || MR_ATTN_STAMP_K == stamp_id) && lane == 0 && wave == 0 && blockIdx.x < 512) g_attn_stamps[(blockIdx.x * 16 + 15) * 8 + (slot)] = t_; \
    } while (0)
#else
#define MR_ASTAMP(slot) do {} while (0)
#define MR_ASTAMP_WG(slot) do {} while (0)
#endif

namespace {

constexpr int TK = 64;        // inner tile (keys in fwd / dQ, queries in dK/dV)
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
constexpr float NEG_BIAS = -8589934592.0f;         // -2^33: stands in for the reference's -1e10 bias (see above)
constexpr float PAD_LSE = 0.25f * NEG_BIAS;        // an LSE below this marks a row with no allowed key
constexpr int CODE_NONE = -2;                      // key beyond the sequence
constexpr int CODE_PADQ = -3;                      // query with no allowed key (PAD, or beyond the sequence): matches no key code
constexpr int CODE_MIXED = INT32_MIN;              // a 64-position tile (a wave's positions) without ONE common code: the general path
constexpr int CODE_DEAD = INT32_MIN + 1;           // backward: a query tile whose rows all have a zero upstream gradient: skipped by every wave
// ---- block-structured masks (round 5).  The mask is code[i] == code[j] && code[i] >= 0 and real sequences carry their codes in RUNS (one video
// source after the other, a PAD tail or gap), so most (wave, 64-position tile) pairs are UNIFORM: every pair allowed -- then the per-score
// compare / select / add of the bias is dead weight and the tile runs the unmasked instruction sequence (bit-identical: the bias is exactly 0) --
// or every pair disallowed while every query of the wave is a valid one (so each has an allowed key: itself): then every weight of the
// tile is exactly 0 (forward: exp2 of -1e10 log2 e below a finite maximum; or, before the row's first allowed key, wiped by the rescale factor
// exp2(-1e10..) = 0 at that key) and the tile is SKIPPED, bit-identically.  Classification: the staging wave votes on the tile's codes
// (one ballot) and leaves its class in LDS beside the codes; each wave votes once on its own positions; PAD queries (uniform rows over all
// S keys), ragged last tiles and mixed tiles take the general path.  Forward, a wave whose queries are ALL PAD (TILE_UNIFORM): every score of
// a tile inside the sequence rounds to the same -1e10 log2 e, so m = that constant, every weight is exp2(0) = 1 and each lane's row sum
// grows by its 16 keys: the Q K^T product and the softmax arithmetic are skipped and the P V product runs on a fragment of ones -- the same
// MFMA sequence on the same operands as the general path, bit-identical.
enum : int { TILE_GENERAL = 0, TILE_FAST = 1, TILE_SKIP = 2, TILE_UNIFORM = 3 };
__device__ __forceinline__ int tile_mode(bool wave_uniform, int wave_code, int tile_code) {
    if (!wave_uniform || tile_code == CODE_MIXED) return TILE_GENERAL;
    return tile_code == wave_code ? TILE_FAST : TILE_SKIP;
}
// class of the 64 positions the staging wave (all 64 lanes active) holds one code each of: their common code, CODE_MIXED otherwise or when
// the tile reaches beyond the sequence
__device__ __forceinline__ int tile_class(int c, bool tile_inside) {
    const int c0 = __builtin_amdgcn_readfirstlane(c);
    return (tile_inside && !__any(c != c0)) ? c0 : CODE_MIXED;
}

typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;

__device__ __forceinline__ bf16x8 pack_acc_pair(const f32x4& a, const f32x4& b) {
    bf16x8 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) { r[e] = (__bf16)a[e]; r[4 + e] = (__bf16)b[e]; }
    return r;
}

// ---- LDS-DMA staged tiles (all three kernels).  A [64 rows][64 dims] bf16 tile is an UNPADDED 8-KiB image (LDS-DMA writes 1 KiB per
// wave instruction, lane-linear: piece p = rows 8p .. 8p+7, lane l = row 8p + (l >> 3), 16-byte slot l & 7), so the bank-conflict
// padding a register-staged tile would use becomes an XOR on the SOURCE chunk: slot = chunk ^ 2 * ((row >> 1) & 3).  The XOR is even, so the
// two chunks a ds_read_b64_tr_b16 row segment spans stay adjacent; with it both read forms are conflict-free (checked over the
// hardware's lane groups: ds_read_b128 rows i = lane & 15 / chunk 4 dd + g, and the transposed 8-byte reads of tr_frag_d_issue).
// No staging registers, no ds_write, no per-load bounds branch: rows past the sequence lie beyond the buffer descriptor's extent and
// arrive as zeros.
constexpr int TILE_B = TK * 64 * 2;      // 8 KiB
__device__ __forceinline__ int dswz(int row) { return 2 * ((row >> 1) & 3); }
// byte offset (from the tile's first row, tile 0) of the 16-byte chunk lane `lane` of piece `p` fetches
__device__ __forceinline__ unsigned dma_src(int p, int lane, int64_t ld) {
    const int row = p * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ dswz(row);
    return (unsigned)((row * ld + chunk * 8) * 2);
}
__device__ __forceinline__ bf16x8 row_frag_d(const char* tile, int row0, int dd, int lane) {
    const int g = lane >> 4, row = row0 + (lane & 15);
    return *reinterpret_cast<const bf16x8*>(tile + row * 128 + (((dd * 4 + g) ^ dswz(row)) << 4));
}
// The same read as inline asm (the caller waits with lgkmcnt): with a ring of three tiles the compiler can no longer prove that the tile being read is
// not the one an LDS-DMA request in flight writes, and puts s_waitcnt vmcnt(0) in front of every plain LDS read of the ring -- which serialises the
// prefetch it is meant to overlap (the two-slot kernels' b / b ^ 1 indexing it does see through).
__device__ __forceinline__ void row_frag_d_issue(const char* tile, int row0, int dd, int lane, u32x4& out) {
    const int g = lane >> 4, row = row0 + (lane & 15);
    const unsigned a0 = (unsigned)(uintptr_t)MR_LDS_PTR(const char, tile + row * 128 + (((dd * 4 + g) ^ dswz(row)) << 4));
    asm volatile("ds_read_b128 %0, %1" : "=v"(out) : "v"(a0));
}
// transposed fragment (rows row0 + 4g + {0..3} and row0 + 16 + 4g + {0..3}; 16 columns from col0; lane receives column lane & 15)
// from a DMA image.  Inline asm: with LDS-DMA in flight hipcc would put s_waitcnt vmcnt(0) in front of
// the ds_read_tr builtin (it cannot tell it from the DMA's LDS writes) and drain the prefetch; the caller waits with lgkmcnt(0).
__device__ __forceinline__ void tr_frag_d_issue(const char* tile, int row0, int col0, int lane, s16x4& lo, s16x4& hi) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int row = row0 + 4 * g + q, col = col0 + 4 * p;
    const unsigned a0 = (unsigned)(uintptr_t)MR_LDS_PTR(const char, tile + row * 128 + ((((col >> 3)) ^ dswz(row)) << 4) + (col & 7) * 2);
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a0));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(hi) : "v"(a0));       // row + 16: same swizzle
}
__device__ __forceinline__ bf16x8 tr_join(const s16x4& lo, const s16x4& hi) {
    s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, both);
}

// per-key metadata of tile position k: its code and the bias a query with a DIFFERENT code adds to the score
__device__ __forceinline__ void key_meta(int64_t k, int64_t S, const int32_t* __restrict__ code_seq, bool masked, int& c, float& nb) {
    const int cr = masked ? code_seq[k < S ? k : S - 1] : 0;      // (no branch around the load: a batch of these stays one basic block, all in flight at once)
    c = k < S ? cr : CODE_NONE;
    nb = k < S ? NEG_BIAS : -INFINITY;
}

// fragment * 2^-3: the query / sqrt(depth) of flax's dot_product_attention_weights (exact in bf16)
__device__ __forceinline__ bf16x8 scale_eighth(u32x4 raw) {
    bf16x8 v = __builtin_bit_cast(bf16x8, raw);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (__bf16)((float)v[e] * 0.125f);
    return v;
}

// s[r] = allowed ? s[r] : s[r] + nk[r]   (allowed = the key's code equals the query's; see the header for the constants)
__device__ __forceinline__ void add_bias(f32x4& s, const i32x4& ck, const f32x4& nk, int cq) {
#pragma unroll
    for (int r = 0; r < 4; ++r) s[r] = (ck[r] == cq) ? s[r] : s[r] + nk[r];
}

// sum over the 16 lanes of a DPP row (lanes 16 g .. 16 g + 15): quad butterflies, then the two mirrors (no LDS)
__device__ __forceinline__ float row16_sum(float v) {
#define MR_DPP(x, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, 0xF, 0xF, true))
    v += MR_DPP(v, 0xB1);      // quad_perm [1,0,3,2]
    v += MR_DPP(v, 0x4E);      // quad_perm [2,3,0,1]
    v += MR_DPP(v, 0x141);     // row_half_mirror
    v += MR_DPP(v, 0x140);     // row_mirror
#undef MR_DPP
    return v;
}

// Column sums of what a workgroup stored (the qkv bias gradient, flax DenseGeneral bias: M:228): cs[db][r] = this lane's sum over
// its own rows of column d = 16 db + 4 g + r (the bf16-ROUNDED values, like a column-sum pass over the stored tensor).  Rows across
// the 16 lanes of a DPP row, then the 4 waves through `red`; thread d < 64 writes dst[d].  All threads must call it.
__device__ __forceinline__ void block_colsum_store(f32x4 (&cs)[4], float (*red)[64], float* __restrict__ dst, int tid) {
    const int lane = tid & 63, wave = tid >> 6, g = lane >> 4, i = lane & 15;
#pragma unroll
    for (int db = 0; db < 4; ++db) {
#pragma unroll
        for (int r = 0; r < 4; ++r) cs[db][r] = row16_sum(cs[db][r]);
        if (i == 0) *reinterpret_cast<f32x4*>(&red[wave][db * 16 + g * 4]) = cs[db];
    }
    __syncthreads();
    if (tid < 64) dst[tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

// Row of the "rotary" scale table for position `pos` (< S) of a sequence whose first row is seq_rot0 = (seq * S) % rot_rows: the table has rot_rows
// rows and repeats (one row per position of a sequence: rot_rows = S; or one per position of the batch).  32-bit, and without a division when the
// table is at least a sequence long -- as (seq * S + pos) % rot_rows in 64 bits this was ~60 vector instructions per use, twice per query tile in
// the one-pass backward kernel's loop.
__device__ __forceinline__ int rot_row_index(int seq_rot0, int pos, int rot_rows, bool table_shorter_than_seq) {
    int r = seq_rot0 + pos;
    if (table_shorter_than_seq) return r % rot_rows;
    return r >= rot_rows ? r - rot_rows : r;
}

// A wave's [16 NB rows][64 dims] bf16 tile, held as the MFMA leaves it (lane: dims 16 db + 4 g .. + 3 of row 16 nb + i), to global memory as WHOLE
// 128-byte rows, 16 bytes per lane, through a 2 NB KiB LDS region private to the wave ([row][128 B], 16-byte chunk c of row r at slot c ^ (r & 7)).  Stored
// straight from the MFMA layout a wave instruction wrote 16 rows x 32 bytes -- 16 partial lines: 7 900 of the one-pass backward's 61 000 cycles per
// workgroup (scripts/attn_bwd1_stamps.py).  Row `row0 + r` goes to gdst + r * ld (elements) if it is < n_rows.  No barrier: the wave's own waits order it.
template <int NB>
__device__ __forceinline__ void store_rows_via_lds(char* stg, const bf16x4 (&v)[NB][4], __bf16* gdst, int64_t ld, int row0, int64_t n_rows, int lane) {
    const int g = lane >> 4, i = lane & 15;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int row = nb * 16 + i;
#pragma unroll
        for (int db = 0; db < 4; ++db)
            *reinterpret_cast<bf16x4*>(stg + row * 128 + (((db * 2 + (g >> 1)) ^ (row & 7)) << 4) + (g & 1) * 8) = v[nb][db];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int it = 0; it < 2 * NB; ++it) {
        const int row = it * 8 + (lane >> 3), ch = lane & 7;
        const u32x4 w = *reinterpret_cast<const u32x4*>(stg + row * 128 + ((ch ^ (row & 7)) << 4));
        if (row0 + row < n_rows) *reinterpret_cast<u32x4*>(gdst + (int64_t)row * ld + ch * 8) = w;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                   // the reads are done before the region is written again
}

// XCD-aware block order.  The hardware hands consecutive workgroup ids to the 8 XCDs round-robin, so with the plain
// (block, head, sequence) grid the query blocks of ONE (sequence, head) -- which all stream the same K / V rows -- landed
// on different XCDs and every XCD's L2 fetched those rows for itself (rocprofv3 FETCH_SIZE: 3.7x the algorithmic bytes
// on the joint tower).  With this bijective remap the ids an XCD receives are consecutive, block index fastest.
struct AttnBlock { int64_t seq, h, blk; };
__device__ __forceinline__ AttnBlock attn_block(int nblk, int nh) {
    const unsigned n = gridDim.x, b = blockIdx.x;
    const unsigned xcd = b & 7, qd = n >> 3, rm = n & 7;
    const unsigned id = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (b >> 3);
    AttnBlock a;
    a.blk = id % nblk;
    const unsigned t = id / nblk;
    a.h = t % nh;
    a.seq = t / nh;
    return a;
}

// ---- forward: scores in the exp2 domain (pre-multiplied by log2 e / 8) with the reference's literal -1e10 bias.  (A forward
// with the backward kernels' cheaper per-score sequence -- 1/8 folded into Q, FMA + exp2, compare/add/select bias, rescale
// only when a maximum moved -- measured the same on the masked joint sequences and 9-30 % SLOWER on the short unmasked ones,
// where its extra registers cost a wave per SIMD: kept as it was.)
constexpr float SCALE2 = 0.125f * LOG2E;           // 1/sqrt(64) in the exp2 domain
constexpr float NEG_BIG2 = -1e10f * LOG2E;         // the reference's -1e10 bias, exp2 domain

// ------------------------------------------------------------------------------------------------ forward
template <int QB, bool MASKED>
__global__ __launch_bounds__(256, MR_ATTN_OCC) void attn_fwd_kernel(const __bf16* __restrict__ qkv, const int32_t* __restrict__ code,
                                                          __bf16* __restrict__ out, float* __restrict__ lse,
                                                          int64_t S, int64_t nh, const int tile_modes) {
    __shared__ __attribute__((aligned(16))) char Ks[2][TILE_B];          // LDS-DMA images (see dma_src)
    __shared__ __attribute__((aligned(16))) char Vs[2][TILE_B];
    // key codes (masked), the additive key bias (exp2 domain, see key_meta2) and the tile classes of the WHOLE sequence: dynamic LDS, written once in the
    // prologue by all four waves (8 bytes per padded position + 4 per tile) -- see the dQ kernel
    extern __shared__ __attribute__((aligned(16))) float fwd_dyn[];
    [[maybe_unused]] constexpr int stamp_id = 1;
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, i = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // provably wave-uniform: LDS-DMA bases stay scalar
    const AttnBlock ab_ = attn_block((int)((S + 64 * QB - 1) / (64 * QB)), (int)nh);
    const int64_t seq = ab_.seq, h = ab_.h, q0 = ab_.blk * (64 * QB);
    const int64_t H = nh * 64, ld = 3 * H;
    const __bf16* base = qkv + seq * S * ld;
    MR_ASTAMP_WG(0);

    // K / V of this (sequence, head): rows beyond S are beyond the descriptors' extent (zero fill); the tile advance is a scalar offset
    const __bf16* Kg = base + H + h * 64;
    const __bf16* Vg = base + 2 * H + h * 64;
    const int ext = (int)(((S - 1) * ld + 64) * 2);
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(Kg), 0, ext, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(Vg), 0, ext, 0x00020000);
    const unsigned so0 = dma_src(wave * 2, lane, ld), so1 = dma_src(wave * 2 + 1, lane, ld);
    const unsigned tile_step = (unsigned)(TK * ld * 2);
    // Per-key staging: the additive bias a score gets unless the key is allowed for the query, in the exp2 domain -- the reference's
    // literal -1e10 for a key inside the sequence (an unmasked tower allows every such key: 0), -inf beyond it -- and, masked only, the
    // key's code.  A score is then fma(raw, SCALE2, allowed ? 0 : bias): compare + select + FMA (unmasked: the FMA alone).  A PAD
    // query's code equals no key's, so all its scores round to the same -1e10 log2 e inside the sequence (-inf outside): uniform over
    // the S keys, as in the reference.  (No branch around the load: see the dQ kernel's prologue.)
    auto key_meta2 = [&](int64_t k, int& c, float& nb) {
        const bool in = k < S;
        const int cr = MASKED ? code[seq * S + (in ? k : S - 1)] : 0;
        c = (MASKED && in) ? cr : CODE_NONE;
        nb = in ? (MASKED ? NEG_BIG2 : 0.f) : -INFINITY;
    };
    auto stage = [&](int t, int b) {             // this wave's 2 + 2 pieces of key tile t -> buffer b
        const unsigned so = (unsigned)t * tile_step;
        char* kd = Ks[b] + wave * 2048;
        char* vd = Vs[b] + wave * 2048;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, MR_LDS_PTR(void, kd), 16, so0, so, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, MR_LDS_PTR(void, kd + 1024), 16, so1, so, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, MR_LDS_PTR(void, vd), 16, so0, so, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, MR_LDS_PTR(void, vd + 1024), 16, so1, so, 0, 0);
    };
    // The prologue's requests, all in flight before anything waits (see the dQ kernel): the first batch of the key table, this lane's query fragments
    // and codes (rows clamped into the sequence, the values discarded afterwards), the first key tile's DMA.  The table is padded to a multiple of
    // 12 tiles so that a batch's stores are unconditional: under `if (tile < nt)` the compiler sinks the batch's LOADS into the branch, behind everything.
    const int nt = (int)((S + TK - 1) / TK);
    const int ntp = (nt + 11) / 12 * 12;
    int32_t* const Cs = reinterpret_cast<int32_t*>(fwd_dyn);
    float* const Ns = fwd_dyn + ntp * TK;
    int32_t* const Ku = reinterpret_cast<int32_t*>(Ns + ntp * TK);
    int c3[3];
    float n3[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) key_meta2((int64_t)(wave + 4 * u) * TK + lane, c3[u], n3[u]);
    auto table_put = [&](int t0) {               // wave w fills and classifies tiles w, w + 4, ... (lane = key)
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int t = t0 + 4 * u;
            Ns[t * TK + lane] = n3[u];
            if (MASKED) { Cs[t * TK + lane] = c3[u]; const int kur = tile_class(c3[u], (int64_t)(t + 1) * TK <= S); if (lane == 0) Ku[t] = kur; }
        }
    };

    bf16x8 qf[QB][2];
    int64_t qi[QB];
    int cq[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        qi[qb] = q0 + (wave * QB + qb) * 16 + i;
        const int64_t row = qi[qb] < S ? qi[qb] : S - 1;
#pragma unroll
        for (int dd = 0; dd < 2; ++dd) qf[qb][dd] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(base + row * ld + h * 64 + dd * 32 + g * 8));
        cq[qb] = MASKED ? code[seq * S + row] : 0;
    }
    stage(0, 0);                                 // (behind the fragment requests: ahead of them the unmasked forward measured 1.2 us slower, scripts/ab_q.sh)
    __builtin_amdgcn_sched_barrier(0);
    table_put(wave);
    for (int t0 = wave + 12; t0 < nt; t0 += 12) {      // (S > 768)
#pragma unroll
        for (int u = 0; u < 3; ++u) key_meta2((int64_t)(t0 + 4 * u) * TK + lane, c3[u], n3[u]);
        table_put(t0);
    }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        if (qi[qb] >= S) {
            cq[qb] = 0;
#pragma unroll
            for (int dd = 0; dd < 2; ++dd) qf[qb][dd] = __builtin_bit_cast(bf16x8, u32x4{0u, 0u, 0u, 0u});
        }
        if (MASKED && cq[qb] < 0) cq[qb] = CODE_PADQ;            // a PAD query matches no key code (a PAD key's is -1)
    }
    // this wave's queries: one common valid code?  (queries beyond the sequence are never stored: wildcards; a wave without any query inside
    // the sequence skips every tile)
    bool wq_uniform = false, wq_allpad = false;
    int wq_code = 0;
    if constexpr (MASKED) {
        wq_code = __builtin_amdgcn_readfirstlane(cq[0]);        // lane 0 = the wave's first query
        bool ok = true;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) ok = ok && (qi[qb] >= S || cq[qb] == wq_code);
        const bool first_inside = q0 + wave * QB * 16 < S;
        wq_uniform = tile_modes != 0 && (!first_inside || (wq_code >= 0 && !__any(!ok)));
        wq_allpad = tile_modes != 0 && first_inside && wq_code == CODE_PADQ && !__any(!ok);
        if (!first_inside) wq_code = CODE_PADQ - 1;             // equals no tile class: every tile is skipped
    }

    float m[QB], l[QB];
    f32x4 ot[QB][4];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        m[qb] = -INFINITY; l[qb] = 0.f;
#pragma unroll
        for (int db = 0; db < 4; ++db) ot[qb][db] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // one key tile's scores, softmax update and P V product; FAST (masked kernels only): every (query, key) pair of this wave and tile is
    // allowed, the bias is 0 for all of them
    // (staging and the end-of-tile wait stay inside the lambda: split off, the same statements cost registers -- see the dQ kernel)
    for (int t = 0; t < nt; ++t) {
        const int b = t & 1;
        // masked kernels: one body, wave-uniform branches around the bias section (TILE_FAST) / the whole compute section (TILE_SKIP) --
        // three instantiated bodies cost this kernel its third wave per SIMD (206 registers)
        int mode = TILE_GENERAL;
        if constexpr (MASKED) {
            mode = tile_mode(wq_uniform, wq_code, __builtin_amdgcn_readfirstlane(Ku[t]));
            if (wq_allpad && (int64_t)(t + 1) * TK <= S) mode = TILE_UNIFORM;
        }
        MR_ASTAMP(0);
        if (t + 1 < nt) stage(t + 1, b ^ 1);      // next tile: straight into the other buffer (last read one iteration ago, behind that iteration's barrier)
        MR_ASTAMP(1);
        if (!MASKED || mode != TILE_SKIP) {
        bf16x8 pf[QB][2];
        constexpr bool V_EARLY = !MASKED;
        s16x4 vlo[2][4], vhi[2][4];
        if (MASKED && mode == TILE_UNIFORM) {
            bf16x8 ones;
#pragma unroll
            for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) { m[qb] = NEG_BIG2; l[qb] += 16.f; pf[qb][0] = ones; pf[qb][1] = ones; }
        } else {
        f32x4 st[QB][4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const bf16x8 k0 = row_frag_d(Ks[b], kb * 16, 0, lane), k1 = row_frag_d(Ks[b], kb * 16, 1, lane);
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                f32x4 a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, qf[qb][0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                st[qb][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, qf[qb][1], a, 0, 0, 0);
            }
        }
        MR_ASTAMP(2);
        // the V fragments do not depend on the softmax: the first half (keys 0-31) is requested now and lands while it runs; the
        // second half after it (its registers are the softmax's), landing under the first half's MFMAs
        // (the masked variant holds the codes as well and would lose its third wave per SIMD to these 16 registers: it asks late)
        if constexpr (V_EARLY) {
#pragma unroll
            for (int db = 0; db < 4; ++db) tr_frag_d_issue(Vs[b], 0, 16 * db, lane, vlo[0][db], vhi[0][db]);
        }
        float tmx[QB];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) tmx[qb] = -INFINITY;
        if (MASKED && mode == TILE_FAST) {       // every pair allowed: the bias is 0, s = fma(raw, SCALE2, 0)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float s = st[qb][kb][r] * SCALE2;
                        st[qb][kb][r] = s;
                        tmx[qb] = fmaxf(tmx[qb], s);
                    }
        } else {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {         // the staged vectors of one 16-key block at a time: 8 registers live, not 32
                const f32x4 nbv = *reinterpret_cast<const f32x4*>(&Ns[t * TK + kb * 16 + g * 4]);
                i32x4 ckv = {0, 0, 0, 0};
                if (MASKED) ckv = *reinterpret_cast<const i32x4*>(&Cs[t * TK + kb * 16 + g * 4]);
#pragma unroll
                for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float bias = MASKED ? ((ckv[r] == cq[qb]) ? 0.f : nbv[r]) : nbv[r];
                        const float s = __builtin_fmaf(st[qb][kb][r], SCALE2, bias);
                        st[qb][kb][r] = s;
                        tmx[qb] = fmaxf(tmx[qb], s);
                    }
            }
        }
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            float tmax = tmx[qb];
            tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            const float mn = fmaxf(m[qb], tmax);
            const float alpha = __builtin_amdgcn_exp2f(m[qb] - mn);      // m = -inf on the first tile -> 0
            m[qb] = mn;
            float psum = 0.f;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(st[qb][kb][r] - mn);
                    st[qb][kb][r] = pv;
                    psum += pv;
                }
            l[qb] = l[qb] * alpha + psum;
#pragma unroll
            for (int db = 0; db < 4; ++db) ot[qb][db] *= alpha;
            pf[qb][0] = pack_acc_pair(st[qb][0], st[qb][1]);
            pf[qb][1] = pack_acc_pair(st[qb][2], st[qb][3]);
        }
        }
        MR_ASTAMP(3);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!V_EARLY) {
#pragma unroll
            for (int db = 0; db < 4; ++db) tr_frag_d_issue(Vs[b], 0, 16 * db, lane, vlo[0][db], vhi[0][db]);
        }
#pragma unroll
        for (int db = 0; db < 4; ++db) tr_frag_d_issue(Vs[b], 32, 16 * db, lane, vlo[1][db], vhi[1][db]);
        // (the compiler does not count the asm reads: explicit waits) everything older than the 8 reads just issued has returned
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            if (t2 == 1) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                const bf16x8 vf = tr_join(vlo[t2][db], vhi[t2][db]);
#pragma unroll
                for (int qb = 0; qb < QB; ++qb)
                    ot[qb][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qb][t2], ot[qb][db], 0, 0, 0);
            }
        }
        MR_ASTAMP(4);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this wave's pieces of tile t + 1 have landed
        MR_ASTAMP(5);
        __syncthreads();
        MR_ASTAMP(6);
    }
    // lane holds O^T[d = 16 db + 4 g + r][query i]: 4 consecutive d of one row -> one 8-byte store per (qb, db)
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        float lt = l[qb];
        lt += __shfl_xor(lt, 16, 64);
        lt += __shfl_xor(lt, 32, 64);
        const float inv = 1.0f / lt;
        if (qi[qb] < S) {
            if (g == 0) lse[(seq * nh + h) * S + qi[qb]] = (m[qb] + log2f(lt)) * LN2;
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                bf16x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = (__bf16)(ot[qb][db][r] * inv);
                *reinterpret_cast<bf16x4*>(out + (seq * S + qi[qb]) * H + h * 64 + db * 16 + g * 4) = v;
            }
        }
    }
    MR_ASTAMP_WG(1);
}

// ------------------------------------------------------------------------------------------------ dQ (+ delta = rowsum(dO * O))
template <int QB, bool MASKED>
__global__ __launch_bounds__(256, (MASKED ? MR_ATTN_OCC_DQ : MR_ATTN_OCC_DQ_UNMASKED)) void attn_bwd_dq_kernel(const __bf16* __restrict__ qkv, const int32_t* __restrict__ code,
                                                             const __bf16* __restrict__ o, const __bf16* __restrict__ dout,
                                                             const float* __restrict__ lse, float* __restrict__ delta,
                                                             __bf16* __restrict__ dqkv, const float* __restrict__ rot_tab,
                                                             int64_t rot_rows, float* __restrict__ colsum, int64_t S, int64_t nh, const int tile_modes) {
    __shared__ __attribute__((aligned(16))) float red[4][64];
    __shared__ __attribute__((aligned(16))) char Ks[2][TILE_B];       // LDS-DMA images: row reads (S^T) and tr reads (dQ^T)
    __shared__ __attribute__((aligned(16))) char Vs[2][TILE_B];       // row reads (dP^T)
    // the key codes, the per-key bias and the tile classes of the WHOLE sequence: written once, in the prologue, by all four waves (dynamic LDS: 8 bytes per
    // padded position + 4 per tile).  Staged tile by tile by wave 0, their global loads parked that wave ~700 cycles at the top of every tile (stamps,
    // scripts/attn_dq_stamps.py) and the other three waited for it at the barrier
    extern __shared__ __attribute__((aligned(16))) float dq_dyn[];
    [[maybe_unused]] constexpr int stamp_id = 2;
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, i = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const AttnBlock ab_ = attn_block((int)((S + 64 * QB - 1) / (64 * QB)), (int)nh);
    const int64_t seq = ab_.seq, h = ab_.h, q0 = ab_.blk * (64 * QB);
    const int64_t H = nh * 64, ld = 3 * H;
    const __bf16* base = qkv + seq * S * ld;
    const int32_t* code_seq = MASKED ? code + seq * S : nullptr;
    const float inv_S = 1.0f / (float)S;
    MR_ASTAMP_WG(0);
    const int seq_rot0 = rot_tab != nullptr ? (int)((seq * S) % rot_rows) : 0;      // (uniform; once per workgroup)
    const bool rot_short = rot_rows < S;

    // The prologue's requests, ALL in flight before anything waits (round 5, stamps: scripts/attn_dq_stamps.py): the first key tile's DMA, then this
    // lane's 12 fragment loads, its LSE and code -- from the row clamped into the sequence, the value discarded afterwards, because a load under `if (row < S)`
    // is a basic block of its own and the compiler waited for each block's three loads before issuing the next: five serial round trips.  Measured: the
    // dK / dV kernel's prologue 10.9 k -> 8.5 k cycles; this kernel's stays at 16.4 k of a first-round workgroup's 59 k -- it is bound by its BYTES (three
    // resident workgroups x 64 KB per CU at the ~11 B / clk / CU a launch's first round gets: 17 k cycles), which later rounds hide under their neighbours.
    const __bf16* Kg = base + H + h * 64;
    const __bf16* Vg = base + 2 * H + h * 64;
    const int ext = (int)(((S - 1) * ld + 64) * 2);
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(Kg), 0, ext, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(Vg), 0, ext, 0x00020000);
    const unsigned so0 = dma_src(wave * 2, lane, ld), so1 = dma_src(wave * 2 + 1, lane, ld);
    const unsigned tile_step = (unsigned)(TK * ld * 2);
    auto stage = [&](int t, int b) {             // this wave's 2 + 2 pieces of key tile t -> buffer b (as in the forward kernel)
        const unsigned so = (unsigned)t * tile_step;
        char* kd = Ks[b] + wave * 2048;
        char* vd = Vs[b] + wave * 2048;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, MR_LDS_PTR(void, kd), 16, so0, so, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, MR_LDS_PTR(void, kd + 1024), 16, so1, so, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, MR_LDS_PTR(void, vd), 16, so0, so, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, MR_LDS_PTR(void, vd + 1024), 16, so1, so, 0, 0);
    };
    stage(0, 0);
    const int nt = (int)((S + TK - 1) / TK);
    const int ntp = (nt + 11) / 12 * 12;
    int32_t* const Cs = reinterpret_cast<int32_t*>(dq_dyn);
    float* const Ns = dq_dyn + ntp * TK;
    int32_t* const Ku = reinterpret_cast<int32_t*>(Ns + ntp * TK);     // masked: class of the key tile (tile_class)
    // the key table: wave w fills and classifies tiles w, w + 4, ... (lane = key), three tiles per batch; the first batch's loads lead the prologue's
    // requests.  Padded to a multiple of 12 tiles, so that a batch's stores are unconditional: under `if (tile < nt)` the compiler sinks the batch's
    // LOADS into the branch, behind the fragment loads, and waits there for all of them
    int c3[3];
    float n3[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) key_meta((int64_t)(wave + 4 * u) * TK + lane, S, code_seq, MASKED, c3[u], n3[u]);
    auto table_put = [&](int t0) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int t = t0 + 4 * u;
            Cs[t * TK + lane] = c3[u];
            Ns[t * TK + lane] = n3[u];
            if (MASKED) { const int kur = tile_class(c3[u], (int64_t)(t + 1) * TK <= S); if (lane == 0) Ku[t] = kur; }
        }
    };

    bf16x8 qf[QB][2], dof[QB][2];
    int qi[QB];      // (32-bit: S < 2^31; address arithmetic widens at the use)
    int cq[QB];
    float nlse2[QB], del[QB];
    bool padq[QB];
    bool any_pad = false, any_nz = false;
    u32x4 raw_q[QB][2], raw_d[QB][2], raw_o[QB][2];
    int raw_c[QB];
    float raw_l[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        qi[qb] = (int)q0 + (wave * QB + qb) * 16 + i;
        const int64_t row = qi[qb] < S ? qi[qb] : S - 1;
#pragma unroll
        for (int dd = 0; dd < 2; ++dd) {
            raw_q[qb][dd] = *reinterpret_cast<const u32x4*>(base + row * ld + h * 64 + dd * 32 + g * 8);
            raw_d[qb][dd] = *reinterpret_cast<const u32x4*>(dout + (seq * S + row) * H + h * 64 + dd * 32 + g * 8);
            raw_o[qb][dd] = *reinterpret_cast<const u32x4*>(o + (seq * S + row) * H + h * 64 + dd * 32 + g * 8);
        }
        raw_c[qb] = MASKED ? code_seq[row] : 0;
        raw_l[qb] = lse[(seq * nh + h) * S + row];
    }
    __builtin_amdgcn_sched_barrier(0);
    table_put(wave);
    for (int t0 = wave + 12; t0 < nt; t0 += 12) {      // (S > 768)
#pragma unroll
        for (int u = 0; u < 3; ++u) key_meta((int64_t)(t0 + 4 * u) * TK + lane, S, code_seq, MASKED, c3[u], n3[u]);
        table_put(t0);
    }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const bool ok = qi[qb] < S;
        float dsum = 0.f;
        unsigned nz = 0u;                             // any non-zero bit pattern (other than -0) in this lane's part of the dO row
#pragma unroll
        for (int dd = 0; dd < 2; ++dd) {
            const u32x4 zero = {0u, 0u, 0u, 0u};
            const u32x4 v = ok ? raw_q[qb][dd] : zero, w = ok ? raw_d[qb][dd] : zero, x = ok ? raw_o[qb][dd] : zero;
            nz |= (w[0] | w[1] | w[2] | w[3]) & 0x7fff7fffu;
            qf[qb][dd] = scale_eighth(v);             // the TRUE q also for PAD rows: the reference differentiates through them
            dof[qb][dd] = __builtin_bit_cast(bf16x8, w);
            float a[8], bb[8];
            unpack8(w, a);
            unpack8(x, bb);
#pragma unroll
            for (int e = 0; e < 8; ++e) dsum += a[e] * bb[e];
        }
        dsum += __shfl_xor(dsum, 16, 64);             // the 4 lanes of a query hold 16 of its 64 dims each
        dsum += __shfl_xor(dsum, 32, 64);
        nz |= __shfl_xor(nz, 16, 64);
        nz |= __shfl_xor(nz, 32, 64);
        any_nz = any_nz || nz != 0u;
        del[qb] = dsum;
        // for the dK / dV kernel, launched behind this one.  A row whose upstream gradient is ENTIRELY zero (the PAD rows of a training step: nothing
        // reads their outputs) contributes exactly nothing to dQ, dK or dV (dS = P (0 - 0), dV += P^T 0) whatever its weights are: it is
        // marked with delta = -0.0 (its delta is +0.0 by arithmetic), so that the dK / dV kernel can skip query tiles made of such rows
        if (ok && g == 0) delta[(seq * nh + h) * S + qi[qb]] = (tile_modes != 0 && nz == 0u) ? -0.0f : dsum;
        int c = ok ? raw_c[qb] : CODE_PADQ;
        if (c < 0) c = CODE_PADQ;
        cq[qb] = c;
        float L = ok ? raw_l[qb] : INFINITY;                          // beyond the sequence: p = exp2(-inf) = 0
        // a query row with no allowed key (PAD): the softmax is uniform over the S keys and its LSE is not representable:
        // take P = 1/S instead of exp(s - lse) -- unless its upstream gradient is zero (every PAD row of a training step): then
        // dS = P (0 - 0) = 0 whatever P is, and the row is treated like one beyond the sequence (p = 0) instead of costing its wave the uniform-row selects
        const bool pad = MASKED && ok && L < PAD_LSE;
        if (pad && tile_modes != 0 && nz == 0u) L = INFINITY;
        padq[qb] = pad && L != INFINITY;
        any_pad = any_pad || padq[qb];
        nlse2[qb] = -L * LOG2E;
    }
    const bool wave_pad = MASKED && __any(any_pad);
    // every row of this wave has a zero upstream gradient (rows beyond the sequence count as such): dQ = 0, no tile is computed
    const bool wave_dead = tile_modes != 0 && !__any(any_nz);
    // this wave's queries: one common valid code (see tile_mode)?  Rows beyond the sequence are wildcards (their P is exp2(-inf) = 0)
    bool wq_uniform = false;
    int wq_code = 0;
    if constexpr (MASKED) {
        // the LIVE rows' codes: a row beyond the sequence or a PAD row with a zero upstream gradient has p = exp2(-inf) = 0 under any bias: a wildcard
        bool live[QB];
        int first_code = CODE_PADQ - 1;                       // (equals no tile class: a wave without live rows skips every tile)
#pragma unroll
        for (int qb = QB - 1; qb >= 0; --qb) {
            live[qb] = qi[qb] < S && nlse2[qb] != -INFINITY;
            const unsigned long long bl = __ballot(live[qb]);
            if (bl != 0ull) first_code = __builtin_amdgcn_readlane(cq[qb], (int)__builtin_ctzll(bl));
        }
        wq_code = first_code;
        bool ok = true;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) ok = ok && (!live[qb] || cq[qb] == wq_code);
        wq_uniform = tile_modes != 0 && (wq_code >= 0 || wq_code == CODE_PADQ - 1) && !__any(!ok);
    }
    f32x4 dq[QB][4];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int db = 0; db < 4; ++db) dq[qb][db] = f32x4{0.f, 0.f, 0.f, 0.f};

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const bool ragged = (S & (TK - 1)) != 0;
    // one key tile; NB (compile time) = the scores of this tile need the per-key bias: when masked unless the tile is uniformly allowed
    // for this wave (tile_mode), otherwise only in a ragged LAST tile (keys beyond the sequence) -- which is peeled below so that the hot
    // loop of the unmasked towers carries neither the test nor the addresses of the staged codes (as one loop with a run-time flag the
    // kernel needed 190 registers)
    // (staging and the end-of-tile wait stay INSIDE this lambda: as three lambdas -- top / compute / end -- the same statements compiled to
    // 168 registers + 31 spilled for the unmasked instance instead of 154 + 0)
    // masked kernels: ONE body with wave-uniform branches on the tile's mode (three instantiated bodies cost registers: 230 -> 256 + 20 spilled in
    // the dK / dV kernel, 168 -> 206 in the forward kernel)
    auto tile_body = [&](int t, auto nb_c, const int mode) {
        const bool need_bias = mode == TILE_GENERAL;         // (masked kernels; the unmasked ones take nb_c)
        const bool skip = MASKED && mode == TILE_SKIP;       // stage, wait and synchronise only: every weight of the tile is 0 for this wave
        const int b = t & 1;
        MR_ASTAMP(0);
        if (t + 1 < nt) stage(t + 1, b ^ 1);
        MR_ASTAMP(1);
        if (!skip) {
        bf16x8 dsf[QB][2];
        // scores, P, dS of the tile (the first two products + the per-score arithmetic).  Masked kernels: instantiated twice under ONE wave-uniform
        // branch per tile -- with the bias arithmetic (general) and without (every pair allowed); a branch per (key block, query block) site instead cost
        // 8 scalar branches and ~40 register moves at their joins per tile, more than the bias instructions it skipped
        auto scores = [&](auto nbs_c) {
            constexpr bool NB = decltype(nbs_c)::value;
            f32x4 ds[QB][2];
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                const bf16x8 k0 = row_frag_d(Ks[b], kb * 16, 0, lane), k1 = row_frag_d(Ks[b], kb * 16, 1, lane);
                const bf16x8 v0 = row_frag_d(Vs[b], kb * 16, 0, lane), v1 = row_frag_d(Vs[b], kb * 16, 1, lane);
                i32x4 ck = {0, 0, 0, 0};
                f32x4 nk = {0.f, 0.f, 0.f, 0.f};
                if constexpr (NB) {
                    ck = *reinterpret_cast<const i32x4*>(&Cs[t * TK + kb * 16 + g * 4]);
                    nk = *reinterpret_cast<const f32x4*>(&Ns[t * TK + kb * 16 + g * 4]);
                }
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
                    f32x4 st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, qf[qb][0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, qf[qb][1], st, 0, 0, 0);
                    f32x4 dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v0, dof[qb][0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v1, dof[qb][1], dp, 0, 0, 0);
                    if constexpr (NB) add_bias(st, ck, nk, cq[qb]);
                    f32x4 pv;
#pragma unroll
                    for (int r = 0; r < 4; ++r) pv[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[r], LOG2E, nlse2[qb]));
                    if (NB && wave_pad) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) pv[r] = padq[qb] ? ((ck[r] == CODE_NONE) ? 0.f : inv_S) : pv[r];
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) ds[qb][kb & 1][r] = pv[r] * (dp[r] - del[qb]);
                    if (kb & 1) dsf[qb][kb >> 1] = pack_acc_pair(ds[qb][0], ds[qb][1]);     // packed pair by pair: 16 fewer live registers
                }
            }
        };
        if constexpr (MASKED) {
            if (need_bias) scores(std::true_type{});
            else scores(std::false_type{});
        } else {
            scores(nb_c);
        }
        MR_ASTAMP(2);
        // K^T fragments (asm reads, explicit waits: see tr_frag_d_issue), one 32-key half at a time: the second half is requested once
        // the first half's MFMAs are issued and lands under them (its 16 registers are the first half's: 3 waves per SIMD)
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            s16x4 klo[4], khi[4];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int db = 0; db < 4; ++db) tr_frag_d_issue(Ks[b], 32 * t2, 16 * db, lane, klo[db], khi[db]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                const bf16x8 kt = tr_join(klo[db], khi[db]);
#pragma unroll
                for (int qb = 0; qb < QB; ++qb)
                    dq[qb][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt, dsf[qb][t2], dq[qb][db], 0, 0, 0);
            }
        }
        }
        MR_ASTAMP(3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this wave's pieces of tile t + 1 have landed
        MR_ASTAMP(4);
        __syncthreads();
        MR_ASTAMP(5);
    };
    if constexpr (MASKED) {
        for (int t = 0; t < nt; ++t)
            tile_body(t, std::true_type{}, wave_dead ? (int)TILE_SKIP : tile_mode(wq_uniform, wq_code, __builtin_amdgcn_readfirstlane(Ku[t])));
    } else {
        for (int t = 0; t < nt - 1; ++t) tile_body(t, std::false_type{}, TILE_GENERAL);
        if (ragged) tile_body(nt - 1, std::true_type{}, TILE_GENERAL);
        else tile_body(nt - 1, std::false_type{}, TILE_GENERAL);
    }
    { const int t = 13; MR_ASTAMP(0); }
    f32x4 cs[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) cs[db] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (MASKED || QB == 1) {   // whole rows through the wave's share of the K buffers (free since the loop's last barrier)
        bf16x4 ov[QB][4];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            const bool ok = qi[qb] < S;
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                const int d = db * 16 + g * 4;
                f32x4 x = dq[qb][db] * 0.125f;
                if (rot_tab != nullptr && d < 32 && ok) x *= *reinterpret_cast<const f32x4*>(rot_tab + (int64_t)rot_row_index(seq_rot0, (int)qi[qb], (int)rot_rows, rot_short) * 32 + d);
#pragma unroll
                for (int r = 0; r < 4; ++r) { ov[qb][db][r] = (__bf16)x[r]; if (ok) cs[db][r] += (float)ov[qb][db][r]; }
            }
        }
        const int row0 = (int)q0 + wave * QB * 16;
        store_rows_via_lds<QB>(&Ks[0][0] + wave * (QB * 2048), ov, dqkv + (seq * S + row0) * ld + h * 64, ld, row0, S, lane);
    } else {                             // (the unmasked two-block instance sits at its register limit: with the staging it spilled into its MFMA blocks)
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            if (qi[qb] < S) {
#pragma unroll
                for (int db = 0; db < 4; ++db) {
                    const int d = db * 16 + g * 4;
                    f32x4 x = dq[qb][db] * 0.125f;
                    if (rot_tab != nullptr && d < 32) x *= *reinterpret_cast<const f32x4*>(rot_tab + (int64_t)rot_row_index(seq_rot0, (int)qi[qb], (int)rot_rows, rot_short) * 32 + d);
                    bf16x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { v[r] = (__bf16)x[r]; cs[db][r] += (float)v[r]; }
                    *reinterpret_cast<bf16x4*>(dqkv + (seq * S + qi[qb]) * ld + h * 64 + d) = v;
                }
            }
        }
    }
    { const int t = 13; MR_ASTAMP(1); }
    if (colsum != nullptr)        // wave-uniform: partial row (sequence, query block), columns of this head's q
        block_colsum_store(cs, red, colsum + (seq * ((S + 64 * QB - 1) / (64 * QB)) + ab_.blk) * ld + h * 64, tid);
    MR_ASTAMP_WG(1);
}

// ------------------------------------------------------------------------------------------------ dK, dV
// Block owns 64*KB keys (wave: 16*KB).  Here scores are NOT transposed (S = Q . K^T: column = key on the lane, 4 queries
// per 16-query block in the registers), so P and dS are the B operands of dV^T = dO^T . P and dK^T = Q^T . dS.
// Per query of the streamed tile the staging threads provide: -lse * log2 e (-inf beyond the sequence: p = 0), delta, the
// code (CODE_PADQ for a row without allowed key) and, for such rows, the uniform weight 1/S.
template <int KB, bool MASKED>
__global__ __launch_bounds__(256, MR_ATTN_OCC) void attn_bwd_dkv_kernel(const __bf16* __restrict__ qkv, const int32_t* __restrict__ code,
                                                              const __bf16* __restrict__ dout, const float* __restrict__ lse,
                                                              const float* __restrict__ delta, __bf16* __restrict__ dqkv,
                                                              const float* __restrict__ rot_tab, int64_t rot_rows,
                                                              float* __restrict__ colsum, int64_t S, int64_t nh, const int tile_modes) {
    __shared__ __attribute__((aligned(16))) float red[4][64];
    // A ring of THREE query tiles, requested TWO tiles ahead (round 5): with two buffers the wait for tile t + 1 at the end of tile t took ~900 of a tile's
    // 4 300 cycles (stamps, scripts/attn_dkv_stamps.py) -- under the load of 512 resident workgroups an LDS-DMA request needs ~3 000 cycles to land, more
    // than a tile's arithmetic.  52 KiB per workgroup: two per CU, as the registers allow anyway.
    constexpr int NB = 3;
    __shared__ __attribute__((aligned(16))) char Qs[NB][TILE_B];       // LDS-DMA images: row reads (S) and tr reads (dK^T)
    __shared__ __attribute__((aligned(16))) char Ds[NB][TILE_B];       // dO: row reads (dP) and tr reads (dV^T)
    // The per-query scalars (-lse log2 e, delta, code, the uniform weight of a live PAD row) of the WHOLE sequence and the classes of its tiles: written once,
    // in the prologue, by all four waves (dynamic LDS: 16 bytes per padded position + 8 per tile; S = 640: 10 KiB, S = 1312: 21 KiB).  Staged tile by tile
    // by wave 0, their global loads -- ~3 000 cycles under load -- were what the staging wave waited for at the end of every tile, and the other
    // three waited for it at the barrier: ~700 of a tile's 4 300 cycles (stamps).
    extern __shared__ __attribute__((aligned(16))) float dkv_dyn[];
    [[maybe_unused]] constexpr int stamp_id = 3;
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, i = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const AttnBlock ab_ = attn_block((int)((S + 64 * KB - 1) / (64 * KB)), (int)nh);
    const int64_t seq = ab_.seq, h = ab_.h, kbase = ab_.blk * (64 * KB);
    const int64_t H = nh * 64, ld = 3 * H;
    const __bf16* base = qkv + seq * S * ld;
    const int32_t* code_seq = MASKED ? code + seq * S : nullptr;
    const float inv_S = 1.0f / (float)S;
    MR_ASTAMP_WG(0);
    const int seq_rot0 = rot_tab != nullptr ? (int)((seq * S) % rot_rows) : 0;      // (uniform; once per workgroup)
    const bool rot_short = rot_rows < S;

    // The prologue's requests, all in flight before anything waits (see the dQ kernel): the first two query tiles' DMA, the first batch of the
    // per-query table, this lane's K / V fragments and codes (rows clamped into the sequence, the values discarded afterwards)
    const __bf16* Qg = base + h * 64;
    const __bf16* Dg = dout + seq * S * H + h * 64;
    const float* Lg = lse + (seq * nh + h) * S;
    const float* Eg = delta + (seq * nh + h) * S;
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(Qg), 0, (int)(((S - 1) * ld + 64) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(Dg), 0, (int)(((S - 1) * H + 64) * 2), 0x00020000);
    const unsigned sq0 = dma_src(wave * 2, lane, ld), sq1 = dma_src(wave * 2 + 1, lane, ld);
    const unsigned sd0 = dma_src(wave * 2, lane, H), sd1 = dma_src(wave * 2 + 1, lane, H);
    const unsigned q_step = (unsigned)(TK * ld * 2), d_step = (unsigned)(TK * H * 2);
    auto stage = [&](int t, int b) {             // this wave's 2 + 2 pieces of query tile t (Q rows, dO rows) -> buffer b
        char* qd = Qs[b] + wave * 2048;
        char* dd_ = Ds[b] + wave * 2048;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, MR_LDS_PTR(void, qd), 16, sq0, (unsigned)t * q_step, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, MR_LDS_PTR(void, qd + 1024), 16, sq1, (unsigned)t * q_step, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, MR_LDS_PTR(void, dd_), 16, sd0, (unsigned)t * d_step, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, MR_LDS_PTR(void, dd_ + 1024), 16, sd1, (unsigned)t * d_step, 0, 0);
    };
    const int nt = (int)((S + TK - 1) / TK);
    const int ntp = (nt + 11) / 12 * 12;                               // (the table is filled in unconditional batches of 12 tiles: see the dQ kernel)
    const int Sp = ntp * TK;
    float* const Ls = dkv_dyn;
    float* const Dl = Ls + Sp;
    float* const Us = Dl + Sp;
    int32_t* const Cs = reinterpret_cast<int32_t*>(Us + Sp);
    int32_t* const Fs = Cs + Sp;                                       // tile has a LIVE row without allowed key
    int32_t* const Qu = Fs + ntp;                                      // masked: the tile's common valid code over its live rows, CODE_MIXED or CODE_DEAD
    stage(0, 0);
    if (nt > 1) stage(1, 1);
    float L3[3], e3[3];
    int c3[3];
    auto table_get = [&](int t0) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int q = (t0 + 4 * u) * TK + lane;
            const int qc = q < S ? q : (int)S - 1;
            L3[u] = Lg[qc];
            e3[u] = Eg[qc];
            c3[u] = MASKED ? code_seq[qc] : 0;
        }
    };
    table_get(wave);

    bf16x8 kf[KB][2], vf[KB][2];
    int64_t ki[KB];
    int ck[KB];
    float nkl[KB], unil[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        ki[kb] = kbase + (wave * KB + kb) * 16 + i;
        const int64_t row = ki[kb] < S ? ki[kb] : S - 1;
#pragma unroll
        for (int dd = 0; dd < 2; ++dd) {
            kf[kb][dd] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(base + row * ld + H + h * 64 + dd * 32 + g * 8));
            vf[kb][dd] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(base + row * ld + 2 * H + h * 64 + dd * 32 + g * 8));
        }
        ck[kb] = MASKED ? code_seq[row] : 0;
    }
    __builtin_amdgcn_sched_barrier(0);
    // wave w fills AND classifies tiles w, w + 4, ... (lane = query): the values are in its registers when it votes, so no barrier sits between the two
    auto table_put = [&](int t0) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int t = t0 + 4 * u;
            const int q = t * TK + lane;
            float L = q < S ? L3[u] : INFINITY;
            const float eq = q < S ? e3[u] : 0.f;
            const int cr = q < S ? c3[u] : CODE_PADQ;
            // a PAD row whose upstream gradient is zero (marked delta = -0.0 by the dQ kernel) weighs nothing in dK / dV: an absent row (p = 0), see there
            bool pad = MASKED && q < S && L < PAD_LSE;
            if (pad && tile_modes != 0 && __float_as_uint(eq) == 0x80000000u) { pad = false; L = INFINITY; }
            const float lrq = -L * LOG2E;
            const int cq_ = (cr < 0) ? CODE_PADQ : cr;
            Ls[q] = lrq;
            Dl[q] = eq;
            Cs[q] = cq_;
            Us[q] = pad ? inv_S : 0.f;
            const int fr = MASKED ? (int)__any(pad) : 0;
            int qur = CODE_MIXED;
            if (MASKED) {
                // the tile's class over its LIVE rows: a row beyond the sequence or a PAD row with a zero upstream gradient has lr = -inf, i.e. p = 0
                // under any bias -- a wildcard; a live PAD row weighs EVERY key: never uniform
                const bool live = lrq != -INFINITY;
                const unsigned long long bl = __ballot(live);
                const int c0 = bl != 0ull ? __builtin_amdgcn_readlane(cq_, (int)__builtin_ctzll(bl)) : 0;
                qur = (tile_modes == 0 || c0 < 0 || __any(live && cq_ != c0)) ? CODE_MIXED : c0;
                // every row of the tile beyond the sequence or marked by the dQ kernel as having a zero upstream gradient (delta = -0.0): such a
                // tile contributes exactly nothing to dK / dV and is skipped by every wave
                if (tile_modes != 0 && !__any(q < S && __float_as_uint(eq) != 0x80000000u)) qur = CODE_DEAD;
            }
            if (lane == 0) { Fs[t] = fr; Qu[t] = qur; }
        }
    };
    table_put(wave);
    for (int t0 = wave + 12; t0 < nt; t0 += 12) { table_get(t0); table_put(t0); }      // (S > 768)
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        const bool ok = ki[kb] < S;
#pragma unroll
        for (int dd = 0; dd < 2; ++dd) {
            const u32x4 zero = {0u, 0u, 0u, 0u};
            kf[kb][dd] = scale_eighth(ok ? __builtin_bit_cast(u32x4, kf[kb][dd]) : zero);             // (q . k) / 8 = q . (k / 8): exact
            if (!ok) vf[kb][dd] = __builtin_bit_cast(bf16x8, zero);
        }
        ck[kb] = ok ? ck[kb] : CODE_NONE;
        nkl[kb] = ok ? NEG_BIAS : -INFINITY;
        unil[kb] = ok ? 1.0f : 0.0f;
    }
    // this wave's keys: all inside the sequence with one common code (a valid one, or -1: PAD keys, which no valid query may see)?
    bool wk_uniform = false;
    int wk_code = 0;
    if constexpr (MASKED) {
        wk_code = __builtin_amdgcn_readfirstlane(ck[0]);
        bool ok = true;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) ok = ok && ck[kb] == wk_code;
        wk_uniform = tile_modes != 0 && wk_code != CODE_NONE && !__any(!ok);
    }
    f32x4 dk[KB][4], dv[KB][4];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int db = 0; db < 4; ++db) { dk[kb][db] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[kb][db] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // one query tile; FAST (masked kernels only): every (query, key) pair of the tile and this wave's keys is allowed
    // (staging and the end-of-tile wait stay inside the lambda: split off, the same statements cost registers -- see the dQ kernel)
    int b = 0;                                   // t % NB
    for (int t = 0; t < nt; ++t, b = (b == NB - 1) ? 0 : b + 1) {
        const int b2 = (b == 0) ? NB - 1 : b - 1;                          // (t + 2) % NB: the slot tile t - 1 was read from
        const int tq = t * TK;                                             // this tile's first position in the sequence-wide scalar arrays
        // masked kernels: wave-uniform branches on the tile's mode inside ONE body (see the dQ kernel)
        int mode = TILE_GENERAL;
        if constexpr (MASKED) {
            const int qc = __builtin_amdgcn_readfirstlane(Qu[t]);
            mode = qc == CODE_DEAD ? (int)TILE_SKIP : tile_mode(wk_uniform, wk_code, qc);
        }
        const bool FAST = MASKED && mode == TILE_FAST;
        MR_ASTAMP(0);
        if (t + 2 < nt) stage(t + 2, b2);
        MR_ASTAMP(1);
        if (!MASKED || mode != TILE_SKIP) {
        const bool tile_pad = MASKED && !FAST && Fs[t] != 0;
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            // this half-tile's P and dS (32 queries) go straight into dV^T / dK^T: only one half's fragments are ever live
            bf16x8 pf[KB], dsf[KB];
            // S, dP, P, dS of the half-tile.  Masked kernels: instantiated twice under ONE wave-uniform branch -- with the code compare / bias add /
            // uniform-row select (general) and without (every pair of the tile and this wave's keys allowed)
            auto scores = [&](auto fast_c) {
                constexpr bool FST = decltype(fast_c)::value;
                f32x4 pp[KB][2], ds[KB][2];
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2) {
                    const int qb = 2 * t2 + q2;
                    u32x4 q0r, q1r, d0r, d1r;                      // (asm reads of the ring: see row_frag_d_issue)
                    row_frag_d_issue(Qs[b], qb * 16, 0, lane, q0r);
                    row_frag_d_issue(Qs[b], qb * 16, 1, lane, q1r);
                    row_frag_d_issue(Ds[b], qb * 16, 0, lane, d0r);
                    row_frag_d_issue(Ds[b], qb * 16, 1, lane, d1r);
                    const f32x4 l4 = *reinterpret_cast<const f32x4*>(&Ls[tq + qb * 16 + g * 4]);
                    const f32x4 e4 = *reinterpret_cast<const f32x4*>(&Dl[tq + qb * 16 + g * 4]);
                    i32x4 c4 = {0, 0, 0, 0};
                    if (MASKED && !FST) c4 = *reinterpret_cast<const i32x4*>(&Cs[tq + qb * 16 + g * 4]);
                    f32x4 u4 = {0.f, 0.f, 0.f, 0.f};
                    if (!FST && tile_pad) u4 = *reinterpret_cast<const f32x4*>(&Us[tq + qb * 16 + g * 4]);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    const bf16x8 q0f = __builtin_bit_cast(bf16x8, q0r), q1f = __builtin_bit_cast(bf16x8, q1r);
                    const bf16x8 d0f = __builtin_bit_cast(bf16x8, d0r), d1f = __builtin_bit_cast(bf16x8, d1r);
#pragma unroll
                    for (int kb = 0; kb < KB; ++kb) {
                        f32x4 st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q0f, kf[kb][0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                        st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q1f, kf[kb][1], st, 0, 0, 0);
                        f32x4 dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(d0f, vf[kb][0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(d1f, vf[kb][1], dp, 0, 0, 0);
                        f32x4 pv;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            // allowed = same code (a PAD key's -1 and a missing key's -2 equal no query code; unmasked: codes are 0,
                            // missing keys -2, missing queries have l4 = -inf)
                            const float s = (FST || c4[r] == ck[kb]) ? st[r] : st[r] + nkl[kb];
                            pv[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s, LOG2E, l4[r]));
                        }
                        if (!FST && tile_pad) {       // rows without allowed key: uniform over the existing keys
#pragma unroll
                            for (int r = 0; r < 4; ++r) pv[r] = (u4[r] > 0.f) ? u4[r] * unil[kb] : pv[r];
                        }
                        pp[kb][q2] = pv;
#pragma unroll
                        for (int r = 0; r < 4; ++r) ds[kb][q2][r] = pv[r] * (dp[r] - e4[r]);
                    }
                }
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) {
                    pf[kb] = pack_acc_pair(pp[kb][0], pp[kb][1]);
                    dsf[kb] = pack_acc_pair(ds[kb][0], ds[kb][1]);
                }
            };
            if constexpr (MASKED) {
                if (FAST) scores(std::true_type{});
                else scores(std::false_type{});
            } else {
                scores(std::false_type{});
            }
            // dO^T / Q^T fragments of this half-tile (asm reads, explicit waits: see tr_frag_d_issue): dims 0-31 and 32-63 as two
            // groups of 8 reads, the second landing under the first's MFMAs
            s16x4 dlo[4], dhi[4], qlo[4], qhi[4];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                tr_frag_d_issue(Ds[b], 32 * t2, 16 * db, lane, dlo[db], dhi[db]);
                tr_frag_d_issue(Qs[b], 32 * t2, 16 * db, lane, qlo[db], qhi[db]);
            }
            asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                if (db == 2) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                }
                const bf16x8 dot = tr_join(dlo[db], dhi[db]);
                const bf16x8 qt = tr_join(qlo[db], qhi[db]);
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) {
                    dv[kb][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dot, pf[kb], dv[kb][db], 0, 0, 0);
                    dk[kb][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt, dsf[kb], dk[kb][db], 0, 0, 0);
                }
            }
        }
        }
        MR_ASTAMP(2);
        if (t + 2 < nt) {
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");          // this wave's pieces of tile t + 1 have landed; the 4 of tile t + 2 stay in flight
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        MR_ASTAMP(3);
        // a RAW barrier: __syncthreads() carries a fence that the compiler lowers to s_waitcnt vmcnt(0) -- it would drain the requests of tile t + 2.
        // What the barrier must order is in LDS: this wave's reads of the slot tile t + 3 will overwrite (all retired: the asm reads were waited for,
        // the others feed arithmetic that has issued)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        MR_ASTAMP(4);
    }
    { const int t = 13; MR_ASTAMP(0); }
    // lane holds dK^T / dV^T [d = 16 db + 4 g + r][key i].  dK: the 1/8 folded into kf was on the OTHER operand of S = q . (k/8), so
    // d(score)/dk = q / 8 still has to be applied here.  Both tiles leave as whole rows through the wave's share of the Q / dO buffers (free since the
    // loop's last barrier): store_rows_via_lds.
    f32x4 csk[4], csv[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) { csk[db] = f32x4{0.f, 0.f, 0.f, 0.f}; csv[db] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    bf16x4 ok_[KB][4], ov_[KB][4];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        const bool ok = ki[kb] < S;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            const int d = db * 16 + g * 4;
            f32x4 x = dk[kb][db] * 0.125f;
            if (rot_tab != nullptr && d < 32 && ok) x *= *reinterpret_cast<const f32x4*>(rot_tab + (int64_t)rot_row_index(seq_rot0, (int)ki[kb], (int)rot_rows, rot_short) * 32 + d);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ok_[kb][db][r] = (__bf16)x[r]; ov_[kb][db][r] = (__bf16)dv[kb][db][r];
                if (ok) { csk[db][r] += (float)ok_[kb][db][r]; csv[db][r] += (float)ov_[kb][db][r]; }
            }
        }
    }
    {
        const int row0 = (int)kbase + wave * KB * 16;
        __bf16* grow = dqkv + (seq * S + row0) * ld + h * 64;
        store_rows_via_lds<KB>(&Qs[0][0] + wave * (KB * 2048), ok_, grow + H, ld, row0, S, lane);
        store_rows_via_lds<KB>(&Ds[0][0] + wave * (KB * 2048), ov_, grow + 2 * H, ld, row0, S, lane);
    }
    { const int t = 13; MR_ASTAMP(1); }
    if (colsum != nullptr) {      // wave-uniform: partial row (sequence, key block), columns of this head's k and v
        float* prow = colsum + (seq * ((S + 64 * KB - 1) / (64 * KB)) + ab_.blk) * ld;
        block_colsum_store(csk, red, prow + H + h * 64, tid);
        __syncthreads();
        block_colsum_store(csv, red, prow + 2 * H + h * 64, tid);
    }
    MR_ASTAMP_WG(1);
}


// ------------------------------------------------------------------------------------------------ dQ, dK, dV in ONE pass (S <= 256)
// One workgroup = EIGHT waves = one (sequence, head): wave w owns keys 32 w .. 32 w + 31 (K / 8 and V fragments and the dK^T / dV^T
// accumulators in registers for the whole kernel, as in the dK / dV kernel above), the workgroup sweeps the queries in tiles of 64.
// Per tile every score is computed ONCE -- S = Q K^T, dP = dO V^T, P = exp2(S log2 e - lse), dS = P (dP - delta) in the
// [query rows, key column] layout, so P and dS are directly the B operands of dV^T += dO^T P and dK^T += Q^T dS -- and only dS
// crosses the LDS, once, for dQ: written TRANSPOSED ([key][query] tiles in the LDS-DMA image format, one 8-byte store per 4 queries)
// and read back with ds_read_b64_tr_b16 as the B operand of dQ^T = K^T dS^T, whose A operand comes from the K tiles staged once per
// workgroup -- the same transposed read on both, so the key permutation of the contraction index matches by construction.  Wave w
// multiplies the 16-dim block w & 3 of dQ^T for the query half w >> 2 over all 256 keys.  5 matrix products and one exponential per
// score instead of the two-pass kernels' 7 and 2, no delta buffer (rowsum(dO * O) is staged per tile by the whole workgroup), and
// dQ is complete inside the workgroup: no atomics, no cross-workgroup sum, bitwise reproducible.  The dQ product of tile t runs at the
// top of iteration t + 1 (its dS buffer is double buffered), so the tile loop keeps ONE barrier per tile.
// Column sums (qkv bias gradient): ONE partial row per sequence, this head's 64 columns of the q, k and v thirds.
constexpr int KP = 256;                         // keys per workgroup
template <bool MASKED>
__global__ __launch_bounds__(512, 2) void attn_bwd1_kernel(const __bf16* __restrict__ qkv, const int32_t* __restrict__ code,
                                                           const __bf16* __restrict__ o, const __bf16* __restrict__ dout,
                                                           const float* __restrict__ lse, __bf16* __restrict__ dqkv,
                                                           const float* __restrict__ rot_tab, int64_t rot_rows,
                                                           float* __restrict__ colsum, int64_t S, int64_t nh) {
    __shared__ __attribute__((aligned(16))) char Kt[4][TILE_B];          // K of the sequence: tr reads (dQ^T)
    __shared__ __attribute__((aligned(16))) char Qs[2][TILE_B];          // row reads (S) and tr reads (dK^T)
    __shared__ __attribute__((aligned(16))) char Ds[2][TILE_B];          // dO: row reads (dP) and tr reads (dV^T)
    __shared__ __attribute__((aligned(16))) char dSs[2][4][TILE_B];      // dS^T: [64 keys][64 queries] images, tr reads (dQ^T)
    __shared__ __attribute__((aligned(16))) float Ls[2][TK], Dl[2][TK], Us[2][TK];
    __shared__ __attribute__((aligned(16))) int32_t Cs[2][TK];
    __shared__ __attribute__((aligned(16))) float red[8][192];          // column-sum partials: [wave][q | k | v third][64]
    [[maybe_unused]] constexpr int stamp_id = 4;
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, i = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const AttnBlock ab_ = attn_block(1, (int)nh);
    const int64_t seq = ab_.seq, h = ab_.h;
    const int64_t H = nh * 64, ld = 3 * H;
    const __bf16* base = qkv + seq * S * ld;
    const int32_t* code_seq = MASKED ? code + seq * S : nullptr;
    const float inv_S = 1.0f / (float)S;
    MR_ASTAMP_WG(0);
    const int seq_rot0 = rot_tab != nullptr ? (int)((seq * S) % rot_rows) : 0;      // (uniform; once per workgroup)
    const bool rot_short = rot_rows < S;

    // ---- this wave's keys: K / 8 and V fragments, codes ----
    bf16x8 kf[2][2], vf[2][2];
    int ki[2], ck[2];
    float nkl[2], unil[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        ki[kb] = wave * 32 + kb * 16 + i;
        const bool ok = ki[kb] < S;
#pragma unroll
        for (int dd = 0; dd < 2; ++dd) {
            u32x4 w = {0u, 0u, 0u, 0u};
            if (ok) w = *reinterpret_cast<const u32x4*>(base + (int64_t)ki[kb] * ld + 2 * H + h * 64 + dd * 32 + g * 8);
            vf[kb][dd] = __builtin_bit_cast(bf16x8, w);          // (kf: from the K images in LDS, below -- K is fetched once)
        }
        ck[kb] = ok ? (MASKED ? code_seq[ki[kb]] : 0) : CODE_NONE;
        nkl[kb] = ok ? NEG_BIAS : -INFINITY;
        unil[kb] = ok ? 1.0f : 0.0f;
    }
    f32x4 dk[2][4], dv[2][4];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int db = 0; db < 4; ++db) { dk[kb][db] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[kb][db] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    // ---- staging: K tiles once (4 pieces per wave), Q / dO tiles per iteration (1 + 1 piece per wave), per-query scalars ----
    const __bf16* Qg = base + h * 64;
    const __bf16* Kg = base + H + h * 64;
    const __bf16* Dg = dout + seq * S * H + h * 64;
    const __bf16* Og = o + seq * S * H + h * 64;
    const float* Lg = lse + (seq * nh + h) * S;
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(Qg), 0, (int)(((S - 1) * ld + 64) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(Kg), 0, (int)(((S - 1) * ld + 64) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(Dg), 0, (int)(((S - 1) * H + 64) * 2), 0x00020000);
    const unsigned sq = dma_src(wave, lane, ld), sd = dma_src(wave, lane, H);
    const unsigned q_step = (unsigned)(TK * ld * 2), d_step = (unsigned)(TK * H * 2);
#pragma unroll
    for (int j = 0; j < 4; ++j) {               // K rows 64 j + 8 wave .. + 7 -> tile j, piece `wave`
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, MR_LDS_PTR(void, Kt[j] + wave * 1024), 16, sq, (unsigned)j * q_step, 0, 0);
    }
    auto stage = [&](int t, int b) {             // this wave's piece of query tile t (Q rows, dO rows) -> buffer b
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, MR_LDS_PTR(void, Qs[b] + wave * 1024), 16, sq, (unsigned)t * q_step, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, MR_LDS_PTR(void, Ds[b] + wave * 1024), 16, sd, (unsigned)t * d_step, 0, 0);
    };
    // per-query scalars of a tile: thread (q = tid >> 3, c = tid & 7) holds 8 dims of dO and O -> delta = rowsum(dO * O) over the 8
    // lanes of a query; lane c == 0 also fetches the query's lse and code
    // (two steps: the loads are issued at the top of a tile, the arithmetic on them runs at its end -- as one step the wave sat ~1 500 cycles at the
    // top of every tile waiting for these global loads: stamps)
    float er = 0.f, lr = 0.f, ur = 0.f;
    int cr = 0;
    u32x4 s_a = {0u, 0u, 0u, 0u}, s_b = {0u, 0u, 0u, 0u};
    float s_L = INFINITY;
    int s_c = CODE_PADQ;
    auto side_issue = [&](int64_t q0) {
        const int64_t q = q0 + (tid >> 3);
        const bool ok = q < S;
        s_a = u32x4{0u, 0u, 0u, 0u};
        s_b = u32x4{0u, 0u, 0u, 0u};
        if (ok) {
            s_a = *reinterpret_cast<const u32x4*>(Dg + q * H + (tid & 7) * 8);
            s_b = *reinterpret_cast<const u32x4*>(Og + q * H + (tid & 7) * 8);
        }
        s_L = ok ? Lg[q] : INFINITY;
        s_c = ok ? (MASKED ? code_seq[q] : 0) : CODE_PADQ;
    };
    auto side_compute = [&]() {
        float x[8], y[8];
        unpack8(s_a, x);
        unpack8(s_b, y);
        float dsum = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) dsum += x[e] * y[e];
        dsum += __shfl_xor(dsum, 1, 64);
        dsum += __shfl_xor(dsum, 2, 64);
        dsum += __shfl_xor(dsum, 4, 64);
        er = dsum;
        const bool pad = MASKED && s_L != INFINITY && s_L < PAD_LSE;
        lr = -s_L * LOG2E;
        cr = (s_c < 0) ? CODE_PADQ : s_c;
        ur = pad ? inv_S : 0.f;
    };
    auto side_store = [&](int b) {
        if ((tid & 7) == 0) { const int q = tid >> 3; Ls[b][q] = lr; Dl[b][q] = er; Cs[b][q] = cr; Us[b][q] = ur; }
    };
    stage(0, 0);
    side_issue(0);
    side_compute();
    side_store(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // this wave's K / 8 fragments from the staged images (rows past the sequence are zeros there): the prologue of every workgroup of a round runs
    // at once and is bound by the bytes it requests (~11 B / clk / CU: stamps, scripts/attn_bwd1_stamps.py); K came twice, 31 of its 117 KB
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int dd = 0; dd < 2; ++dd)
            kf[kb][dd] = scale_eighth(__builtin_bit_cast(u32x4, row_frag_d(Kt[wave >> 1], (wave & 1) * 32 + kb * 16, dd, lane)));

    const bool keys_inside = !MASKED && 32 * wave + 32 <= S;
    // dQ^T of one tile: wave (db = wave & 3, query half wave >> 2) over all 256 keys
    const int qdb = wave & 3, qh = wave >> 2;
    f32x4 csq = {0.f, 0.f, 0.f, 0.f};
    auto dq_tile = [&](int t, int b) {
        f32x4 dq[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        // the "rotary" scales of this wave's two query blocks (dims < 32 only: waves with qdb < 2), requested BEFORE the products they are applied
        // behind: loaded where they are used, the dependent global load sat in front of every tile's stores (~6 us of the ViT launch)
        f32x4 rv[2] = {f32x4{1.f, 1.f, 1.f, 1.f}, f32x4{1.f, 1.f, 1.f, 1.f}};
        if (rot_tab != nullptr && qdb < 2) {
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2) {
                int q = t * TK + 16 * (2 * qh + q2) + i;
                q = q < S ? q : (int)S - 1;
                rv[q2] = *reinterpret_cast<const f32x4*>(rot_tab + (int64_t)rot_row_index(seq_rot0, q, (int)rot_rows, rot_short) * 32 + qdb * 16 + g * 4);
            }
        }
#pragma unroll
        for (int ks2 = 0; ks2 < 4; ++ks2) {          // two 32-key steps per group of reads: 12 transposed reads in flight
            s16x4 klo[2], khi[2], slo[2][2], shi[2][2];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int ks = 2 * ks2 + u;
                tr_frag_d_issue(Kt[ks >> 1], 32 * (ks & 1), 16 * qdb, lane, klo[u], khi[u]);
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2) tr_frag_d_issue(dSs[b][ks >> 1], 32 * (ks & 1), 16 * (2 * qh + q2), lane, slo[u][q2], shi[u][q2]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const bf16x8 kt = tr_join(klo[u], khi[u]);
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2)
                    dq[q2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt, tr_join(slo[u][q2], shi[u][q2]), dq[q2], 0, 0, 0);
            }
        }
        // lane holds dQ^T[d = 16 qdb + 4 g + r][query 64 t + 16 (2 qh + q2) + i]
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2) {
            const int64_t q = (int64_t)t * TK + 16 * (2 * qh + q2) + i;
            if (q < S) {
                const int d = qdb * 16 + g * 4;
                f32x4 x = dq[q2] * 0.125f;
                x *= rv[q2];
                bf16x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) { v[r] = (__bf16)x[r]; csq[r] += (float)v[r]; }
                *reinterpret_cast<bf16x4*>(dqkv + (seq * S + q) * ld + h * 64 + d) = v;
            }
        }
    };

    const int nt = (int)((S + TK - 1) / TK);
#ifdef MR_ATTN_BWD1_PRIO
    // experiment (MI355X_MICROARCH.md, "two waves per SIMD", item 4): the second-dispatched half of the workgroup is the arbitration loser of every
    // segment; one static s_setprio 1 for it, no per-segment flips.  Same instructions, same results.  Measured (round 6, scripts/build_diag.sh prio attention
    // -fno-slp-vectorize -DMR_ATTN_BWD1_PRIO, scripts/attn_bench.py, alternating on one box): base ViT 77.4-77.5 us against 76.3-77.1, large ViT 100.5-101.1 against
    // 99.0-99.5, S = 130 58.8 against 56.8-57.3 -- 0.5-3 % SLOWER here (this kernel's halves are not in the compute / load alternation the guide measured): off.
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
    for (int t = 0; t < nt; ++t) {
        const int b = t & 1;
        MR_ASTAMP(0);
        if (t + 1 < nt) {
            stage(t + 1, b ^ 1);
            side_issue((int64_t)(t + 1) * TK);
        }
        MR_ASTAMP(1);
        if (t > 0) dq_tile(t - 1, b ^ 1);
        MR_ASTAMP(2);
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            // P and dS of this 32-query half, rounded to bf16 as soon as they exist (they are MFMA operands and the LDS payload as bf16
            // anyway; as fp32 the four row blocks' values pushed the masked variant past 256 registers, and a spilled or copied result of
            // the asm transposed reads is read before it lands)
            bf16x4 pb[2][2], sb[2][2];
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2) {
                const int qb = 2 * t2 + q2;
                const bf16x8 q0f = row_frag_d(Qs[b], qb * 16, 0, lane), q1f = row_frag_d(Qs[b], qb * 16, 1, lane);
                const bf16x8 d0f = row_frag_d(Ds[b], qb * 16, 0, lane), d1f = row_frag_d(Ds[b], qb * 16, 1, lane);
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(&Ls[b][qb * 16 + g * 4]);
                const f32x4 e4 = *reinterpret_cast<const f32x4*>(&Dl[b][qb * 16 + g * 4]);
                i32x4 c4 = {0, 0, 0, 0};
                if (MASKED) c4 = *reinterpret_cast<const i32x4*>(&Cs[b][qb * 16 + g * 4]);
                // rows without allowed key (PAD queries): uniform weight 1 / S over the existing keys, staged per query (0 for every other row)
                f32x4 u4 = {0.f, 0.f, 0.f, 0.f};
                if (MASKED) u4 = *reinterpret_cast<const f32x4*>(&Us[b][qb * 16 + g * 4]);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    f32x4 st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q0f, kf[kb][0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q1f, kf[kb][1], st, 0, 0, 0);
                    f32x4 dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(d0f, vf[kb][0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(d1f, vf[kb][1], dp, 0, 0, 0);
                    f32x4 pv;
                    if (keys_inside) {    // (wave-uniform) unmasked, every key of this wave inside the sequence: every pair is allowed (a query beyond the sequence has l4 = -inf: p = 0)
#pragma unroll
                        for (int r = 0; r < 4; ++r) pv[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[r], LOG2E, l4[r]));
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float sc = (c4[r] == ck[kb]) ? st[r] : st[r] + nkl[kb];
                            pv[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sc, LOG2E, l4[r]));
                        }
                    }
                    if (MASKED) {         // (always, no wave-uniform shortcut: this variant only runs masked towers of <= 256 positions)
#pragma unroll
                        for (int r = 0; r < 4; ++r) pv[r] = (u4[r] > 0.f) ? u4[r] * unil[kb] : pv[r];
                    }
                    bf16x4 p4, w4;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { p4[r] = (__bf16)pv[r]; w4[r] = (__bf16)(pv[r] * (dp[r] - e4[r])); }
                    pb[kb][q2] = p4;
                    sb[kb][q2] = w4;
                    // dS^T -> LDS: key row 32 wave + 16 kb + i, queries 16 qb + 4 g .. + 3 (8 bytes), in the tiles' swizzled image
                    {
                        const int row = 32 * (wave & 1) + 16 * kb + i, col = 16 * qb + 4 * g;
                        char* dst = dSs[b][wave >> 1] + row * 128 + (((col >> 3) ^ dswz(row)) << 4) + (col & 7) * 2;
                        *reinterpret_cast<bf16x4*>(dst) = w4;
                    }
                }
            }
            bf16x8 pf[2], dsf[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                pf[kb] = __builtin_shufflevector(pb[kb][0], pb[kb][1], 0, 1, 2, 3, 4, 5, 6, 7);
                dsf[kb] = __builtin_shufflevector(sb[kb][0], sb[kb][1], 0, 1, 2, 3, 4, 5, 6, 7);
            }
            s16x4 dlo[4], dhi[4], qlo[4], qhi[4];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                tr_frag_d_issue(Ds[b], 32 * t2, 16 * db, lane, dlo[db], dhi[db]);
                tr_frag_d_issue(Qs[b], 32 * t2, 16 * db, lane, qlo[db], qhi[db]);
            }
            asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                if (db == 2) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                }
                const bf16x8 dot = tr_join(dlo[db], dhi[db]);
                const bf16x8 qt = tr_join(qlo[db], qhi[db]);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    dv[kb][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dot, pf[kb], dv[kb][db], 0, 0, 0);
                    dk[kb][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt, dsf[kb], dk[kb][db], 0, 0, 0);
                }
            }
            MR_ASTAMP(3 + t2);
        }
        if (t + 1 < nt) { side_compute(); side_store(b ^ 1); }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // this wave's pieces of tile t + 1 have landed; its dS stores are out
        MR_ASTAMP(5);
        __syncthreads();
        MR_ASTAMP(6);
    }
    { const int t = 13; MR_ASTAMP(0); }
    dq_tile(nt - 1, (nt - 1) & 1);
    { const int t = 13; MR_ASTAMP(1); }

    // ---- dK / dV of this wave's 32 keys + the column sums of everything this workgroup stored ----
    // The lane holds [d = 16 db + 4 g + r][key i]: stored from there, a wave instruction wrote 16 rows x 32 bytes -- 16 partial lines -- and the 16 such
    // stores per lane took 7 900 cycles of a 61 000-cycle workgroup (stamps).  The tile goes through the wave's share of the Q / dO buffers instead (free
    // since the loop's last barrier; [32 keys][128 B], 16-byte chunk c of row r at slot c ^ (r & 7)) and leaves as whole 128-byte rows, 16 bytes per lane.
    f32x4 csk[4], csv[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) { csk[db] = f32x4{0.f, 0.f, 0.f, 0.f}; csv[db] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    char* const stg = (wave < 4 ? &Qs[0][0] : &Ds[0][0]) + (wave & 3) * 4096;
#pragma unroll
    for (int part = 0; part < 2; ++part) {           // 0: dK (x 1/8, x "rotary" scale), 1: dV
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const bool ok = ki[kb] < S;
            const int row = kb * 16 + i;
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                const int d = db * 16 + g * 4;
                f32x4 x = part ? dv[kb][db] : dk[kb][db] * 0.125f;
                if (part == 0 && rot_tab != nullptr && d < 32 && ok)
                    x *= *reinterpret_cast<const f32x4*>(rot_tab + (int64_t)rot_row_index(seq_rot0, ki[kb], (int)rot_rows, rot_short) * 32 + d);
                bf16x4 a;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    a[r] = (__bf16)x[r];
                    if (ok) { if (part) csv[db][r] += (float)a[r]; else csk[db][r] += (float)a[r]; }
                }
                *reinterpret_cast<bf16x4*>(stg + row * 128 + (((db * 2 + (g >> 1)) ^ (row & 7)) << 4) + (g & 1) * 8) = a;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = it * 8 + (lane >> 3), ch = lane & 7;
            const u32x4 v = *reinterpret_cast<const u32x4*>(stg + row * 128 + ((ch ^ (row & 7)) << 4));
            const int key = wave * 32 + row;
            if (key < S) *reinterpret_cast<u32x4*>(dqkv + (seq * S + key) * ld + (1 + part) * H + h * 64 + ch * 8) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // the reads are done before the next part overwrites the region
    }
    { const int t = 13; MR_ASTAMP(2); }
    if (colsum != nullptr) {      // wave-uniform: ONE partial row per sequence; this head's 64 columns of q, k, v
        float* prow = colsum + seq * ld;
        // k and v: every wave holds all 64 dims for its keys -> sum over the 16 lanes of a DPP row, then over the 8 waves; q: wave (qdb, qh) holds
        // dims 16 qdb + 4 g .. + 3 for its query half.  All three thirds go to LDS behind ONE barrier (three rounds of write / barrier / sum / barrier
        // were 4 750 cycles of the workgroup's 54 600: stamps)
#pragma unroll
        for (int db = 0; db < 4; ++db) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { csk[db][r] = row16_sum(csk[db][r]); csv[db][r] = row16_sum(csv[db][r]); }
            if (i == 0) {
                *reinterpret_cast<f32x4*>(&red[wave][64 + db * 16 + g * 4]) = csk[db];
                *reinterpret_cast<f32x4*>(&red[wave][128 + db * 16 + g * 4]) = csv[db];
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) csq[r] = row16_sum(csq[r]);
        if (i == 0) *reinterpret_cast<f32x4*>(&red[qh][qdb * 16 + g * 4]) = csq;
        __syncthreads();
        if (tid < 64) prow[h * 64 + tid] = red[0][tid] + red[1][tid];
        else if (tid < 192) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) sum += red[w][tid];
            prow[(tid >> 6) * H + h * 64 + (tid & 63)] = sum;
        }
    }
    MR_ASTAMP_WG(1);
}

// ------------------------------------------------------------------------------------------------ dQ, dK, dV of a SHORT sequence (S <= 32): one WAVE per (sequence, head)
// The audio (S = 31) and span (S = 16) towers: 2304 (sequence, head) pairs of <= 32 positions each.  On the two-pass kernels a pair was a workgroup of
// four waves for <= 32 queries -- two of them idle, two barriers per tile, two launches, every operand fetched twice -- and the pair of launches took
// 33 us against ~14 us of HBM time.  Here a wave owns a pair outright: Q, K and dO are fetched once by LDS-DMA into a region of LDS private to the
// wave (4 KiB images in the tile format of dma_src; no barrier anywhere, the wave's own vmcnt / lgkmcnt waits order everything), K / 8 and V fragments
// and the per-query scalars come straight from global memory, and the arithmetic is the one-pass kernel's (attn_bwd1_kernel) for a single 32-query
// half-tile: S = Q K^T and dP = dO V^T with the key on the lane, P / dS as the B operands of dV^T += dO^T P and dK^T += Q^T dS, dS^T through the
// wave's LDS region once for dQ^T = K^T dS^T.  40 MFMAs per pair: the kernel is a latency chain per wave, so what matters is that every load of a
// wave is in flight at once and that a CU holds many waves (16 KiB of LDS and < 128 registers each: 8-10 per CU).
// Column sums: one partial row per sequence, this head's 64 columns of the q, k and v thirds, written by the wave alone.
constexpr int SMALL_S = 32;
template <bool MASKED>
__global__ __launch_bounds__(256, 2) void attn_bwd_small_kernel(const __bf16* __restrict__ qkv, const int32_t* __restrict__ code,
                                                                const __bf16* __restrict__ o, const __bf16* __restrict__ dout,
                                                                const float* __restrict__ lse, __bf16* __restrict__ dqkv,
                                                                const float* __restrict__ rot_tab, int64_t rot_rows,
                                                                float* __restrict__ colsum, int64_t S, int64_t nh, int64_t npairs) {
    constexpr int IMG = SMALL_S * 128;                                   // [32 rows][64 dims] bf16 in the 128-byte-row tile format
    __shared__ __attribute__((aligned(16))) char Qs[4][IMG];             // row reads (S) and tr reads (dK^T)
    __shared__ __attribute__((aligned(16))) char Ks[4][IMG];             // tr reads (dQ^T)
    __shared__ __attribute__((aligned(16))) char Ds[4][IMG];             // dO: row reads (dP) and tr reads (dV^T)
    __shared__ __attribute__((aligned(16))) char dSs[4][IMG];            // dS^T [32 keys][32 queries (64-wide rows)]: tr reads (dQ^T)
    __shared__ __attribute__((aligned(16))) float Ls[4][SMALL_S], Dl[4][SMALL_S], Us[4][SMALL_S];
    __shared__ __attribute__((aligned(16))) int32_t Cs[4][SMALL_S];
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, i = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t pair = (int64_t)blockIdx.x * 4 + wave;
    if (pair >= npairs) return;                                          // (wave-uniform; no barrier in this kernel)
    const int64_t seq = pair / nh, h = pair % nh;
    const int64_t H = nh * 64, ld = 3 * H;
    const __bf16* base = qkv + seq * S * ld;
    const int32_t* code_seq = MASKED ? code + seq * S : nullptr;
    const float inv_S = 1.0f / (float)S;
    const int seq_rot0 = rot_tab != nullptr ? (int)((seq * S) % rot_rows) : 0;
    const bool rot_short = rot_rows < S;
    char* const qs = Qs[wave];
    char* const ks = Ks[wave];
    char* const dsm = Ds[wave];
    char* const dss = dSs[wave];

    // ---- everything this wave reads, requested at once: Q, K, dO images by LDS-DMA (4 pieces of 8 rows each; rows past S are beyond the
    // descriptors' extent: zeros), K / 8 and V fragments, the per-query scalars ----
    const __bf16* Qg = base + h * 64;
    const __bf16* Kg = base + H + h * 64;
    const __bf16* Dg = dout + seq * S * H + h * 64;
    const __bf16* Og = o + seq * S * H + h * 64;
    const float* Lg = lse + (seq * nh + h) * S;
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(Qg), 0, (int)(((S - 1) * ld + 64) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(Kg), 0, (int)(((S - 1) * ld + 64) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(Dg), 0, (int)(((S - 1) * H + 64) * 2), 0x00020000);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, MR_LDS_PTR(void, qs + p * 1024), 16, dma_src(p, lane, ld), 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, MR_LDS_PTR(void, ks + p * 1024), 16, dma_src(p, lane, ld), 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, MR_LDS_PTR(void, dsm + p * 1024), 16, dma_src(p, lane, H), 0, 0, 0);
    }
    bf16x8 kf[2][2], vf[2][2];
    int ki[2], ck[2];
    float nkl[2], unil[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        ki[kb] = kb * 16 + i;
        const bool ok = ki[kb] < S;
#pragma unroll
        for (int dd = 0; dd < 2; ++dd) {
            u32x4 v = {0u, 0u, 0u, 0u}, w = {0u, 0u, 0u, 0u};
            if (ok) {
                v = *reinterpret_cast<const u32x4*>(base + (int64_t)ki[kb] * ld + H + h * 64 + dd * 32 + g * 8);
                w = *reinterpret_cast<const u32x4*>(base + (int64_t)ki[kb] * ld + 2 * H + h * 64 + dd * 32 + g * 8);
            }
            kf[kb][dd] = scale_eighth(v);
            vf[kb][dd] = __builtin_bit_cast(bf16x8, w);
        }
        ck[kb] = ok ? (MASKED ? code_seq[ki[kb]] : 0) : CODE_NONE;
        nkl[kb] = ok ? NEG_BIAS : -INFINITY;
        unil[kb] = ok ? 1.0f : 0.0f;
    }
    {   // per-query scalars: lane (q = lane >> 1, half = lane & 1) holds 32 dims of dO and O -> delta = rowsum(dO * O); the even lane stores
        const int q = lane >> 1, hf = lane & 1;
        const bool ok = q < S;
        float dsum = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            u32x4 a = {0u, 0u, 0u, 0u}, bb = {0u, 0u, 0u, 0u};
            if (ok) {
                a = *reinterpret_cast<const u32x4*>(Dg + (int64_t)q * H + hf * 32 + c * 8);
                bb = *reinterpret_cast<const u32x4*>(Og + (int64_t)q * H + hf * 32 + c * 8);
            }
            float x[8], y[8];
            unpack8(a, x);
            unpack8(bb, y);
#pragma unroll
            for (int e = 0; e < 8; ++e) dsum += x[e] * y[e];
        }
        dsum += __shfl_xor(dsum, 1, 64);
        const float L = ok ? Lg[q] : INFINITY;
        const bool pad = MASKED && ok && L < PAD_LSE;
        int c = ok ? (MASKED ? code_seq[q] : 0) : CODE_PADQ;
        if (hf == 0) {
            Ls[wave][q] = -L * LOG2E;
            Dl[wave][q] = dsum;
            Cs[wave][q] = (c < 0) ? CODE_PADQ : c;
            Us[wave][q] = pad ? inv_S : 0.f;
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");          // the wave's images have landed, its scalars are stored
    __builtin_amdgcn_sched_barrier(0);

    // ---- S, dP, P, dS of the 32 x 32 block; dS^T -> the wave's LDS region ----
    const bool keys_inside = !MASKED && SMALL_S <= S;
    bf16x4 pb[2][2], sb[2][2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const bf16x8 q0f = row_frag_d(qs, qb * 16, 0, lane), q1f = row_frag_d(qs, qb * 16, 1, lane);
        const bf16x8 d0f = row_frag_d(dsm, qb * 16, 0, lane), d1f = row_frag_d(dsm, qb * 16, 1, lane);
        const f32x4 l4 = *reinterpret_cast<const f32x4*>(&Ls[wave][qb * 16 + g * 4]);
        const f32x4 e4 = *reinterpret_cast<const f32x4*>(&Dl[wave][qb * 16 + g * 4]);
        const i32x4 c4 = *reinterpret_cast<const i32x4*>(&Cs[wave][qb * 16 + g * 4]);
        f32x4 u4 = {0.f, 0.f, 0.f, 0.f};
        if (MASKED) u4 = *reinterpret_cast<const f32x4*>(&Us[wave][qb * 16 + g * 4]);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x4 st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q0f, kf[kb][0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q1f, kf[kb][1], st, 0, 0, 0);
            f32x4 dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(d0f, vf[kb][0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(d1f, vf[kb][1], dp, 0, 0, 0);
            f32x4 pv;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // allowed = same code (a PAD key's -1 and a missing key's -2 equal no query code; unmasked: codes are 0, missing keys -2, a
                // missing query has l4 = -inf: p = 0)
                const float sc = (keys_inside || c4[r] == ck[kb]) ? st[r] : st[r] + nkl[kb];
                pv[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sc, LOG2E, l4[r]));
            }
            if (MASKED) {         // rows without allowed key (PAD queries): uniform weight 1 / S over the existing keys
#pragma unroll
                for (int r = 0; r < 4; ++r) pv[r] = (u4[r] > 0.f) ? u4[r] * unil[kb] : pv[r];
            }
            bf16x4 p4, w4;
#pragma unroll
            for (int r = 0; r < 4; ++r) { p4[r] = (__bf16)pv[r]; w4[r] = (__bf16)(pv[r] * (dp[r] - e4[r])); }
            pb[kb][qb] = p4;
            sb[kb][qb] = w4;
            {   // dS^T: key row 16 kb + i, queries 16 qb + 4 g .. + 3 (8 bytes), in the images' swizzled format
                const int row = 16 * kb + i, col = 16 * qb + 4 * g;
                char* dst = dss + row * 128 + (((col >> 3) ^ dswz(row)) << 4) + (col & 7) * 2;
                *reinterpret_cast<bf16x4*>(dst) = w4;
            }
        }
    }
    // ---- dV^T += dO^T P, dK^T += Q^T dS over the 32 queries (one k-step) ----
    f32x4 dk[2][4], dv[2][4];
    {
        bf16x8 pf[2], dsf[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            pf[kb] = __builtin_shufflevector(pb[kb][0], pb[kb][1], 0, 1, 2, 3, 4, 5, 6, 7);
            dsf[kb] = __builtin_shufflevector(sb[kb][0], sb[kb][1], 0, 1, 2, 3, 4, 5, 6, 7);
        }
        s16x4 dlo[4], dhi[4], qlo[4], qhi[4];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            tr_frag_d_issue(dsm, 0, 16 * db, lane, dlo[db], dhi[db]);
            tr_frag_d_issue(qs, 0, 16 * db, lane, qlo[db], qhi[db]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // (also: the dS^T stores above are out)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            const bf16x8 dot = tr_join(dlo[db], dhi[db]);
            const bf16x8 qt = tr_join(qlo[db], qhi[db]);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                dv[kb][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dot, pf[kb], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                dk[kb][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt, dsf[kb], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            }
        }
    }
    // ---- dQ^T = K^T dS^T over the 32 keys (one k-step): all four 16-dim blocks, both query blocks ----
    f32x4 dq[2][4];
    {
        s16x4 klo[4], khi[4], slo[2], shi[2];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int db = 0; db < 4; ++db) tr_frag_d_issue(ks, 0, 16 * db, lane, klo[db], khi[db]);
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) tr_frag_d_issue(dss, 0, 16 * qb, lane, slo[qb], shi[qb]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            const bf16x8 kt = tr_join(klo[db], khi[db]);
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
                dq[qb][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt, tr_join(slo[qb], shi[qb]), f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        }
    }
    // ---- stores: the lane holds [d = 16 db + 4 g + r][position i] of each 16-position block; column sums over the positions ----
    f32x4 csq[4], csk[4], csv[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) { csq[db] = f32x4{0.f, 0.f, 0.f, 0.f}; csk[db] = f32x4{0.f, 0.f, 0.f, 0.f}; csv[db] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int pbk = 0; pbk < 2; ++pbk) {
        const int pos = pbk * 16 + i;
        if (pos < S) {
            const int rr = rot_tab != nullptr ? rot_row_index(seq_rot0, pos, (int)rot_rows, rot_short) : 0;
            __bf16* row = dqkv + (seq * S + pos) * ld + h * 64;
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                const int d = db * 16 + g * 4;
                f32x4 xq = dq[pbk][db] * 0.125f, xk = dk[pbk][db] * 0.125f;
                if (rot_tab != nullptr && d < 32) {
                    const f32x4 tv = *reinterpret_cast<const f32x4*>(rot_tab + (int64_t)rr * 32 + d);
                    xq *= tv;
                    xk *= tv;
                }
                bf16x4 a, b, c;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    a[r] = (__bf16)xq[r]; b[r] = (__bf16)xk[r]; c[r] = (__bf16)dv[pbk][db][r];
                    csq[db][r] += (float)a[r]; csk[db][r] += (float)b[r]; csv[db][r] += (float)c[r];
                }
                *reinterpret_cast<bf16x4*>(row + d) = a;
                *reinterpret_cast<bf16x4*>(row + H + d) = b;
                *reinterpret_cast<bf16x4*>(row + 2 * H + d) = c;
            }
        }
    }
    if (colsum != nullptr) {      // wave-uniform: ONE partial row per sequence; this head's 64 columns of q, k, v
        float* prow = colsum + seq * ld + h * 64;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { csq[db][r] = row16_sum(csq[db][r]); csk[db][r] = row16_sum(csk[db][r]); csv[db][r] = row16_sum(csv[db][r]); }
            if (i == 0) {
                *reinterpret_cast<f32x4*>(prow + db * 16 + g * 4) = csq[db];
                *reinterpret_cast<f32x4*>(prow + H + db * 16 + g * 4) = csk[db];
                *reinterpret_cast<f32x4*>(prow + 2 * H + db * 16 + g * 4) = csv[db];
            }
        }
    }
}

// MR_ATTN_QB_S (diagnostic): sequences longer than this use two query / key blocks per workgroup (default 64)
static int64_t attn_qb_threshold() {
    static int64_t v = -1;
    if (v < 0) v = mr_env_int("MR_ATTN_QB_S", 64);
    return v;
}

template <int QB>
static dim3 attn_grid(int64_t S, int64_t nh, int64_t nseq) {
    return dim3((unsigned)(((S + 64 * QB - 1) / (64 * QB)) * nh * nseq));      // 1-D: see attn_block()
}

}  // namespace

extern "C" int mr_attention_fwd(const void* qkv, const int32_t* code, void* out, float* lse, int64_t nseq, int64_t S,
                                int64_t nh, void* stream) {
    MR_CHECK_ARG(qkv && out && lse, "mr_attention_fwd: null pointer");
    MR_CHECK_ARG(nseq > 0 && S > 0 && nh > 0, "mr_attention_fwd: bad shape nseq=%ld S=%ld nh=%ld", (long)nseq, (long)S, (long)nh);
    MR_CHECK_ARG(nseq <= 65535 && nh <= 65535, "mr_attention_fwd: nseq / nh exceed grid limits");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const __bf16* q = static_cast<const __bf16*>(qkv);
    __bf16* o = static_cast<__bf16*>(out);
    const int tm = mr_opts().attn_tile_modes;
    const int64_t nt_ = (S + TK - 1) / TK;
    const int64_t ntp_ = (nt_ + 11) / 12 * 12;                            // (the kernels fill their tables in unconditional batches of 12 tiles)
    const size_t fwd_smem = (size_t)(ntp_ * TK * 8 + ntp_ * 4);             // key codes, bias and tile classes of a sequence (dynamic LDS)
    MR_CHECK_ARG(fwd_smem <= 100 * 1024, "mr_attention_fwd: S = %ld is beyond what one workgroup's LDS holds (S <= 12 700)", (long)S);
    const bool two = S > attn_qb_threshold();        // short sequences (audio 31, span 16): one 16-query block per wave
    if (two && code) hipLaunchKernelGGL((attn_fwd_kernel<2, true>), attn_grid<2>(S, nh, nseq), dim3(256), fwd_smem, s, q, code, o, lse, S, nh, tm);
    else if (two) hipLaunchKernelGGL((attn_fwd_kernel<2, false>), attn_grid<2>(S, nh, nseq), dim3(256), fwd_smem, s, q, code, o, lse, S, nh, tm);
    else if (code) hipLaunchKernelGGL((attn_fwd_kernel<1, true>), attn_grid<1>(S, nh, nseq), dim3(256), fwd_smem, s, q, code, o, lse, S, nh, tm);
    else hipLaunchKernelGGL((attn_fwd_kernel<1, false>), attn_grid<1>(S, nh, nseq), dim3(256), fwd_smem, s, q, code, o, lse, S, nh, tm);
    MR_CHECK_LAUNCH("mr_attention_fwd");
    return MR_OK;
}

// The one-pass kernel takes sequences that one workgroup's 256 keys cover and that are long enough to fill its eight waves (the short
// towers, S = 31 / 16, stay on the two-pass kernels' 64-key workgroups).  Option "attn_onepass": -1 = this rule | 0 = never | 1 = whenever S <= 256.
static bool attn_onepass(int64_t S) {
    const int v = mr_opts().attn_onepass;
    if (v == 0 || S > KP) return false;
    return v == 1 || S > 128;
}
// ... and sequences of at most 32 positions go to the one-wave-per-pair kernel (option "attn_onepass": -1 or 1; 0 = the kernel pair)
static bool attn_small(int64_t S) { return mr_opts().attn_onepass != 0 && S <= SMALL_S; }

extern "C" int64_t mr_attention_bwd_colsum_rows(int64_t nseq, int64_t S) {
    if (attn_small(S) || attn_onepass(S)) return nseq;                  // one partial row per sequence
    const int64_t per = (S > attn_qb_threshold()) ? 128 : 64;          // queries (keys) per workgroup
    return nseq * ((S + per - 1) / per);
}

extern "C" int mr_attention_bwd(const void* qkv, const int32_t* code, const void* out, const void* dout, const float* lse,
                                float* delta, void* dqkv, const float* rot_tab, int64_t rot_rows, float* colsum, int64_t nseq,
                                int64_t S, int64_t nh, void* stream) {
    MR_CHECK_ARG(qkv && out && dout && lse && delta && dqkv, "mr_attention_bwd: null pointer");
    MR_CHECK_ARG(nseq > 0 && S > 0 && nh > 0, "mr_attention_bwd: bad shape");
    MR_CHECK_ARG(nseq <= 65535 && nh <= 65535, "mr_attention_bwd: nseq / nh exceed grid limits");
    MR_CHECK_ARG(!rot_tab || rot_rows > 0, "mr_attention_bwd: rot_rows must be > 0");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const __bf16* q = static_cast<const __bf16*>(qkv);
    const __bf16* d = static_cast<const __bf16*>(dout);
    __bf16* g = static_cast<__bf16*>(dqkv);
    const __bf16* oo = static_cast<const __bf16*>(out);       // delta = rowsum(dO * O) is computed by the dQ kernel (for itself and for dK / dV)
    if (attn_small(S)) {
        const int64_t npairs = nh * nseq;
        const dim3 grid((unsigned)((npairs + 3) / 4));
        if (code) hipLaunchKernelGGL((attn_bwd_small_kernel<true>), grid, dim3(256), 0, s, q, code, oo, d, lse, g, rot_tab, rot_rows, colsum, S, nh, npairs);
        else hipLaunchKernelGGL((attn_bwd_small_kernel<false>), grid, dim3(256), 0, s, q, code, oo, d, lse, g, rot_tab, rot_rows, colsum, S, nh, npairs);
        MR_CHECK_LAUNCH("mr_attention_bwd (short sequences)");
        return MR_OK;
    }
    if (attn_onepass(S)) {
        const dim3 grid((unsigned)(nh * nseq));
        if (code) hipLaunchKernelGGL((attn_bwd1_kernel<true>), grid, dim3(512), 0, s, q, code, oo, d, lse, g, rot_tab, rot_rows, colsum, S, nh);
        else hipLaunchKernelGGL((attn_bwd1_kernel<false>), grid, dim3(512), 0, s, q, code, oo, d, lse, g, rot_tab, rot_rows, colsum, S, nh);
        MR_CHECK_LAUNCH("mr_attention_bwd (one pass)");
        return MR_OK;
    }
    const bool two = S > attn_qb_threshold();
    const int tm = mr_opts().attn_tile_modes;
    // the dK / dV kernel keeps a sequence's per-query scalars in dynamic LDS: 16 B per padded position + 8 B per tile (beside 52 KiB of static tiles)
    const int64_t nt_ = (S + TK - 1) / TK;
    const int64_t ntp_ = (nt_ + 11) / 12 * 12;                            // (the kernels fill their tables in unconditional batches of 12 tiles)
    const size_t dkv_smem = (size_t)(ntp_ * TK * 16 + ntp_ * 8);
    const size_t dq_smem = (size_t)(ntp_ * TK * 8 + ntp_ * 4);              // the dQ kernel: codes, bias and classes of the sequence's keys
    MR_CHECK_ARG(dkv_smem <= 100 * 1024, "mr_attention_bwd: S = %ld is beyond what one workgroup's LDS holds (S <= 6336)", (long)S);
#define MR_LAUNCH_BWD(QB, M)                                                                                                  \
    do {                                                                                                                      \
        hipLaunchKernelGGL((attn_bwd_dq_kernel<QB, M>), attn_grid<QB>(S, nh, nseq), dim3(256), dq_smem, s, q, code, oo, d, lse, delta, g, \
                           rot_tab, rot_rows, colsum, S, nh, tm);                                                             \
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<QB, M>), attn_grid<QB>(S, nh, nseq), dim3(256), dkv_smem, s, q, code, d, lse, delta, g, \
                           rot_tab, rot_rows, colsum, S, nh, tm);                                                             \
    } while (0)
    if (two && code) MR_LAUNCH_BWD(2, true);
    else if (two) MR_LAUNCH_BWD(2, false);
    else if (code) MR_LAUNCH_BWD(1, true);
    else MR_LAUNCH_BWD(1, false);
#undef MR_LAUNCH_BWD
    MR_CHECK_LAUNCH("mr_attention_bwd");
    return MR_OK;
}

#ifdef MR_ATTN_STAMPS
extern "C" int mr_diag_attn_stamps(unsigned long long* host_out) {      // diagnostic build only
    (void)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_attn_stamps), sizeof(unsigned long long) * 512 * 16 * 8);
    unsigned long long z = 0;
    (void)z;
    return 0;
}
#endif
