// bf16 MFMA GEMM with fused epilogue for gfx950 (MI355X).
//
// Replaces what XLA generates for the reference's flax Dense/DenseGeneral calls
// (mreserve/modeling.py:228-236, 252-255, 371, 402, 453, 631) and their transposes in backward.
//
// Structure (round 1): 128x128x64 block tile, 256 threads = 4 waves in 2x2, each wave a 64x64
// sub-tile = 4x4 v_mfma_f32_16x16x32_bf16 accumulators.  Operands are staged global -> VGPR -> LDS
// (the next K-tile's global loads are in flight while the current tile is multiplied).
// An operand whose contraction index is contiguous in memory is read from LDS with ds_read_b128;
// one whose contraction index is the ROW index (flax [in,out] kernels in forward, both operands in
// wgrad) is kept row-major in LDS and read with ds_read_b64_tr_b16, so no transposed copy of any
// weight or activation is ever materialised in HBM.
// Epilogue: bias / "rotary" diagonal scale / activation in the MFMA layout, then the tile goes
// through LDS so that residual / gelu' operands are read and C is written as 16-byte row segments.
#include <stdlib.h>
#include "mr_common.h"
#include "mr_options.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int LDK = 72;    // K-contiguous tile  [128 rows][64 k + 8 pad]   (144-B rows)
constexpr int LDT = 144;   // K-strided tile     [64 k][128 cols + 16 pad]  (288-B rows: 8 k-rows tile all 64 banks)
constexpr int TILE_ELEMS = 128 * LDK;  // == 64 * LDT == 9216
constexpr int LDC = 136;   // epilogue staging tile [128][128 + 8 pad] bf16
static_assert(128 * LDK == 64 * LDT, "tile sizes");
static_assert(128 * LDC <= 2 * TILE_ELEMS, "epilogue tile must fit the staging LDS");

// One operand tile's 4 16-byte chunks per thread.
//  TR = false: tile rows = the operand's own rows (128 of them), 8 chunks along K per row.
//  TR = true : tile rows = K (64 of them), 16 chunks along the operand's own index per row.
template <bool TR>
__device__ __forceinline__ void load_tile(const __bf16* __restrict__ base, int64_t ld, int64_t own0, int64_t k0,
                                          int64_t own_n, int64_t K, int tid, u32x4 (&r)[4]) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int c = tid + 256 * it;
        int64_t grow, gcol;
        bool ok;
        if (!TR) {
            const int row = c >> 3, ch = c & 7;
            grow = own0 + row;
            gcol = k0 + 8 * ch;
            ok = (grow < own_n) && (gcol < K);
        } else {
            const int row = c >> 4, ch = c & 15;
            grow = k0 + row;
            gcol = own0 + 8 * ch;
            ok = (grow < K) && (gcol < own_n);
        }
        u32x4 v = {0u, 0u, 0u, 0u};
        if (ok) v = *reinterpret_cast<const u32x4*>(base + grow * ld + gcol);
        r[it] = v;
    }
}

template <bool TR>
__device__ __forceinline__ void store_tile(__bf16* tile, int tid, const u32x4 (&r)[4]) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int c = tid + 256 * it;
        int off;
        if (!TR) off = (c >> 3) * LDK + 8 * (c & 7);
        else     off = (c >> 4) * LDT + 8 * (c & 15);
        *reinterpret_cast<u32x4*>(tile + off) = r[it];
    }
}

// Fragment of 16 "own" indices x 32 k for v_mfma_f32_16x16x32_bf16: lane l holds own = own0 + (l & 15),
// k = 8*(l >> 4) + j, j = 0..7.
template <bool TR>
__device__ __forceinline__ bf16x8 read_frag(const __bf16* tile, int own0, int kk, int lane) {
    const int g = lane >> 4, i = lane & 15;
    if (!TR) {
        return *reinterpret_cast<const bf16x8*>(tile + (own0 + i) * LDK + kk * 32 + g * 8);
    } else {
        // ds_read_b64_tr_b16: the 16-lane group reads a 4(k) x 16(own) block; lane 4q+p gives the address of
        // row q, columns 4p..4p+3 and receives column (l & 15), rows 0..3.
        const int q = i >> 2, p = i & 3;
        const __bf16* a0 = tile + (kk * 32 + g * 8 + q) * LDT + own0 + 4 * p;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(MR_LDS_PTR(s16x4, a0));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(MR_LDS_PTR(s16x4, a0 + 4 * LDT));
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, both);
    }
}

// gridDim.y = number of K splits: split s handles k-tiles [s*kt_per_split, min((s+1)*kt_per_split, nk)) and, when
// gridDim.y > 1, stores its fp32 partial tile to p.workspace[s][M][N] (summed by splitk_reduce_kernel).
// One 128 x 128 output tile (`wgid` of a problem with tiles_n column tiles), K range blockIdx.y when the grid has a y extent.
template <bool TA, bool TB>
__device__ __forceinline__ void gemm_bf16_tile(const mr_gemm_args& p, __bf16* smem, int wgid, int tiles_n, int kt_per_split) {
    __bf16* As = smem;
    __bf16* Bs = smem + TILE_ELEMS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int g = lane >> 4, li = lane & 15;
    const int64_t m0 = (int64_t)(wgid / tiles_n) * BM;
    const int64_t n0 = (int64_t)(wgid % tiles_n) * BN;

    const __bf16* A = static_cast<const __bf16*>(p.A);
    const __bf16* B = static_cast<const __bf16*>(p.B);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    u32x4 ra[4], rb[4];
    const int64_t nk_all = (p.K + BK - 1) / BK;
    const int64_t kt0 = (int64_t)blockIdx.y * kt_per_split;
    const int64_t nk = (kt0 + kt_per_split < nk_all) ? kt0 + kt_per_split : nk_all;
    load_tile<TA>(A, p.lda, m0, kt0 * BK, p.M, p.K, tid, ra);
    load_tile<!TB>(B, p.ldb, n0, kt0 * BK, p.N, p.K, tid, rb);
    store_tile<TA>(As, tid, ra);
    store_tile<!TB>(Bs, tid, rb);
    __syncthreads();

    for (int64_t kt = kt0; kt < nk; ++kt) {
        const bool more = (kt + 1 < nk);
        if (more) {
            load_tile<TA>(A, p.lda, m0, (kt + 1) * BK, p.M, p.K, tid, ra);
            load_tile<!TB>(B, p.ldb, n0, (kt + 1) * BK, p.N, p.K, tid, rb);
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = read_frag<TA>(As, wm * 64 + i * 16, kk, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[j] = read_frag<!TB>(Bs, wn * 64 + j * 16, kk, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (more) {
            store_tile<TA>(As, tid, ra);
            store_tile<!TB>(Bs, tid, rb);
            __syncthreads();
        }
    }

    if (gridDim.y > 1) {   // split-K partial: raw fp32 accumulators
        float* Wp = static_cast<float*>(p.workspace) + (int64_t)blockIdx.y * p.M * p.N;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t n = n0 + wn * 64 + j * 16 + li;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t m = m0 + wm * 64 + i * 16 + g * 4 + r;
                    if (m < p.M && n < p.N) Wp[m * p.N + n] = acc[i][j][r];
                }
            }
        return;
    }

    // ---------------- epilogue, phase 1: MFMA layout (col = lane & 15, row = 4*(lane >> 4) + r) ----------------
    const __bf16* bias = static_cast<const __bf16*>(p.bias);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t n = n0 + wn * 64 + j * 16 + li;
        float bv = 0.f;
        if (bias != nullptr && n < p.N) bv = (float)bias[n];
        const bool rot = (p.rot_tab != nullptr) && (n < p.rot_cols) && ((n & 63) < 32);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[i][j][r] + bv;
                if (rot) {
                    const int64_t m = m0 + wm * 64 + i * 16 + g * 4 + r;
                    if (m < p.M) v *= p.rot_tab[(m % p.rot_rows) * 32 + (n & 63)];
                }
                acc[i][j][r] = v;
            }
        }
    }

    if (p.c_dtype == MR_DT_F32) {
        float* C = static_cast<float*>(p.C);
        int li_here, g_here;              // (opaque copies: keeps this path's address arithmetic out of the common path, see phase 2 below)
        asm volatile("v_mov_b32 %0, %1" : "=v"(li_here) : "v"(li));
        asm volatile("v_mov_b32 %0, %1" : "=v"(g_here) : "v"(g));
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t n = n0 + wn * 64 + j * 16 + li_here;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t m = m0 + wm * 64 + i * 16 + g_here * 4 + r;
                    if (m < p.M && n < p.N) C[m * p.ldc + n] = acc[i][j][r];
                }
            }
        return;
    }

    __bf16* Cs = smem;
    const int npass = (p.c2 != nullptr) ? 2 : 1;
    for (int pass = 0; pass < npass; ++pass) {
        const bool final_pass = (pass == npass - 1);
        const bool do_act = final_pass && (p.act == MR_ACT_GELU1702);
        const bool do_dact = !final_pass && (p.act == MR_ACT_GELU1702);     // c2 pass under GELU stores gelu'(v)
        // stage the tile as bf16
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[i][j][r];
                    if (do_act) v = gelu1702(v);
                    if (do_dact) v = gelu1702_grad(v);
                    Cs[(wm * 64 + i * 16 + g * 4 + r) * LDC + wn * 64 + j * 16 + li] = (__bf16)v;
                }
        __syncthreads();
        // phase 2: 16-byte row segments
        __bf16* Cout = static_cast<__bf16*>(final_pass ? p.C : p.c2);
        const __bf16* R = final_pass ? static_cast<const __bf16*>(p.residual) : nullptr;
        const __bf16* X = final_pass ? static_cast<const __bf16*>(p.aux) : nullptr;
        // the row / column / address arithmetic below depends only on the thread and the arguments, so the optimiser computed all of it ahead of the
        // staging loop above and spilled it across that loop (22-42 registers): an opaque zero added to the thread index pins it here
        int tid_here;
        asm volatile("v_mov_b32 %0, %1" : "=v"(tid_here) : "v"(tid));
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int c = tid_here + 256 * it;
            const int row = c >> 4, ch = c & 15;
            const int64_t gm = m0 + row, gn = n0 + 8 * ch;
            if (gm < p.M && gn < p.N) {
                u32x4 raw = *reinterpret_cast<const u32x4*>(Cs + row * LDC + 8 * ch);
                int64_t orow = gm;
                if (p.out_grp > 0) orow = (gm / p.out_grp) * p.out_grp_stride + p.out_grp_off + gm % p.out_grp;
                if (R != nullptr || X != nullptr) {
                    float f[8];
                    unpack8(raw, f);
                    if (R != nullptr) {
                        float rr[8];
                        unpack8(*reinterpret_cast<const u32x4*>(R + orow * p.ldr + gn), rr);
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] += rr[e];
                    }
                    if (X != nullptr) {
                        float xx[8];
                        unpack8(*reinterpret_cast<const u32x4*>(X + orow * p.ldaux + gn), xx);
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] = (float)(__bf16)f[e] * xx[e];
                    }
                    raw = pack8(f);
                }
                *reinterpret_cast<u32x4*>(Cout + orow * p.ldc + gn) = raw;
            }
        }
        if (!final_pass) __syncthreads();
    }
}

// gridDim.y = number of K splits (see gemm_bf16_tile)
template <bool TA, bool TB>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(const mr_gemm_args p, int tiles_n, int kt_per_split) {
    __shared__ __attribute__((aligned(16))) __bf16 smem[2 * TILE_ELEMS];
    // XCD-aware (bijective) block -> tile map: blocks that share an XCD's L2 get neighbouring tiles.
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int xcd = orig & 7, qd = nwg >> 3, rm = nwg & 7;
    const int wgid = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (orig >> 3);
    gemm_bf16_tile<TA, TB>(p, smem, wgid, tiles_n, kt_per_split);
}

// Several small problems of ONE operand layout in one launch (the contrastive loss's six logits / d-query / d-key products, each a
// 17-23 us latency chain of 12 k-tiles on one or two workgroups when launched alone): block -> (problem, tile), no split-K.
constexpr int SMALL_MAXG = 8;
struct SmallGroup {
    int count;
    int tile_start[SMALL_MAXG + 1];
    int tiles_n[SMALL_MAXG];
    mr_gemm_args p[SMALL_MAXG];
};
template <bool TA, bool TB>
__global__ __launch_bounds__(256, 2) void gemm_bf16_grouped_kernel(const SmallGroup ga) {
    __shared__ __attribute__((aligned(16))) __bf16 smem[2 * TILE_ELEMS];
    int j = 0;
#pragma unroll
    for (int k = 1; k < SMALL_MAXG; ++k) j += (k < ga.count && (int)blockIdx.x >= ga.tile_start[k]);
    const mr_gemm_args& p = ga.p[j];
    gemm_bf16_tile<TA, TB>(p, smem, (int)blockIdx.x - ga.tile_start[j], ga.tiles_n[j], (int)((p.K + BK - 1) / BK));
}

// C[m,n] = sum_s partial[s][m][n] (+ bias[n]); bf16 or fp32 output; N % 4 == 0
__global__ void splitk_reduce_kernel(const float* __restrict__ ws, int splits, int64_t M, int64_t N, const __bf16* __restrict__ bias,
                                     void* __restrict__ C, int64_t ldc, int c_dtype) {
    const int64_t MN = M * N;
    for (int64_t i4 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i4 < MN; i4 += (int64_t)gridDim.x * blockDim.x * 4) {
        f32x4 acc = *reinterpret_cast<const f32x4*>(ws + i4);
        for (int s = 1; s < splits; ++s) acc += *reinterpret_cast<const f32x4*>(ws + (int64_t)s * MN + i4);
        const int64_t m = i4 / N, n = i4 % N;
        if (bias != nullptr) {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += (float)bias[n + e];
        }
        if (c_dtype == MR_DT_F32) {
            *reinterpret_cast<f32x4*>(static_cast<float*>(C) + m * ldc + n) = acc;
        } else {
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (__bf16)acc[e];
            *reinterpret_cast<bf16x4*>(static_cast<__bf16*>(C) + m * ldc + n) = o;
        }
    }
}

void launch_splitk_reduce(const mr_gemm_args* a, int64_t splits, hipStream_t s) {
    const int64_t n4 = a->M * a->N / 4;
    int64_t blocks = (n4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const float*>(a->workspace),
                       (int)splits, a->M, a->N, static_cast<const __bf16*>(a->bias), a->C, a->ldc, (int)a->c_dtype);
}

}  // namespace

// gemm256.hip
bool mr_gemm256_eligible(const mr_gemm_args* a);
int mr_gemm256_launch(const mr_gemm_args* a, hipStream_t s, void (*reduce)(const mr_gemm_args*, int64_t, hipStream_t));

bool mr_gemm256_grouped(const mr_gemm_args* list, int count, hipStream_t s);
// gemm3.hip
bool mr_gemm3_eligible(const mr_gemm_args* a);
int mr_gemm3_launch(const mr_gemm_args* a, hipStream_t s);
bool mr_gemm3_tn_grouped(const mr_gemm_args* list, int count, hipStream_t s);
// gemm5.hip
bool mr_gemm5_wanted(const mr_gemm_args* a);
int mr_gemm5_launch(const mr_gemm_args* a, hipStream_t s);

static int use_gemm256() {
    static int v = -1;
    if (v < 0) v = mr_env_int("MR_GEMM_V1_ONLY", 0) ? 0 : 1;
    return v && !g_mr_opt_v1_only;
}

extern "C" int mr_gemm(const mr_gemm_args* a, void* stream) {
    MR_CHECK_ARG(a != nullptr, "mr_gemm: null args");
    MR_CHECK_ARG(a->M > 0 && a->N > 0 && a->K > 0, "mr_gemm: empty problem M=%ld N=%ld K=%ld", (long)a->M, (long)a->N, (long)a->K);
    MR_CHECK_ARG(a->A && a->B && a->C, "mr_gemm: null operand");
    MR_CHECK_ARG(a->lda % 8 == 0 && a->ldb % 8 == 0, "mr_gemm: lda/ldb must be multiples of 8 (got %ld, %ld)", (long)a->lda, (long)a->ldb);
    // the contiguous dimension of each operand is consumed in 16-byte chunks: a ragged extent is allowed when the
    // row is padded (with zeros, by the caller) up to the next multiple of 8 inside the leading dimension
    MR_CHECK_ARG(((a->transA ? a->M : a->K) + 7) / 8 * 8 <= a->lda, "mr_gemm: A rows must be padded to a multiple of 8 elements");
    MR_CHECK_ARG(((a->transB ? a->K : a->N) + 7) / 8 * 8 <= a->ldb, "mr_gemm: B rows must be padded to a multiple of 8 elements");
    MR_CHECK_ARG(((uintptr_t)a->A % 16) == 0 && ((uintptr_t)a->B % 16) == 0 && ((uintptr_t)a->C % 16) == 0,
                 "mr_gemm: operands must be 16-byte aligned");
    if (a->c_dtype == MR_DT_BF16) {
        MR_CHECK_ARG(a->N % 8 == 0 && a->ldc % 8 == 0, "mr_gemm: bf16 output needs N, ldc multiples of 8");
        MR_CHECK_ARG(!a->residual || a->ldr % 8 == 0, "mr_gemm: ldr must be a multiple of 8");
        MR_CHECK_ARG(!a->aux || a->ldaux % 8 == 0, "mr_gemm: ldaux must be a multiple of 8");
    } else {
        MR_CHECK_ARG(a->c_dtype == MR_DT_F32, "mr_gemm: bad c_dtype %d", a->c_dtype);
        MR_CHECK_ARG(!a->residual && !a->aux && !a->c2 && a->act == MR_ACT_NONE && a->out_grp == 0,
                     "mr_gemm: fp32 output supports bias only");
    }
    MR_CHECK_ARG(!a->rot_tab || a->rot_rows > 0, "mr_gemm: rot_rows must be > 0");
    MR_CHECK_ARG(!a->colsum || (use_gemm256() && mr_gemm_colsum_supported(a) && a->ldcs >= a->N && ((uintptr_t)a->colsum % 16) == 0 && a->ldcs % 4 == 0),
                 "mr_gemm: colsum is only produced by the 256-row kernel with the aux epilogue (ask mr_gemm_colsum_supported)");
    mr_gemm_args with_ws;                    // the current handle's split-K workspace when the caller brings none
    if (a->workspace == nullptr && mr_current_handle() != nullptr && mr_current_handle()->ws != nullptr) {
        int dev_now = -1;                    // the workspace lives on the handle's device: never hand it to a launch on another one
        (void)hipGetDevice(&dev_now);
        MR_CHECK_ARG(dev_now == mr_current_handle()->device, "mr_gemm: the current handle was created for device %d, this thread launches on device %d",
                     mr_current_handle()->device, dev_now);
        with_ws = *a;
        with_ws.workspace = mr_current_handle()->ws;
        with_ws.workspace_bytes = mr_current_handle()->ws_bytes;
        a = &with_ws;
    }
    if (use_gemm256() && mr_gemm5_wanted(a)) return mr_gemm5_launch(a, static_cast<hipStream_t>(stream));      // two workgroups per CU
    if (use_gemm256() && mr_gemm3_eligible(a)) {
        mr_gemm3_launch(a, static_cast<hipStream_t>(stream));
        MR_CHECK_LAUNCH("mr_gemm (ping-pong kernel)");
        return MR_OK;
    }
    if (use_gemm256() && mr_gemm256_eligible(a)) {
        mr_gemm256_launch(a, static_cast<hipStream_t>(stream), launch_splitk_reduce);
        MR_CHECK_LAUNCH("mr_gemm (256-row kernel)");
        return MR_OK;
    }
    const int64_t tm = (a->M + BM - 1) / BM, tn = (a->N + BN - 1) / BN;
    MR_CHECK_ARG(tm * tn < (1LL << 30), "mr_gemm: grid too large");
    // split-K: problems with few output tiles and a long contraction (every wgrad) cannot fill 256 CUs otherwise
    const int64_t nk = (a->K + BK - 1) / BK;
    int64_t splits = 1;
    const bool plain = !a->rot_tab && !a->c2 && a->act == MR_ACT_NONE && !a->residual && !a->aux && a->out_grp == 0;
    if (a->workspace && plain && a->N % 4 == 0 && tm * tn < 384 && nk >= 16) {
        splits = (768 + tm * tn - 1) / (tm * tn);
        if (splits > nk / 8) splits = nk / 8;
        if (splits > 32) splits = 32;
        const int64_t fit = a->workspace_bytes / (a->M * a->N * (int64_t)sizeof(float));
        if (splits > fit) splits = fit;
        if (splits < 2) splits = 1;
    }
    int64_t kt_per_split = (nk + splits - 1) / splits;
    splits = (nk + kt_per_split - 1) / kt_per_split;
    dim3 grid((unsigned)(tm * tn), (unsigned)splits), block(256);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (!a->transA && !a->transB) hipLaunchKernelGGL((gemm_bf16_kernel<false, false>), grid, block, 0, s, *a, (int)tn, (int)kt_per_split);
    else if (!a->transA && a->transB) hipLaunchKernelGGL((gemm_bf16_kernel<false, true>), grid, block, 0, s, *a, (int)tn, (int)kt_per_split);
    else if (a->transA && !a->transB) hipLaunchKernelGGL((gemm_bf16_kernel<true, false>), grid, block, 0, s, *a, (int)tn, (int)kt_per_split);
    else hipLaunchKernelGGL((gemm_bf16_kernel<true, true>), grid, block, 0, s, *a, (int)tn, (int)kt_per_split);
    if (splits > 1) launch_splitk_reduce(a, splits, s);
    mr_note_route("gemm_bf16_kernel<%d,%d>%s", (int)a->transA, (int)a->transB, splits > 1 ? " + splitk_reduce" : "");
    MR_CHECK_LAUNCH("mr_gemm");
    return MR_OK;
}

extern "C" int mr_gemm_grouped(const mr_gemm_args* list, int32_t count, void* stream) {
    MR_CHECK_ARG(list != nullptr && count >= 1, "mr_gemm_grouped: empty list");
    for (int k = 0; k < count; ++k) {          // the same operand checks mr_gemm makes, BEFORE any kernel sees the list (16-byte LDS-DMA loads / b128 stores)
        const mr_gemm_args* a = &list[k];
        MR_CHECK_ARG(a->M > 0 && a->N > 0 && a->K > 0, "mr_gemm_grouped: problem %d is empty (M=%ld N=%ld K=%ld)", k, (long)a->M, (long)a->N, (long)a->K);
        MR_CHECK_ARG(a->A && a->B && a->C, "mr_gemm_grouped: problem %d has a null operand", k);
        MR_CHECK_ARG(a->lda % 8 == 0 && a->ldb % 8 == 0, "mr_gemm_grouped: problem %d: lda / ldb must be multiples of 8", k);
        MR_CHECK_ARG(((uintptr_t)a->A % 16) == 0 && ((uintptr_t)a->B % 16) == 0 && ((uintptr_t)a->C % 16) == 0,
                     "mr_gemm_grouped: problem %d: operands must be 16-byte aligned", k);
        MR_CHECK_ARG(a->c_dtype != MR_DT_BF16 || (a->N % 8 == 0 && a->ldc % 8 == 0), "mr_gemm_grouped: problem %d: bf16 output needs N, ldc multiples of 8", k);
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (use_gemm256() && count <= 20 && mr_gemm3_tn_grouped(list, count, s)) {      // weight gradients: the TN ping-pong kernel
        MR_CHECK_LAUNCH("mr_gemm_grouped (ping-pong kernel)");
        return MR_OK;
    }
    // small problems of one layout (none of which the big kernels would take): ONE launch of the 128 x 128 kernel over all their tiles
    if (count > 1 && count <= SMALL_MAXG) {
        bool small = true;
        int64_t tiles = 0;
        SmallGroup sg;
        for (int k = 0; small && k < count; ++k) {
            const mr_gemm_args* a = &list[k];
            small = a->transA == list[0].transA && a->transB == list[0].transB && !(use_gemm256() && (mr_gemm5_wanted(a) || mr_gemm3_eligible(a) || mr_gemm256_eligible(a)));
            small = small && ((a->transA ? a->M : a->K) + 7) / 8 * 8 <= a->lda && ((a->transB ? a->K : a->N) + 7) / 8 * 8 <= a->ldb;
            small = small && (a->c_dtype == MR_DT_BF16 || (a->c_dtype == MR_DT_F32 && !a->residual && !a->aux && !a->c2 && a->act == MR_ACT_NONE && a->out_grp == 0));
            small = small && !a->colsum && (!a->rot_tab || a->rot_rows > 0);
            small = small && (a->c_dtype != MR_DT_BF16 || ((!a->residual || a->ldr % 8 == 0) && (!a->aux || a->ldaux % 8 == 0)));
            if (!small) break;
            const int64_t tm = (a->M + BM - 1) / BM, tn = (a->N + BN - 1) / BN;
            sg.tile_start[k] = (int)tiles;
            sg.tiles_n[k] = (int)tn;
            sg.p[k] = *a;
            tiles += tm * tn;
        }
        if (small && tiles <= 4096) {
            sg.count = count;
            for (int k = count; k <= SMALL_MAXG; ++k) sg.tile_start[k] = (int)tiles;
            dim3 grid((unsigned)tiles), block(256);
            const bool ta = list[0].transA, tb = list[0].transB;
            if (!ta && !tb) hipLaunchKernelGGL((gemm_bf16_grouped_kernel<false, false>), grid, block, 0, s, sg);
            else if (!ta && tb) hipLaunchKernelGGL((gemm_bf16_grouped_kernel<false, true>), grid, block, 0, s, sg);
            else if (ta && !tb) hipLaunchKernelGGL((gemm_bf16_grouped_kernel<true, false>), grid, block, 0, s, sg);
            else hipLaunchKernelGGL((gemm_bf16_grouped_kernel<true, true>), grid, block, 0, s, sg);
            mr_note_route("gemm_bf16_grouped_kernel<%d,%d> x%d", (int)ta, (int)tb, count);
            MR_CHECK_LAUNCH("mr_gemm_grouped (small problems)");
            return MR_OK;
        }
    }
    if (count > 4) {        // the one-barrier kernel groups <= 4 problems per launch
        for (int k = 0; k < count; k += 4) {
            const int rc = mr_gemm_grouped(list + k, count - k < 4 ? count - k : 4, stream);
            if (rc != MR_OK) return rc;
        }
        return MR_OK;
    }
    bool all_big = use_gemm256() && count > 1 && count <= 4;
    for (int k = 0; all_big && k < count; ++k) all_big = list[k].M > 0 && list[k].N % 128 == 0;     // (256-wide tiles: N % 256, checked inside)
    if (all_big && mr_gemm256_grouped(list, count, s)) {
        MR_CHECK_LAUNCH("mr_gemm_grouped");
        return MR_OK;
    }
    for (int k = 0; k < count; ++k) {          // not groupable: one launch per problem (identical results)
        const int rc = mr_gemm(&list[k], stream);
        if (rc != MR_OK) return rc;
    }
    return MR_OK;
}
