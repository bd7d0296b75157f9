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
#include "mr_common.h"

namespace {

constexpr int TQ = 64;        // queries (or keys) per block: 4 waves x 16
constexpr int TK = 64;        // keys (or queries) per inner tile
constexpr int LDR = 72;       // row-read-only tile stride (elements): 144 B
constexpr int LDV = 80;       // tiles that are tr-read: 160 B rows (8 rows x 32 B tile the 64 banks)
constexpr float NEG_BIG = -1e10f;

typedef __attribute__((ext_vector_type(8))) short s16x8;

__device__ __forceinline__ bf16x8 tr_frag(const __bf16* tile, int ld, int row0, int col0, int lane) {
    // rows row0 + 4g + {0..3} and row0 + 16 + 4g + {0..3}; 16 columns from col0; lane receives column (lane & 15)
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const __bf16* a0 = tile + (row0 + 4 * g + q) * ld + col0 + 4 * p;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(MR_LDS_PTR(s16x4, a0));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(MR_LDS_PTR(s16x4, a0 + 16 * ld));
    s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, both);
}

__device__ __forceinline__ bf16x8 row_frag(const __bf16* tile, int ld, int row0, int dd, int lane) {
    const int g = lane >> 4, i = lane & 15;
    return *reinterpret_cast<const bf16x8*>(tile + (row0 + i) * ld + dd * 32 + g * 8);
}

__device__ __forceinline__ bf16x8 pack_acc_pair(const f32x4& a, const f32x4& b) {
    bf16x8 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) { r[e] = (__bf16)a[e]; r[4 + e] = (__bf16)b[e]; }
    return r;
}

// cooperative load of a [64 rows][64 cols] bf16 tile (rows row0.., zero beyond nrows) from a matrix with leading dim ld
__device__ __forceinline__ void load_tile64(const __bf16* __restrict__ src, int64_t ld, int64_t row0, int64_t nrows,
                                            __bf16* tile, int tld, int tid) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int c = tid + 256 * it, r = c >> 3, ch = c & 7;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (row0 + r < nrows) v = *reinterpret_cast<const u32x4*>(src + (row0 + r) * ld + 8 * ch);
        *reinterpret_cast<u32x4*>(tile + r * tld + 8 * ch) = v;
    }
}

// score for (query-side code cq, key index kidx): bias semantics of modeling.py:353-356
__device__ __forceinline__ float biased(float raw, bool exists, bool has_code, int cq, int ck) {
    if (!exists) return -INFINITY;
    float s = raw * 0.125f;
    if (has_code && !(cq == ck && cq >= 0)) s += NEG_BIG;
    return s;
}

// ------------------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const __bf16* __restrict__ qkv, const int32_t* __restrict__ code,
                                                          __bf16* __restrict__ out, float* __restrict__ lse,
                                                          int64_t S, int64_t nh) {
    __shared__ __attribute__((aligned(16))) __bf16 Ks[TK * LDR];
    __shared__ __attribute__((aligned(16))) __bf16 Vs[TK * LDV];
    __shared__ int32_t Cs[TK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, i = lane & 15;
    const int64_t seq = blockIdx.z, h = blockIdx.y, q0 = (int64_t)blockIdx.x * TQ;
    const int64_t H = nh * 64, ld = 3 * H;
    const __bf16* base = qkv + seq * S * ld;
    const int64_t qi = q0 + wave * 16 + i;
    const bool has_code = code != nullptr;

    bf16x8 qf[2];
#pragma unroll
    for (int dd = 0; dd < 2; ++dd) {
        u32x4 v = {0u, 0u, 0u, 0u};
        if (qi < S) v = *reinterpret_cast<const u32x4*>(base + qi * ld + h * 64 + dd * 32 + g * 8);
        qf[dd] = __builtin_bit_cast(bf16x8, v);
    }
    const int cq = (has_code && qi < S) ? code[seq * S + qi] : 0;

    float m = -INFINITY, l = 0.f;
    f32x4 ot[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) ot[db] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int64_t k0 = 0; k0 < S; k0 += TK) {
        __syncthreads();
        load_tile64(base + H + h * 64, ld, k0, S, Ks, LDR, tid);
        load_tile64(base + 2 * H + h * 64, ld, k0, S, Vs, LDV, tid);
        if (has_code && tid < TK) Cs[tid] = (k0 + tid < S) ? code[seq * S + k0 + tid] : -1;
        __syncthreads();

        f32x4 st[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            st[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dd = 0; dd < 2; ++dd)
                st[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Ks, LDR, kb * 16, dd, lane), qf[dd], st[kb], 0, 0, 0);
        }
        float tmax = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kl = kb * 16 + g * 4 + r;
                const float s = biased(st[kb][r], k0 + kl < S, has_code, cq, has_code ? Cs[kl] : 0);
                st[kb][r] = s;
                tmax = fmaxf(tmax, s);
            }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float mn = fmaxf(m, tmax);
        const float alpha = __expf(m - mn);      // m = -inf on the first tile -> 0
        m = mn;
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = __expf(st[kb][r] - mn);
                st[kb][r] = pv;
                psum += pv;
            }
        l = l * alpha + psum;
#pragma unroll
        for (int db = 0; db < 4; ++db) ot[db] *= alpha;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const bf16x8 pf = pack_acc_pair(st[2 * t], st[2 * t + 1]);
#pragma unroll
            for (int db = 0; db < 4; ++db)
                ot[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(Vs, LDV, 32 * t, 16 * db, lane), pf, ot[db], 0, 0, 0);
        }
    }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    if (g == 0 && qi < S) lse[(seq * nh + h) * S + qi] = m + __logf(l);

    __syncthreads();
    __bf16* Os = Ks;  // [64 queries][LDR]
#pragma unroll
    for (int db = 0; db < 4; ++db) {
        bf16x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (__bf16)(ot[db][r] * inv);
        *reinterpret_cast<bf16x4*>(Os + (wave * 16 + i) * LDR + db * 16 + g * 4) = v;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int c = tid + 256 * it, r = c >> 3, ch = c & 7;
        if (q0 + r < S)
            *reinterpret_cast<u32x4*>(out + (seq * S + q0 + r) * H + h * 64 + 8 * ch) =
                *reinterpret_cast<const u32x4*>(Os + r * LDR + 8 * ch);
    }
}

// ------------------------------------------------------------------------------------------------ delta = rowsum(dO * O)
__global__ void attn_delta_kernel(const __bf16* __restrict__ o, const __bf16* __restrict__ dout, float* __restrict__ delta,
                                  int64_t rows, int64_t S, int64_t nh) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int64_t H = nh * 64, seq = row / S, s = row % S;
    for (int64_t c = lane; c < H / 8; c += 64) {
        float a[8], b[8];
        unpack8(*reinterpret_cast<const u32x4*>(o + row * H + 8 * c), a);
        unpack8(*reinterpret_cast<const u32x4*>(dout + row * H + 8 * c), b);
        float acc = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc += a[e] * b[e];
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        acc += __shfl_xor(acc, 4, 64);
        if ((lane & 7) == 0) delta[(seq * nh + c / 8) * S + s] = acc;
    }
}

// ------------------------------------------------------------------------------------------------ dQ
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const __bf16* __restrict__ qkv, const int32_t* __restrict__ code,
                                                             const __bf16* __restrict__ dout, const float* __restrict__ lse,
                                                             const float* __restrict__ delta, __bf16* __restrict__ dqkv,
                                                             const float* __restrict__ rot_tab, int64_t rot_rows,
                                                             int64_t S, int64_t nh) {
    __shared__ __attribute__((aligned(16))) __bf16 Ks[TK * LDV];   // row reads (S^T) and tr reads (dQ^T)
    __shared__ __attribute__((aligned(16))) __bf16 Vs[TK * LDR];   // row reads (dP^T)
    __shared__ int32_t Cs[TK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, i = lane & 15;
    const int64_t seq = blockIdx.z, h = blockIdx.y, q0 = (int64_t)blockIdx.x * TQ;
    const int64_t H = nh * 64, ld = 3 * H;
    const __bf16* base = qkv + seq * S * ld;
    const int64_t qi = q0 + wave * 16 + i;
    const bool has_code = code != nullptr;

    bf16x8 qf[2], dof[2];
#pragma unroll
    for (int dd = 0; dd < 2; ++dd) {
        u32x4 v = {0u, 0u, 0u, 0u}, w = {0u, 0u, 0u, 0u};
        if (qi < S) {
            v = *reinterpret_cast<const u32x4*>(base + qi * ld + h * 64 + dd * 32 + g * 8);
            w = *reinterpret_cast<const u32x4*>(dout + (seq * S + qi) * H + h * 64 + dd * 32 + g * 8);
        }
        qf[dd] = __builtin_bit_cast(bf16x8, v);
        dof[dd] = __builtin_bit_cast(bf16x8, w);
    }
    const int cq = (has_code && qi < S) ? code[seq * S + qi] : 0;
    const float lse_q = (qi < S) ? lse[(seq * nh + h) * S + qi] : 0.f;
    const float del_q = (qi < S) ? delta[(seq * nh + h) * S + qi] : 0.f;

    f32x4 dq[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) dq[db] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int64_t k0 = 0; k0 < S; k0 += TK) {
        __syncthreads();
        load_tile64(base + H + h * 64, ld, k0, S, Ks, LDV, tid);
        load_tile64(base + 2 * H + h * 64, ld, k0, S, Vs, LDR, tid);
        if (has_code && tid < TK) Cs[tid] = (k0 + tid < S) ? code[seq * S + k0 + tid] : -1;
        __syncthreads();
        f32x4 ds[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            f32x4 st = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dd = 0; dd < 2; ++dd) {
                st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Ks, LDV, kb * 16, dd, lane), qf[dd], st, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Vs, LDR, kb * 16, dd, lane), dof[dd], dp, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kl = kb * 16 + g * 4 + r;
                const float s = biased(st[r], k0 + kl < S, has_code, cq, has_code ? Cs[kl] : 0);
                const float pv = __expf(s - lse_q);
                ds[kb][r] = pv * (dp[r] - del_q);
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const bf16x8 dsf = pack_acc_pair(ds[2 * t], ds[2 * t + 1]);
#pragma unroll
            for (int db = 0; db < 4; ++db)
                dq[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(Ks, LDV, 32 * t, 16 * db, lane), dsf, dq[db], 0, 0, 0);
        }
    }
    __syncthreads();
    __bf16* Os = Ks;   // [64][LDV]
#pragma unroll
    for (int db = 0; db < 4; ++db) {
        bf16x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int d = db * 16 + g * 4 + r;
            float x = dq[db][r] * 0.125f;
            if (rot_tab != nullptr && d < 32 && qi < S) x *= rot_tab[((seq * S + qi) % rot_rows) * 32 + d];
            v[r] = (__bf16)x;
        }
        *reinterpret_cast<bf16x4*>(Os + (wave * 16 + i) * LDV + db * 16 + g * 4) = v;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int c = tid + 256 * it, r = c >> 3, ch = c & 7;
        if (q0 + r < S)
            *reinterpret_cast<u32x4*>(dqkv + (seq * S + q0 + r) * ld + h * 64 + 8 * ch) =
                *reinterpret_cast<const u32x4*>(Os + r * LDV + 8 * ch);
    }
}

// ------------------------------------------------------------------------------------------------ dK, dV
// Block owns 64 keys (wave: 16).  Here scores are NOT transposed (S = Q . K^T: column = key on the lane, 4 queries per
// 16-query block in the registers), so P and dS are the B operands of dV^T = dO^T . P and dK^T = Q^T . dS.
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(const __bf16* __restrict__ qkv, const int32_t* __restrict__ code,
                                                              const __bf16* __restrict__ dout, const float* __restrict__ lse,
                                                              const float* __restrict__ delta, __bf16* __restrict__ dqkv,
                                                              const float* __restrict__ rot_tab, int64_t rot_rows,
                                                              int64_t S, int64_t nh) {
    __shared__ __attribute__((aligned(16))) __bf16 Qs[TK * LDV];    // row reads (S) and tr reads (dK^T)
    __shared__ __attribute__((aligned(16))) __bf16 Ds[TK * LDV];    // dO: row reads (dP) and tr reads (dV^T)
    __shared__ float Ls[TK], Dl[TK];
    __shared__ int32_t Cs[TK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, i = lane & 15;
    const int64_t seq = blockIdx.z, h = blockIdx.y, kbase = (int64_t)blockIdx.x * TQ;
    const int64_t H = nh * 64, ld = 3 * H;
    const __bf16* base = qkv + seq * S * ld;
    const int64_t ki = kbase + wave * 16 + i;
    const bool has_code = code != nullptr;

    bf16x8 kf[2], vf[2];
#pragma unroll
    for (int dd = 0; dd < 2; ++dd) {
        u32x4 v = {0u, 0u, 0u, 0u}, w = {0u, 0u, 0u, 0u};
        if (ki < S) {
            v = *reinterpret_cast<const u32x4*>(base + ki * ld + H + h * 64 + dd * 32 + g * 8);
            w = *reinterpret_cast<const u32x4*>(base + ki * ld + 2 * H + h * 64 + dd * 32 + g * 8);
        }
        kf[dd] = __builtin_bit_cast(bf16x8, v);
        vf[dd] = __builtin_bit_cast(bf16x8, w);
    }
    const int ck = (has_code && ki < S) ? code[seq * S + ki] : -1;

    f32x4 dk[4], dv[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) { dk[db] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[db] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    for (int64_t q0 = 0; q0 < S; q0 += TK) {
        __syncthreads();
        load_tile64(base + h * 64, ld, q0, S, Qs, LDV, tid);
        load_tile64(dout + seq * S * H + h * 64, H, q0, S, Ds, LDV, tid);
        if (tid < TK) {
            const bool ok = q0 + tid < S;
            Ls[tid] = ok ? lse[(seq * nh + h) * S + q0 + tid] : 0.f;
            Dl[tid] = ok ? delta[(seq * nh + h) * S + q0 + tid] : 0.f;
            Cs[tid] = (ok && has_code) ? code[seq * S + q0 + tid] : -1;
        }
        __syncthreads();
        f32x4 pp[4], ds[4];
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) {
            f32x4 st = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dd = 0; dd < 2; ++dd) {
                st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Qs, LDV, qb * 16, dd, lane), kf[dd], st, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(Ds, LDV, qb * 16, dd, lane), vf[dd], dp, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ql = qb * 16 + g * 4 + r;
                const bool qok = q0 + ql < S;
                const float s = biased(st[r], ki < S, has_code, has_code ? Cs[ql] : 0, ck);
                const float pv = qok ? __expf(s - Ls[ql]) : 0.f;
                pp[qb][r] = pv;
                ds[qb][r] = pv * (dp[r] - Dl[ql]);
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const bf16x8 pf = pack_acc_pair(pp[2 * t], pp[2 * t + 1]);
            const bf16x8 dsf = pack_acc_pair(ds[2 * t], ds[2 * t + 1]);
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                dv[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(Ds, LDV, 32 * t, 16 * db, lane), pf, dv[db], 0, 0, 0);
                dk[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(Qs, LDV, 32 * t, 16 * db, lane), dsf, dk[db], 0, 0, 0);
            }
        }
    }
    // store dK (scaled, "rotary"-scaled) and dV through LDS as 16-byte row segments
    for (int which = 0; which < 2; ++which) {
        __syncthreads();
        __bf16* Os = Qs;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            bf16x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int d = db * 16 + g * 4 + r;
                float x = (which == 0) ? dk[db][r] * 0.125f : dv[db][r];
                if (which == 0 && rot_tab != nullptr && d < 32 && ki < S) x *= rot_tab[((seq * S + ki) % rot_rows) * 32 + d];
                v[r] = (__bf16)x;
            }
            *reinterpret_cast<bf16x4*>(Os + (wave * 16 + i) * LDV + db * 16 + g * 4) = v;
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int c = tid + 256 * it, r = c >> 3, ch = c & 7;
            if (kbase + r < S)
                *reinterpret_cast<u32x4*>(dqkv + (seq * S + kbase + r) * ld + (which + 1) * H + h * 64 + 8 * ch) =
                    *reinterpret_cast<const u32x4*>(Os + r * LDV + 8 * ch);
        }
    }
}

}  // namespace

extern "C" int mr_attention_fwd(const void* qkv, const int32_t* code, void* out, float* lse, int64_t nseq, int64_t S,
                                int64_t nh, void* stream) {
    MR_CHECK_ARG(qkv && out && lse, "mr_attention_fwd: null pointer");
    MR_CHECK_ARG(nseq > 0 && S > 0 && nh > 0, "mr_attention_fwd: bad shape nseq=%ld S=%ld nh=%ld", (long)nseq, (long)S, (long)nh);
    MR_CHECK_ARG(nseq <= 65535 && nh <= 65535, "mr_attention_fwd: nseq / nh exceed grid limits");
    dim3 grid((unsigned)((S + TQ - 1) / TQ), (unsigned)nh, (unsigned)nseq);
    hipLaunchKernelGGL(attn_fwd_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const __bf16*>(qkv), code, static_cast<__bf16*>(out), lse, S, nh);
    MR_CHECK_LAUNCH("mr_attention_fwd");
    return MR_OK;
}

extern "C" int mr_attention_bwd(const void* qkv, const int32_t* code, const void* out, const void* dout, const float* lse,
                                float* delta, void* dqkv, const float* rot_tab, int64_t rot_rows, int64_t nseq, int64_t S,
                                int64_t nh, void* stream) {
    MR_CHECK_ARG(qkv && out && dout && lse && delta && dqkv, "mr_attention_bwd: null pointer");
    MR_CHECK_ARG(nseq > 0 && S > 0 && nh > 0, "mr_attention_bwd: bad shape");
    MR_CHECK_ARG(nseq <= 65535 && nh <= 65535, "mr_attention_bwd: nseq / nh exceed grid limits");
    MR_CHECK_ARG(!rot_tab || rot_rows > 0, "mr_attention_bwd: rot_rows must be > 0");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t rows = nseq * S;
    hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s,
                       static_cast<const __bf16*>(out), static_cast<const __bf16*>(dout), delta, rows, S, nh);
    dim3 grid((unsigned)((S + TQ - 1) / TQ), (unsigned)nh, (unsigned)nseq);
    hipLaunchKernelGGL(attn_bwd_dq_kernel, grid, dim3(256), 0, s, static_cast<const __bf16*>(qkv), code,
                       static_cast<const __bf16*>(dout), lse, delta, static_cast<__bf16*>(dqkv), rot_tab, rot_rows, S, nh);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel, grid, dim3(256), 0, s, static_cast<const __bf16*>(qkv), code,
                       static_cast<const __bf16*>(dout), lse, delta, static_cast<__bf16*>(dqkv), rot_tab, rot_rows, S, nh);
    MR_CHECK_LAUNCH("mr_attention_bwd");
    return MR_OK;
}
