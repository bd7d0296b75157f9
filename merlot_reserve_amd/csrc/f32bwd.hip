// fp32 BACKWARD kernels: the reference's `use_bfloat16 = False` training arithmetic (pretrain/train.py:61-67 sets it for its GPU-debug
// mode; pretrain/pretrain_model.py:323-333 is the `use_bfloat16_grads = False` branch of train_step).  Written for clarity and exactness,
// not speed -- this is the debug / parity mode, the benchmarked program is the bf16 one.  Same layouts and entry-point shapes as the bf16
// kernels they mirror (layernorm.hip, attention.hip, rowops.hip); every reduction has a fixed order (no atomics).
#include "mr_common.h"

namespace {

// ------------------------------------------------------------------------------------------------ LayerNorm backward
// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma; statistics recomputed from x (E[x^2] - E[x]^2, eps inside the
// rsqrt: flax nn.LayerNorm as used at mreserve/modeling.py:272,277,360,366).  One wave per row.
__global__ __launch_bounds__(256) void f32_ln_bwd_kernel(const float* __restrict__ dy, int64_t lddy, const float* __restrict__ x, int64_t ldx,
                                                         const float* __restrict__ gamma, float* dx, int64_t lddx, const float* dx_add,
                                                         int64_t ldadd, int64_t rows, int H, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * ldx;
    const float* dr = dy + row * lddy;
    float s = 0.f, s2 = 0.f;
    for (int c = lane; c < H; c += 64) { const float v = xr[c]; s += v; s2 += v * v; }
    s = wave_sum(s); s2 = wave_sum(s2);
    const float mean = s / (float)H, rstd = rsqrtf(s2 / (float)H - mean * mean + eps);
    float a = 0.f, b = 0.f;
    for (int c = lane; c < H; c += 64) {
        const float gg = dr[c] * gamma[c], xh = (xr[c] - mean) * rstd;
        a += gg; b += gg * xh;
    }
    a = wave_sum(a) / (float)H; b = wave_sum(b) / (float)H;
    for (int c = lane; c < H; c += 64) {
        const float gg = dr[c] * gamma[c], xh = (xr[c] - mean) * rstd;
        float o = rstd * (gg - a - xh * b);
        if (dx_add != nullptr) o += dx_add[row * ldadd + c];
        dx[row * lddx + c] = o;
    }
}

// Column reductions over all rows in a fixed order: block = 64 columns x 4 row groups (rows r, r + 4, ...), LDS over the groups.
// MODE 0: out0[c] = sum_r x[r, c] (colsum).  MODE 1: out0[c] = sum_r dy[r, c] * xhat[r, c] (dgamma), out1[c] = sum_r dy[r, c] (dbeta),
// with xhat recomputed from the row statistics in `stat` ([rows][2]: mean, rstd; written by f32_row_stats_kernel).
__global__ __launch_bounds__(256) void f32_row_stats_kernel(const float* __restrict__ x, int64_t ldx, float* __restrict__ stat, int64_t rows, int H, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float s = 0.f, s2 = 0.f;
    for (int c = lane; c < H; c += 64) { const float v = x[row * ldx + c]; s += v; s2 += v * v; }
    s = wave_sum(s); s2 = wave_sum(s2);
    const float mean = s / (float)H;
    if (lane == 0) { stat[2 * row] = mean; stat[2 * row + 1] = rsqrtf(s2 / (float)H - mean * mean + eps); }
}
template <int MODE>
__global__ __launch_bounds__(256) void f32_colreduce_kernel(const float* __restrict__ a, int64_t lda, const float* __restrict__ x, int64_t ldx,
                                                            const float* __restrict__ stat, int64_t rows, int N, float* __restrict__ out0,
                                                            float* __restrict__ out1) {
    __shared__ float red[2][4][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float s0 = 0.f, s1 = 0.f;
    if (c < N)
        for (int64_t r = grp; r < rows; r += 4) {
            const float v = a[r * lda + c];
            if (MODE == 0) s0 += v;
            else { s0 += v * ((x[r * ldx + c] - stat[2 * r]) * stat[2 * r + 1]); s1 += v; }
        }
    red[0][grp][lane] = s0; red[1][grp][lane] = s1;
    __syncthreads();
    if (grp == 0 && c < N) {
        out0[c] = (red[0][0][lane] + red[0][1][lane]) + (red[0][2][lane] + red[0][3][lane]);
        if (MODE == 1) out1[c] = (red[1][0][lane] + red[1][1][lane]) + (red[1][2][lane] + red[1][3][lane]);
    }
}

// ------------------------------------------------------------------------------------------------ attention backward
// Two kernels like the bf16 pair (no atomics): dQ (+ delta = rowsum(dO * O)) per 64-query block, dK / dV per 64-key block, both on
// v_mfma_f32_16x16x4_f32 in the layouts of f32_attn_fwd_kernel (f32path.hip).  Bias semantics of the reference (additive -1e10,
// mreserve/modeling.py:353-356): a disallowed key of a row with an allowed key weighs exactly 0; a row with NO allowed key (its LSE is
// -1e10: ln S is below fp32 resolution there) is uniform over the S keys, taken as P = 1 / S.
constexpr int TKB = 64, KLD = 66, VLD = 68;
constexpr float PAD_LSE_F = -1e9f;

__device__ __forceinline__ float attn_p(float s, int cq, int ck, bool masked, bool key_ok, float lse_q, bool pad_q, float inv_S) {
    if (!key_ok) return 0.f;
    if (pad_q) return inv_S;
    if (masked && !((cq >= 0) && (cq == ck))) return 0.f;      // exp(s - 1e10 - lse) underflows to exactly 0
    return expf(s - lse_q);
}

__global__ __launch_bounds__(256) void f32_attn_bwd_dq_kernel(const float* __restrict__ qkv, const int32_t* __restrict__ code,
                                                              const float* __restrict__ o, const float* __restrict__ dout,
                                                              const float* __restrict__ lse, float* __restrict__ delta,
                                                              float* __restrict__ dqkv, const float* __restrict__ rot_tab, int64_t rot_rows,
                                                              int S, int nh) {
    __shared__ __attribute__((aligned(16))) float Ks[TKB * KLD];
    __shared__ __attribute__((aligned(16))) float Vs[TKB * KLD];
    __shared__ int kcode[TKB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
    const int seq = blockIdx.z, head = blockIdx.y, H = nh * 64;
    const int64_t rowbase = (int64_t)seq * S;
    const int qi = blockIdx.x * 64 + wave * 16 + li;
    const bool qok = qi < S;
    const int64_t qrow = rowbase + (qok ? qi : 0);
    float qreg[16], doreg[16];                      // Q^T / 8 and dO^T operands: d = 4 s + g
    float dpart = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        qreg[s] = qok ? qkv[qrow * (3 * H) + head * 64 + 4 * s + g] * 0.125f : 0.f;
        doreg[s] = qok ? dout[qrow * H + head * 64 + 4 * s + g] : 0.f;
        dpart += doreg[s] * (qok ? o[qrow * H + head * 64 + 4 * s + g] : 0.f);
    }
    dpart += __shfl_xor(dpart, 16, 64);
    dpart += __shfl_xor(dpart, 32, 64);
    const float del = dpart;
    if (qok && g == 0) delta[((int64_t)seq * nh + head) * S + qi] = del;
    const int cq = (code != nullptr && qok) ? code[rowbase + qi] : 0;
    const float L = qok ? lse[((int64_t)seq * nh + head) * S + qi] : 0.f;
    const bool padq = code != nullptr && qok && L < PAD_LSE_F;
    const float inv_S = 1.0f / (float)S;
    f32x4 dq[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) dq[db] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < S; k0 += TKB) {
        const int nkeys = min(TKB, S - k0);
        __syncthreads();
        for (int c = tid; c < TKB * 64; c += 256) {
            const int kr = c >> 6, d = c & 63;
            float kv = 0.f, vv = 0.f;
            if (kr < nkeys) {
                const float* bp = qkv + (rowbase + k0 + kr) * (3 * H) + head * 64 + d;
                kv = bp[H]; vv = bp[2 * H];
            }
            Ks[kr * KLD + d] = kv; Vs[kr * KLD + d] = vv;
        }
        if (tid < TKB) kcode[tid] = (code != nullptr && tid < nkeys) ? code[rowbase + k0 + tid] : 0;
        __syncthreads();
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            f32x4 st = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
            const float* kp = Ks + (16 * kb + li) * KLD + g;
            const float* vp = Vs + (16 * kb + li) * KLD + g;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                st = __builtin_amdgcn_mfma_f32_16x16x4f32(kp[4 * s], qreg[s], st, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x4f32(vp[4 * s], doreg[s], dp, 0, 0, 0);
            }
            // lane (g, q = li) holds keys 16 kb + 4 g + r
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = 16 * kb + 4 * g + r;
                const float pv = attn_p(st[r], cq, kcode[key], code != nullptr, key < nkeys, L, padq, inv_S);
                const float ds = qok ? pv * (dp[r] - del) : 0.f;
                // dQ^T[d][q] += K^T[d][key] * dS^T[key][q]: the lane's own dS value is the B operand (k slot g <-> key 16 kb + 4 g + r)
                const float* kt = Ks + key * KLD + li;
#pragma unroll
                for (int db = 0; db < 4; ++db) dq[db] = __builtin_amdgcn_mfma_f32_16x16x4f32(kt[16 * db], ds, dq[db], 0, 0, 0);
            }
        }
    }
    if (qok) {          // lane holds dQ^T[d = 16 db + 4 g + r][q]; d(score)/dq = k / 8
        float* op = dqkv + (rowbase + qi) * (3 * H) + head * 64;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            const int d = 16 * db + 4 * g;
            f32x4 x = dq[db] * 0.125f;
            if (rot_tab != nullptr && d < 32) x *= *reinterpret_cast<const f32x4*>(rot_tab + ((rowbase + qi) % rot_rows) * 32 + d);
            *reinterpret_cast<f32x4*>(op + d) = x;
        }
    }
}

__global__ __launch_bounds__(256) void f32_attn_bwd_dkv_kernel(const float* __restrict__ qkv, const int32_t* __restrict__ code,
                                                               const float* __restrict__ dout, const float* __restrict__ lse,
                                                               const float* __restrict__ delta, float* __restrict__ dqkv,
                                                               const float* __restrict__ rot_tab, int64_t rot_rows, int S, int nh) {
    __shared__ __attribute__((aligned(16))) float Qs[TKB * KLD];
    __shared__ __attribute__((aligned(16))) float Ds[TKB * KLD];
    __shared__ float Ls[TKB], Es[TKB];
    __shared__ int qcode[TKB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
    const int seq = blockIdx.z, head = blockIdx.y, H = nh * 64;
    const int64_t rowbase = (int64_t)seq * S;
    const int ki = blockIdx.x * 64 + wave * 16 + li;          // this lane's key (accumulator column)
    const bool kok = ki < S;
    float kreg[16], vreg[16];                        // K^T / 8 and V^T operands: d = 4 s + g, key = li
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const float* bp = qkv + (rowbase + (kok ? ki : 0)) * (3 * H) + head * 64 + 4 * s + g;
        kreg[s] = kok ? bp[H] * 0.125f : 0.f;
        vreg[s] = kok ? bp[2 * H] : 0.f;
    }
    const int ck = (code != nullptr && kok) ? code[rowbase + ki] : 0;
    const float inv_S = 1.0f / (float)S;
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) { dk[db] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[db] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    for (int q0 = 0; q0 < S; q0 += TKB) {
        const int nq = min(TKB, S - q0);
        __syncthreads();
        for (int c = tid; c < TKB * 64; c += 256) {
            const int qr = c >> 6, d = c & 63;
            float qv = 0.f, dv_ = 0.f;
            if (qr < nq) {
                qv = qkv[(rowbase + q0 + qr) * (3 * H) + head * 64 + d];
                dv_ = dout[(rowbase + q0 + qr) * H + head * 64 + d];
            }
            Qs[qr * KLD + d] = qv; Ds[qr * KLD + d] = dv_;
        }
        if (tid < TKB) {
            const bool ok = tid < nq;
            Ls[tid] = ok ? lse[((int64_t)seq * nh + head) * S + q0 + tid] : 0.f;
            Es[tid] = ok ? delta[((int64_t)seq * nh + head) * S + q0 + tid] : 0.f;
            qcode[tid] = (code != nullptr && ok) ? code[rowbase + q0 + tid] : 0;
        }
        __syncthreads();
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) {
            // S[q][key] = Q . K^T (A = Q rows from LDS: q = 16 qb + li, d = 4 s + g; B = K^T registers): lane (g, key = li) holds q = 16 qb + 4 g + r
            f32x4 st = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
            const float* qp = Qs + (16 * qb + li) * KLD + g;
            const float* dpp = Ds + (16 * qb + li) * KLD + g;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                st = __builtin_amdgcn_mfma_f32_16x16x4f32(qp[4 * s], kreg[s], st, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x4f32(dpp[4 * s], vreg[s], dp, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = 16 * qb + 4 * g + r;
                const bool qok = q < nq;
                const float L = Ls[q];
                const bool padq = code != nullptr && qok && L < PAD_LSE_F;
                const float pv = qok ? attn_p(st[r], qcode[q], ck, code != nullptr, kok, L, padq, inv_S) : 0.f;
                const float ds = pv * (dp[r] - Es[q]);
                // dV^T[d][key] += dO^T[d][q] * P[q][key];  dK^T[d][key] += Q^T[d][q] * dS[q][key]  (k slot g <-> query 16 qb + 4 g + r)
                const float* dt = Ds + q * KLD + li;
                const float* qt = Qs + q * KLD + li;
#pragma unroll
                for (int db = 0; db < 4; ++db) {
                    dv[db] = __builtin_amdgcn_mfma_f32_16x16x4f32(dt[16 * db], pv, dv[db], 0, 0, 0);
                    dk[db] = __builtin_amdgcn_mfma_f32_16x16x4f32(qt[16 * db], ds, dk[db], 0, 0, 0);
                }
            }
        }
    }
    if (kok) {          // lane holds dK^T / dV^T [d = 16 db + 4 g + r][key]
        float* op = dqkv + (rowbase + ki) * (3 * H) + head * 64;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            const int d = 16 * db + 4 * g;
            f32x4 x = dk[db] * 0.125f;
            if (rot_tab != nullptr && d < 32) x *= *reinterpret_cast<const f32x4*>(rot_tab + ((rowbase + ki) % rot_rows) * 32 + d);
            *reinterpret_cast<f32x4*>(op + H + d) = x;
            *reinterpret_cast<f32x4*>(op + 2 * H + d) = dv[db];
        }
    }
}

// ------------------------------------------------------------------------------------------------ row kernels
constexpr int FMAXR = 8;
// backward of f32_poolattn_kernel (one query, R keys per group; probabilities recomputed): d_q, and d_k / d_v SCATTERED to the key rows
// (each key row belongs to exactly one group: mreserve/modeling.py:419-427, 467-472 pool disjoint windows), other rows untouched.
__global__ __launch_bounds__(256) void f32_poolattn_bwd_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                               int64_t ldkv, const int32_t* __restrict__ key_rows, const float* __restrict__ dout,
                                                               float* __restrict__ dq, float* __restrict__ dk, float* __restrict__ dv,
                                                               int64_t G, int R, int nh) {
    const int lane = threadIdx.x & 63;
    const int64_t gidx = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gidx >= G) return;
    const int H = nh * 64;
    for (int c = lane * 4; c < H; c += 256) {
        const f32x4 qv = *reinterpret_cast<const f32x4*>(q + gidx * H + c);
        const f32x4 dov = *reinterpret_cast<const f32x4*>(dout + gidx * H + c);
        float sc[FMAXR], dp[FMAXR];
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < FMAXR; ++r) {
            sc[r] = -INFINITY; dp[r] = 0.f;
            if (r < R) {
                const int64_t kr = key_rows[gidx * R + r];
                const f32x4 kv = *reinterpret_cast<const f32x4*>(k + kr * ldkv + c);
                const f32x4 vv = *reinterpret_cast<const f32x4*>(v + kr * ldkv + c);
                float d = (qv[0] * 0.125f) * kv[0] + (qv[1] * 0.125f) * kv[1] + (qv[2] * 0.125f) * kv[2] + (qv[3] * 0.125f) * kv[3];
                float e = dov[0] * vv[0] + dov[1] * vv[1] + dov[2] * vv[2] + dov[3] * vv[3];
#pragma unroll
                for (int o_ = 1; o_ < 16; o_ <<= 1) { d += __shfl_xor(d, o_, 64); e += __shfl_xor(e, o_, 64); }
                sc[r] = d; dp[r] = e;
                mx = fmaxf(mx, d);
            }
        }
        float den = 0.f;
#pragma unroll
        for (int r = 0; r < FMAXR; ++r) { sc[r] = (r < R) ? expf(sc[r] - mx) : 0.f; den += sc[r]; }
        const float inv = 1.0f / den;
        float dsum = 0.f;
#pragma unroll
        for (int r = 0; r < FMAXR; ++r) { sc[r] *= inv; dsum += sc[r] * dp[r]; }
        f32x4 dqv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < FMAXR; ++r)
            if (r < R) {
                const int64_t kr = key_rows[gidx * R + r];
                const float ds = sc[r] * (dp[r] - dsum);
                const f32x4 kv = *reinterpret_cast<const f32x4*>(k + kr * ldkv + c);
                dqv += (ds * 0.125f) * kv;
                *reinterpret_cast<f32x4*>(dk + kr * ldkv + c) = (ds * 0.125f) * qv;
                *reinterpret_cast<f32x4*>(dv + kr * ldkv + c) = sc[r] * dov;
            }
        *reinterpret_cast<f32x4*>(dq + gidx * H + c) = dqv;
    }
}

// dst[rows[g, r]] += src[g] / R  (backward of f32_rows_mean_kernel; each destination row belongs to one group)
__global__ __launch_bounds__(256) void f32_rows_mean_bwd_kernel(const float* __restrict__ dsrc, const int32_t* __restrict__ rows, float* __restrict__ dst,
                                                                int64_t ldd, int64_t G, int R, int H) {
    const int lane = threadIdx.x & 63;
    const int64_t g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= G) return;
    const float inv = 1.0f / (float)R;
    for (int c = lane * 4; c < H; c += 256) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(dsrc + g * H + c) * inv;
        for (int r = 0; r < R; ++r) {
            float* p = dst + (int64_t)rows[g * R + r] * ldd + c;
            *reinterpret_cast<f32x4*>(p) = *reinterpret_cast<const f32x4*>(p) + a;
        }
    }
}

// y = x / sqrt(|x|^2 + 1e-5) * t, t = exp(min(ls, ln 100) / 2):  dx = t * inv * (dy - xn * (xn . dy)) with xn = x * inv;
// d ls = 0.5 * sum(dy * y) where ls < ln 100, one partial per row (summed in order by the caller's reduce).
__global__ __launch_bounds__(256) void f32_unit_norm_bwd_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ log_scale,
                                                                const float* __restrict__ dy, int64_t lddy, float* __restrict__ dx, int64_t lddx,
                                                                int accumulate, float* __restrict__ dls_rows, int64_t rows, int H) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float ls = log_scale[0];
    const float temp = expf(fminf(ls, 4.605170185988092f) * 0.5f);
    float ss = 0.f, xd = 0.f;
    for (int c = lane; c < H; c += 64) { const float v = x[row * ldx + c]; ss += v * v; xd += v * dy[row * lddy + c]; }
    ss = wave_sum(ss); xd = wave_sum(xd);
    const float inv = 1.0f / sqrtf(ss + 1e-5f);
    for (int c = lane; c < H; c += 64) {
        const float xn = x[row * ldx + c] * inv;
        const float o = temp * inv * (dy[row * lddy + c] - xn * (xd * inv));
        dx[row * lddx + c] = accumulate ? dx[row * lddx + c] + o : o;
    }
    if (lane == 0) dls_rows[row] = (ls < 4.605170185988092f) ? 0.5f * temp * inv * xd : 0.f;
}

__global__ void f32_sum_ordered_kernel(const float* __restrict__ src, int64_t n, float* __restrict__ dst, int accumulate) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float a = 0.f;
        for (int64_t i = 0; i < n; ++i) a += src[i];
        dst[0] = accumulate ? dst[0] + a : a;
    }
}

// out[c] = sum over groups of x[(g * grp_stride + off), c]   (the CLS parameter's gradient: rows `off` of every sequence)
__global__ __launch_bounds__(256) void f32_sum_rows_strided_kernel(const float* __restrict__ x, int64_t ldx, int64_t ngroups, int64_t grp_stride,
                                                                   int64_t off, int H, float* __restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= H) return;
    float a = 0.f;
    for (int64_t g = 0; g < ngroups; ++g) a += x[(g * grp_stride + off) * ldx + c];
    out[c] = a;
}

__global__ void f32_axpby_kernel(float* __restrict__ y, const float* __restrict__ x, float a, float b, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) y[i] = a * x[i] + b * y[i];
}
__global__ void f32_nan_to_num_kernel(float* __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = y[i];
        y[i] = (v != v) ? 0.f : fminf(fmaxf(v, -3.4028234663852886e38f), 3.4028234663852886e38f);     // jnp.nan_to_num defaults
    }
}


// ---------------------------------------------------------------------------------------------- attention backward under an ARBITRARY [L, L] mask
// The backward twin of dense_mask_attn_fwd_kernel (f32path.hip): TransformerEncoder takes any boolean attention_mask [*, L, L] and the reference's
// autodiff differentiates through it (mreserve/modeling.py:343-358).  Plain fp32 arithmetic on bf16 or fp32 operands, two kernels, no atomics:
//   q pass : one wave per (sequence, head, query i).  Scores on the lanes exactly as in the forward (the literal -1e10 where the mask byte is 0, so a row
//            with no allowed key is uniform over all L keys and its gradient flows, as in the reference), p = softmax, dp_j = dO_i . v_j,
//            delta = sum_j p_j dp_j, ds_j = p_j (dp_j - delta); P[row, :] and dS[row, :] go to the fp32 workspace; then, one head dim per lane,
//            dQ_i = (1 / 8) sum_j ds_j K_j.
//   kv pass: one wave per (sequence, head, key j), one head dim per lane: dV_j = sum_i P[i, j] dO_i, dK_j = (1 / 8) sum_i dS[i, j] Q_i.
// dq / dk are multiplied by the "rotary" scale table on the way out when one is given (gradient wrt the pre-"rotary" qkv, like mr_attention_bwd).
// An API-completeness path (a correctness kernel, O(L^2) floats of workspace), not a training kernel of the stock towers, whose masks are of the block form.
template <typename T>
__global__ __launch_bounds__(256) void dense_mask_attn_bwd_q_kernel(const T* __restrict__ qkv, const uint8_t* __restrict__ mask, const T* __restrict__ dout,
                                                                    T* __restrict__ dqkv, const float* __restrict__ rot_tab, int64_t rot_rows,
                                                                    float* __restrict__ Pw, float* __restrict__ dSw, int64_t S, int64_t nh, int64_t nrows) {
    extern __shared__ float dmb_smem[];                // per wave: q / 8 [64], dO_i [64], then S probabilities, S values of dp / ds
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;          // (sequence, head, query)
    if (row >= nrows) return;                                     // (wave-uniform; no barrier in this kernel)
    float* qs = dmb_smem + (size_t)wave * (128 + 2 * S);
    float* dos = qs + 64;
    float* sc = dos + 64;
    float* ds = sc + S;
    const int64_t qi = row % S, h = (row / S) % nh, seq = row / (S * nh);
    const int64_t H = nh * 64, ld = 3 * H;
    const T* base = qkv + seq * S * ld;
    qs[lane] = (float)base[qi * ld + h * 64 + lane] * 0.125f;
    dos[lane] = (float)dout[(seq * S + qi) * H + h * 64 + lane];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // the wave's own LDS writes (wave-private region)
    const uint8_t* mrow = mask + (seq * S + qi) * S;
    float mx = -INFINITY;
    for (int64_t j = lane; j < S; j += 64) {
        const T* kr = base + j * ld + H + h * 64;
        float a = 0.f;
#pragma unroll 8
        for (int d = 0; d < 64; ++d) a += qs[d] * (float)kr[d];
        a += mrow[j] ? 0.f : -1e10f;
        sc[j] = a;
        mx = fmaxf(mx, a);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int64_t j = lane; j < S; j += 64) {
        const float pv = __expf(sc[j] - mx);
        sc[j] = pv;
        sum += pv;
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    float delta = 0.f;
    for (int64_t j = lane; j < S; j += 64) {
        const T* vr = base + j * ld + 2 * H + h * 64;
        float a = 0.f;
#pragma unroll 8
        for (int d = 0; d < 64; ++d) a += dos[d] * (float)vr[d];
        const float pj = sc[j] * inv;
        sc[j] = pj;
        ds[j] = a;
        delta += pj * a;
    }
    delta = wave_sum(delta);
    float* prow = Pw + row * S;
    float* drow = dSw + row * S;
    for (int64_t j = lane; j < S; j += 64) {
        const float v = sc[j] * (ds[j] - delta);
        ds[j] = v;
        prow[j] = sc[j];
        drow[j] = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float acc = 0.f;
    const T* kcol = base + H + h * 64 + lane;
    for (int64_t j = 0; j < S; ++j) acc += ds[j] * (float)kcol[j * ld];
    acc *= 0.125f;
    if (rot_tab != nullptr && lane < 32) acc *= rot_tab[((seq * S + qi) % rot_rows) * 32 + lane];
    dqkv[(seq * S + qi) * ld + h * 64 + lane] = (T)acc;
}

template <typename T>
__global__ __launch_bounds__(256) void dense_mask_attn_bwd_kv_kernel(const T* __restrict__ qkv, const T* __restrict__ dout, T* __restrict__ dqkv,
                                                                     const float* __restrict__ rot_tab, int64_t rot_rows, const float* __restrict__ Pw,
                                                                     const float* __restrict__ dSw, int64_t S, int64_t nh, int64_t nrows) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;          // (sequence, head, key)
    if (row >= nrows) return;
    const int64_t kj = row % S, h = (row / S) % nh, seq = row / (S * nh);
    const int64_t H = nh * 64, ld = 3 * H;
    const T* qcol = qkv + seq * S * ld + h * 64 + lane;
    const T* docol = dout + seq * S * H + h * 64 + lane;
    const float* pcol = Pw + ((seq * nh + h) * S) * S + kj;        // P[(seq, h, i), kj]: one value per query, the same for every lane
    const float* dcol = dSw + ((seq * nh + h) * S) * S + kj;
    float dv = 0.f, dk = 0.f;
    for (int64_t i = 0; i < S; ++i) {
        dv += pcol[i * S] * (float)docol[i * H];
        dk += dcol[i * S] * (float)qcol[i * ld];
    }
    dk *= 0.125f;                                                    // s = (q / 8) . k: ds / dk = q / 8 (Q is read as stored, unscaled)
    if (rot_tab != nullptr && lane < 32) dk *= rot_tab[((seq * S + kj) % rot_rows) * 32 + lane];
    T* orow = dqkv + (seq * S + kj) * ld + h * 64 + lane;
    orow[H] = (T)dk;
    orow[2 * H] = (T)dv;
}

}  // namespace

extern "C" int mr_f32_layernorm_bwd(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma, float* dx, int64_t lddx,
                                    const float* dx_add, int64_t ldadd, float* dgamma, float* dbeta, float* stat_ws, int64_t rows, int64_t H,
                                    float eps, void* stream) {
    MR_CHECK_ARG(dy && x && gamma && dx && stat_ws && rows > 0 && H > 0, "mr_f32_layernorm_bwd: bad args");
    MR_CHECK_ARG((dgamma == nullptr) == (dbeta == nullptr), "mr_f32_layernorm_bwd: dgamma and dbeta must both be given or both be NULL");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dgamma != nullptr) {       // parameter gradients first: dx may alias dy
        hipLaunchKernelGGL(f32_row_stats_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, ldx, stat_ws, rows, (int)H, eps);
        hipLaunchKernelGGL((f32_colreduce_kernel<1>), dim3((unsigned)((H + 63) / 64)), dim3(256), 0, s, dy, lddy, x, ldx, stat_ws, rows, (int)H, dgamma, dbeta);
    }
    hipLaunchKernelGGL(f32_ln_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, dy, lddy, x, ldx, gamma, dx, lddx, dx_add, ldadd, rows, (int)H, eps);
    MR_CHECK_LAUNCH("mr_f32_layernorm_bwd");
    return MR_OK;
}

extern "C" int mr_f32_colsum(const float* x, int64_t ldx, int64_t rows, int64_t N, float* out, void* stream) {
    MR_CHECK_ARG(x && out && rows > 0 && N > 0, "mr_f32_colsum: bad args");
    hipLaunchKernelGGL((f32_colreduce_kernel<0>), dim3((unsigned)((N + 63) / 64)), dim3(256), 0, static_cast<hipStream_t>(stream), x, ldx,
                       (const float*)nullptr, (int64_t)0, (const float*)nullptr, rows, (int)N, out, (float*)nullptr);
    MR_CHECK_LAUNCH("mr_f32_colsum");
    return MR_OK;
}

extern "C" int mr_f32_attention_bwd(const float* qkv, const int32_t* code, const float* out, const float* dout, const float* lse,
                                    float* delta, float* dqkv, const float* rot_tab, int64_t rot_rows, int64_t nseq, int64_t S, int64_t nh,
                                    void* stream) {
    MR_CHECK_ARG(qkv && out && dout && lse && delta && dqkv && nseq > 0 && S > 0 && nh > 0, "mr_f32_attention_bwd: bad args");
    MR_CHECK_ARG(nseq <= 65535 && nh <= 65535, "mr_f32_attention_bwd: nseq / nh exceed the grid limits");
    MR_CHECK_ARG(!rot_tab || rot_rows > 0, "mr_f32_attention_bwd: rot_rows must be > 0");
    dim3 grid((unsigned)((S + 63) / 64), (unsigned)nh, (unsigned)nseq);
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(f32_attn_bwd_dq_kernel, grid, dim3(256), 0, s, qkv, code, out, dout, lse, delta, dqkv, rot_tab, rot_rows, (int)S, (int)nh);
    hipLaunchKernelGGL(f32_attn_bwd_dkv_kernel, grid, dim3(256), 0, s, qkv, code, dout, lse, delta, dqkv, rot_tab, rot_rows, (int)S, (int)nh);
    MR_CHECK_LAUNCH("mr_f32_attention_bwd");
    return MR_OK;
}

extern "C" int mr_f32_poolattn_bwd(const float* q, const float* k, const float* v, int64_t ldkv, const int32_t* key_rows, const float* dout,
                                   float* dq, float* dk, float* dv, int64_t G, int64_t R, int64_t nh, void* stream) {
    MR_CHECK_ARG(q && k && v && key_rows && dout && dq && dk && dv && G > 0 && R > 0 && R <= FMAXR && nh > 0 && ldkv % 4 == 0,
                 "mr_f32_poolattn_bwd: bad args (R <= %d)", FMAXR);
    hipLaunchKernelGGL(f32_poolattn_bwd_kernel, dim3((unsigned)((G + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), q, k, v, ldkv,
                       key_rows, dout, dq, dk, dv, G, (int)R, (int)nh);
    MR_CHECK_LAUNCH("mr_f32_poolattn_bwd");
    return MR_OK;
}

extern "C" int mr_f32_rows_mean_bwd(const float* dsrc, const int32_t* rows, float* dst, int64_t ldd, int64_t G, int64_t R, int64_t H, void* stream) {
    MR_CHECK_ARG(dsrc && rows && dst && G > 0 && R > 0 && H % 4 == 0 && ldd % 4 == 0, "mr_f32_rows_mean_bwd: bad args");
    hipLaunchKernelGGL(f32_rows_mean_bwd_kernel, dim3((unsigned)((G + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), dsrc, rows, dst, ldd, G, (int)R, (int)H);
    MR_CHECK_LAUNCH("mr_f32_rows_mean_bwd");
    return MR_OK;
}

extern "C" int mr_f32_unit_norm_scale_bwd(const float* x, int64_t ldx, const float* log_scale, const float* dy, int64_t lddy, float* dx, int64_t lddx,
                                          int32_t accumulate, float* dls, float* dls_rows, int64_t rows, int64_t H, void* stream) {
    MR_CHECK_ARG(x && log_scale && dy && dx && dls && dls_rows && rows > 0 && H > 0, "mr_f32_unit_norm_scale_bwd: bad args");
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(f32_unit_norm_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, ldx, log_scale, dy, lddy, dx, lddx, (int)accumulate,
                       dls_rows, rows, (int)H);
    hipLaunchKernelGGL(f32_sum_ordered_kernel, dim3(1), dim3(64), 0, s, dls_rows, rows, dls, 1);
    MR_CHECK_LAUNCH("mr_f32_unit_norm_scale_bwd");
    return MR_OK;
}

extern "C" int mr_f32_sum_rows_strided(const float* x, int64_t ldx, int64_t ngroups, int64_t grp_stride, int64_t off, int64_t H, float* out, void* stream) {
    MR_CHECK_ARG(x && out && ngroups > 0 && H > 0, "mr_f32_sum_rows_strided: bad args");
    hipLaunchKernelGGL(f32_sum_rows_strided_kernel, dim3((unsigned)((H + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), x, ldx, ngroups,
                       grp_stride, off, (int)H, out);
    MR_CHECK_LAUNCH("mr_f32_sum_rows_strided");
    return MR_OK;
}

/* y = a * x + b * y */
extern "C" int mr_f32_axpby(float* y, const float* x, float a, float b, int64_t n, void* stream) {
    MR_CHECK_ARG(y && x && n > 0, "mr_f32_axpby: bad args");
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(f32_axpby_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), y, x, a, b, n);
    MR_CHECK_LAUNCH("mr_f32_axpby");
    return MR_OK;
}

extern "C" int mr_f32_nan_to_num(float* y, int64_t n, void* stream) {
    MR_CHECK_ARG(y && n > 0, "mr_f32_nan_to_num: bad args");
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(f32_nan_to_num_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), y, n);
    MR_CHECK_LAUNCH("mr_f32_nan_to_num");
    return MR_OK;
}

extern "C" int64_t mr_attention_bwd_dense_mask_workspace(int64_t nseq, int64_t S, int64_t nh) {
    return (nseq > 0 && S > 0 && nh > 0) ? 2 * nseq * nh * S * S * (int64_t)sizeof(float) : 0;
}

extern "C" int mr_attention_bwd_dense_mask(const void* qkv, int32_t dtype, const uint8_t* mask, const void* dout, void* dqkv, const float* rot_tab,
                                           int64_t rot_rows, void* workspace, int64_t nseq, int64_t S, int64_t nh, void* stream) {
    MR_CHECK_ARG(qkv && mask && dout && dqkv && workspace, "mr_attention_bwd_dense_mask: null pointer");
    MR_CHECK_ARG(nseq > 0 && S > 0 && nh > 0 && S <= 1920, "mr_attention_bwd_dense_mask: bad shape (S <= 1920: a query's probabilities and score gradients live in LDS)");
    MR_CHECK_ARG(dtype == MR_DT_BF16 || dtype == MR_DT_F32, "mr_attention_bwd_dense_mask: dtype must be MR_DT_BF16 or MR_DT_F32");
    MR_CHECK_ARG(!rot_tab || rot_rows > 0, "mr_attention_bwd_dense_mask: rot_rows must be > 0");
    const int64_t nrows = nseq * nh * S;
    MR_CHECK_ARG((nrows + 3) / 4 < (1LL << 31), "mr_attention_bwd_dense_mask: too many rows");
    const dim3 grid((unsigned)((nrows + 3) / 4));
    const size_t smem = 4 * (128 + 2 * (size_t)S) * sizeof(float);
    float* Pw = static_cast<float*>(workspace);
    float* dSw = Pw + nrows * S;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == MR_DT_F32) {
        hipLaunchKernelGGL(dense_mask_attn_bwd_q_kernel<float>, grid, dim3(256), smem, s, static_cast<const float*>(qkv), mask, static_cast<const float*>(dout),
                           static_cast<float*>(dqkv), rot_tab, rot_rows, Pw, dSw, S, nh, nrows);
        hipLaunchKernelGGL(dense_mask_attn_bwd_kv_kernel<float>, grid, dim3(256), 0, s, static_cast<const float*>(qkv), static_cast<const float*>(dout),
                           static_cast<float*>(dqkv), rot_tab, rot_rows, Pw, dSw, S, nh, nrows);
    } else {
        hipLaunchKernelGGL(dense_mask_attn_bwd_q_kernel<__bf16>, grid, dim3(256), smem, s, static_cast<const __bf16*>(qkv), mask, static_cast<const __bf16*>(dout),
                           static_cast<__bf16*>(dqkv), rot_tab, rot_rows, Pw, dSw, S, nh, nrows);
        hipLaunchKernelGGL(dense_mask_attn_bwd_kv_kernel<__bf16>, grid, dim3(256), 0, s, static_cast<const __bf16*>(qkv), static_cast<const __bf16*>(dout),
                           static_cast<__bf16*>(dqkv), rot_tab, rot_rows, Pw, dSw, S, nh, nrows);
    }
    MR_CHECK_LAUNCH("mr_attention_bwd_dense_mask");
    return MR_OK;
}
