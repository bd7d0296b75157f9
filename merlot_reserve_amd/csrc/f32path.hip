// fp32 forward path for gfx950: the arithmetic of the reference when `use_bfloat16` is false
// (mreserve/modeling.py:594, 999-1000 -- every zero-shot / feature-extraction caller runs the model in fp32), and
// the mode in which the pretraining forward is checked against the fp32 oracle at 1e-3.
// Forward only.  fp32 storage, fp32 MFMA (v_mfma_f32_16x16x4_f32) for the Dense layers, fp32 VALU attention.
//
//   f32_gemm_kernel      128x128x16 tile, 4 waves in 2x2, each 64x64 = 4x4 MFMA accumulators; operands staged
//                        global -> VGPR -> LDS k-major ([k][own + pad]) so that the MFMA operand reads are conflict-free
//                        ds_read_b32; next k-tile's global loads in flight during the multiply; epilogue
//                        bias / "rotary" scale / gelu / residual in the MFMA layout.
//   f32_attn_fwd_kernel  flash attention on the same fp32 MFMA, both products transposed so that the softmax stays in
//                        registers (see the kernel); exact reference mask semantics (additive -1e10, modeling.py:353-356).
#include "mr_common.h"

namespace {

__device__ __forceinline__ void load8f(const float* p, float (&v)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
}
__device__ __forceinline__ void store8f(float* p, const float (&v)[8]) {
    *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
}

// ---------------------------------------------------------------------------------------------- GEMM
// Square T x T x 16 tiles, T = 128 (waves 64 x 64) or 64 (waves 32 x 32): the small tile is chosen when the large one
// would leave most of the 256 CUs without a workgroup (zero-shot shapes: M = 1312 joint rows, 4616 ViT rows).
constexpr int FBK = 16;
template <int T> struct F32Geo {
    static constexpr int LD = T + 16;          // LDS row stride in floats: (T + 16) mod 32 = 16 -> the two k-rows of a 32-lane read hit disjoint banks
    static constexpr int EPT = T / 16;         // elements per thread and operand tile (256 threads)
    static constexpr int NI = T / 32;          // 16 x 16 MFMA blocks per wave and dimension
};

// [T own][16 k] tile of an operand whose k index is contiguous (A [M,K], or B stored [N,K])
template <int T>
__device__ __forceinline__ void f32_load_kcontig(const float* __restrict__ base, int64_t ld, int64_t own0, int64_t k0,
                                                 int64_t own_n, int64_t K, int tid, bool vec, float (&r)[F32Geo<T>::EPT]) {
    constexpr int E = F32Geo<T>::EPT, TPR = 16 / E;          // threads per row
    const int64_t row = own0 + tid / TPR;
    const int64_t k = k0 + E * (tid % TPR);
#pragma unroll
    for (int e = 0; e < E; ++e) r[e] = 0.f;
    if (row >= own_n) return;
    const float* p = base + row * ld + k;
    if (vec && k + E <= K) {
#pragma unroll
        for (int e = 0; e < E; e += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(p + e);
            r[e] = v[0]; r[e + 1] = v[1]; r[e + 2] = v[2]; r[e + 3] = v[3];
        }
    } else {
#pragma unroll
        for (int e = 0; e < E; ++e)
            if (k + e < K) r[e] = p[e];
    }
}
template <int T>
__device__ __forceinline__ void f32_store_kcontig(float* tile, int tid, const float (&r)[F32Geo<T>::EPT]) {
    constexpr int E = F32Geo<T>::EPT, TPR = 16 / E;
    const int row = tid / TPR, kb = E * (tid % TPR);
#pragma unroll
    for (int e = 0; e < E; ++e) tile[(kb + e) * F32Geo<T>::LD + row] = r[e];
}
// [16 k][T own] tile of an operand whose own index is contiguous (B stored [K,N], or A stored [K,M])
template <int T>
__device__ __forceinline__ void f32_load_kstrided(const float* __restrict__ base, int64_t ld, int64_t own0, int64_t k0,
                                                  int64_t own_n, int64_t K, int tid, bool vec, float (&r)[F32Geo<T>::EPT]) {
    constexpr int E = F32Geo<T>::EPT;
    const int64_t k = k0 + (tid >> 4);
    const int64_t o = own0 + E * (tid & 15);
#pragma unroll
    for (int e = 0; e < E; ++e) r[e] = 0.f;
    if (k >= K) return;
    const float* p = base + k * ld + o;
    if (vec && o + E <= own_n) {
#pragma unroll
        for (int e = 0; e < E; e += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(p + e);
            r[e] = v[0]; r[e + 1] = v[1]; r[e + 2] = v[2]; r[e + 3] = v[3];
        }
    } else {
#pragma unroll
        for (int e = 0; e < E; ++e)
            if (o + e < own_n) r[e] = p[e];
    }
}
template <int T>
__device__ __forceinline__ void f32_store_kstrided(float* tile, int tid, const float (&r)[F32Geo<T>::EPT]) {
    constexpr int E = F32Geo<T>::EPT;
    float* d = tile + (tid >> 4) * F32Geo<T>::LD + E * (tid & 15);
#pragma unroll
    for (int e = 0; e < E; e += 4) *reinterpret_cast<f32x4*>(d + e) = f32x4{r[e], r[e + 1], r[e + 2], r[e + 3]};
}

template <int T, bool TA, bool TB>
__global__ __launch_bounds__(256) void f32_gemm_kernel(const mr_gemm_args p, int tiles_n, int vecA, int vecB) {
    constexpr int LD = F32Geo<T>::LD, E = F32Geo<T>::EPT, NI = F32Geo<T>::NI;
    __shared__ __attribute__((aligned(16))) float smem[2][2][FBK * LD];   // [buffer][A|B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int g = lane >> 4, li = lane & 15;
    const int64_t m0 = (int64_t)(blockIdx.x / tiles_n) * T;
    const int64_t n0 = (int64_t)(blockIdx.x % tiles_n) * T;
    const float* A = static_cast<const float*>(p.A);
    const float* B = static_cast<const float*>(p.B);

    f32x4 acc[NI][NI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    float ra[E], rb[E];
    const int64_t nk = (p.K + FBK - 1) / FBK;
    auto loadA = [&](int64_t kt) {
        if (TA) f32_load_kstrided<T>(A, p.lda, m0, kt * FBK, p.M, p.K, tid, vecA, ra);
        else    f32_load_kcontig<T>(A, p.lda, m0, kt * FBK, p.M, p.K, tid, vecA, ra);
    };
    auto loadB = [&](int64_t kt) {
        if (TB) f32_load_kcontig<T>(B, p.ldb, n0, kt * FBK, p.N, p.K, tid, vecB, rb);
        else    f32_load_kstrided<T>(B, p.ldb, n0, kt * FBK, p.N, p.K, tid, vecB, rb);
    };
    auto storeAB = [&](int buf) {
        if (TA) f32_store_kstrided<T>(smem[buf][0], tid, ra); else f32_store_kcontig<T>(smem[buf][0], tid, ra);
        if (TB) f32_store_kcontig<T>(smem[buf][1], tid, rb); else f32_store_kstrided<T>(smem[buf][1], tid, rb);
    };
    loadA(0); loadB(0);
    storeAB(0);
    __syncthreads();
    for (int64_t kt = 0; kt < nk; ++kt) {
        const int buf = (int)(kt & 1);
        const bool more = kt + 1 < nk;
        if (more) { loadA(kt + 1); loadB(kt + 1); }
        const float* As = smem[buf][0];
        const float* Bs = smem[buf][1];
#pragma unroll
        for (int kk = 0; kk < FBK; kk += 4) {
            float af[NI], bf[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) af[i] = As[(kk + g) * LD + wm * (T / 2) + i * 16 + li];
#pragma unroll
            for (int j = 0; j < NI; ++j) bf[j] = Bs[(kk + g) * LD + wn * (T / 2) + j * 16 + li];
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (more) storeAB(buf ^ 1);     // the other buffer was last read before the previous barrier
        __syncthreads();
    }

    // epilogue in the MFMA layout: col = lane & 15, row = 4 * (lane >> 4) + r
    const float* bias = static_cast<const float*>(p.bias);
    const float* R = static_cast<const float*>(p.residual);
    float* C = static_cast<float*>(p.C);
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int64_t n = n0 + wn * (T / 2) + j * 16 + li;
        if (n >= p.N) continue;
        const float bv = bias != nullptr ? bias[n] : 0.f;
        const bool rot = (p.rot_tab != nullptr) && (n < p.rot_cols) && ((n & 63) < 32);
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t m = m0 + wm * (T / 2) + i * 16 + g * 4 + r;
                if (m >= p.M) continue;
                float v = acc[i][j][r] + bv;
                if (rot) v *= p.rot_tab[(m % p.rot_rows) * 32 + (n & 63)];
                int64_t orow = m;
                if (p.out_grp > 0) orow = (m / p.out_grp) * p.out_grp_stride + p.out_grp_off + m % p.out_grp;
                if (p.act == MR_ACT_GELU1702) {
                    const float sg = 1.0f / (1.0f + expf(-1.702f * v));
                    // training: c2 = gelu'(v), the factor the backward multiplies d(activation) with (the bf16 path stores the same)
                    if (p.c2 != nullptr) static_cast<float*>(p.c2)[orow * p.ldc + n] = sg + 1.702f * v * sg * (1.0f - sg);
                    v = v * sg;
                } else if (p.c2 != nullptr) {
                    static_cast<float*>(p.c2)[orow * p.ldc + n] = v;
                }
                if (R != nullptr) v += R[orow * p.ldr + n];
                if (p.aux != nullptr) v *= static_cast<const float*>(p.aux)[orow * p.ldaux + n];
                C[orow * p.ldc + n] = v;
            }
    }
}

// ---------------------------------------------------------------------------------------------- LayerNorm
// flax nn.LayerNorm (eps 1e-5): mean = E[x], var = E[x^2] - E[x]^2, y = (x - mean) * rsqrt(var + eps) * scale + bias
__global__ __launch_bounds__(256) void f32_ln_fwd_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ y, int64_t ldy,
                                                         int64_t rows, int H, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * ldx;
    float s = 0.f, ss = 0.f;
    for (int c = lane * 4; c < H; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xr + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) { s += v[e]; ss += v[e] * v[e]; }
    }
    s = wave_sum(s);
    ss = wave_sum(ss);
    const float mean = s / (float)H;
    const float var = ss / (float)H - mean * mean;
    const float rstd = rsqrtf(var + eps);
    for (int c = lane * 4; c < H; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xr + c);
        const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c);
        const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + c);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[e] - mean) * (rstd * gm[e]) + bt[e];
        *reinterpret_cast<f32x4*>(y + row * ldy + c) = o;
    }
}

// ---------------------------------------------------------------------------------------------- attention
// fp32 flash attention on v_mfma_f32_16x16x4_f32.  One wave = 16 queries of one (sequence, head); a workgroup = 4 waves = 64
// queries sharing the K / V tiles (64 keys) staged in LDS.  Everything is computed TRANSPOSED so that no layout change is
// needed between the two products:
//   S^T[key][q] = K . Q^T      A = K  (lane: key = 16 kb + (l & 15), d = 4 s + (l >> 4)),  B = Q^T (lane: d = 4 s + g, q = l & 15)
//   accumulator: lane (g, q) holds keys 16 kb + 4 g + r  (r = 0..3)  of its query q
//   O^T[d][q]  += V^T . P^T    one k-step per (kb, r): the lane's own P value is the B operand (k slot g <-> key 16 kb + 4 g + r),
//                              A = V^T (lane: key = 16 kb + 4 g + r, d = 16 db + (l & 15))
// so the softmax runs on registers: per query, over the lane's 16 values and the 4 lane groups (two xor-shuffles).
// Exact reference mask semantics (additive -1e10, modeling.py:353-356); keys beyond S are excluded.
constexpr int FA_TK = 64;          // keys per LDS tile
constexpr int FA_KLD = 66;         // K row stride (floats): (2 key + g) mod 32 distinct over a 32-lane read group
constexpr int FA_VLD = 68;         // V row stride: rows 4 apart land 16 banks apart

__global__ __launch_bounds__(256) void f32_attn_fwd_kernel(const float* __restrict__ qkv, const int32_t* __restrict__ code,
                                                           float* __restrict__ out, float* __restrict__ lse, int S, int nh) {
    __shared__ __attribute__((aligned(16))) float Ks[FA_TK * FA_KLD];
    __shared__ __attribute__((aligned(16))) float Vs[FA_TK * FA_VLD];
    __shared__ int kcode[FA_TK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int seq = blockIdx.z, head = blockIdx.y;
    const int H = nh * 64;
    const int64_t rowbase = (int64_t)seq * S;
    const int qi = blockIdx.x * 64 + wave * 16 + li;          // this lane's query (same for the 4 lane groups)
    const bool qok = qi < S;
    float qreg[16];                                            // Q^T operand: d = 4 s + g, pre-scaled by 1/sqrt(64) (flax: before the product)
    {
        const float* qp = qkv + (rowbase + (qok ? qi : 0)) * (3 * H) + head * 64;
#pragma unroll
        for (int s = 0; s < 16; ++s) qreg[s] = qok ? qp[4 * s + g] * 0.125f : 0.f;
    }
    const int cq = (code != nullptr && qok) ? code[rowbase + qi] : 0;
    f32x4 ot[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) ot[db] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, l = 0.f;
    for (int k0 = 0; k0 < S; k0 += FA_TK) {
        const int nkeys = min(FA_TK, S - k0);
        __syncthreads();
        for (int c = tid; c < FA_TK * 16; c += 256) {          // 16 float4 chunks per key row
            const int kr = c >> 4, ch = c & 15;
            f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
            if (kr < nkeys) {
                const float* base = qkv + (rowbase + k0 + kr) * (3 * H) + head * 64 + 4 * ch;
                kv = *reinterpret_cast<const f32x4*>(base + H);
                vv = *reinterpret_cast<const f32x4*>(base + 2 * H);
            }
            float* kd = Ks + kr * FA_KLD + 4 * ch;             // 264-byte rows: 8-byte aligned
            kd[0] = kv[0]; kd[1] = kv[1]; kd[2] = kv[2]; kd[3] = kv[3];
            *reinterpret_cast<f32x4*>(Vs + kr * FA_VLD + 4 * ch) = vv;
        }
        if (tid < FA_TK) kcode[tid] = (code != nullptr && tid < nkeys) ? code[rowbase + k0 + tid] : 0;
        __syncthreads();
        // S^T = K . Q^T
        f32x4 st[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            const float* kp = Ks + (16 * kb + li) * FA_KLD + g;
#pragma unroll
            for (int s = 0; s < 16; ++s) a = __builtin_amdgcn_mfma_f32_16x16x4f32(kp[4 * s], qreg[s], a, 0, 0, 0);
            st[kb] = a;
        }
        // bias / tail, running max
        float tmax = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = 16 * kb + 4 * g + r;
                float v = st[kb][r];
                if (code != nullptr) v += ((cq >= 0) && (cq == kcode[key])) ? 0.f : -1e10f;
                if (key >= nkeys) v = -INFINITY;
                st[kb][r] = v;
                tmax = fmaxf(tmax, v);
            }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float mn = fmaxf(m, tmax);
        const float alpha = __expf(m - mn);                    // exp(-inf) = 0 on the first tile
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = __expf(st[kb][r] - mn);
                st[kb][r] = pv;
                psum += pv;
            }
        psum += __shfl_xor(psum, 16, 64);
        psum += __shfl_xor(psum, 32, 64);
        l = l * alpha + psum;
        m = mn;
#pragma unroll
        for (int db = 0; db < 4; ++db) ot[db] *= alpha;
        // O^T += V^T . P^T
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float* vp = Vs + (16 * kb + 4 * g + r) * FA_VLD + li;
#pragma unroll
                for (int db = 0; db < 4; ++db) ot[db] = __builtin_amdgcn_mfma_f32_16x16x4f32(vp[16 * db], st[kb][r], ot[db], 0, 0, 0);
            }
    }
    if (qok) {
        const float inv = 1.0f / l;
        float* op = out + (rowbase + qi) * H + head * 64;       // lane holds d = 16 db + 4 g + r
#pragma unroll
        for (int db = 0; db < 4; ++db) *reinterpret_cast<f32x4*>(op + 16 * db + 4 * g) = ot[db] * inv;
        if (lse != nullptr && g == 0) lse[((int64_t)seq * nh + head) * S + qi] = m + __logf(l);
    }
}

// ---------------------------------------------------------------------------------------------- row kernels
struct SegSrcF {
    const float* p0; int64_t ld0, n0;
    const float* p1; int64_t ld1, n1;
    const float* p2; int64_t ld2;
};

__global__ __launch_bounds__(256) void f32_segment_sum_kernel(SegSrcF src, const int32_t* __restrict__ indptr,
                                                              const int32_t* __restrict__ indices, float* __restrict__ dst,
                                                              int64_t ldd, int64_t n_dst, int H, float scale) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_dst) return;
    const int beg = indptr[row], end = indptr[row + 1];
    for (int c = lane * 4; c < H; c += 256) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        for (int j = beg; j < end; ++j) {
            const int64_t cd = indices[j];
            const float* p;
            if (cd < src.n0) p = src.p0 + cd * src.ld0;
            else if (cd < src.n0 + src.n1) p = src.p1 + (cd - src.n0) * src.ld1;
            else p = src.p2 + (cd - src.n0 - src.n1) * src.ld2;
            a += *reinterpret_cast<const f32x4*>(p + c);
        }
        *reinterpret_cast<f32x4*>(dst + row * ldd + c) = a * scale;
    }
}

__global__ __launch_bounds__(256) void f32_rows_mean_kernel(const float* __restrict__ src, int64_t lds, const int32_t* __restrict__ rows,
                                                            float* __restrict__ dst, int64_t G, int R, int H) {
    const int lane = threadIdx.x & 63;
    const int64_t g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= G) return;
    const float inv = 1.0f / (float)R;
    for (int c = lane * 4; c < H; c += 256) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < R; ++r) a += *reinterpret_cast<const f32x4*>(src + (int64_t)rows[g * R + r] * lds + c);
        *reinterpret_cast<f32x4*>(dst + g * H + c) = a * inv;
    }
}

constexpr int FMAXR = 8;
// flax MultiHeadDotProductAttention core with one query and R keys per group (modeling.py:419-427, 467-472); 16 lanes per head
__global__ __launch_bounds__(256) void f32_poolattn_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                           const float* __restrict__ v, int64_t ldkv, const int32_t* __restrict__ key_rows,
                                                           float* __restrict__ out, int64_t G, int R, int nh) {
    const int lane = threadIdx.x & 63;
    const int64_t g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= G) return;
    const int H = nh * 64;
    for (int c = lane * 4; c < H; c += 256) {        // 16 lanes cover one head: whole heads stay inside a pass
        const f32x4 qv = *reinterpret_cast<const f32x4*>(q + g * H + c);
        float sc[FMAXR];
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < FMAXR; ++r) {
            sc[r] = -INFINITY;
            if (r < R) {
                const f32x4 kv = *reinterpret_cast<const f32x4*>(k + (int64_t)key_rows[g * R + r] * ldkv + c);
                float d = (qv[0] * 0.125f) * kv[0] + (qv[1] * 0.125f) * kv[1] + (qv[2] * 0.125f) * kv[2] + (qv[3] * 0.125f) * kv[3];
                d += __shfl_xor(d, 1, 64);
                d += __shfl_xor(d, 2, 64);
                d += __shfl_xor(d, 4, 64);
                d += __shfl_xor(d, 8, 64);
                sc[r] = d;
                mx = fmaxf(mx, d);
            }
        }
        float den = 0.f;
#pragma unroll
        for (int r = 0; r < FMAXR; ++r) { sc[r] = (r < R) ? expf(sc[r] - mx) : 0.f; den += sc[r]; }
        const float inv = 1.0f / den;
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < FMAXR; ++r)
            if (r < R) o += (sc[r] * inv) * *reinterpret_cast<const f32x4*>(v + (int64_t)key_rows[g * R + r] * ldkv + c);
        *reinterpret_cast<f32x4*>(out + g * H + c) = o;
    }
}

// unit_normalize (modeling.py:570-578) times exp(min(log_scale, ln 100) / 2) (pretrain_model.py:246-257); log_scale NULL -> 1
__global__ __launch_bounds__(256) void f32_unit_norm_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ log_scale,
                                                            float* __restrict__ y, int64_t ldy, int64_t rows, int H) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float temp = log_scale != nullptr ? expf(fminf(log_scale[0], 4.605170185988092f) * 0.5f) : 1.0f;
    float ss = 0.f;
    for (int c = lane * 4; c < H; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + row * ldx + c);
        ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    ss = wave_sum(ss);
    const float inv = 1.0f / sqrtf(ss + 1e-5f);
    for (int c = lane * 4; c < H; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + row * ldx + c);
        *reinterpret_cast<f32x4*>(y + row * ldy + c) = (v * inv) * temp;
    }
}

__global__ void f32_fill_rows_kernel(const float* __restrict__ vec, float* __restrict__ dst, int64_t ldd, int64_t ngroups,
                                     int64_t grp_stride, int64_t off, int H) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int nch = H >> 2;
    if (idx >= ngroups * nch) return;
    const int64_t gi = idx / nch;
    const int c = (int)(idx % nch);
    *reinterpret_cast<f32x4*>(dst + (gi * grp_stride + off) * ldd + 4 * c) = *reinterpret_cast<const f32x4*>(vec + 4 * c);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int mr_f32_gemm(const mr_gemm_args* a, void* stream) {
    MR_CHECK_ARG(a && a->A && a->B && a->C, "mr_f32_gemm: null pointer");
    MR_CHECK_ARG(a->M > 0 && a->N > 0 && a->K > 0, "mr_f32_gemm: empty problem %lld x %lld x %lld", (long long)a->M, (long long)a->N, (long long)a->K);
    MR_CHECK_ARG(a->c_dtype == MR_DT_F32, "mr_f32_gemm: c_dtype must be MR_DT_F32");
    MR_CHECK_ARG(a->rot_tab == nullptr || a->rot_rows > 0, "mr_f32_gemm: rot_rows must be positive with a rot_tab");
    const int vecA = aligned16(a->A) && a->lda % 4 == 0, vecB = aligned16(a->B) && a->ldb % 4 == 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t t128 = ((a->M + 127) / 128) * ((a->N + 127) / 128);
    const bool small = t128 < 512;             // fewer than two 128-tiles per CU: 64 x 64 tiles (4x the workgroups)
    const int T = small ? 64 : 128;
    const int tiles_m = (int)((a->M + T - 1) / T), tiles_n = (int)((a->N + T - 1) / T);
    dim3 grid((unsigned)(tiles_m * tiles_n));
#define MR_F32_LAUNCH(TT)                                                                                                              \
    do {                                                                                                                               \
        if (!a->transA && !a->transB) hipLaunchKernelGGL((f32_gemm_kernel<TT, false, false>), grid, dim3(256), 0, st, *a, tiles_n, vecA, vecB); \
        else if (!a->transA && a->transB) hipLaunchKernelGGL((f32_gemm_kernel<TT, false, true>), grid, dim3(256), 0, st, *a, tiles_n, vecA, vecB); \
        else if (a->transA && !a->transB) hipLaunchKernelGGL((f32_gemm_kernel<TT, true, false>), grid, dim3(256), 0, st, *a, tiles_n, vecA, vecB); \
        else hipLaunchKernelGGL((f32_gemm_kernel<TT, true, true>), grid, dim3(256), 0, st, *a, tiles_n, vecA, vecB);                   \
    } while (0)
    if (small) MR_F32_LAUNCH(64); else MR_F32_LAUNCH(128);
#undef MR_F32_LAUNCH
    MR_CHECK_LAUNCH("mr_f32_gemm");
    return MR_OK;
}

extern "C" int mr_f32_layernorm_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, float* y, int64_t ldy,
                                    int64_t rows, int64_t H, float eps, void* stream) {
    MR_CHECK_ARG(x && gamma && beta && y && rows > 0 && H > 0 && H % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0,
                 "mr_f32_layernorm_fwd: H and leading dims must be multiples of 4");
    hipLaunchKernelGGL(f32_ln_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), x, ldx,
                       gamma, beta, y, ldy, rows, (int)H, eps);
    MR_CHECK_LAUNCH("mr_f32_layernorm_fwd");
    return MR_OK;
}

// ---------------------------------------------------------------------------------------------- attention forward under an ARBITRARY [L, L] mask
// TransformerEncoder accepts any boolean attention_mask [*, L, L] (mreserve/modeling.py:303, 350-356: bias = 0 where allowed, -1e10 elsewhere).  Every mask the
// model itself builds is of the block form the other attention kernels take as one code per position; anything else comes here: a plain fp32 kernel, one wave
// per (sequence, head, query) -- scores of its keys on the lanes (q broadcast from LDS, K rows read per lane), the reference's literal -1e10 added where the
// mask byte is 0 (so a row without an allowed key is uniform over all L keys, as in the reference), softmax by wave reductions, then the output with one
// head dim per lane.  An API-completeness path (forward only, the zero-shot / feature-extraction surface), not a training kernel.
template <typename T>
__global__ __launch_bounds__(256) void dense_mask_attn_fwd_kernel(const T* __restrict__ qkv, const uint8_t* __restrict__ mask, T* __restrict__ out,
                                                                  int64_t S, int64_t nh, int64_t nrows) {
    extern __shared__ float dm_smem[];                 // per wave: 64 floats of q, then S scores
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;          // (sequence, head, query)
    if (row >= nrows) return;                                     // (wave-uniform; no barrier in this kernel)
    float* qs = dm_smem + (size_t)wave * (64 + S);
    float* sc = qs + 64;
    const int64_t qi = row % S, h = (row / S) % nh, seq = row / (S * nh);
    const int64_t H = nh * 64, ld = 3 * H;
    const T* base = qkv + seq * S * ld;
    qs[lane] = (float)base[qi * ld + h * 64 + lane] * 0.125f;
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
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float acc = 0.f;
    const T* vcol = base + 2 * H + h * 64 + lane;
    for (int64_t j = 0; j < S; ++j) acc += sc[j] * (float)vcol[j * ld];
    out[(seq * S + qi) * H + h * 64 + lane] = (T)(acc / sum);
}

extern "C" int mr_attention_fwd_dense_mask(const void* qkv, int32_t dtype, const uint8_t* mask, void* out, int64_t nseq, int64_t S, int64_t nh,
                                           void* stream) {
    MR_CHECK_ARG(qkv && mask && out, "mr_attention_fwd_dense_mask: null pointer");
    MR_CHECK_ARG(nseq > 0 && S > 0 && nh > 0 && S <= 3968, "mr_attention_fwd_dense_mask: bad shape (S <= 3968: the scores of a query live in LDS)");
    MR_CHECK_ARG(dtype == MR_DT_BF16 || dtype == MR_DT_F32, "mr_attention_fwd_dense_mask: dtype must be MR_DT_BF16 or MR_DT_F32");
    const int64_t nrows = nseq * nh * S;
    MR_CHECK_ARG((nrows + 3) / 4 < (1LL << 31), "mr_attention_fwd_dense_mask: too many rows");
    const dim3 grid((unsigned)((nrows + 3) / 4));
    const size_t smem = 4 * (64 + (size_t)S) * sizeof(float);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == MR_DT_F32)
        hipLaunchKernelGGL(dense_mask_attn_fwd_kernel<float>, grid, dim3(256), smem, s, static_cast<const float*>(qkv), mask, static_cast<float*>(out), S, nh, nrows);
    else
        hipLaunchKernelGGL(dense_mask_attn_fwd_kernel<__bf16>, grid, dim3(256), smem, s, static_cast<const __bf16*>(qkv), mask, static_cast<__bf16*>(out), S, nh, nrows);
    MR_CHECK_LAUNCH("mr_attention_fwd_dense_mask");
    return MR_OK;
}

extern "C" int mr_f32_attention_fwd(const float* qkv, const int32_t* code, float* out, float* lse, int64_t nseq, int64_t S,
                                    int64_t nh, void* stream) {
    MR_CHECK_ARG(qkv && out && nseq > 0 && S > 0 && nh > 0, "mr_f32_attention_fwd: bad args");
    MR_CHECK_ARG(nseq <= 65535 && nh <= 65535, "mr_f32_attention_fwd: nseq / nh exceed the grid limits");
    dim3 grid((unsigned)((S + 63) / 64), (unsigned)nh, (unsigned)nseq);
    hipLaunchKernelGGL(f32_attn_fwd_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), qkv, code, out, lse, (int)S, (int)nh);
    MR_CHECK_LAUNCH("mr_f32_attention_fwd");
    return MR_OK;
}

extern "C" int mr_f32_poolattn_fwd(const float* q, const float* k, const float* v, int64_t ldkv, const int32_t* key_rows, float* out,
                                   int64_t G, int64_t R, int64_t nh, void* stream) {
    MR_CHECK_ARG(q && k && v && key_rows && out && G > 0 && R > 0 && R <= FMAXR && nh > 0 && ldkv % 4 == 0,
                 "mr_f32_poolattn_fwd: bad args (R <= %d)", FMAXR);
    hipLaunchKernelGGL(f32_poolattn_kernel, dim3((unsigned)((G + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), q, k, v,
                       ldkv, key_rows, out, G, (int)R, (int)nh);
    MR_CHECK_LAUNCH("mr_f32_poolattn_fwd");
    return MR_OK;
}

extern "C" int mr_f32_segment_sum(const float* src0, int64_t ld0, int64_t n0, const float* src1, int64_t ld1, int64_t n1,
                                  const float* src2, int64_t ld2, int64_t n2, const int32_t* indptr, const int32_t* indices,
                                  float* dst, int64_t ldd, int64_t n_dst, int64_t H, float scale, void* stream) {
    MR_CHECK_ARG(src0 && indptr && indices && dst && (n1 == 0 || src1) && (n2 == 0 || src2), "mr_f32_segment_sum: null pointer");
    MR_CHECK_ARG(n_dst > 0 && H > 0 && H % 4 == 0 && ldd % 4 == 0 && ld0 % 4 == 0 && ld1 % 4 == 0 && ld2 % 4 == 0,
                 "mr_f32_segment_sum: H and leading dims must be multiples of 4");
    SegSrcF s{src0, ld0, n0, src1, ld1, n1, src2, ld2};
    hipLaunchKernelGGL(f32_segment_sum_kernel, dim3((unsigned)((n_dst + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), s,
                       indptr, indices, dst, ldd, n_dst, (int)H, scale);
    MR_CHECK_LAUNCH("mr_f32_segment_sum");
    return MR_OK;
}

extern "C" int mr_f32_rows_mean_fwd(const float* src, int64_t lds, const int32_t* rows, float* dst, int64_t G, int64_t R, int64_t H,
                                    void* stream) {
    MR_CHECK_ARG(src && rows && dst && G > 0 && R > 0 && H % 4 == 0 && lds % 4 == 0, "mr_f32_rows_mean_fwd: bad args");
    hipLaunchKernelGGL(f32_rows_mean_kernel, dim3((unsigned)((G + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), src, lds,
                       rows, dst, G, (int)R, (int)H);
    MR_CHECK_LAUNCH("mr_f32_rows_mean_fwd");
    return MR_OK;
}

extern "C" int mr_f32_unit_norm_scale_fwd(const float* x, int64_t ldx, const float* log_scale, float* y, int64_t ldy, int64_t rows,
                                          int64_t H, void* stream) {
    MR_CHECK_ARG(x && y && rows > 0 && H % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0, "mr_f32_unit_norm_scale_fwd: bad args");
    hipLaunchKernelGGL(f32_unit_norm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), x, ldx,
                       log_scale, y, ldy, rows, (int)H);
    MR_CHECK_LAUNCH("mr_f32_unit_norm_scale_fwd");
    return MR_OK;
}

extern "C" int mr_f32_fill_rows(const float* vec, float* dst, int64_t ldd, int64_t ngroups, int64_t grp_stride, int64_t off,
                                int64_t H, void* stream) {
    MR_CHECK_ARG(vec && dst && ngroups > 0 && H % 4 == 0 && ldd % 4 == 0, "mr_f32_fill_rows: bad args");
    const int64_t n = ngroups * (H / 4);
    hipLaunchKernelGGL(f32_fill_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), vec,
                       dst, ldd, ngroups, grp_stride, off, (int)H);
    MR_CHECK_LAUNCH("mr_f32_fill_rows");
    return MR_OK;
}
