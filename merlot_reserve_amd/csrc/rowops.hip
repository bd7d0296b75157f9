// Row-granular HBM-bound kernels for gfx950: gathers / segment sums, attention pooling core, unit-normalise,
// contrastive LSE, small elementwise helpers.  One wave per row, 16-byte vectors, wavefront reductions.
#include "mr_common.h"

namespace {

// ---------------------------------------------------------------------------------------------- segment sum / gather
// Used for nn.Embed (modeling.py:527), the audio-span substitution (:685-695), vision tiling (pretrain_model.py:104),
// one_hot_pool (modeling.py:541-567), the gathers of pretrain_model.py:183-190,233-236, and every transpose of those
// in backward (a scatter-add becomes a segment sum over an inverted index built by the host planner: fixed
// summation order, no atomics, bitwise reproducible).
struct SegSrc {
    const __bf16* p0; int64_t ld0, n0;
    const __bf16* p1; int64_t ld1, n1;
    const __bf16* p2; int64_t ld2;
};

template <bool F32OUT>
__global__ __launch_bounds__(256) void segment_sum_kernel(SegSrc src, const int32_t* __restrict__ indptr,
                                                          const int32_t* __restrict__ indices, void* __restrict__ dst,
                                                          int64_t ldd, int64_t n_dst, int H, float scale, int accumulate) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_dst) return;
    const int beg = indptr[row], end = indptr[row + 1];
    const int nch = H >> 3;
    for (int c = lane; c < nch; c += 64) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int j = beg; j < end; ++j) {
            int64_t code = indices[j];
            const __bf16* p;
            if (code < src.n0) p = src.p0 + code * src.ld0;
            else if (code < src.n0 + src.n1) p = src.p1 + (code - src.n0) * src.ld1;
            else p = src.p2 + (code - src.n0 - src.n1) * src.ld2;
            float v[8];
            unpack8(*reinterpret_cast<const u32x4*>(p + 8 * c), v);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += v[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] *= scale;
        if (F32OUT) {
            float* d = static_cast<float*>(dst) + row * ldd + 8 * c;
#pragma unroll
            for (int e = 0; e < 8; ++e) d[e] = accumulate ? d[e] + acc[e] : acc[e];
        } else {
            __bf16* d = static_cast<__bf16*>(dst) + row * ldd + 8 * c;
            if (accumulate) {
                float old[8];
                unpack8(*reinterpret_cast<const u32x4*>(d), old);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += old[e];
            }
            *reinterpret_cast<u32x4*>(d) = pack8(acc);
        }
    }
}

// ---------------------------------------------------------------------------------------------- mean over R rows
__global__ __launch_bounds__(256) void rows_mean_fwd_kernel(const __bf16* __restrict__ src, int64_t lds,
                                                            const int32_t* __restrict__ rows, __bf16* __restrict__ dst,
                                                            int64_t G, int R, int H) {
    const int lane = threadIdx.x & 63;
    const int64_t g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= G) return;
    const float inv = 1.0f / (float)R;
    for (int c = lane; c < (H >> 3); c += 64) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < R; ++r) {
            float v[8];
            unpack8(*reinterpret_cast<const u32x4*>(src + (int64_t)rows[g * R + r] * lds + 8 * c), v);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += v[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] *= inv;
        *reinterpret_cast<u32x4*>(dst + g * H + 8 * c) = pack8(acc);
    }
}

__global__ __launch_bounds__(256) void rows_mean_bwd_kernel(const __bf16* __restrict__ ddst, const int32_t* __restrict__ rows,
                                                            __bf16* __restrict__ dsrc, int64_t lds, int64_t G, int R, int H) {
    const int lane = threadIdx.x & 63;
    const int64_t g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= G) return;
    const float inv = 1.0f / (float)R;
    for (int c = lane; c < (H >> 3); c += 64) {
        float d[8];
        unpack8(*reinterpret_cast<const u32x4*>(ddst + g * H + 8 * c), d);
        for (int r = 0; r < R; ++r) {
            __bf16* p = dsrc + (int64_t)rows[g * R + r] * lds + 8 * c;
            float o[8];
            unpack8(*reinterpret_cast<const u32x4*>(p), o);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] += d[e] * inv;
            *reinterpret_cast<u32x4*>(p) = pack8(o);
        }
    }
}

// ---------------------------------------------------------------------------------------------- attention pooling core
// flax MultiHeadDotProductAttention with one query and R keys per group (modeling.py:419-427, 467-472).
constexpr int MAXR = 8;

__global__ __launch_bounds__(256) void poolattn_fwd_kernel(const __bf16* __restrict__ q, const __bf16* __restrict__ k,
                                                           const __bf16* __restrict__ v, int64_t ldkv,
                                                           const int32_t* __restrict__ key_rows, __bf16* __restrict__ out,
                                                           float* __restrict__ probs, int64_t G, int R, int nh) {
    const int lane = threadIdx.x & 63;
    const int64_t g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= G) return;
    const int H = nh * 64;
    int64_t kr[MAXR];
#pragma unroll
    for (int r = 0; r < MAXR; ++r) kr[r] = (r < R) ? key_rows[g * R + r] : 0;
    for (int c = lane; c < (H >> 3); c += 64) {       // 8 lanes (chunks) per head: whole heads stay inside a pass
        float qv[8];
        unpack8(*reinterpret_cast<const u32x4*>(q + g * H + 8 * c), qv);
        float sc[MAXR];
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < MAXR; ++r) {
            sc[r] = -INFINITY;
            if (r < R) {
                float kv[8];
                unpack8(*reinterpret_cast<const u32x4*>(k + kr[r] * ldkv + 8 * c), kv);
                float d = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) d += (qv[e] * 0.125f) * kv[e];
                d += __shfl_xor(d, 1, 64);
                d += __shfl_xor(d, 2, 64);
                d += __shfl_xor(d, 4, 64);
                sc[r] = d;
                mx = fmaxf(mx, d);
            }
        }
        float den = 0.f;
#pragma unroll
        for (int r = 0; r < MAXR; ++r) { sc[r] = (r < R) ? __expf(sc[r] - mx) : 0.f; den += sc[r]; }
        const float inv = 1.0f / den;
        float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < MAXR; ++r) {
            if (r < R) {
                const float p = sc[r] * inv;
                float vv[8];
                unpack8(*reinterpret_cast<const u32x4*>(v + kr[r] * ldkv + 8 * c), vv);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] += p * vv[e];
                if ((lane & 7) == 0) probs[(g * nh + (c >> 3)) * R + r] = p;
            }
        }
        *reinterpret_cast<u32x4*>(out + g * H + 8 * c) = pack8(o);
    }
}

__global__ __launch_bounds__(256) void poolattn_bwd_kernel(const __bf16* __restrict__ q, const __bf16* __restrict__ k,
                                                           const __bf16* __restrict__ v, int64_t ldkv,
                                                           const int32_t* __restrict__ key_rows, const float* __restrict__ probs,
                                                           const __bf16* __restrict__ dout, __bf16* __restrict__ dq,
                                                           __bf16* __restrict__ dk, __bf16* __restrict__ dv, int64_t G, int R,
                                                           int nh) {
    const int lane = threadIdx.x & 63;
    const int64_t g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= G) return;
    const int H = nh * 64;
    int64_t kr[MAXR];
#pragma unroll
    for (int r = 0; r < MAXR; ++r) kr[r] = (r < R) ? key_rows[g * R + r] : 0;
    for (int c = lane; c < (H >> 3); c += 64) {
        float qv[8], dov[8];
        unpack8(*reinterpret_cast<const u32x4*>(q + g * H + 8 * c), qv);
        unpack8(*reinterpret_cast<const u32x4*>(dout + g * H + 8 * c), dov);
        float p[MAXR], dp[MAXR];
        float dot = 0.f;
#pragma unroll
        for (int r = 0; r < MAXR; ++r) {
            p[r] = 0.f; dp[r] = 0.f;
            if (r < R) {
                p[r] = probs[(g * nh + (c >> 3)) * R + r];
                float vv[8];
                unpack8(*reinterpret_cast<const u32x4*>(v + kr[r] * ldkv + 8 * c), vv);
                float d = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) d += dov[e] * vv[e];
                d += __shfl_xor(d, 1, 64);
                d += __shfl_xor(d, 2, 64);
                d += __shfl_xor(d, 4, 64);
                dp[r] = d;
                dot += p[r] * d;
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = p[r] * dov[e];
                *reinterpret_cast<u32x4*>(dv + kr[r] * ldkv + 8 * c) = pack8(o);
            }
        }
        float dqv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < MAXR; ++r) {
            if (r < R) {
                const float ds = p[r] * (dp[r] - dot) * 0.125f;
                float kv[8], o[8];
                unpack8(*reinterpret_cast<const u32x4*>(k + kr[r] * ldkv + 8 * c), kv);
#pragma unroll
                for (int e = 0; e < 8; ++e) { dqv[e] += ds * kv[e]; o[e] = ds * qv[e]; }
                *reinterpret_cast<u32x4*>(dk + kr[r] * ldkv + 8 * c) = pack8(o);
            }
        }
        *reinterpret_cast<u32x4*>(dq + g * H + 8 * c) = pack8(dqv);
    }
}

// ---------------------------------------------------------------------------------------------- small helpers
__global__ void pad_cols_kernel(const __bf16* __restrict__ src, int64_t cols_in, __bf16* __restrict__ dst, int64_t cols_out,
                                int64_t rows) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * cols_out) return;
    const int64_t r = idx / cols_out, c = idx % cols_out;
    dst[idx] = (c < cols_in) ? src[r * cols_in + c] : (__bf16)0.f;
}

__global__ void fill_rows_kernel(const __bf16* __restrict__ vec, __bf16* __restrict__ dst, int64_t ldd, int64_t ngroups,
                                 int64_t grp_stride, int64_t off, int H) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int nch = H >> 3;
    if (idx >= ngroups * nch) return;
    const int64_t gi = idx / nch;
    const int c = (int)(idx % nch);
    *reinterpret_cast<u32x4*>(dst + (gi * grp_stride + off) * ldd + 8 * c) = *reinterpret_cast<const u32x4*>(vec + 8 * c);
}

// block = 64 columns x 16 row groups: a group's loads are independent of the other groups' (one thread per column walking all the
// rows was a chain of ngroups load latencies: 47 us for 192 rows), partial sums combined in a fixed order
__global__ __launch_bounds__(1024) void sum_rows_strided_kernel(const __bf16* __restrict__ src, int64_t lds, int64_t ngroups,
                                                                int64_t grp_stride, int64_t off, int H, __bf16* __restrict__ out) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (col < H)
        for (int64_t gi = grp; gi < ngroups; gi += 16) s += (float)src[(gi * grp_stride + off) * lds + col];
    red[grp][lane] = s;
    __syncthreads();
    if (grp == 0 && col < H) {
        s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += red[k][lane];
        out[col] = (__bf16)s;
    }
}

__global__ void add_bf16_kernel(const __bf16* __restrict__ a, const __bf16* __restrict__ b, __bf16* __restrict__ y, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        float x[8], z[8];
        unpack8(*reinterpret_cast<const u32x4*>(a + 8 * i), x);
        unpack8(*reinterpret_cast<const u32x4*>(b + 8 * i), z);
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] += z[e];
        *reinterpret_cast<u32x4*>(y + 8 * i) = pack8(x);
    }
}

__global__ void cast_f32_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = (__bf16)src[i];
}

// hi = bf16(x), lo = bf16(x - hi): a 16-bit-mantissa split, so that a bf16 MFMA GEMM run on hi and on lo reproduces an
// fp32-operand product.  Used for dL/dlogits, whose rows sum to zero: rounding them to 8 bits breaks the cancellation.
__global__ void split_hilo_kernel(const float* __restrict__ src, __bf16* __restrict__ hi, __bf16* __restrict__ lo, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float x = src[i];
        const __bf16 h = (__bf16)x;
        hi[i] = h;
        lo[i] = (__bf16)(x - (float)h);
    }
}

// the same split for a [rows, cols] fp32 matrix (row stride lds) into two bf16 matrices with row stride ldo
__global__ void split_hilo_rows_kernel(const float* __restrict__ src, int64_t lds, __bf16* __restrict__ hi, __bf16* __restrict__ lo,
                                       int64_t ldo, int64_t rows, int64_t cols) {
    const int64_t n = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols, c = i - r * cols;
        const float x = src[r * lds + c];
        const __bf16 h = (__bf16)x;
        hi[r * ldo + c] = h;
        lo[r * ldo + c] = (__bf16)(x - (float)h);
    }
}

// ---------------------------------------------------------------------------------------------- unit-normalise * temperature
constexpr float LN100 = 4.605170185988092f;

__global__ __launch_bounds__(256) void unit_norm_fwd_kernel(const __bf16* __restrict__ x, int64_t ldx,
                                                            const __bf16* __restrict__ log_scale, __bf16* __restrict__ y,
                                                            int64_t ldy, float* __restrict__ inv_norm, int64_t rows, int H) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float temp = __expf(fminf((float)log_scale[0], LN100) * 0.5f);
    float ss = 0.f;
    for (int c = lane; c < (H >> 3); c += 64) {
        float v[8];
        unpack8(*reinterpret_cast<const u32x4*>(x + row * ldx + 8 * c), v);
#pragma unroll
        for (int e = 0; e < 8; ++e) ss += v[e] * v[e];
    }
    ss = wave_sum(ss);
    const float inv = 1.0f / sqrtf(ss + 1e-5f);
    if (lane == 0) inv_norm[row] = inv;
    for (int c = lane; c < (H >> 3); c += 64) {
        float v[8];
        unpack8(*reinterpret_cast<const u32x4*>(x + row * ldx + 8 * c), v);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)(__bf16)(v[e] * inv) * temp;   // unit_normalize casts to bf16 first
        *reinterpret_cast<u32x4*>(y + row * ldy + 8 * c) = pack8(v);
    }
}

__global__ __launch_bounds__(256) void unit_norm_bwd_kernel(const __bf16* __restrict__ x, int64_t ldx,
                                                            const __bf16* __restrict__ log_scale,
                                                            const float* __restrict__ inv_norm, const __bf16* __restrict__ dy,
                                                            int64_t lddy, __bf16* __restrict__ dx, int64_t lddx,
                                                            float* __restrict__ dls_partials, int64_t rows, int H) {
    __shared__ float red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    const float ls = (float)log_scale[0];
    const float temp = __expf(fminf(ls, LN100) * 0.5f);
    float ndn = 0.f;
    float inv = 0.f;
    if (row < rows) {
        inv = inv_norm[row];
        for (int c = lane; c < (H >> 3); c += 64) {
            float v[8], d[8];
            unpack8(*reinterpret_cast<const u32x4*>(x + row * ldx + 8 * c), v);
            unpack8(*reinterpret_cast<const u32x4*>(dy + row * lddy + 8 * c), d);
#pragma unroll
            for (int e = 0; e < 8; ++e) ndn += (v[e] * inv) * d[e];
        }
        ndn = wave_sum(ndn);      // = sum_h n * dy
        for (int c = lane; c < (H >> 3); c += 64) {
            float v[8], d[8], o[8];
            unpack8(*reinterpret_cast<const u32x4*>(x + row * ldx + 8 * c), v);
            unpack8(*reinterpret_cast<const u32x4*>(dy + row * lddy + 8 * c), d);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = inv * temp * (d[e] - (v[e] * inv) * ndn);
            *reinterpret_cast<u32x4*>(dx + row * lddx + 8 * c) = pack8(o);
        }
    }
    if (lane == 0) red[wave] = (row < rows) ? ndn : 0.f;
    __syncthreads();
    // d/dls [n * exp(ls/2)] = n * temp / 2  ->  dls = (temp/2) * sum n*dy; the clip (P:239) passes gradient only below ln 100.
    // One partial per block, summed in a fixed order by mr_reduce_f32_ordered: bitwise reproducible (no float atomics).
    if (threadIdx.x == 0) dls_partials[blockIdx.x] = (ls < LN100) ? 0.5f * temp * (red[0] + red[1] + red[2] + red[3]) : 0.f;
}

// dst[0] (+)= sum of src[0..n) in a FIXED order: one block, thread t sums elements t, t + 256, ... then a fixed tree.
__global__ __launch_bounds__(256) void reduce_f32_ordered_kernel(const float* __restrict__ src, int64_t n, float scale,
                                                                 float* __restrict__ dst, int accumulate) {
    __shared__ float red[256];
    float a = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 256) a += src[i];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) dst[0] = (accumulate ? dst[0] : 0.f) + scale * red[0];
}

// ---------------------------------------------------------------------------------------------- contrastive LSE
// pretrain_model.py:276-300 for one direction of one objective: row_out[l] = lse_l - logits[l, own_off + l],
// logits overwritten by dL/dlogits = coef * (softmax - onehot).  The loss (and the per-source diagnostics) are summed from
// row_out in a fixed order by contrastive_reduce_kernel: bitwise reproducible (no float atomics).
__global__ __launch_bounds__(256) void contrastive_lse_kernel(float* __restrict__ logits, int64_t ldl, int64_t L, int64_t V,
                                                              int64_t own_off, float coef, float* __restrict__ row_out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= L) return;
    float* x = logits + row * ldl;
    float mx = -INFINITY;
    for (int64_t c = lane; c < V; c += 64) mx = fmaxf(mx, x[c]);
    mx = wave_max(mx);
    float den = 0.f;
    for (int64_t c = lane; c < V; c += 64) den += __expf(x[c] - mx);
    den = wave_sum(den);
    const float lse = mx + __logf(den);
    const float numer = x[own_off + row];
    const float inv = 1.0f / den;
    for (int64_t c = lane; c < V; c += 64) {
        float g = __expf(x[c] - mx) * inv;
        if (c == own_off + row) g -= 1.0f;
        x[c] = coef * g;
    }
    if (lane == 0) row_out[row] = lse - numer;
}

// The mask-LM branch of loss_fn_given_preds (pretrain_model.py:265-274; no forward of the reference emits 'text_preds', the branch is kept for the
// API's completeness): one wave per row.  scratch[r] = -log_softmax(logits[r])[label_r] for a row whose label is not 0, else 0; scratch[n + r] = 1 / 0
// (the mask); scratch[2 n + r] = the row's log-sum-exp.
__global__ __launch_bounds__(256) void masked_lm_rows_kernel(const float* __restrict__ logits, int64_t ldl, int64_t n, int64_t V,
                                                             const int32_t* __restrict__ labels, float* __restrict__ scratch) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const float* x = logits + row * ldl;
    float mx = -INFINITY;
    for (int64_t c = lane; c < V; c += 64) mx = fmaxf(mx, x[c]);
    mx = wave_max(mx);
    float den = 0.f;
    for (int64_t c = lane; c < V; c += 64) den += expf(x[c] - mx);
    den = wave_sum(den);
    const float lse = mx + logf(den);
    const int lab = labels[row];
    if (lane == 0) {
        // P:268-272: nll = -(log_softmax * one_hot(label)).sum(), mask = label != 0.  A label outside [0, V) (an "ignore index" such as -100) has an
        // all-zero one_hot row in jax.nn.one_hot: its nll is 0, it still counts in the mask -- and nothing outside the logits row is read.
        const bool on = lab != 0, in_range = lab >= 0 && (int64_t)lab < V;
        scratch[row] = (on && in_range) ? lse - x[lab] : 0.f;
        scratch[n + row] = on ? 1.f : 0.f;
        scratch[2 * n + row] = lse;
    }
}

// out[0] = sum(nll) / sum(mask), out[1] = sum(mask): fixed-order tree, bitwise reproducible
__global__ __launch_bounds__(256) void masked_lm_reduce_kernel(const float* __restrict__ scratch, int64_t n, float* __restrict__ out) {
    __shared__ float red[2][256];
    float a = 0.f, b = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 256) { a += scratch[i]; b += scratch[n + i]; }
    red[0][threadIdx.x] = a;
    red[1][threadIdx.x] = b;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = red[0][0] / red[1][0]; out[1] = red[1][0]; }
}

// dL/dlogits[r, c] = mask_r (softmax(logits[r])[c] - [c == label_r]) / sum(mask)
__global__ __launch_bounds__(256) void masked_lm_grad_kernel(const float* __restrict__ logits, int64_t ldl, int64_t n, int64_t V,
                                                             const int32_t* __restrict__ labels, const float* __restrict__ scratch,
                                                             const float* __restrict__ out, float* __restrict__ dlogits) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const float* x = logits + row * ldl;
    float* d = dlogits + row * ldl;
    const int lab = labels[row];
    // (a label outside [0, V) contributes a constant 0 to the loss: zero gradient.  With EVERY row masked out[1] is 0 and both the loss and this
    // gradient are 0 / 0 = NaN, as the reference's `.sum() / mask.sum()` is.)
    const bool in_range = lab >= 0 && (int64_t)lab < V;
    const float w = in_range ? scratch[n + row] / out[1] : 0.f, lse = scratch[2 * n + row];
    for (int64_t c = lane; c < V; c += 64) d[c] = in_range ? w * (expf(x[c] - lse) - (c == lab ? 1.0f : 0.0f)) : 0.f;
}

// loss_out[0] += coef * sum_l row[l];  diag[s] = sum over rows with source s of row[l], diag[3 + s] = their count (P:296-300)
__global__ __launch_bounds__(256) void contrastive_reduce_kernel(const float* __restrict__ row, int64_t L, float coef,
                                                                 const int32_t* __restrict__ src, float* __restrict__ loss_out,
                                                                 float* __restrict__ diag) {
    __shared__ float red[7][256];
    float a[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int64_t i = threadIdx.x; i < L; i += 256) {
        const float v = row[i];
        a[0] += v;
        if (src != nullptr) {
            const int s = src[i];
            if (s >= 0 && s < 3) { a[1 + s] += v; a[4 + s] += 1.0f; }
        }
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) red[k][threadIdx.x] = a[k];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
#pragma unroll
            for (int k = 0; k < 7; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        loss_out[0] += coef * red[0][0];
        if (src != nullptr && diag != nullptr) {
#pragma unroll
            for (int k = 0; k < 6; ++k) diag[k] = red[1 + k][0];
        }
    }
}

// ---------------------------------------------------------------------------------------------- softmax cross-entropy
// finetune/vcr/qa_qar_joint_finetune.py:188-195: loss = -mean_r log_softmax(logits[r])[label[r]], is_right = mean(argmax == label);
// one lane per row (C <= 64 classes); dlogits = coef * (softmax - onehot) written as bf16 at the same strides.  ONE workgroup walks the
// rows (thread t: rows t, t + 256, ...) and the two sums leave it through a fixed-order LDS tree: no float atomics, bitwise reproducible.
__global__ __launch_bounds__(256) void softmax_xent_kernel(const float* __restrict__ logits, int64_t row_stride, int64_t class_stride,
                                    const int32_t* __restrict__ labels, int64_t rows, int C, float coef,
                                    float* __restrict__ loss_out, float* __restrict__ correct_out, __bf16* __restrict__ dlogits) {
    __shared__ float red[2][256];
    float lsum = 0.f, csum = 0.f;
    for (int64_t r = threadIdx.x; r < rows; r += 256) {
        const float* x = logits + r * row_stride;
        float mx = -INFINITY;
        int arg = 0;
        for (int c = 0; c < C; ++c) {
            const float v = x[c * class_stride];
            if (v > mx) { mx = v; arg = c; }          // first maximum, like argmax
        }
        float den = 0.f;
        for (int c = 0; c < C; ++c) den += expf(x[c * class_stride] - mx);
        const int lab = labels[r];
        const float logp = x[lab * class_stride] - mx - logf(den);
        lsum += -coef * logp;
        csum += coef * (arg == lab ? 1.0f : 0.0f);
        if (dlogits != nullptr) {
            const float inv = 1.0f / den;
            for (int c = 0; c < C; ++c)
                dlogits[r * row_stride + c * class_stride] = (__bf16)(coef * (expf(x[c * class_stride] - mx) * inv - (c == lab ? 1.0f : 0.0f)));
        }
    }
    red[0][threadIdx.x] = lsum;
    red[1][threadIdx.x] = csum;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            red[0][threadIdx.x] += red[0][threadIdx.x + o];
            red[1][threadIdx.x] += red[1][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        loss_out[0] += red[0][0];
        if (correct_out != nullptr) correct_out[0] += red[1][0];
    }
}

}  // namespace

extern "C" int mr_softmax_xent(const float* logits, int64_t row_stride, int64_t class_stride, const int32_t* labels, int64_t rows,
                               int64_t C, float coef, float* loss_out, float* correct_out, void* dlogits_bf16, void* stream) {
    MR_CHECK_ARG(logits && labels && loss_out && rows > 0 && C > 0 && C <= 64, "mr_softmax_xent: bad args (C <= 64)");
    hipLaunchKernelGGL(softmax_xent_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), logits,
                       row_stride, class_stride, labels, rows, (int)C, coef, loss_out, correct_out, static_cast<__bf16*>(dlogits_bf16));
    MR_CHECK_LAUNCH("mr_softmax_xent");
    return MR_OK;
}

extern "C" int mr_segment_sum(const void* src0, int64_t ld0, int64_t n0, const void* src1, int64_t ld1, int64_t n1,
                              const void* src2, int64_t ld2, int64_t n2, const int32_t* indptr, const int32_t* indices,
                              void* dst, int64_t ldd, int32_t dst_dtype, int64_t n_dst, int64_t H, float scale,
                              int32_t accumulate, void* stream) {
    MR_CHECK_ARG(src0 && indptr && indices && dst, "mr_segment_sum: null pointer");
    MR_CHECK_ARG((n1 == 0 || src1) && (n2 == 0 || src2), "mr_segment_sum: missing source table");
    MR_CHECK_ARG(n_dst > 0 && H > 0 && H % 8 == 0 && ldd % 8 == 0 && ld0 % 8 == 0 && ld1 % 8 == 0 && ld2 % 8 == 0,
                 "mr_segment_sum: H and leading dims must be multiples of 8");
    SegSrc s{static_cast<const __bf16*>(src0), ld0, n0, static_cast<const __bf16*>(src1), ld1, n1,
             static_cast<const __bf16*>(src2), ld2};
    dim3 grid((unsigned)((n_dst + 3) / 4));
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dst_dtype == MR_DT_F32)
        hipLaunchKernelGGL(segment_sum_kernel<true>, grid, dim3(256), 0, st, s, indptr, indices, dst, ldd, n_dst, (int)H, scale, (int)accumulate);
    else
        hipLaunchKernelGGL(segment_sum_kernel<false>, grid, dim3(256), 0, st, s, indptr, indices, dst, ldd, n_dst, (int)H, scale, (int)accumulate);
    MR_CHECK_LAUNCH("mr_segment_sum");
    return MR_OK;
}

extern "C" int mr_rows_mean_fwd(const void* src, int64_t lds, const int32_t* rows, void* dst, int64_t G, int64_t R, int64_t H,
                                void* stream) {
    MR_CHECK_ARG(src && rows && dst && G > 0 && R > 0 && H % 8 == 0 && lds % 8 == 0, "mr_rows_mean_fwd: bad args");
    hipLaunchKernelGGL(rows_mean_fwd_kernel, dim3((unsigned)((G + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const __bf16*>(src), lds, rows, static_cast<__bf16*>(dst), G, (int)R, (int)H);
    MR_CHECK_LAUNCH("mr_rows_mean_fwd");
    return MR_OK;
}

extern "C" int mr_rows_mean_bwd(const void* ddst, const int32_t* rows, void* dsrc, int64_t lds, int64_t G, int64_t R, int64_t H,
                                void* stream) {
    MR_CHECK_ARG(ddst && rows && dsrc && G > 0 && R > 0 && H % 8 == 0 && lds % 8 == 0, "mr_rows_mean_bwd: bad args");
    hipLaunchKernelGGL(rows_mean_bwd_kernel, dim3((unsigned)((G + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const __bf16*>(ddst), rows, static_cast<__bf16*>(dsrc), lds, G, (int)R, (int)H);
    MR_CHECK_LAUNCH("mr_rows_mean_bwd");
    return MR_OK;
}

extern "C" int mr_poolattn_fwd(const void* q, const void* k, const void* v, int64_t ldkv, const int32_t* key_rows, void* out,
                               float* probs, int64_t G, int64_t R, int64_t nh, void* stream) {
    MR_CHECK_ARG(q && k && v && key_rows && out && probs, "mr_poolattn_fwd: null pointer");
    MR_CHECK_ARG(G > 0 && R > 0 && R <= MAXR && nh > 0 && ldkv % 8 == 0, "mr_poolattn_fwd: bad shape (R <= %d)", MAXR);
    hipLaunchKernelGGL(poolattn_fwd_kernel, dim3((unsigned)((G + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const __bf16*>(q), static_cast<const __bf16*>(k), static_cast<const __bf16*>(v), ldkv,
                       key_rows, static_cast<__bf16*>(out), probs, G, (int)R, (int)nh);
    MR_CHECK_LAUNCH("mr_poolattn_fwd");
    return MR_OK;
}

extern "C" int mr_poolattn_bwd(const void* q, const void* k, const void* v, int64_t ldkv, const int32_t* key_rows,
                               const float* probs, const void* dout, void* dq, void* dk, void* dv, int64_t G, int64_t R,
                               int64_t nh, void* stream) {
    MR_CHECK_ARG(q && k && v && key_rows && probs && dout && dq && dk && dv, "mr_poolattn_bwd: null pointer");
    MR_CHECK_ARG(G > 0 && R > 0 && R <= MAXR && nh > 0 && ldkv % 8 == 0, "mr_poolattn_bwd: bad shape (R <= %d)", MAXR);
    hipLaunchKernelGGL(poolattn_bwd_kernel, dim3((unsigned)((G + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const __bf16*>(q), static_cast<const __bf16*>(k), static_cast<const __bf16*>(v), ldkv,
                       key_rows, probs, static_cast<const __bf16*>(dout), static_cast<__bf16*>(dq), static_cast<__bf16*>(dk),
                       static_cast<__bf16*>(dv), G, (int)R, (int)nh);
    MR_CHECK_LAUNCH("mr_poolattn_bwd");
    return MR_OK;
}

extern "C" int mr_pad_cols(const void* src, int64_t cols_in, void* dst, int64_t cols_out, int64_t rows, void* stream) {
    MR_CHECK_ARG(src && dst && rows > 0 && cols_out >= cols_in && cols_in > 0, "mr_pad_cols: bad args");
    const int64_t n = rows * cols_out;
    hipLaunchKernelGGL(pad_cols_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const __bf16*>(src), cols_in, static_cast<__bf16*>(dst), cols_out, rows);
    MR_CHECK_LAUNCH("mr_pad_cols");
    return MR_OK;
}

extern "C" int mr_fill_rows(const void* vec, void* dst, int64_t ldd, int64_t ngroups, int64_t grp_stride, int64_t off, int64_t H,
                            void* stream) {
    MR_CHECK_ARG(vec && dst && ngroups > 0 && H % 8 == 0 && ldd % 8 == 0, "mr_fill_rows: bad args");
    const int64_t n = ngroups * (H / 8);
    hipLaunchKernelGGL(fill_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const __bf16*>(vec), static_cast<__bf16*>(dst), ldd, ngroups, grp_stride, off, (int)H);
    MR_CHECK_LAUNCH("mr_fill_rows");
    return MR_OK;
}

extern "C" int mr_sum_rows_strided(const void* src, int64_t lds, int64_t ngroups, int64_t grp_stride, int64_t off, int64_t H,
                                   void* out, void* stream) {
    MR_CHECK_ARG(src && out && ngroups > 0 && H > 0, "mr_sum_rows_strided: bad args");
    hipLaunchKernelGGL(sum_rows_strided_kernel, dim3((unsigned)((H + 63) / 64)), dim3(1024), 0, static_cast<hipStream_t>(stream),
                       static_cast<const __bf16*>(src), lds, ngroups, grp_stride, off, (int)H, static_cast<__bf16*>(out));
    MR_CHECK_LAUNCH("mr_sum_rows_strided");
    return MR_OK;
}

extern "C" int mr_add_bf16(const void* a, const void* b, void* y, int64_t n, void* stream) {
    MR_CHECK_ARG(a && b && y && n > 0 && n % 8 == 0, "mr_add_bf16: n must be a positive multiple of 8");
    int64_t blocks = (n / 8 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(add_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const __bf16*>(a), static_cast<const __bf16*>(b), static_cast<__bf16*>(y), n / 8);
    MR_CHECK_LAUNCH("mr_add_bf16");
    return MR_OK;
}

extern "C" int mr_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream) {
    MR_CHECK_ARG(src && dst && n > 0, "mr_cast_f32_to_bf16: bad args");
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), src,
                       static_cast<__bf16*>(dst), n);
    MR_CHECK_LAUNCH("mr_cast_f32_to_bf16");
    return MR_OK;
}

extern "C" int mr_split_f32_to_bf16_hilo(const float* src, void* hi, void* lo, int64_t n, void* stream) {
    MR_CHECK_ARG(src && hi && lo && n > 0, "mr_split_f32_to_bf16_hilo: bad args");
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(split_hilo_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), src,
                       static_cast<__bf16*>(hi), static_cast<__bf16*>(lo), n);
    MR_CHECK_LAUNCH("mr_split_f32_to_bf16_hilo");
    return MR_OK;
}

extern "C" int mr_split_f32_to_bf16_hilo_rows(const float* src, int64_t lds, void* hi, void* lo, int64_t ldo, int64_t rows,
                                              int64_t cols, void* stream) {
    MR_CHECK_ARG(src && hi && lo && rows > 0 && cols > 0 && lds >= cols && ldo >= cols, "mr_split_f32_to_bf16_hilo_rows: bad args");
    int64_t blocks = (rows * cols + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(split_hilo_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), src, lds,
                       static_cast<__bf16*>(hi), static_cast<__bf16*>(lo), ldo, rows, cols);
    MR_CHECK_LAUNCH("mr_split_f32_to_bf16_hilo_rows");
    return MR_OK;
}

extern "C" int mr_unit_norm_scale_fwd(const void* x, int64_t ldx, const void* log_scale, void* y, int64_t ldy, float* inv_norm,
                                      int64_t rows, int64_t H, void* stream) {
    MR_CHECK_ARG(x && log_scale && y && inv_norm && rows > 0 && H % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0,
                 "mr_unit_norm_scale_fwd: bad args");
    hipLaunchKernelGGL(unit_norm_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const __bf16*>(x), ldx, static_cast<const __bf16*>(log_scale), static_cast<__bf16*>(y), ldy,
                       inv_norm, rows, (int)H);
    MR_CHECK_LAUNCH("mr_unit_norm_scale_fwd");
    return MR_OK;
}

extern "C" int mr_unit_norm_scale_bwd(const void* x, int64_t ldx, const void* log_scale, const float* inv_norm, const void* dy,
                                      int64_t lddy, void* dx, int64_t lddx, float* dlog_scale, int32_t accumulate,
                                      float* partials, int64_t rows, int64_t H, void* stream) {
    MR_CHECK_ARG(x && log_scale && inv_norm && dy && dx && dlog_scale && partials && rows > 0 && H % 8 == 0 && ldx % 8 == 0 &&
                     lddy % 8 == 0 && lddx % 8 == 0, "mr_unit_norm_scale_bwd: bad args");
    const int64_t nblk = (rows + 3) / 4;                 // = the number of floats `partials` must hold
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(unit_norm_bwd_kernel, dim3((unsigned)nblk), dim3(256), 0, st,
                       static_cast<const __bf16*>(x), ldx, static_cast<const __bf16*>(log_scale), inv_norm,
                       static_cast<const __bf16*>(dy), lddy, static_cast<__bf16*>(dx), lddx, partials, rows, (int)H);
    hipLaunchKernelGGL(reduce_f32_ordered_kernel, dim3(1), dim3(256), 0, st, partials, nblk, 1.0f, dlog_scale, (int)accumulate);
    MR_CHECK_LAUNCH("mr_unit_norm_scale_bwd");
    return MR_OK;
}

extern "C" int mr_masked_lm_xent(const float* logits, int64_t ldl, int64_t n, int64_t V, const int32_t* labels, float* out2,
                                 float* dlogits, float* row_scratch, void* stream) {
    MR_CHECK_ARG(logits && labels && out2 && row_scratch && n > 0 && V > 0 && ldl >= V, "mr_masked_lm_xent: bad args");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)((n + 3) / 4));
    hipLaunchKernelGGL(masked_lm_rows_kernel, grid, dim3(256), 0, st, logits, ldl, n, V, labels, row_scratch);
    hipLaunchKernelGGL(masked_lm_reduce_kernel, dim3(1), dim3(256), 0, st, row_scratch, n, out2);
    if (dlogits != nullptr)
        hipLaunchKernelGGL(masked_lm_grad_kernel, grid, dim3(256), 0, st, logits, ldl, n, V, labels, row_scratch, out2, dlogits);
    MR_CHECK_LAUNCH("mr_masked_lm_xent");
    return MR_OK;
}

extern "C" int mr_contrastive_lse(float* logits, int64_t ldl, int64_t L, int64_t V, int64_t own_off, float coef,
                                  const int32_t* src, float* loss_out, float* diag, float* row_scratch, void* stream) {
    MR_CHECK_ARG(logits && loss_out && row_scratch && L > 0 && V > 0 && own_off >= 0 && own_off + L <= V, "mr_contrastive_lse: bad args");
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(contrastive_lse_kernel, dim3((unsigned)((L + 3) / 4)), dim3(256), 0, st, logits, ldl, L, V, own_off, coef,
                       row_scratch);
    hipLaunchKernelGGL(contrastive_reduce_kernel, dim3(1), dim3(256), 0, st, row_scratch, L, coef, src, loss_out, diag);
    MR_CHECK_LAUNCH("mr_contrastive_lse");
    return MR_OK;
}
