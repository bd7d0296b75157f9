// Fused optimizer step for gfx950 (HBM-bound, 20 B/param): nan_to_num -> bf16-state Adam (momentum bf16, second moment
// cube-coded "unsigned bf16") -> weight decay (leaves with ndim > 1) -> schedule -> -lr -> apply, and the bf16 working
// copy of the parameters for the next forward.
// Replaces pretrain/pretrain_model.py:323-324,328,339 + pretrain/optimization.py:36-51 (codec), :54-114 (Adam),
// :180-195 (optax chain).
#include "mr_common.h"

namespace {

constexpr float MISSING_PRECISION = 1.0f + 1.0f / 512.0f;   // optimization.py:36

__device__ __forceinline__ float nan_to_num_f(float g) {
    // jnp.nan_to_num on a bf16 leaf: nan -> 0, +-inf -> +-max finite bf16
    if (g != g) return 0.f;
    const float BF16_MAX = 3.3895313892515355e38f;
    if (g > BF16_MAX) return BF16_MAX;
    if (g < -BF16_MAX) return -BF16_MAX;
    return g;
}

// FT = the finetuning chain (finetune/optimization.py:77-90): after Adam, u -= wd * bf16(initial parameter), then
// u += wd * parameter, both under the same mask.
// DEV: the four per-step scalars (schedule value, -lr, 1/bias_corr1, 1/bias_corr2) are read from device memory, so the
// launch can live inside a captured hipGraph (or be issued before the host has computed them for a later step).
// GF: the gradients are fp32 (the reference's use_bfloat16_grads = False branch, pretrain_model.py:323-333: no bf16 round trip of
// the gradients; nan_to_num at fp32 range) -- same chain otherwise.
template <bool FT, bool DEV = false, bool GF = false>
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ master, __bf16* __restrict__ work,
                                                   const void* __restrict__ grad_, __bf16* __restrict__ mu,
                                                   __bf16* __restrict__ nu, const __bf16* __restrict__ orig,
                                                   const uint8_t* __restrict__ decay_flag, float c1,
                                                   float b1, float c2, float b2, float eps, float wd, float sched, float neg_lr,
                                                   float inv_bc1, float inv_bc2, const float* __restrict__ hyper = nullptr) {
    if (DEV) { sched = hyper[0]; neg_lr = hyper[1]; inv_bc1 = hyper[2]; inv_bc2 = hyper[3]; }
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    const bool decay = decay_flag[blockIdx.x] != 0;
    float g[8], m[8], v[8], p[8];
    // streaming (non-temporal) accesses for everything that is touched once per step -- gradients, both moments, the fp32 master
    // copy: 5.5 -> 5.9-6.1 TB/s for the kernel alone; the bf16 working copy, which the next forward reads, is stored normally
#define MR_LD(T, p) __builtin_nontemporal_load(reinterpret_cast<const T*>(p))
#define MR_ST(T, p, v) __builtin_nontemporal_store((v), reinterpret_cast<T*>(p))
    if (GF) {
        const float* gp = static_cast<const float*>(grad_) + i;
        const f32x4 g0 = MR_LD(f32x4, gp), g1 = MR_LD(f32x4, gp + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { g[e] = g0[e]; g[4 + e] = g1[e]; }
    } else {
        unpack8(MR_LD(u32x4, static_cast<const __bf16*>(grad_) + i), g);
    }
    unpack8(MR_LD(u32x4, mu + i), m);
    unpack8(MR_LD(u32x4, nu + i), v);
    const f32x4 p0 = MR_LD(f32x4, master + i), p1 = MR_LD(f32x4, master + i + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { p[e] = p0[e]; p[4 + e] = p1[e]; }
    float mo[8], vo[8], wo[8], og[8];
    if (FT) unpack8(*reinterpret_cast<const u32x4*>(orig + i), og);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float ge = GF ? ((g[e] != g[e]) ? 0.f : fminf(fmaxf(g[e], -3.4028234663852886e38f), 3.4028234663852886e38f)) : nan_to_num_f(g[e]);
        const float nm = c1 * ge + b1 * m[e];                                   // optimization.py:83-87
        float va = fabsf(v[e]);
        if (!(v[e] >= 0.f)) va *= MISSING_PRECISION;                             // :38-41 (v >= 0 also true for -0.0)
        const float nv = c2 * (ge * ge) + b2 * cbrtf(va);                        // :89-92
        mo[e] = nm;
        const float v3 = nv * nv * nv;                                           // :44-51
        const float vb = (float)(__bf16)v3;
        const float err0 = fabsf(vb - v3), err1 = fabsf(vb * MISSING_PRECISION - v3);
        vo[e] = (err0 < err1) ? vb : -vb;
        float u = (nm * inv_bc1) / (sqrtf(nv * inv_bc2) + eps);                  // :104-110
        if (FT && decay) u -= wd * og[e];                                        // finetune/optimization.py:30-31
        if (decay) u += wd * p[e];                                               // :182-184
        u = (u * sched) * neg_lr;                                                // :185-189
        p[e] += u;
        wo[e] = p[e];
    }
    MR_ST(u32x4, mu + i, pack8(mo));
    // -0.0 must survive as the sign-coded zero: pack via the sign-preserving conversion
    MR_ST(u32x4, nu + i, pack8(vo));
    *reinterpret_cast<u32x4*>(work + i) = pack8(wo);
    MR_ST(f32x4, master + i, (f32x4{p[0], p[1], p[2], p[3]}));
    MR_ST(f32x4, master + i + 4, (f32x4{p[4], p[5], p[6], p[7]}));
}

__global__ void nan_to_num_kernel(__bf16* __restrict__ g, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        float v[8];
        unpack8(*reinterpret_cast<const u32x4*>(g + 8 * i), v);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = nan_to_num_f(v[e]);
        *reinterpret_cast<u32x4*>(g + 8 * i) = pack8(v);
    }
}

__global__ void cast_params_kernel(const float* __restrict__ master, __bf16* __restrict__ work, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(master + 8 * i), b = *reinterpret_cast<const f32x4*>(master + 8 * i + 4);
        const float f[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        *reinterpret_cast<u32x4*>(work + 8 * i) = pack8(f);
    }
}

// Transposed bf16 working copies of the Dense kernels ([K, N] as flax stores them -> [N, K]) so that FORWARD GEMMs read a
// K-contiguous B operand like the dgrads do (gemm3.hip takes "NT" operands only; K-contiguous fragments are plain
// ds_read_b128).  One 64 x 64 tile per workgroup through LDS; the leaf of a tile is found by bisection over the leaves'
// first-tile table.  HBM-bound: 4 B / parameter, ~1 % of a training step.
struct TrLeaf { int off, rows, cols, tile0; };     // element offset of the leaf in the flat buffers; [rows, cols] source; its first tile
__global__ __launch_bounds__(256) void transpose_leaves_kernel(const __bf16* __restrict__ work, __bf16* __restrict__ workT,
                                                               const TrLeaf* __restrict__ leaves, int nleaf, int tile_lo) {
    __shared__ __bf16 tl[64][72];
    const int tile = tile_lo + blockIdx.x;
    int lo = 0, hi = nleaf - 1;
    while (lo < hi) {                     // last leaf whose first tile is <= tile
        const int mid = (lo + hi + 1) >> 1;
        if (leaves[mid].tile0 <= tile) lo = mid; else hi = mid - 1;
    }
    const TrLeaf lf = leaves[lo];
    const int tc = lf.cols >> 6, t = tile - lf.tile0, r0 = (t / tc) << 6, c0 = (t % tc) << 6;
    const int tid = threadIdx.x, r = tid >> 2, q = tid & 3;
    const __bf16* src = work + lf.off + (int64_t)(r0 + r) * lf.cols + c0 + q * 16;
    const bf16x8 v0 = *reinterpret_cast<const bf16x8*>(src), v1 = *reinterpret_cast<const bf16x8*>(src + 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) { tl[q * 16 + e][r] = v0[e]; tl[q * 16 + 8 + e][r] = v1[e]; }
    __syncthreads();
    __bf16* dst = workT + lf.off + (int64_t)(c0 + r) * lf.rows + r0 + q * 16;
    *reinterpret_cast<bf16x8*>(dst) = *reinterpret_cast<const bf16x8*>(&tl[r][q * 16]);
    *reinterpret_cast<bf16x8*>(dst + 8) = *reinterpret_cast<const bf16x8*>(&tl[r][q * 16 + 8]);
}

}  // namespace

extern "C" int mr_adam_bf16_update(float* master, void* work_bf16, const void* grad_bf16, void* mu_bf16, void* nu_bf16,
                                   const uint8_t* decay_flag_per_block, int64_t n, double b1, double b2, float eps,
                                   float weight_decay, float sched, float neg_lr, float bias_corr1, float bias_corr2,
                                   void* stream) {
    MR_CHECK_ARG(master && work_bf16 && grad_bf16 && mu_bf16 && nu_bf16 && decay_flag_per_block, "mr_adam_bf16_update: null pointer");
    MR_CHECK_ARG(n > 0 && n % 2048 == 0, "mr_adam_bf16_update: n must be a positive multiple of 2048 (got %ld)", (long)n);
    MR_CHECK_ARG(bias_corr1 > 0.f && bias_corr2 > 0.f, "mr_adam_bf16_update: bias corrections must be > 0 (1 disables)");
    // (1 - b) evaluated like the reference: Python double subtraction, then cast to f32 (optimization.py:86, 91)
    const float c1 = (float)(1.0 - b1), c2 = (float)(1.0 - b2);      // f32(1 - b) of the Python float b, as jnp evaluates (1 - b1) * g
    hipLaunchKernelGGL(adam_kernel<false>, dim3((unsigned)(n / 2048)), dim3(256), 0, static_cast<hipStream_t>(stream), master,
                       static_cast<__bf16*>(work_bf16), static_cast<const __bf16*>(grad_bf16), static_cast<__bf16*>(mu_bf16),
                       static_cast<__bf16*>(nu_bf16), static_cast<const __bf16*>(nullptr), decay_flag_per_block, c1, (float)b1, c2, (float)b2,
                       eps, weight_decay, sched, neg_lr, 1.0f / bias_corr1, 1.0f / bias_corr2);
    MR_CHECK_LAUNCH("mr_adam_bf16_update");
    return MR_OK;
}

extern "C" int mr_adam_bf16_update_finetune(float* master, void* work_bf16, const void* grad_bf16, void* mu_bf16, void* nu_bf16,
                                            const void* orig_bf16, const uint8_t* decay_flag_per_block, int64_t n, double b1,
                                            double b2, float eps, float weight_decay, float sched, float neg_lr,
                                            float bias_corr1, float bias_corr2, void* stream) {
    MR_CHECK_ARG(master && work_bf16 && grad_bf16 && mu_bf16 && nu_bf16 && orig_bf16 && decay_flag_per_block,
                 "mr_adam_bf16_update_finetune: null pointer");
    MR_CHECK_ARG(n > 0 && n % 2048 == 0, "mr_adam_bf16_update_finetune: n must be a positive multiple of 2048 (got %ld)", (long)n);
    MR_CHECK_ARG(bias_corr1 > 0.f && bias_corr2 > 0.f, "mr_adam_bf16_update_finetune: bias corrections must be > 0");
    const float c1 = (float)(1.0 - b1), c2 = (float)(1.0 - b2);      // f32(1 - b) of the Python float b, as jnp evaluates (1 - b1) * g
    hipLaunchKernelGGL(adam_kernel<true>, dim3((unsigned)(n / 2048)), dim3(256), 0, static_cast<hipStream_t>(stream), master,
                       static_cast<__bf16*>(work_bf16), static_cast<const __bf16*>(grad_bf16), static_cast<__bf16*>(mu_bf16),
                       static_cast<__bf16*>(nu_bf16), static_cast<const __bf16*>(orig_bf16), decay_flag_per_block, c1, (float)b1, c2,
                       (float)b2, eps, weight_decay, sched, neg_lr, 1.0f / bias_corr1, 1.0f / bias_corr2);
    MR_CHECK_LAUNCH("mr_adam_bf16_update_finetune");
    return MR_OK;
}

extern "C" int mr_adam_bf16_update_dev(float* master, void* work_bf16, const void* grad_bf16, void* mu_bf16, void* nu_bf16,
                                       const void* orig_bf16, const uint8_t* decay_flag_per_block, int64_t n, double b1, double b2,
                                       float eps, float weight_decay, const float* hyper_dev, void* stream) {
    MR_CHECK_ARG(master && work_bf16 && grad_bf16 && mu_bf16 && nu_bf16 && decay_flag_per_block && hyper_dev,
                 "mr_adam_bf16_update_dev: null pointer");
    MR_CHECK_ARG(n > 0 && n % 2048 == 0, "mr_adam_bf16_update_dev: n must be a positive multiple of 2048 (got %ld)", (long)n);
    const float c1 = (float)(1.0 - b1), c2 = (float)(1.0 - b2);      // f32(1 - b) of the Python float b, as jnp evaluates (1 - b1) * g
    hipStream_t st = static_cast<hipStream_t>(stream);
    dim3 grid((unsigned)(n / 2048));
    if (orig_bf16 != nullptr)
        hipLaunchKernelGGL((adam_kernel<true, true>), grid, dim3(256), 0, st, master, static_cast<__bf16*>(work_bf16),
                           static_cast<const __bf16*>(grad_bf16), static_cast<__bf16*>(mu_bf16), static_cast<__bf16*>(nu_bf16),
                           static_cast<const __bf16*>(orig_bf16), decay_flag_per_block, c1, (float)b1, c2, (float)b2, eps, weight_decay, 0.f, 0.f,
                           1.f, 1.f, hyper_dev);
    else
        hipLaunchKernelGGL((adam_kernel<false, true>), grid, dim3(256), 0, st, master, static_cast<__bf16*>(work_bf16),
                           static_cast<const __bf16*>(grad_bf16), static_cast<__bf16*>(mu_bf16), static_cast<__bf16*>(nu_bf16),
                           static_cast<const __bf16*>(nullptr), decay_flag_per_block, c1, (float)b1, c2, (float)b2, eps, weight_decay, 0.f, 0.f,
                           1.f, 1.f, hyper_dev);
    MR_CHECK_LAUNCH("mr_adam_bf16_update_dev");
    return MR_OK;
}

// fp32 gradients (use_bfloat16_grads = False), per-step scalars in device memory; pretraining chain only.
extern "C" int mr_adam_f32grad_update_dev(float* master, void* work_bf16, const float* grad_f32, void* mu_bf16, void* nu_bf16,
                                          const uint8_t* decay_flag_per_block, int64_t n, double b1, double b2, float eps, float weight_decay,
                                          const float* hyper_dev, void* stream) {
    MR_CHECK_ARG(master && work_bf16 && grad_f32 && mu_bf16 && nu_bf16 && decay_flag_per_block && hyper_dev, "mr_adam_f32grad_update_dev: null pointer");
    MR_CHECK_ARG(n > 0 && n % 2048 == 0, "mr_adam_f32grad_update_dev: n must be a positive multiple of 2048 (got %ld)", (long)n);
    const float c1 = (float)(1.0 - b1), c2 = (float)(1.0 - b2);
    hipLaunchKernelGGL((adam_kernel<false, true, true>), dim3((unsigned)(n / 2048)), dim3(256), 0, static_cast<hipStream_t>(stream), master,
                       static_cast<__bf16*>(work_bf16), static_cast<const void*>(grad_f32), static_cast<__bf16*>(mu_bf16), static_cast<__bf16*>(nu_bf16),
                       (const __bf16*)nullptr, decay_flag_per_block, c1, (float)b1, c2, (float)b2, eps, weight_decay, 0.f, 0.f, 1.f, 1.f, hyper_dev);
    MR_CHECK_LAUNCH("mr_adam_f32grad_update_dev");
    return MR_OK;
}

extern "C" int mr_nan_to_num_bf16(void* g, int64_t n, void* stream) {
    MR_CHECK_ARG(g && n > 0 && n % 8 == 0, "mr_nan_to_num_bf16: n must be a positive multiple of 8");
    int64_t blocks = (n / 8 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(nan_to_num_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<__bf16*>(g), n / 8);
    MR_CHECK_LAUNCH("mr_nan_to_num_bf16");
    return MR_OK;
}

extern "C" int mr_cast_f32_to_bf16_params(const float* master, void* work_bf16, int64_t n, void* stream) {
    MR_CHECK_ARG(master && work_bf16 && n > 0 && n % 8 == 0, "mr_cast_f32_to_bf16_params: n must be a positive multiple of 8");
    int64_t blocks = (n / 8 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(cast_params_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), master,
                       static_cast<__bf16*>(work_bf16), n / 8);
    MR_CHECK_LAUNCH("mr_cast_f32_to_bf16_params");
    return MR_OK;
}

extern "C" int mr_transpose_leaves(const void* work_bf16, void* workT_bf16, const int32_t* leaves_dev, int32_t nleaf, int32_t tile_lo,
                                   int32_t tile_hi, void* stream) {
    MR_CHECK_ARG(work_bf16 && workT_bf16 && leaves_dev && nleaf > 0, "mr_transpose_leaves: null pointer / empty table");
    MR_CHECK_ARG(tile_lo >= 0 && tile_hi >= tile_lo, "mr_transpose_leaves: bad tile range [%d, %d)", tile_lo, tile_hi);
    if (tile_hi == tile_lo) return MR_OK;
    hipLaunchKernelGGL(transpose_leaves_kernel, dim3((unsigned)(tile_hi - tile_lo)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const __bf16*>(work_bf16), static_cast<__bf16*>(workT_bf16),
                       reinterpret_cast<const TrLeaf*>(leaves_dev), nleaf, tile_lo);
    MR_CHECK_LAUNCH("mr_transpose_leaves");
    return MR_OK;
}
