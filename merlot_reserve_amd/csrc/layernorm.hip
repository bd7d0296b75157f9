// LayerNorm forward / backward and column sums for gfx950 (HBM-bound: one wave per row, 16-byte vectors,
// wavefront reductions).  Replaces flax nn.LayerNorm(epsilon=1e-5) as used at mreserve/modeling.py:272,277,360,366:
// statistics in fp32 with var = E[x^2] - E[x]^2, y = (x - mean) * (rsqrt(var + eps) * scale) + bias.
#include <string.h>
#include "mr_common.h"
#include "mr_options.h"

namespace {

constexpr int MAXC = 4;           // 16-byte chunks per lane (template MC <= MAXC): H <= 64 * 8 * 4 = 2048
constexpr int PART_ROWS = 1024;   // ln_bwd grid: 4 blocks per CU    // rows of the fp32 partial-sum workspace

// Two rows per wave: both rows' chunks are requested before either row's reductions run (with one row per wave the kernel sat at
// 4.1 TB/s: a wave had 1.5 loads in flight for H = 768), gamma / beta are fetched once for both.
template <int MC>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const __bf16* __restrict__ x, int64_t ldx, const __bf16* __restrict__ gamma,
                                                     const __bf16* __restrict__ beta, __bf16* __restrict__ y, int64_t ldy,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out, int64_t rows,
                                                     int H, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) * 2;     // (uniform: row bases in scalar registers)
    if (row0 >= rows) return;
    const bool two = row0 + 1 < rows;
    const int nch = H >> 3;
    u32x4 raw[2][MC];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int k = 0; k < MC; ++k) {
            const int c = lane + 64 * k;
            raw[r][k] = u32x4{0u, 0u, 0u, 0u};
            if (c < nch && (r == 0 || two)) raw[r][k] = *reinterpret_cast<const u32x4*>(x + (row0 + r) * ldx + 8 * c);
        }
    u32x4 graw[MC], braw[MC];
#pragma unroll
    for (int k = 0; k < MC; ++k) {
        const int c = lane + 64 * k;
        graw[k] = u32x4{0u, 0u, 0u, 0u}; braw[k] = graw[k];
        if (c < nch) {
            graw[k] = *reinterpret_cast<const u32x4*>(gamma + 8 * c);
            braw[k] = *reinterpret_cast<const u32x4*>(beta + 8 * c);
        }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        if (r == 1 && !two) break;
        float v[MC][8];
        float s = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < MC; ++k) {
            unpack8(raw[r][k], v[k]);                     // chunks beyond H are zeros: they add nothing
#pragma unroll
            for (int e = 0; e < 8; ++e) { s += v[k][e]; s2 += v[k][e] * v[k][e]; }
        }
        s = wave_sum(s);
        s2 = wave_sum(s2);
        const float mean = s / (float)H;
        const float var = s2 / (float)H - mean * mean;
        const float rstd = rsqrtf(var + eps);
        if (lane == 0) {
            if (mean_out) mean_out[row0 + r] = mean;
            if (rstd_out) rstd_out[row0 + r] = rstd;
        }
#pragma unroll
        for (int k = 0; k < MC; ++k) {
            const int c = lane + 64 * k;
            if (c < nch) {
                float gm[8], bt[8], o[8];
                unpack8(graw[k], gm);
                unpack8(braw[k], bt);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (v[k][e] - mean) * (rstd * gm[e]) + bt[e];
                *reinterpret_cast<u32x4*>(y + (row0 + r) * ldy + 8 * c) = pack8(o);
            }
        }
    }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma;  partial sums of dgamma = dy * xhat, dbeta = dy.
template <int MC>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const __bf16* __restrict__ dy, int64_t lddy, const __bf16* __restrict__ x,
                                                     int64_t ldx, const __bf16* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, __bf16* dx, int64_t lddx,
                                                     const __bf16* dx_add, int64_t ldadd, float* __restrict__ partials,
                                                     int64_t rows, int H) {
    extern __shared__ __attribute__((aligned(16))) float red[];   // [4 waves][2H]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nch = H >> 3;
    float pg[MC][8], pb[MC][8], gm[MC][8];
#pragma unroll
    for (int k = 0; k < MC; ++k) {
        const int c = lane + 64 * k;
#pragma unroll
        for (int e = 0; e < 8; ++e) { pg[k][e] = 0.f; pb[k][e] = 0.f; gm[k][e] = 0.f; }
        if (c < nch) unpack8(*reinterpret_cast<const u32x4*>(gamma + 8 * c), gm[k]);
    }
    // Rows are software-pipelined: the next row's x / dy / dx_add chunks are requested before the current row's two
    // dependent phases (row reductions, then the dx formula) run, so a wave keeps two rows of loads in flight instead of
    // one (the kernel is HBM-latency-bound at 16 waves per CU: 3.8 TB/s before).
    const int64_t rstride = (int64_t)gridDim.x * 4;
    int64_t row = (int64_t)blockIdx.x * 4 + wave;
    u32x4 rx[MC], rd[MC], ra[MC];
    float mu = 0.f, rs = 0.f;
    auto fetch = [&](int64_t r, u32x4 (&X)[MC], u32x4 (&D)[MC], u32x4 (&A)[MC], float& m_, float& r_) {
        m_ = mean[r];
        r_ = rstd[r];
#pragma unroll
        for (int k = 0; k < MC; ++k) {
            const int c = lane + 64 * k;
            X[k] = u32x4{0u, 0u, 0u, 0u}; D[k] = X[k]; A[k] = X[k];
            if (c < nch) {
                X[k] = *reinterpret_cast<const u32x4*>(x + r * ldx + 8 * c);
                D[k] = *reinterpret_cast<const u32x4*>(dy + r * lddy + 8 * c);
                if (dx_add != nullptr) A[k] = *reinterpret_cast<const u32x4*>(dx_add + r * ldadd + 8 * c);
            }
        }
    };
    if (row < rows) fetch(row, rx, rd, ra, mu, rs);
    for (; row < rows; row += rstride) {
        u32x4 nx[MC], nd[MC], na[MC];
        float nmu = 0.f, nrs = 0.f;
        const bool more = row + rstride < rows;
        if (more) fetch(row + rstride, nx, nd, na, nmu, nrs);
        float xh[MC][8], gg[MC][8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < MC; ++k) {
            const int c = lane + 64 * k;
            if (c < nch) {
                float xv[8], dv[8];
                unpack8(rx[k], xv);
                unpack8(rd[k], dv);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    xh[k][e] = (xv[e] - mu) * rs;
                    gg[k][e] = dv[e] * gm[k][e];
                    s1 += gg[k][e];
                    s2 += gg[k][e] * xh[k][e];
                    pg[k][e] += dv[e] * xh[k][e];
                    pb[k][e] += dv[e];
                }
            }
        }
        s1 = wave_sum(s1) / (float)H;
        s2 = wave_sum(s2) / (float)H;
#pragma unroll
        for (int k = 0; k < MC; ++k) {
            const int c = lane + 64 * k;
            if (c < nch) {
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = rs * (gg[k][e] - s1 - xh[k][e] * s2);
                if (dx_add != nullptr) {
                    float old[8];
                    unpack8(ra[k], old);
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] += old[e];
                }
                *reinterpret_cast<u32x4*>(dx + row * lddx + 8 * c) = pack8(o);
            }
        }
        if (more) {
#pragma unroll
            for (int k = 0; k < MC; ++k) { rx[k] = nx[k]; rd[k] = nd[k]; ra[k] = na[k]; }
            mu = nmu;
            rs = nrs;
        }
    }
    // reduce the 4 waves' column partials through LDS, one fp32 row [2H] per block
#pragma unroll
    for (int k = 0; k < MC; ++k) {
        const int c = lane + 64 * k;
        if (c < nch) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                red[wave * 2 * H + 8 * c + e] = pg[k][e];
                red[wave * 2 * H + H + 8 * c + e] = pb[k][e];
            }
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * H; c += 256)
        partials[(int64_t)blockIdx.x * 2 * H + c] = red[c] + red[2 * H + c] + red[4 * H + c] + red[6 * H + c];
}

// Round 6.  The kernel above holds 161 registers (three waves per SIMD), so of its 1024 workgroups only 768 were resident and the last
// 256 ran as a second round on a quarter of the chip's wave slots: 3.5 TB/s.  This one keeps <= 128 registers -- gamma stays packed, xhat and
// g = dy * gamma are formed twice (before and after the row sums) instead of being held across them -- in 512-thread workgroups, two per CU, so
// the WHOLE grid is resident from the first cycle (every wave has two rows of loads in flight at once) and a launch leaves at most 512 partial rows
// (half the bytes for the deferred reduction to read).  Same arithmetic, same order per row and per column partial as the kernel above.
constexpr int PART_ROWS2 = 512;
template <int MC, bool ADD>
__global__ __launch_bounds__(512, 4) void ln_bwd2_kernel(const __bf16* __restrict__ dy, int64_t lddy, const __bf16* __restrict__ x,
                                                         int64_t ldx, const __bf16* __restrict__ gamma, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, __bf16* dx, int64_t lddx,
                                                         const __bf16* dx_add, int64_t ldadd, float* __restrict__ partials,
                                                         int64_t rows, int H) {
    extern __shared__ __attribute__((aligned(16))) float red[];   // [4][2H]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (uniform: row bases live in scalar registers)
    const int nch = H >> 3;
    float pg[MC][8], pb[MC][8];
    u32x4 gmr[MC];
#pragma unroll
    for (int k = 0; k < MC; ++k) {
        const int c = lane + 64 * k;
#pragma unroll
        for (int e = 0; e < 8; ++e) { pg[k][e] = 0.f; pb[k][e] = 0.f; }
        gmr[k] = u32x4{0u, 0u, 0u, 0u};
        if (c < nch) gmr[k] = *reinterpret_cast<const u32x4*>(gamma + 8 * c);
    }
    const int64_t rstride = (int64_t)gridDim.x * 8;
    int64_t row = (int64_t)blockIdx.x * 8 + wave;
    u32x4 rx[MC], rd[MC];
    float mu = 0.f, rs = 0.f;
    auto fetch = [&](int64_t r, u32x4 (&X)[MC], u32x4 (&D)[MC], float& m_, float& r_) {
        m_ = mean[r];
        r_ = rstd[r];
#pragma unroll
        for (int k = 0; k < MC; ++k) {
            const int c = lane + 64 * k;
            X[k] = u32x4{0u, 0u, 0u, 0u}; D[k] = X[k];
            if (c < nch) {
                X[k] = *reinterpret_cast<const u32x4*>(x + r * ldx + 8 * c);
                D[k] = *reinterpret_cast<const u32x4*>(dy + r * lddy + 8 * c);
            }
        }
    };
    if (row < rows) fetch(row, rx, rd, mu, rs);
    for (; row < rows; row += rstride) {
        // this row's residual-path gradient (wanted after the row sums) and the NEXT row's x / dy are requested before any arithmetic
        u32x4 ra[MC], nx[MC], nd[MC];
        if (ADD) {
#pragma unroll
            for (int k = 0; k < MC; ++k) {
                const int c = lane + 64 * k;
                ra[k] = u32x4{0u, 0u, 0u, 0u};
                if (c < nch) ra[k] = *reinterpret_cast<const u32x4*>(dx_add + row * ldadd + 8 * c);
            }
        }
        float nmu = 0.f, nrs = 0.f;
        const bool more = row + rstride < rows;
        if (more) fetch(row + rstride, nx, nd, nmu, nrs);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < MC; ++k) {
            const int c = lane + 64 * k;
            if (c < nch) {
                float xv[8], dv[8], gm[8];
                unpack8(rx[k], xv);
                unpack8(rd[k], dv);
                unpack8(gmr[k], gm);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xh = (xv[e] - mu) * rs;
                    const float gg = dv[e] * gm[e];
                    s1 += gg;
                    s2 += gg * xh;
                    pg[k][e] += dv[e] * xh;
                    pb[k][e] += dv[e];
                }
            }
        }
        s1 = wave_sum(s1) / (float)H;
        s2 = wave_sum(s2) / (float)H;
        // (opaque to the optimiser: otherwise common-subexpression elimination keeps the 32 floats of xhat / g alive across the sums)
#pragma unroll
        for (int k = 0; k < MC; ++k) asm volatile("" : "+v"(rx[k]), "+v"(rd[k]));
#pragma unroll
        for (int k = 0; k < MC; ++k) {
            const int c = lane + 64 * k;
            if (c < nch) {
                float xv[8], dv[8], gm[8], o[8];
                unpack8(rx[k], xv);
                unpack8(rd[k], dv);
                unpack8(gmr[k], gm);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xh = (xv[e] - mu) * rs;
                    const float gg = dv[e] * gm[e];
                    o[e] = rs * (gg - s1 - xh * s2);
                }
                if (ADD) {
                    float old[8];
                    unpack8(ra[k], old);
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] += old[e];
                }
                *reinterpret_cast<u32x4*>(dx + row * lddx + 8 * c) = pack8(o);
            }
        }
        if (more) {
#pragma unroll
            for (int k = 0; k < MC; ++k) { rx[k] = nx[k]; rd[k] = nd[k]; }
            mu = nmu;
            rs = nrs;
        }
    }
    // the 8 waves' column partials through LDS in two steps (4 rows of [2H] floats), fixed order ((w0 + w4) + (w1 + w5)) + ((w2 + w6) + (w3 + w7))
    auto put = [&](bool add) {
        float* dst = red + (wave & 3) * 2 * H;
#pragma unroll
        for (int k = 0; k < MC; ++k) {
            const int c = lane + 64 * k;
            if (c < nch) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    dst[8 * c + e] = add ? dst[8 * c + e] + pg[k][e] : pg[k][e];
                    dst[H + 8 * c + e] = add ? dst[H + 8 * c + e] + pb[k][e] : pb[k][e];
                }
            }
        }
    };
    if (wave < 4) put(false);
    __syncthreads();
    if (wave >= 4) put(true);
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * H; c += 512)
        partials[(int64_t)blockIdx.x * 2 * H + c] = (red[c] + red[2 * H + c]) + (red[4 * H + c] + red[6 * H + c]);
}

// Column reduction of fp32 partial rows, two deterministic levels:
//   level 1 (grid.y = RED_Y): block (x, y) sums rows y, y + RED_Y, ... of its 64 columns into mid[y, c]
//   level 2 (grid.y = 1, nparts = RED_Y): sums mid and writes bf16; columns [0, split) -> out0, [split, ncols) -> out1.
// Block = 64 columns x 16 row groups (coalesced 256-B row segments), LDS reduce over the groups.
constexpr int RED_Y = 8;
__global__ __launch_bounds__(1024) void reduce_partials_kernel(const float* __restrict__ partials, int nparts, int ncols, int split,
                                                               float* __restrict__ mid, __bf16* __restrict__ out0,
                                                               __bf16* __restrict__ out1) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (c < ncols)
        for (int p = blockIdx.y + gridDim.y * grp; p < nparts; p += gridDim.y * 16) s += partials[(int64_t)p * ncols + c];
    red[grp][lane] = s;
    __syncthreads();
    if (grp == 0 && c < ncols) {
        s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += red[k][lane];
        if (mid != nullptr) mid[(int64_t)blockIdx.y * ncols + c] = s;
        else if (c < split) out0[c] = (__bf16)s;
        else out1[c - split] = (__bf16)s;
    }
}

static void launch_reduce(const float* partials, int nparts, int ncols, int split, float* mid, __bf16* out0, __bf16* out1, hipStream_t s) {
    const unsigned gx = (unsigned)((ncols + 63) / 64);
    if (nparts > 64) {
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(gx, RED_Y), dim3(1024), 0, s, partials, nparts, ncols, split, mid, out0, out1);
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(gx, 1), dim3(1024), 0, s, mid, RED_Y, ncols, split, (float*)nullptr, out0, out1);
    } else {
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(gx, 1), dim3(1024), 0, s, partials, nparts, ncols, split, (float*)nullptr, out0, out1);
    }
}

// Several independent reductions in ONE launch (the LayerNorm and bias gradients of a transformer layer): single level,
// block = 32 columns of one job x 32 row groups (a row segment = one 128-byte line), fixed summation order.  (64 columns x 16 groups
// left half of the CUs without a block and every thread with 4 dependent rounds of loads for the 244 partial rows of an fc1 bias: 11.7 us.)
constexpr int MAX_JOBS = 16;
struct ReduceBatch {
    int count;
    int blk_start[MAX_JOBS + 1];
    mr_reduce_job job[MAX_JOBS];
};
constexpr int RB_COLS = 32, RB_GRPS = 32;
__global__ __launch_bounds__(1024) void reduce_batch_kernel(const ReduceBatch rb) {
    __shared__ float red[RB_GRPS][RB_COLS];
    const int lane = threadIdx.x & (RB_COLS - 1), grp = threadIdx.x / RB_COLS;
    int j = 0;
#pragma unroll
    for (int k = 1; k < MAX_JOBS; ++k) j += (k < rb.count && (int)blockIdx.x >= rb.blk_start[k]);
    const mr_reduce_job& q = rb.job[j];
    const int c = ((int)blockIdx.x - rb.blk_start[j]) * RB_COLS + lane, ncols = q.ncols, nparts = q.nparts;
    float s = 0.f;
    if (c < ncols) {
        // four independent chains: the loads of a group's partial rows are all in flight at once, summed in a fixed order
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int p = grp;
        for (; p + 3 * RB_GRPS < nparts; p += 4 * RB_GRPS) {
            s0 += q.partials[(int64_t)p * ncols + c];
            s1 += q.partials[(int64_t)(p + RB_GRPS) * ncols + c];
            s2 += q.partials[(int64_t)(p + 2 * RB_GRPS) * ncols + c];
            s3 += q.partials[(int64_t)(p + 3 * RB_GRPS) * ncols + c];
        }
        for (; p < nparts; p += RB_GRPS) s0 += q.partials[(int64_t)p * ncols + c];
        s = (s0 + s1) + (s2 + s3);
    }
    red[grp][lane] = s;
    __syncthreads();
    if (grp == 0 && c < ncols) {
        s = 0.f;
#pragma unroll
        for (int k = 0; k < RB_GRPS; ++k) s += red[k][lane];
        if (c < q.split) static_cast<__bf16*>(q.out0)[c] = (__bf16)s;
        else static_cast<__bf16*>(q.out1)[c - q.split] = (__bf16)s;
    }
}

// column sums: grid.x = 512-column groups, grid.y = row strips; partial[strip, N]
__global__ __launch_bounds__(256) void colsum_kernel(const __bf16* __restrict__ x, int64_t ldx, int64_t rows, int N,
                                                     float* __restrict__ partials) {
    __shared__ float red[4][512];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;     // chunk index
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c < (N >> 3)) {
        for (int64_t row = (int64_t)blockIdx.y * 4 + wave; row < rows; row += (int64_t)gridDim.y * 4) {
            float v[8];
            unpack8(*reinterpret_cast<const u32x4*>(x + row * ldx + 8 * c), v);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += v[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[wave][lane * 8 + e] = acc[e];
    __syncthreads();
    for (int t = threadIdx.x; t < 512; t += 256) {
        const int col = blockIdx.x * 512 + t;
        if (col < N) partials[(int64_t)blockIdx.y * N + col] = red[0][t] + red[1][t] + red[2][t] + red[3][t];
    }
}

}  // namespace

extern "C" int mr_layernorm_fwd(const void* x, int64_t ldx, const void* gamma, const void* beta, void* y, int64_t ldy,
                                float* mean, float* rstd, int64_t rows, int64_t H, float eps, void* stream) {
    MR_CHECK_ARG(x && gamma && beta && y, "mr_layernorm_fwd: null pointer");
    MR_CHECK_ARG(rows > 0 && H > 0 && H % 8 == 0 && H <= 64 * 8 * MAXC, "mr_layernorm_fwd: H=%ld unsupported", (long)H);
    MR_CHECK_ARG(ldx % 8 == 0 && ldy % 8 == 0, "mr_layernorm_fwd: leading dims must be multiples of 8");
    auto kern = (H <= 1024) ? ln_fwd_kernel<2> : ln_fwd_kernel<MAXC>;
    hipLaunchKernelGGL(kern, dim3((unsigned)((rows + 7) / 8)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const __bf16*>(x), ldx, static_cast<const __bf16*>(gamma), static_cast<const __bf16*>(beta),
                       static_cast<__bf16*>(y), ldy, mean, rstd, rows, (int)H, eps);
    MR_CHECK_LAUNCH("mr_layernorm_fwd");
    return MR_OK;
}

// workgroups (= partial rows) of a backward launch: the round-6 kernel (option "ln_impl" = 1, the default; H <= 1024) runs 8 rows per workgroup
// and step, at most 512 workgroups; the round-5 kernel 4 rows, at most 1024
// (H > 1024 keeps the round-5 kernel on the same grid: the count depends on `rows` and the option alone, as the C-ABI's nparts query does)
static int64_t ln_bwd_blocks(int64_t rows) {
    if (mr_opts().ln_impl != 0) { const int64_t n = (rows + 7) / 8; return n > PART_ROWS2 ? PART_ROWS2 : n; }
    const int64_t n = (rows + 3) / 4;
    return n > PART_ROWS ? PART_ROWS : n;
}
extern "C" int64_t mr_layernorm_bwd_nparts(int64_t rows) { return ln_bwd_blocks(rows); }
extern "C" int64_t mr_layernorm_bwd_workspace(int64_t H) { return (int64_t)(PART_ROWS + RED_Y) * 2 * H * sizeof(float); }

extern "C" int mr_layernorm_bwd(const void* dy, int64_t lddy, const void* x, int64_t ldx, const void* gamma, const float* mean,
                                const float* rstd, void* dx, int64_t lddx, const void* dx_add, int64_t ldadd, void* dgamma,
                                void* dbeta, void* partials, int64_t rows, int64_t H, void* stream) {
    MR_CHECK_ARG(dy && x && gamma && mean && rstd && dx && partials, "mr_layernorm_bwd: null pointer");
    MR_CHECK_ARG((dgamma == nullptr) == (dbeta == nullptr), "mr_layernorm_bwd: dgamma and dbeta must both be given or both be NULL");
    MR_CHECK_ARG(rows > 0 && H > 0 && H % 8 == 0 && H <= 64 * 8 * MAXC, "mr_layernorm_bwd: H=%ld unsupported", (long)H);
    MR_CHECK_ARG(lddy % 8 == 0 && ldx % 8 == 0 && lddx % 8 == 0 && ldadd % 8 == 0, "mr_layernorm_bwd: leading dims must be multiples of 8");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t nblk = ln_bwd_blocks(rows);
    if ((mr_opts().ln_impl != 0) && H <= 1024) {
        auto kern = dx_add != nullptr ? ln_bwd2_kernel<2, true> : ln_bwd2_kernel<2, false>;
        hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(512), (size_t)(8 * H * sizeof(float)), s,
                           static_cast<const __bf16*>(dy), lddy, static_cast<const __bf16*>(x), ldx,
                           static_cast<const __bf16*>(gamma), mean, rstd, static_cast<__bf16*>(dx), lddx, static_cast<const __bf16*>(dx_add),
                           ldadd, static_cast<float*>(partials), rows, (int)H);
    } else {
        auto kern = (H <= 1024) ? ln_bwd_kernel<2> : ln_bwd_kernel<MAXC>;
        hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), (size_t)(8 * H * sizeof(float)), s,
                           static_cast<const __bf16*>(dy), lddy, static_cast<const __bf16*>(x), ldx,
                           static_cast<const __bf16*>(gamma), mean, rstd, static_cast<__bf16*>(dx), lddx, static_cast<const __bf16*>(dx_add),
                           ldadd, static_cast<float*>(partials), rows, (int)H);
    }
    if (dgamma != nullptr)      // else: deferred, the caller reduces `partials` with mr_reduce_partials
        launch_reduce(static_cast<const float*>(partials), (int)nblk, (int)(2 * H), (int)H,
                      static_cast<float*>(partials) + (int64_t)PART_ROWS * 2 * H, static_cast<__bf16*>(dgamma), static_cast<__bf16*>(dbeta), s);
    MR_CHECK_LAUNCH("mr_layernorm_bwd");
    return MR_OK;
}

constexpr int COLSUM_STRIPS = 128;
extern "C" int64_t mr_colsum_nparts(int64_t rows) { const int64_t n = (rows + 3) / 4; return n > COLSUM_STRIPS ? COLSUM_STRIPS : n; }
extern "C" int64_t mr_colsum_workspace(int64_t N) { return (int64_t)(COLSUM_STRIPS + RED_Y) * N * sizeof(float); }

extern "C" int mr_colsum(const void* x, int64_t ldx, int64_t rows, int64_t N, void* out, void* partials, void* stream) {
    MR_CHECK_ARG(x && partials, "mr_colsum: null pointer");
    MR_CHECK_ARG(rows > 0 && N > 0 && N % 8 == 0 && ldx % 8 == 0, "mr_colsum: N and ldx must be multiples of 8");
    hipStream_t s = static_cast<hipStream_t>(stream);
    int64_t strips = (rows + 3) / 4;
    if (strips > COLSUM_STRIPS) strips = COLSUM_STRIPS;
    dim3 grid((unsigned)((N / 8 + 63) / 64), (unsigned)strips);
    hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, s, static_cast<const __bf16*>(x), ldx, rows, (int)N,
                       static_cast<float*>(partials));
    if (out != nullptr)         // else: deferred, the caller reduces `partials` with mr_reduce_partials
        launch_reduce(static_cast<const float*>(partials), (int)strips, (int)N, (int)N,
                      static_cast<float*>(partials) + (int64_t)COLSUM_STRIPS * N, static_cast<__bf16*>(out), static_cast<__bf16*>(out), s);
    MR_CHECK_LAUNCH("mr_colsum");
    return MR_OK;
}

extern "C" int mr_reduce_partials(const mr_reduce_job* jobs, int32_t count, void* stream) {
    MR_CHECK_ARG(jobs != nullptr && count >= 1 && count <= MAX_JOBS, "mr_reduce_partials: 1..%d jobs per call (got %d)", MAX_JOBS, (int)count);
    ReduceBatch rb;
    memset(&rb, 0, sizeof(rb));
    rb.count = count;
    int blocks = 0;
    for (int k = 0; k < count; ++k) {
        const mr_reduce_job& q = jobs[k];
        MR_CHECK_ARG(q.partials && q.out0 && (q.out1 || q.split >= q.ncols) && q.nparts > 0 && q.ncols > 0 && q.split >= 0,
                     "mr_reduce_partials: bad job %d", k);
        rb.blk_start[k] = blocks;
        rb.job[k] = q;
        blocks += (q.ncols + RB_COLS - 1) / RB_COLS;
    }
    rb.blk_start[count] = blocks;
    hipLaunchKernelGGL(reduce_batch_kernel, dim3((unsigned)blocks), dim3(1024), 0, static_cast<hipStream_t>(stream), rb);
    MR_CHECK_LAUNCH("mr_reduce_partials");
    return MR_OK;
}
