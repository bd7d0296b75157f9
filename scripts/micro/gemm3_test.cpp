// Standalone check + timing of the NT GEMM paths of libmreserve_hip.so (no torch): every epilogue mode of the ping-pong kernel
// (gemm3.hip) against a naive device reference, and its time beside the one-barrier kernel (gemm256.hip) on the step's shapes.
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/gemm3_test.cpp -o scripts/micro/gemm3_test -Lmerlot_reserve_amd -lmreserve_hip
//   LD_LIBRARY_PATH=merlot_reserve_amd scripts/micro/gemm3_test [check|time|all] [reps]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <string>
#include "../../include/mreserve_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

typedef uint16_t bf16_t;
__host__ __device__ static inline float bf2f(bf16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }
__host__ __device__ static inline bf16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (bf16_t)(u >> 16); }

__global__ void fill_kernel(bf16_t* p, int64_t n, uint32_t seed, float scale, float offset) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u + seed;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        const float u = (float)(x >> 8) * (1.0f / 16777216.0f) * 2.0f - 1.0f;
        p[i] = f2bf(u * scale + offset);
    }
}
__global__ void fill_f32_kernel(float* p, int64_t n, uint32_t seed) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u + seed;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15;
        p[i] = 0.5f + (float)(x >> 8) * (1.0f / 16777216.0f);
    }
}

// reference: fp32 accumulation in k order, the epilogue of include/mreserve_hip.h (mr_gemm) in fp32 / bf16 roundings as documented
__global__ void ref_kernel(const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, int64_t M, int64_t N, int64_t K,
                           const bf16_t* bias, const float* rot, int64_t rot_rows, int64_t rot_cols, int act, const bf16_t* residual,
                           const bf16_t* aux, int64_t ldx, int64_t grp, int64_t gstride, int64_t goff, bf16_t* C, bf16_t* C2, int64_t ldc) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= N || m >= M) return;
    float acc = 0.f;
    const bf16_t* a = A + m * lda;
    const bf16_t* b = B + n * ldb;
    for (int64_t k = 0; k < K; ++k) acc += bf2f(a[k]) * bf2f(b[k]);
    float v = acc + (bias ? bf2f(bias[n]) : 0.f);
    if (rot && n < rot_cols && (n & 63) < 32) v *= rot[(m % rot_rows) * 32 + (n & 63)];
    const int64_t orow = grp > 0 ? (m / grp) * gstride + goff + m % grp : m;
    float o = v;
    if (act == MR_ACT_GELU1702) {
        const float sg = 1.0f / (1.0f + exp2f(-1.702f * 1.4426950408889634f * v));
        o = v * sg;
        if (C2) C2[orow * ldc + n] = f2bf(sg + 1.702f * v * sg * (1.0f - sg));
    }
    bf16_t ob = f2bf(o);
    if (residual) ob = f2bf(bf2f(ob) + bf2f(residual[orow * ldx + n]));
    if (aux) ob = f2bf(bf2f(ob) * bf2f(aux[orow * ldx + n]));
    C[orow * ldc + n] = ob;
}

__global__ void ref_tn_kernel(const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, int64_t M, int64_t N, int64_t K, bf16_t* C, int64_t ldc) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= N || m >= M) return;
    float acc = 0.f;
    for (int64_t k = 0; k < K; ++k) acc += bf2f(A[k * lda + m]) * bf2f(B[k * ldb + n]);
    C[m * ldc + n] = f2bf(acc);
}

__global__ void cmp_kernel(const uint4* a, const uint4* b, long n16, unsigned long long* bad) {
    unsigned long long nb = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) {
        const long j = n16 - 1 - i;          // not the writer's block -> data mapping
        const uint4 x = a[j], y = b[j];
        if (x.x != y.x || x.y != y.y || x.z != y.z || x.w != y.w) ++nb;
    }
    if (nb) atomicAdd(bad, nb);
}

struct Case { int64_t M, N, K; int mode; const char* name; };   // mode 0 bias | 1 rot | 2 gelu + c2 | 3 residual | 4 aux + colsum | 5 plain (no bias) | 6 bias + row map

static bf16_t *dA[3], *dB, *dC[3], *dC2[3], *dX, *dBias, *dRefC, *dRefC2;
static float *dRot, *dCs, *dWs;
static int64_t capA, capB, capC;

static bool g_cmp_ph = false;    // time_variants: [1]/[2] = one-phase, [0]/[3] = two-phase ping-pong (default: [1]/[2] = gemm4, [0]/[3] = gemm3)
static bool g_cmp_old = false;   // time_variants: compare with the one-barrier kernel (NT, NN) instead of the 2-phase ping-pong schedule
static bool g_nn = false;        // time the [K, N] (flax forward) operand layout instead: the one-barrier kernel's NN variants
static void setup_args(mr_gemm_args* g, const Case& c, int set, bool colsum) {
    memset(g, 0, sizeof(*g));
    g->M = c.M; g->N = c.N; g->K = c.K;
    g->A = dA[set]; g->lda = c.K; g->transA = 0;
    g->B = dB; g->ldb = g_nn ? c.N : c.K; g->transB = g_nn ? 0 : 1;
    g->C = dC[set]; g->ldc = c.N; g->c_dtype = MR_DT_BF16;
    g->act = MR_ACT_NONE;
    if (c.mode != 5) g->bias = dBias;
    if (c.mode == 1) { g->rot_tab = dRot; g->rot_rows = 241 * 16; g->rot_cols = (c.N / 3) * 2; }
    if (c.mode == 2) { g->c2 = dC2[set]; g->act = MR_ACT_GELU1702; }
    if (c.mode == 3) { g->residual = dX; g->ldr = c.N; g->bias = nullptr; }
    if (c.mode == 4) { g->aux = dX; g->ldaux = c.N; g->bias = nullptr; if (colsum) { g->colsum = dCs; g->ldcs = c.N; } }
    if (c.mode == 6) { g->out_grp = 240; g->out_grp_stride = 241; g->out_grp_off = 1; }
    g->workspace = dWs; g->workspace_bytes = 64LL << 20;
}

// Sustained, interleaved timing: the variants take turns (one launch each, round robin) for ~`budget_ms` of GPU time after a
// warm-up of the same length, so that they share the clock the chip settles at under load (a cold 3-ms burst runs ~15 % faster
// than the same kernel inside a training step).  variants: 1 = ping-pong 256, 2 = ping-pong 192 (one phase per k-tile), and
// 0 / 3 = the two-phase schedule at 256 / 192 -- or, with G3_CMP_OLD=1, the one-barrier kernel NT / NN (flax forward layout).
static void time_variants(const Case& c, bool colsum, double budget_ms, double out_us[4]) {
    mr_gemm_args g;
    hipEvent_t e0[4], e1[4];
    for (int v = 0; v < 4; ++v) { CK(hipEventCreate(&e0[v])); CK(hipEventCreate(&e1[v])); }
    auto launch = [&](int v, int set) {
        if (g_cmp_old) { g_nn = v == 3; mr_set_option("gemm3", v == 1 ? 256 : v == 2 ? 192 : 0); }
        else if (g_cmp_ph) { mr_set_option("gemm3", (v == 0 || v == 1) ? 256 : 192); mr_set_option("gemm3_phases", (v == 1 || v == 2) ? 1 : 2); }
        else { mr_set_option("gemm3", (v == 0 || v == 1) ? 256 : 192); mr_set_option("gemm4", (v == 1 || v == 2) ? 1 : 0); }
        setup_args(&g, c, set, colsum);
        if (mr_gemm(&g, nullptr) != 0) { printf("mr_gemm failed: %s\n", mr_last_error()); exit(3); }
    };
    double tot[4] = {0, 0, 0, 0}; int n[4] = {0, 0, 0, 0};
    for (int phase = 0; phase < 2; ++phase) {           // phase 0 = warm-up (discarded)
        double spent = 0; int it = 0;
        while (spent < budget_ms) {
            for (int v = 0; v < 4; ++v) {
                CK(hipEventRecord(e0[v], nullptr));
                for (int r = 0; r < 4; ++r) launch(v, (it + r) % 3);
                CK(hipEventRecord(e1[v], nullptr));
            }
            CK(hipEventSynchronize(e1[3]));
            for (int v = 0; v < 4; ++v) { float ms; CK(hipEventElapsedTime(&ms, e0[v], e1[v])); spent += ms; if (phase) { tot[v] += ms; n[v] += 4; } }
            ++it;
        }
    }
    for (int v = 0; v < 4; ++v) out_us[v] = tot[v] * 1000.0 / n[v];
    g_nn = false;
    mr_set_option("gemm3", 1);
    mr_set_option("gemm3_phases", 0);
    mr_set_option("gemm4", -1);
}


// gemm5 (two workgroups per CU): [0] = the shipped default path, [1..3] = gemm5 with stagger mode 0 / 1 / 2
static void time_g5(const Case& c, bool colsum, double budget_ms, double out_us[4]) {
    mr_gemm_args g;
    hipEvent_t e0[4], e1[4];
    for (int v = 0; v < 4; ++v) { CK(hipEventCreate(&e0[v])); CK(hipEventCreate(&e1[v])); }
    auto launch = [&](int v, int set) {
        // columns 1..3 = "gemm5" option values, G5_COLS (default 1,3,-1: 256 x 128 two per CU | 128 x 128 deep ring | the default policy)
        static int cols[4] = {0, 1, 3, -1};
        static bool init = false;
        if (!init) { init = true; if (const char* e = getenv("G5_COLS")) sscanf(e, "%d,%d,%d", &cols[1], &cols[2], &cols[3]); }
        mr_set_option("gemm5", cols[v]);
        mr_set_option("gemm5_stagger", -1);
        { const char* e = getenv("G5_CUS"); mr_set_option("gemm_cus", (v && e) ? atoi(e) : 0); }      // G5_CUS=128: 256 workgroups = one per CU
        setup_args(&g, c, set, colsum);
        if (mr_gemm(&g, nullptr) != 0) { printf("mr_gemm failed: %s\n", mr_last_error()); exit(3); }
    };
    double tot[4] = {0, 0, 0, 0}; int n[4] = {0, 0, 0, 0};
    for (int phase = 0; phase < 2; ++phase) {
        double spent = 0; int it = 0;
        while (spent < budget_ms) {
            for (int v = 0; v < 4; ++v) {
                CK(hipEventRecord(e0[v], nullptr));
                for (int r = 0; r < 4; ++r) launch(v, (it + r) % 3);
                CK(hipEventRecord(e1[v], nullptr));
            }
            CK(hipEventSynchronize(e1[3]));
            for (int v = 0; v < 4; ++v) { float ms; CK(hipEventElapsedTime(&ms, e0[v], e1[v])); spent += ms; if (phase) { tot[v] += ms; n[v] += 4; } }
            ++it;
        }
    }
    for (int v = 0; v < 4; ++v) out_us[v] = tot[v] * 1000.0 / n[v];
    mr_set_option("gemm5", -1);
    mr_set_option("gemm5_stagger", -1);
    mr_set_option("gemm_cus", 0);
}

static int check_case(const Case& c, const char* label) {
    const int64_t rows_out = c.mode == 6 ? (c.M / 240 + 1) * 241 : c.M;
    if (rows_out * c.N > capC) { printf("  (skip check: output too large)\n"); return 0; }
    mr_gemm_args g;
    setup_args(&g, c, 0, true);
    CK(hipMemset(dC[0], 0xFF, rows_out * c.N * 2)); CK(hipMemset(dRefC, 0xFF, rows_out * c.N * 2));
    if (c.mode == 2) { CK(hipMemset(dC2[0], 0xFF, rows_out * c.N * 2)); CK(hipMemset(dRefC2, 0xFF, rows_out * c.N * 2)); }
    if (mr_gemm(&g, nullptr) != 0) { printf("mr_gemm failed: %s\n", mr_last_error()); return 1; }
    dim3 grid((unsigned)((c.N + 255) / 256), (unsigned)c.M);
    hipLaunchKernelGGL(ref_kernel, grid, dim3(256), 0, nullptr, dA[0], c.K, dB, c.K, c.M, c.N, c.K, (const bf16_t*)g.bias, g.rot_tab, g.rot_rows, g.rot_cols,
                       g.act, (const bf16_t*)g.residual, (const bf16_t*)g.aux, c.N, g.out_grp, g.out_grp_stride, g.out_grp_off, dRefC,
                       c.mode == 2 ? dRefC2 : nullptr, c.N);
    CK(hipDeviceSynchronize());
    std::vector<bf16_t> got(rows_out * c.N), ref(rows_out * c.N);
    int bad = 0;
    double maxerr = 0;
    for (int pass = 0; pass < (c.mode == 2 ? 2 : 1); ++pass) {
        CK(hipMemcpy(got.data(), pass ? dC2[0] : dC[0], got.size() * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(ref.data(), pass ? dRefC2 : dRefC, ref.size() * 2, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < got.size(); ++i) {
            if (got[i] == ref[i]) continue;                        // (also the untouched 0xFFFF rows of a row map)
            const float a = bf2f(got[i]), b = bf2f(ref[i]);
            const double err = fabs((double)a - b), tol = 0.01 * fabs((double)b) + (c.mode == 3 ? 0.017 : 0.004);   // residual: the 1-ulp spread of bf16(acc) survives the add
            if (!(err <= tol)) { if (bad < 5) printf("    MISMATCH %s[%zu] (row %zu col %zu): got %g ref %g\n", pass ? "c2" : "C", i, i / c.N, i % c.N, a, b); ++bad; }
            if (err > maxerr) maxerr = err;
        }
    }
    if (c.mode == 4) {        // column sums per 64-row band of the stored output
        const int64_t nrows = mr_gemm_colsum_rows(c.M);
        std::vector<float> cs(nrows * c.N);
        CK(hipMemcpy(cs.data(), dCs, cs.size() * 4, hipMemcpyDeviceToHost));
        for (int64_t band = 0; band < nrows; ++band)
            for (int64_t n = 0; n < c.N; n += 37) {
                double s = 0;
                for (int64_t m = band * 64; m < band * 64 + 64 && m < c.M; ++m) s += bf2f(got[m * c.N + n]);
                if (fabs(s - cs[band * c.N + n]) > 1e-3 * (1 + fabs(s))) { if (bad < 5) printf("    COLSUM mismatch band %ld col %ld: got %g ref %g\n", (long)band, (long)n, cs[band * c.N + n], s); ++bad; }
            }
    }
    printf("  check %-10s %-28s M=%ld N=%ld K=%ld: %s (max abs err %.4g, %d bad)\n", label, c.name, (long)c.M, (long)c.N, (long)c.K, bad ? "FAIL" : "ok", maxerr, bad);
    return bad != 0;
}

int main(int argc, char** argv) {
    const char* what = argc > 1 ? argv[1] : "all";
    g_cmp_old = getenv("G3_CMP_OLD") != nullptr;
    g_cmp_ph = getenv("G3_CMP_PH") != nullptr;
    const int reps = argc > 2 ? atoi(argv[2]) : 12;
    capA = 15424LL * 4096; capB = 8192LL * 8192; capC = 15488LL * 4096;
    if (capA < 8192LL * 8192) capA = 8192LL * 8192;
    if (capC < 8192LL * 8192) capC = 8192LL * 8192;
    for (int s = 0; s < 3; ++s) { CK(hipMalloc(&dA[s], capA * 2)); CK(hipMalloc(&dC[s], capC * 2)); CK(hipMalloc(&dC2[s], capC * 2)); }
    CK(hipMalloc(&dB, capB * 2)); CK(hipMalloc(&dX, capC * 2)); CK(hipMalloc(&dBias, 8192 * 2)); CK(hipMalloc(&dRefC, capC * 2)); CK(hipMalloc(&dRefC2, capC * 2));
    CK(hipMalloc(&dRot, 241 * 16 * 32 * 4)); CK(hipMalloc(&dCs, 4 * 64 * 8192 * 4)); CK(hipMalloc(&dWs, 64LL << 20));
    for (int s = 0; s < 3; ++s) hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, nullptr, dA[s], capA, 1234u + s, 1.0f, 0.0f);
    hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, nullptr, dB, capB, 99u, 0.03f, 0.0f);
    hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, nullptr, dX, capC, 7u, 1.0f, 0.0f);
    hipLaunchKernelGGL(fill_kernel, dim3(8), dim3(256), 0, nullptr, dBias, (int64_t)8192, 5u, 0.5f, 0.0f);
    hipLaunchKernelGGL(fill_f32_kernel, dim3(64), dim3(256), 0, nullptr, dRot, (int64_t)241 * 16 * 32, 3u);
    CK(hipDeviceSynchronize());

    int fails = 0;
    if (!strcmp(what, "check") || !strcmp(what, "all")) {
        const Case checks[] = {
            {15424, 3072, 768, 2, "fc1 fwd gelu+c2"}, {15424, 3072, 768, 4, "fc1 dgrad aux+colsum"}, {15424, 768, 3072, 3, "fc2 fwd residual"},
            {15424, 768, 3072, 5, "d ln2 plain"}, {15424, 2304, 768, 1, "qkv fwd bias+rot"}, {15424, 768, 2304, 0, "bias"},
            {15360, 768, 768, 6, "patch embed row map"}, {5952, 3072, 768, 2, "audio fc1"}, {5952, 768, 768, 3, "audio proj"},
            {3072, 2304, 768, 1, "span qkv"}, {4616, 4096, 1024, 4, "ragged M aux"}, {2000, 1000, 192, 0, "ragged M N"},
            {1024, 256, 128, 3, "small"}, {1312, 3072, 64, 0, "one k-tile"}, {40000, 192, 128, 5, "two k-tiles, narrow"},
        };
        const int widths[] = {256, 192};
        for (int gen = 4; gen >= 3; --gen)
            for (int w : widths) {
                mr_set_option("gemm3", w);
                mr_set_option("gemm4", gen == 4);
                char label[32]; snprintf(label, sizeof label, "g%d/%d", gen, w);
                for (const Case& c : checks) fails += check_case(c, label);
            }
        mr_set_option("gemm4", -1);
        mr_set_option("gemm3", 0);
        fails += check_case(checks[0], "old");       // the harness itself against the shipped kernel
        fails += check_case(checks[4], "old");
        mr_set_option("gemm3", 1);
    }
    if (!strcmp(what, "time") || !strcmp(what, "all")) {
        const Case shapes[] = {
            {15424, 3072, 768, 2, "fc1 fwd gelu+c2"}, {15424, 3072, 768, 4, "fc1 dgrad aux+colsum"}, {15424, 3072, 768, 5, "fc1 plain"},
            {15424, 768, 3072, 3, "fc2 fwd residual"}, {15424, 768, 3072, 5, "d ln2 plain"}, {15424, 2304, 768, 1, "qkv fwd bias+rot"},
            {15424, 2304, 768, 5, "qkv plain"}, {15424, 768, 2304, 5, "d ln1 plain"}, {15424, 768, 768, 3, "proj fwd residual"}, {15424, 768, 768, 5, "d att plain"},
            {5952, 3072, 768, 2, "audio fc1 fwd"}, {5952, 768, 3072, 3, "audio fc2 fwd"}, {5952, 2304, 768, 1, "audio qkv"}, {5952, 768, 768, 3, "audio proj"},
            {3072, 3072, 768, 2, "span fc1 fwd"}, {3072, 768, 3072, 3, "span fc2 fwd"},
            {15424, 4096, 1024, 2, "large fc1 fwd"}, {15424, 4096, 1024, 5, "large fc1 plain"}, {15424, 1024, 4096, 3, "large fc2 fwd"}, {15424, 1024, 4096, 5, "large fc2 plain"},
            {15424, 3072, 1024, 1, "large qkv"}, {15424, 1024, 1024, 3, "large proj"}, {8192, 8192, 8192, 5, "8192^3"},
        };
        printf("%-26s %18s | %9s %9s %9s %9s | TF/s: [0], [3] -> best new   (sustained, interleaved; [0]/[3] = %s)\n", "shape", "M x N x K", "[0] us", "[3] us", "g3/256 us", "g3/192 us",
               g_cmp_old ? "one-barrier kernel NT / NN" : g_cmp_ph ? "two-phase ping-pong 256 / 192" : "ping-pong (gemm3) 256 / 192; g3/ columns = gemm4");
        for (const Case& c : shapes) {
            double t[4];
            time_variants(c, c.mode == 4, reps * 10.0, t);
            const double fl = 2.0 * c.M * c.N * c.K, tb = t[1] < t[2] ? t[1] : t[2];
            printf("%-26s %6ldx%5ldx%5ld | %9.1f %9.1f %9.1f %9.1f | %7.0f %7.0f -> %7.0f\n", c.name, (long)c.M, (long)c.N, (long)c.K, t[0], t[3], t[1], t[2],
                   fl / t[0] * 1e-6, fl / t[3] * 1e-6, fl / tb * 1e-6);
        }
        mr_set_option("gemm3", 1);
    }

    if (!strcmp(what, "g5check") || !strcmp(what, "g5")) {
        const Case checks[] = {
            {15424, 3072, 768, 2, "fc1 fwd gelu+c2"}, {15424, 3072, 768, 4, "fc1 dgrad aux+colsum"}, {15424, 768, 3072, 3, "fc2 fwd residual"},
            {15424, 768, 3072, 5, "d ln2 plain"}, {15424, 2304, 768, 1, "qkv fwd bias+rot"}, {15424, 768, 2304, 0, "bias"},
            {15360, 768, 768, 6, "patch embed row map"}, {5952, 3072, 768, 2, "audio fc1"}, {5952, 768, 768, 3, "audio proj"},
            {3072, 2304, 768, 1, "span qkv"}, {4616, 4096, 1024, 4, "ragged M aux"}, {2000, 1000, 192, 0, "ragged M N"},
            {1024, 256, 128, 3, "small"}, {1312, 3072, 128, 0, "four steps"}, {40000, 192, 128, 5, "four steps, narrow"}, {300, 192, 256, 1, "tiny ragged rot"},
        };
        mr_set_option("gemm5", getenv("G5_R128") ? atoi(getenv("G5_R128")) : 1);      // G5_R128=3: the 128 x 128 geometry
        for (int st = 0; st < (getenv("G5_R128") ? 1 : 3); ++st) {
            mr_set_option("gemm5_stagger", st);
            char label[32]; snprintf(label, sizeof label, "g5/st%d", st);
            for (const Case& c : checks) fails += check_case(c, label);
        }
        mr_set_option("gemm5", -1);
        mr_set_option("gemm5_stagger", -1);
    }
    if (!strcmp(what, "g5time") || !strcmp(what, "g5")) {
        const Case shapes[] = {
            {15424, 3072, 768, 2, "fc1 fwd gelu+c2"}, {15424, 3072, 768, 4, "fc1 dgrad aux+colsum"}, {15424, 3072, 768, 5, "fc1 plain"},
            {15424, 2304, 768, 1, "qkv fwd bias+rot"}, {15424, 768, 3072, 3, "fc2 fwd residual"}, {15424, 768, 3072, 5, "d ln2 plain"},
            {15424, 768, 2304, 5, "d ln1 plain"}, {15424, 768, 768, 3, "proj fwd residual"}, {15424, 768, 768, 5, "d att plain"},
            {5952, 3072, 768, 2, "audio fc1 fwd"}, {5952, 3072, 768, 4, "audio fc1 dgrad"}, {5952, 768, 3072, 3, "audio fc2 fwd"}, {5952, 2304, 768, 1, "audio qkv"}, {5952, 768, 768, 3, "audio proj"},
            {3072, 3072, 768, 2, "span fc1 fwd"}, {3072, 768, 3072, 3, "span fc2 fwd"}, {3072, 2304, 768, 1, "span qkv"}, {3072, 768, 768, 3, "span proj"},
            {2308, 4096, 1024, 2, "vcr vit fc1"}, {2308, 1024, 4096, 3, "vcr vit fc2"}, {2308, 3072, 1024, 1, "vcr vit qkv"},
            {2308, 1024, 4096, 5, "vcr vit d ln2"}, {2308, 1024, 3072, 5, "vcr vit d ln1"}, {2308, 1024, 1024, 3, "vcr vit proj"}, {2308, 4096, 1024, 4, "vcr vit fc1 dgrad"},
            {5952, 768, 3072, 5, "audio d ln2"}, {5952, 768, 2304, 5, "audio d ln1"}, {5952, 768, 768, 5, "audio d att"}, {3072, 768, 3072, 5, "span d ln2"}, {3072, 3072, 768, 4, "span fc1 dgrad"},
            {15424, 4096, 1024, 2, "large fc1 fwd"}, {15424, 4096, 1024, 4, "large fc1 dgrad"}, {15424, 1024, 4096, 3, "large fc2 fwd"},
            {15424, 3072, 1024, 1, "large qkv"}, {15424, 1024, 1024, 3, "large proj"}, {8192, 8192, 8192, 5, "8192^3"},
        };
        printf("%-26s %18s | %9s %9s %9s %9s | TF/s default -> best g5   (sustained, interleaved; [0] gemm5 off (ping-pong / one-wave / one-barrier kernels), then the gemm5 option values of G5_COLS, default 1,3,-1 = 256 x 128 two per CU | 128 x 128 deep ring | the default policy)\n", "shape", "M x N x K", "g5 off", "col1", "col2", "col3");
        const char* sel = getenv("G5_SHAPES");          // e.g. "0,2,26": only these rows of the table
        int idx = -1;
        for (const Case& c : shapes) {
            ++idx;
            if (sel) { char key[16]; snprintf(key, sizeof key, ",%d,", idx); std::string ss = std::string(",") + sel + ","; if (ss.find(key) == std::string::npos) continue; }
            double t[4];
            time_g5(c, c.mode == 4, reps * 10.0, t);
            const double fl = 2.0 * c.M * c.N * c.K;
            double tb = t[1]; if (t[2] < tb) tb = t[2]; if (t[3] < tb) tb = t[3];
            printf("%-26s %6ldx%5ldx%5ld | %9.1f %9.1f %9.1f %9.1f | %7.0f -> %7.0f\n", c.name, (long)c.M, (long)c.N, (long)c.K, t[0], t[1], t[2], t[3], fl / t[0] * 1e-6, fl / tb * 1e-6);
            fflush(stdout);
        }
    }

    if (!strcmp(what, "g5stamps")) {
        // in-kernel timeline of the two workgroups of a CU (library built with -DMR_G5_STAMPS): per tile, k-loop start / end, epilogue end
        const int mode = argc > 2 ? atoi(argv[2]) : 2, st = argc > 3 ? atoi(argv[3]) : 1;
        const Case c = {15424, 3072, 768, mode, "stamps"};
        mr_gemm_args g;
        mr_set_option("gemm5", 1);
        mr_set_option("gemm5_stagger", st);
        for (int r = 0; r < 20; ++r) { setup_args(&g, c, r % 3, mode == 4); mr_gemm(&g, nullptr); }
        CK(hipDeviceSynchronize());
        CK(hipMemset(dWs, 0, 512 * 8 * 4 * 8));
        setup_args(&g, c, 0, mode == 4);
        mr_gemm(&g, nullptr);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(512 * 8 * 4);
        CK(hipMemcpy(h.data(), dWs, h.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull;
        for (int b = 0; b < 512; ++b) if (h[b * 32] && h[b * 32] < t0) t0 = h[b * 32];
        printf("mode %d stagger %d: blocks b and b + 256 share a CU; cycles from the first k-loop start\n", mode, st);
        for (int b : {0, 1, 8, 100, 255}) {
            for (int w = 0; w < 2; ++w) {
                const int bb = b + 256 * w;
                printf("  block %3d:", bb);
                for (int q = 0; q < 4; ++q) {
                    const unsigned long long* e = &h[(bb * 8 + q) * 4];
                    if (!e[0]) break;
                    printf("  [k %6llu-%6llu e -%6llu]", e[0] - t0, e[1] - t0, e[2] - t0);
                }
                printf("\n");
            }
        }
        mr_set_option("gemm5", -1); mr_set_option("gemm5_stagger", -1);
    }
    if (!strcmp(what, "tn") || !strcmp(what, "all")) {
        // weight gradients of a layer (x 2 layers): A = activations [K = tokens, in], B = upstream gradients [K, out]
        struct WG { int64_t M, N; };
        const int64_t H = 768, Ks[] = {15424, 5952, 1000};
        for (int64_t K : Ks) {
            const WG one[4] = {{4 * H, H}, {H, 4 * H}, {H, H}, {H, 3 * H}};
            mr_gemm_args list[8];
            bf16_t* outs[8];
            int64_t coff = 0;
            for (int k = 0; k < 8; ++k) {
                const WG w = one[k % 4];
                mr_gemm_args* g = &list[k];
                memset(g, 0, sizeof(*g));
                g->M = w.M; g->N = w.N; g->K = K;
                g->A = dA[k / 4] + (k % 4) * 1000; g->lda = w.M; g->transA = 1;         // distinct (overlapping, read-only) operands
                g->B = dX + (k % 4) * 3000; g->ldb = w.N; g->transB = 0;
                outs[k] = dC[0] + coff; coff += w.M * w.N;
                g->C = outs[k]; g->ldc = w.N; g->c_dtype = MR_DT_BF16;
                g->workspace = dWs; g->workspace_bytes = 64LL << 20;
            }
            // check every problem against the naive reference
            CK(hipMemset(dC[0], 0xFF, coff * 2));
            mr_set_option("gemm3", 256);
            if (mr_gemm_grouped(list, 8, nullptr) != 0) { printf("mr_gemm_grouped failed: %s\n", mr_last_error()); return 1; }
            int bad = 0; double maxerr = 0;
            for (int k = 0; k < 8; ++k) {
                const mr_gemm_args& g = list[k];
                dim3 grid((unsigned)((g.N + 255) / 256), (unsigned)g.M);
                hipLaunchKernelGGL(ref_tn_kernel, grid, dim3(256), 0, nullptr, (const bf16_t*)g.A, g.lda, (const bf16_t*)g.B, g.ldb, g.M, g.N, g.K, dRefC, g.N);
                CK(hipDeviceSynchronize());
                std::vector<bf16_t> got(g.M * g.N), ref(g.M * g.N);
                CK(hipMemcpy(got.data(), g.C, got.size() * 2, hipMemcpyDeviceToHost));
                CK(hipMemcpy(ref.data(), dRefC, ref.size() * 2, hipMemcpyDeviceToHost));
                for (size_t i = 0; i < got.size(); ++i) {
                    const float a = bf2f(got[i]), b = bf2f(ref[i]);
                    const double err = fabs((double)a - b), tol = 0.01 * fabs((double)b) + 0.02 * sqrt((double)K / 1000.0);
                    if (!(err <= tol)) { if (bad < 5) printf("    TN MISMATCH problem %d [%zu]: got %g ref %g\n", k, i, a, b); ++bad; }
                    if (err > maxerr) maxerr = err;
                }
            }
            printf("  check TN grouped x8, K=%ld: %s (max abs err %.4g, %d bad)\n", (long)K, bad ? "FAIL" : "ok", maxerr, bad);
            fails += bad != 0;
            // time: one-barrier kernel (two launches of 4) vs ping-pong (one launch of 8), sustained + interleaved
            hipEvent_t e0[2], e1[2];
            for (int v = 0; v < 2; ++v) { CK(hipEventCreate(&e0[v])); CK(hipEventCreate(&e1[v])); }
            double tot[2] = {0, 0}; int n[2] = {0, 0};
            for (int phase = 0; phase < 2; ++phase) {
                double spent = 0;
                while (spent < 150.0) {
                    for (int v = 0; v < 2; ++v) {
                        mr_set_option("gemm3", v ? 256 : 0);
                        CK(hipEventRecord(e0[v], nullptr));
                        for (int r = 0; r < 3; ++r) mr_gemm_grouped(list, 8, nullptr);
                        CK(hipEventRecord(e1[v], nullptr));
                    }
                    CK(hipEventSynchronize(e1[1]));
                    for (int v = 0; v < 2; ++v) { float ms; CK(hipEventElapsedTime(&ms, e0[v], e1[v])); spent += ms; if (phase) { tot[v] += ms; n[v] += 3; } }
                }
            }
            const double fl = 2.0 * K * 2 * (4 * H * H * 2 + H * H + 3 * H * H);
            printf("  time  TN grouped x8, K=%ld: one-barrier %.1f us (%.0f TF/s)   ping-pong %.1f us (%.0f TF/s)\n", (long)K, tot[0] * 1e3 / n[0], fl / (tot[0] / n[0]) * 1e-9,
                   tot[1] * 1e3 / n[1], fl / (tot[1] / n[1]) * 1e-9);
        }
        mr_set_option("gemm3", 1);
    }
    if (!strcmp(what, "race")) {
        // A consumer kernel right behind the GEMM in the SAME stream must see every store of the GEMM, also while another stream's
        // GEMMs are resident: step t computes C = A[t % 2] . B into the same C buffer and a checker compares it with the reference
        // output of A[t % 2] (computed once by the same kernel, alone on the GPU).
        const Case c = {15424, 768, 3072, 3, "fc2 fwd residual"};
        const int variants[] = {192, 256, 0};
        unsigned long long* dBad; CK(hipMalloc(&dBad, 8));
        hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
        for (int v : variants) for (int other = 0; other < 2; ++other) {
            mr_set_option("gemm3", v);
            mr_gemm_args g, g2;
            const long n16 = c.M * c.N * 2 / 16;
            for (int set = 0; set < 2; ++set) {          // references -> dC2[set]
                setup_args(&g, c, set, false); g.C = dC2[set]; mr_gemm(&g, nullptr);
            }
            CK(hipDeviceSynchronize());
            CK(hipMemset(dBad, 0, 8));
            const Case o = {5952, 3072, 768, 2, "other"};
            for (int t = 0; t < 400; ++t) {
                if (other) { setup_args(&g2, o, 2, false); mr_gemm(&g2, s2); }
                setup_args(&g, c, t % 2, false); g.C = dC[0];
                mr_gemm(&g, s1);
                hipLaunchKernelGGL(cmp_kernel, dim3(512), dim3(256), 0, s1, (const uint4*)dC[0], (const uint4*)dC2[t % 2], n16, dBad);
            }
            CK(hipDeviceSynchronize());
            unsigned long long hb; CK(hipMemcpy(&hb, dBad, 8, hipMemcpyDeviceToHost));
            printf("race: kernel %s, other stream %d: %llu stale 16-byte words in 400 steps\n", v == 0 ? "one-barrier" : v == 192 ? "ping-pong/192" : "ping-pong/256", other, hb);
        }
        mr_set_option("gemm3", 1);
    }
    if (!strcmp(what, "quick")) {
        const Case shapes[] = {{15424, 3072, 768, 2, "fc1 fwd gelu+c2"}, {15424, 3072, 768, 4, "fc1 dgrad aux+colsum"}, {15424, 3072, 768, 5, "fc1 plain"},
                               {15424, 768, 3072, 3, "fc2 fwd residual"}, {15424, 2304, 768, 1, "qkv fwd bias+rot"}, {15424, 768, 768, 3, "proj fwd residual"}};
        for (const Case& c : shapes) {
            double t[4];
            time_variants(c, c.mode == 4, reps * 10.0, t);
            printf("quick %-24s g3/256 %7.1f us   g3/192 %7.1f us   | [0] %7.1f us  [3] %7.1f us\n", c.name, t[1], t[2], t[0], t[3]);
        }
    }
    if (!strcmp(what, "kloop")) {        // plain products only: the k-loop (and the plain epilogue) of the two generations
        const Case shapes[] = {{8192, 8192, 8192, 5, "8192^3"}, {15424, 768, 3072, 5, "d ln2 plain"}, {15424, 3072, 768, 5, "fc1 plain"}};
        for (const Case& c : shapes) {
            double t[4];
            time_variants(c, false, reps * 10.0, t);
            printf("kloop %-16s g4/256 %7.1f us   g4/192 %7.1f us   | g3/256 %7.1f us  g3/192 %7.1f us\n", c.name, t[1], t[2], t[0], t[3]);
        }
    }
    if (!strcmp(what, "stamps")) {       // needs the MR_G3_STAMPS build of the library
        const Case shapes[] = {{15424, 3072, 768, 5, "fc1 plain"}, {15424, 3072, 768, 2, "fc1 fwd gelu+c2"}, {15424, 768, 3072, 3, "fc2 fwd residual"}, {8192, 8192, 8192, 5, "8192^3"}};
        const int widths[] = {256, 192};
        for (const Case& c : shapes) for (int w : widths) {
            mr_set_option("gemm3", w);
            mr_gemm_args g;
            for (int r = 0; r < 3; ++r) { setup_args(&g, c, r, false); mr_gemm(&g, nullptr); }
            CK(hipMemset(dWs, 0, 2 << 20));
            setup_args(&g, c, 0, false);
            mr_gemm(&g, nullptr);
            CK(hipDeviceSynchronize());
            std::vector<unsigned long long> st(256 * 2 * 8 * 4);
            CK(hipMemcpy(st.data(), dWs, st.size() * 8, hipMemcpyDeviceToHost));
            const int nkt = (int)(c.K / 64);
            double kl[2] = {0, 0}, ep[2] = {0, 0}, gap[2] = {0, 0}, span = 0; int n[2] = {0, 0}, ng[2] = {0, 0}, nb = 0;
            for (int b = 0; b < 256; ++b) for (int wr = 0; wr < 2; ++wr) {
                const unsigned long long* q = &st[((b * 2 + wr) * 8) * 4];
                int last = -1;
                for (int i = 0; i < 8 && q[i * 4]; ++i) {
                    kl[wr] += (double)(q[i * 4 + 1] - q[i * 4]); ep[wr] += (double)(q[i * 4 + 2] - q[i * 4 + 1]); ++n[wr];
                    if (i > 0) { gap[wr] += (double)(q[i * 4] - q[(i - 1) * 4 + 2]); ++ng[wr]; }
                    last = i;
                }
                if (wr == 0 && last >= 0) { span += (double)(q[last * 4 + 2] - q[0]); ++nb; }
            }
            if (w == 256) {
                std::vector<unsigned long long> ks(8 * 4 * 16);
                CK(hipMemcpy(ks.data(), (unsigned long long*)dWs + 65536, ks.size() * 8, hipMemcpyDeviceToHost));
                for (int b = 0; b < 2; ++b) for (int i = 0; i < 3; ++i) {
                    printf("   k-tile cycles, block %d tile %d:", b, i);
                    for (int t = 1; t < 16 && t < nkt && ks[(b * 4 + i) * 16 + t]; ++t) printf(" %5llu", ks[(b * 4 + i) * 16 + t] - ks[(b * 4 + i) * 16 + t - 1]);
                    printf("\n");
                }
            }
            printf("stamps %-18s g3/%d: per tile  k-loop %8.0f / %8.0f cyc (%5.0f per k-tile)  epilogue %7.0f / %7.0f  gap %6.0f / %6.0f   (group 0 / 1; s_memtime ticks = 10 ns);  block span %8.0f\n",
                   c.name, w, kl[0] / n[0], kl[1] / n[1], kl[0] / n[0] / nkt, ep[0] / n[0], ep[1] / n[1], ng[0] ? gap[0] / ng[0] : 0., ng[1] ? gap[1] / ng[1] : 0., span / nb);
        }
        mr_set_option("gemm3", 1);
    }
    printf(fails ? "FAILED (%d)\n" : "ALL OK\n", fails);
    return fails ? 1 : 0;
}
