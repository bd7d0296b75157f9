// Store-pattern microbenchmark (gfx950): how fast can 256 workgroups write 256x128 bf16 tiles of a [M, N] matrix when
// a wave-instruction writes (A) 16 rows x 64 B  (the register epilogue of gemm256), (B) 8 rows x 128 B (full lines),
// (C) 4 rows x 256 B?   hipcc --offload-arch=gfx950 -O3 -o store_bench store_bench.cpp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int PAT>
__global__ __launch_bounds__(512) void k(unsigned short* C, int M, int N, int tiles_m, int tiles_n, int delay) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int g = lane >> 4, li = lane & 15;
    const u32x4 v = {1u, 2u, 3u, (unsigned)tid};
    for (int t = blockIdx.x; t < tiles_m * tiles_n; t += gridDim.x) {
        const int m0 = (t / tiles_n) * 256, n0 = (t % tiles_n) * 128;
        if (PAT == 0) {
            for (int i = 0; i < 4; ++i)
                for (int jp = 0; jp < 2; ++jp) {
                    const int row = m0 + wm * 64 + i * 16 + li, col = n0 + wn * 64 + (2 * jp + (g & 1)) * 16 + (g >> 1) * 8;
                    if (row < M) *reinterpret_cast<u32x4*>(C + (size_t)row * N + col) = v;
                }
        } else if (PAT == 1) {
            for (int q = 0; q < 8; ++q) {
                const int row = m0 + wm * 64 + q * 8 + (lane >> 3), col = n0 + wn * 64 + (lane & 7) * 8;
                if (row < M) *reinterpret_cast<u32x4*>(C + (size_t)row * N + col) = v;
            }
        } else {   // 4 rows x 256 B: the two wn waves of a row group split rows instead of columns
            for (int q = 0; q < 8; ++q) {
                const int row = m0 + wm * 64 + wn * 32 + q * 4 + (lane >> 4), col = n0 + (lane & 15) * 8;
                if (row < M) *reinterpret_cast<u32x4*>(C + (size_t)row * N + col) = v;
            }
        }
        // stand-in for the k-loop between two epilogues
        if (delay) __builtin_amdgcn_s_sleep(0);
        for (int d = 0; d < delay; ++d) __builtin_amdgcn_s_sleep(127);
    }
}

int main(int argc, char** argv) {
    const int M = argc > 5 ? atoi(argv[5]) : 15424, N = argc > 1 ? atoi(argv[1]) : 3072, delay = argc > 2 ? atoi(argv[2]) : 0, grid = argc > 3 ? atoi(argv[3]) : 256;
    const int Mw = argc > 4 ? atoi(argv[4]) : M;          // rows actually written (fewer tiles for a small grid)
    unsigned short* C;
    hipMalloc(&C, (size_t)M * N * 2);
    const int tm = (Mw + 255) / 256, tn = N / 128;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pat = 0; pat < 3; ++pat) {
        float best = 1e9;
        for (int r = 0; r < 5; ++r) {
            hipEventRecord(e0);
            if (pat == 0) k<0><<<grid, 512>>>(C, M, N, tm, tn, delay);
            if (pat == 1) k<1><<<grid, 512>>>(C, M, N, tm, tn, delay);
            if (pat == 2) k<2><<<grid, 512>>>(C, M, N, tm, tn, delay);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("N=%d delay=%d grid=%d rows=%d pattern %d: %.1f us  %.2f TB/s  (%.1f B/clk/CU at 2.1 GHz)\n", N, delay, grid, Mw, pat, best * 1e3, (double)Mw * N * 2 / (best * 1e-3) / 1e12,
               (double)Mw * N * 2 / (best * 1e-3) / grid / 2.1e9);
    }
    return 0;
}
