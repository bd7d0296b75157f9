// LDS-DMA fill-rate microbenchmark (gfx950): per-CU rate of `buffer_load_dwordx4 ... lds` streaming a 256-row strip of a
// K-contiguous bf16 matrix out of L2, when a 1-KiB piece is (A) 8 rows x 128 B (whole lines) or (B) 16 rows x 64 B
// (half lines; the other halves are fetched by the next pieces / (C) only a whole k-step later).
//   hipcc --offload-arch=gfx950 -O3 -o dma_bench dma_bench.cpp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int PAT>
__global__ __launch_bounds__(512) void k(const unsigned short* A, int K, int passes, int* sink) {
    __shared__ __attribute__((aligned(16))) char smem[3 * 32768];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    const int strip = (b & 7) * 4 + ((b >> 3) & 3);               // the 32 blocks of an XCD share 4 strips
    const unsigned short* base = A + (size_t)strip * 256 * K;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(base), 0, 256 * K * 2, 0x00020000);
    unsigned off[4];
    for (int q = 0; q < 4; ++q) {
        if (PAT == 0) {               // piece = 8 rows x 128 B
            const int row = wave * 32 + q * 8 + (lane >> 3);
            off[q] = (unsigned)((row * K + (lane & 7) * 8) * 2);
        } else {                      // piece = 16 rows x 64 B; q&1 = row half, q>>1 = k half
            const int row = wave * 32 + (q & 1) * 16 + (lane >> 2);
            off[q] = (unsigned)((row * K + (q >> 1) * 32 + (lane & 3) * 8) * 2);
        }
    }
    const int nk = K / 64;
    int stage = 0;
    for (int p = 0; p < passes; ++p)
        for (int t = 0; t < nk; ++t) {
            char* st = smem + stage * 32768 + wave * 4096;
            const unsigned so = (unsigned)(t * 128);
            if (PAT != 2) {
                for (int q = 0; q < 4; ++q) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDS_PTR(st + q * 1024), 16, off[q], so, 0, 0);
            } else {                  // half lines, the second halves one k-step later
                const unsigned so1 = (unsigned)((t ? t - 1 : 0) * 128);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDS_PTR(st), 16, off[0], so, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDS_PTR(st + 1024), 16, off[1], so, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDS_PTR(st + 2048), 16, off[2], so1, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDS_PTR(st + 3072), 16, off[3], so1, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // two k-steps stay in flight
            __builtin_amdgcn_s_barrier();
            stage = stage == 2 ? 0 : stage + 1;
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0 && sink) sink[b] = smem[5];
}

int main(int argc, char** argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 1024, passes = argc > 2 ? atoi(argv[2]) : 64;
    unsigned short* A;
    hipMalloc(&A, (size_t)32 * 256 * K * 2);
    hipMemset(A, 1, (size_t)32 * 256 * K * 2);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pat = 0; pat < 3; ++pat) {
        float best = 1e9;
        for (int rr = 0; rr < 4; ++rr) {
            hipEventRecord(e0);
            if (pat == 0) k<0><<<256, 512>>>(A, K, passes, nullptr);
            if (pat == 1) k<1><<<256, 512>>>(A, K, passes, nullptr);
            if (pat == 2) k<2><<<256, 512>>>(A, K, passes, nullptr);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double bytes = 256.0 * passes * (K / 64) * 32768;
        printf("K=%d pattern %d: %.1f us  %.1f GB/s per CU  %.2f TB/s chip\n", K, pat, best * 1e3, bytes / 256 / (best * 1e-3) / 1e9, bytes / (best * 1e-3) / 1e12);
    }
    return 0;
}
