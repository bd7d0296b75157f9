// Per-CU L2 -> LDS fill rate on gfx950 with three transports (data L2-resident, 256 workgroups of 512 threads, two k-steps
// of 64 KiB in flight): (0) LDS-DMA only (`buffer_load_dwordx4 ... lds`), (1) register-staged only (buffer_load_dwordx4 ->
// VGPR -> ds_write_b128), (2) half and half.  Question: is the ~65 GB/s per CU that bounds the GEMM k-loop a limit of the
// LDS-DMA path or of the CU's vector memory path as a whole?
//   hipcc --offload-arch=gfx950 -O3 -o fill_bench fill_bench.cpp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int MODE>
__global__ __launch_bounds__(512) void k(const unsigned short* A, int K, int passes, int* sink) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 65536];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    const int strip = (b & 7) * 4 + ((b >> 3) & 3);               // the 32 blocks of an XCD share 4 strips of 512 rows
    const unsigned short* base = A + (size_t)strip * 512 * K;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(base), 0, 512 * K * 2, 0x00020000);
    unsigned off[8];
    for (int q = 0; q < 8; ++q) {                                 // piece = 8 rows x 128 B; 64 pieces = 512 rows x 64 k
        const int row = wave * 64 + q * 8 + (lane >> 3);
        off[q] = (unsigned)((row * K + (lane & 7) * 8) * 2);
    }
    const int nk = K / 64;
    int stage = 0;
    u32x4 regs[2][8];
    int acc = 0;
    for (int p = 0; p < passes; ++p)
        for (int t = 0; t < nk; ++t) {
            char* st = smem + stage * 65536 + wave * 8192;
            const unsigned so = (unsigned)(t * 128);
            constexpr int NDMA = MODE == 0 ? 8 : (MODE == 1 ? 0 : 4);
#pragma unroll
            for (int q = 0; q < NDMA; ++q) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDS_PTR(st + q * 1024), 16, off[q], so, 0, 0);
#pragma unroll
            for (int q = NDMA; q < 8; ++q) regs[stage][q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off[q], so, 0));
            // one k-step stays in flight: wait for the OTHER stage's loads, write its registers to LDS
            if (MODE == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (MODE == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            char* so_ = smem + (stage ^ 1) * 65536 + wave * 8192;
#pragma unroll
            for (int q = NDMA; q < 8; ++q) *reinterpret_cast<u32x4*>(so_ + q * 1024 + lane * 16) = regs[stage ^ 1][q];
            __builtin_amdgcn_s_barrier();
            acc += so_[lane];
            stage ^= 1;
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (sink && acc == 12345) sink[b] = acc;
}

int main(int argc, char** argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 512, passes = argc > 2 ? atoi(argv[2]) : 64;
    unsigned short* A;
    int* sink;
    hipMalloc(&A, (size_t)32 * 512 * K * 2);
    hipMalloc(&sink, 4096);
    hipMemset(A, 1, (size_t)32 * 512 * K * 2);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode) {
        float best = 1e9;
        for (int rr = 0; rr < 4; ++rr) {
            hipEventRecord(e0);
            if (mode == 0) k<0><<<256, 512>>>(A, K, passes, sink);
            if (mode == 1) k<1><<<256, 512>>>(A, K, passes, sink);
            if (mode == 2) k<2><<<256, 512>>>(A, K, passes, sink);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double bytes = 256.0 * passes * (K / 64) * 65536;
        printf("K=%d mode %d (%s): %.1f us  %.1f GB/s per CU  %.2f TB/s chip\n", K, mode, mode == 0 ? "LDS-DMA" : mode == 1 ? "VGPR + ds_write" : "half / half",
               best * 1e3, bytes / 256 / (best * 1e-3) / 1e9, bytes / (best * 1e-3) / 1e12);
    }
    return 0;
}
