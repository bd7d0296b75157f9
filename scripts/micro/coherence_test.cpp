// Does a kernel see every store of the PREVIOUS kernel of its own stream while another stream's kernel is resident?
//   stream 1:  write(step) -> check(step) -> write(step+1) -> ...   (check counts words that are not what write(step) stored)
//   stream 2:  a streaming kernel over another buffer, back to back (optional)
// hipcc --offload-arch=gfx950 -O3 -o coherence_test coherence_test.cpp ; ./coherence_test [steps] [other_stream 0|1] [mbytes]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__global__ __launch_bounds__(512) void write_k(u32x4* x, long n4, unsigned step) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const unsigned b = (unsigned)(i * 4) + step * 0x9E3779B1u;
        x[i] = u32x4{b, b + 1, b + 2, b + 3};
    }
}
// reads in a DIFFERENT block -> data mapping than the writer (reversed), so most lines are read through another XCD's L2
__global__ __launch_bounds__(256) void check_k(const u32x4* x, long n4, unsigned step, unsigned long long* bad, unsigned* first) {
    unsigned long long nb = 0;
    for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < n4; j += (long)gridDim.x * blockDim.x) {
        const long i = n4 - 1 - j;
        const u32x4 v = x[i];
        const unsigned b = (unsigned)(i * 4) + step * 0x9E3779B1u;
        if (v[0] != b || v[1] != b + 1 || v[2] != b + 2 || v[3] != b + 3) {
            if (nb == 0 && atomicCAS(first, 0u, 1u) == 0u) { first[1] = (unsigned)i; first[2] = v[0]; first[3] = b; first[4] = step; }
            ++nb;
        }
    }
    if (nb) atomicAdd(bad, nb);
}
__global__ __launch_bounds__(512) void busy_k(u32x4* y, long n4, int reps) {
    for (int r = 0; r < reps; ++r)
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
            u32x4 v = y[i];
            v[0] += 1;
            y[i] = v;
        }
}

int main(int argc, char** argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 300, other = argc > 2 ? atoi(argv[2]) : 1;
    const long mb = argc > 3 ? atol(argv[3]) : 24;
    const long n4 = mb * 1024 * 1024 / 16;
    u32x4 *x, *y; unsigned long long* bad; unsigned* first;
    CK(hipMalloc(&x, n4 * 16)); CK(hipMalloc(&y, 64L << 20)); CK(hipMalloc(&bad, 8)); CK(hipMalloc(&first, 32));
    CK(hipMemset(bad, 0, 8)); CK(hipMemset(first, 0, 32)); CK(hipMemset(y, 0, 64L << 20));
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    for (int t = 0; t < steps; ++t) {
        if (other == 1) hipLaunchKernelGGL(busy_k, dim3(128), dim3(512), 0, s2, y, (64L << 20) / 16, 2);
        if (other == 2) for (int q = 0; q < 24; ++q) hipLaunchKernelGGL(busy_k, dim3(64), dim3(256), 0, s2, y + (q % 8) * 65536, (1L << 20) / 16, 1);   // many SHORT kernels: many kernel boundaries on the other queue
        hipLaunchKernelGGL(write_k, dim3(256), dim3(512), 0, s1, x, n4, (unsigned)t);
        hipLaunchKernelGGL(check_k, dim3(1024), dim3(256), 0, s1, x, n4, (unsigned)t, bad, first);
    }
    CK(hipDeviceSynchronize());
    unsigned long long hb; unsigned hf[8];
    CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hf, first, 32, hipMemcpyDeviceToHost));
    printf("steps %d, %ld MB, other stream %d: %llu stale / wrong 16-byte words", steps, mb, other, hb);
    if (hb) printf("  (first: word %u held %08x, expected %08x at step %u)", hf[1], hf[2], hf[3], hf[4]);
    printf("\n");
    return 0;
}
