// Test helper (not part of the product library): a kernel that holds `nblocks` CUs for `usec` microseconds with 64 KiB of
// LDS per workgroup, to emulate a resident RCCL kernel next to the persistent GEMMs.
#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ __launch_bounds__(256) void occupier_kernel(long long ticks, int* sink) {
    __shared__ int hog[16 * 1024];
    hog[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    if (hog[(threadIdx.x * 7) & 16383] == -1) sink[0] = 1;
}
extern "C" int occupy(int nblocks, double usec, int* sink, void* stream) {
    hipLaunchKernelGGL(occupier_kernel, dim3(nblocks), dim3(256), 0, static_cast<hipStream_t>(stream), (long long)(usec * 100.0), sink);   // wall_clock64: 100 MHz
    return (int)hipGetLastError();
}
