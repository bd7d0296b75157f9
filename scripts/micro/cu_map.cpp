// Which workgroups of a 512-block launch (256 threads, 72 KiB of LDS each: two per CU) share a CU?  Every block records HW_REG_HW_ID and
// HW_REG_XCC_ID of its first wave and its start time; the host groups blocks by (XCC, SE, SH, CU) and prints the pairs.
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/cu_map.cpp -o scripts/micro/cu_map && scripts/micro/cu_map
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <map>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)
__global__ __launch_bounds__(256, 2) void probe(unsigned* out, int spin) {
    __shared__ char smem[72 * 1024];
    unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);      // HW_REG_HW_ID, all 32 bits
    unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);    // HW_REG_XCC_ID
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    smem[threadIdx.x] = (char)hw;
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);
    __syncthreads();
    if (threadIdx.x % 64 == 0) {
        unsigned* o = out + (blockIdx.x * 4 + threadIdx.x / 64) * 4;
        o[0] = hw; o[1] = xcc; o[2] = (unsigned)t0; o[3] = (unsigned)smem[0];
    }
}
int main() {
    const int G = 512;
    unsigned* d; CK(hipMalloc(&d, G * 16 * 4));
    hipLaunchKernelGGL(probe, dim3(G), dim3(256), 0, nullptr, d, 50);
    CK(hipDeviceSynchronize());
    std::vector<unsigned> h(G * 16);
    CK(hipMemcpy(h.data(), d, G * 16 * 4, hipMemcpyDeviceToHost));
    std::map<unsigned long long, std::vector<int>> cu;
    for (int b = 0; b < G; ++b) {
        const unsigned hw = h[b * 16], xcc = h[b * 16 + 1] & 0xf;
        const unsigned cuid = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        cu[((unsigned long long)xcc << 16) | (se << 8) | (sh << 4) | cuid].push_back(b);
    }
    printf("%zu distinct (xcc, se, sh, cu) among %d blocks\n", cu.size(), G);
    int shown = 0;
    std::map<int, int> diffhist;
    for (auto& kv : cu) {
        auto& v = kv.second;
        if (shown < 24) { printf("  xcc %llu se %llu sh %llu cu %2llu :", kv.first >> 16, (kv.first >> 8) & 0xff, (kv.first >> 4) & 0xf, kv.first & 0xf); for (int b : v) { unsigned hw = h[b * 16]; printf(" b%d(wave_id %u,%u,%u,%u simd %u tg %u t0 %u)", b, hw & 0xf, h[b*16+4]&0xf, h[b*16+8]&0xf, h[b*16+12]&0xf, (hw >> 4) & 3, (hw >> 16) & 0xf, h[b * 16 + 2]); } printf("\n"); ++shown; }
        if (v.size() == 2) diffhist[std::abs(v[1] - v[0])]++;
    }
    printf("blockIdx difference of the two residents of a CU: ");
    for (auto& kv : diffhist) printf(" %d x%d", kv.first, kv.second);
    printf("\n");
    return 0;
}
