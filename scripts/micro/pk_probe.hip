// Instruction-level probe for the packed-fp32 hazard of DESIGN.md section 4: every lane evaluates  lo = a.x - c.y,  hi = a.y - c.y  (and variants) with ONE
// packed instruction in a loop and compares with the scalar result; mismatching (lane, iteration) pairs are counted per lane.
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC scripts/micro/pk_probe.hip -o scripts/micro/libpk_probe.so
// variant 0: v_pk_add_f32 d, a, c op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]      (the LayerNorm form: both halves subtract the HI register of c)
// variant 1: v_pk_add_f32 d, a, c neg_lo:[0,1] neg_hi:[0,1]                   (no op_sel)
// variant 2: v_pk_add_f32 d, a, c op_sel:[0,1]                                 (no neg)
// variant 3: v_pk_mul_f32 d, s, a op_sel_hi:[0,1]                              (the affine part's broadcast form)
// variant 4: v_pk_add_f32 d, a, c                                              (plain)
// variant 5: v_pk_mov_b32 d, a, c op_sel:[1,0]                                 (d = {a.hi, c.lo}: the form the compiler emits in the GEMM epilogues)
// variant 6: v_pk_fma_f32 d, a, c, a op_sel:[0,1,0]                            (src1 low element from the high register)
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int V>
__global__ __launch_bounds__(256) void pk_probe_kernel(const float* __restrict__ in, unsigned* __restrict__ bad_per_lane, int iters) {
    const int lane = threadIdx.x & 63;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    f32x2 a = {in[(gid * 4) & 0xfffff], in[(gid * 4 + 1) & 0xfffff]};
    f32x2 c = {in[(gid * 4 + 2) & 0xfffff], in[(gid * 4 + 3) & 0xfffff]};
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        f32x2 d, want;
        if (V == 0) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(c)); want = f32x2{a.x - c.y, a.y - c.y}; }
        if (V == 1) { asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(c)); want = f32x2{a.x - c.x, a.y - c.y}; }
        if (V == 2) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(a), "v"(c)); want = f32x2{a.x + c.y, a.y + c.y}; }
        if (V == 3) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(d) : "v"(c), "v"(a)); want = f32x2{c.x * a.x, c.x * a.y}; }
        if (V == 4) { asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(c)); want = f32x2{a.x + c.x, a.y + c.y}; }
        if (V == 5) { asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(d) : "v"(a), "v"(c)); want = f32x2{a.y, c.x}; }
        if (V == 6) { asm volatile("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[0,1,0]" : "=v"(d) : "v"(a), "v"(c)); want = f32x2{__builtin_fmaf(a.x, c.y, a.x), __builtin_fmaf(a.y, c.y, a.y)}; }
        bad += (__float_as_uint(d.x) != __float_as_uint(want.x)) + (__float_as_uint(d.y) != __float_as_uint(want.y));
        // new operands every iteration (dependent on the results, so nothing is hoisted)
        a = f32x2{d.y * 0.5f + 0.25f, d.x * 0.5f - 0.125f};
        c = f32x2{want.y * 0.25f + 1.0f, want.x * 0.25f - 1.0f};
    }
    if (bad) atomicAdd(&bad_per_lane[lane], bad);
}

extern "C" int pk_probe(int variant, int blocks, int iters, const float* in, unsigned* bad_per_lane, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (variant) {
        case 0: hipLaunchKernelGGL(pk_probe_kernel<0>, dim3(blocks), dim3(256), 0, s, in, bad_per_lane, iters); break;
        case 1: hipLaunchKernelGGL(pk_probe_kernel<1>, dim3(blocks), dim3(256), 0, s, in, bad_per_lane, iters); break;
        case 2: hipLaunchKernelGGL(pk_probe_kernel<2>, dim3(blocks), dim3(256), 0, s, in, bad_per_lane, iters); break;
        case 3: hipLaunchKernelGGL(pk_probe_kernel<3>, dim3(blocks), dim3(256), 0, s, in, bad_per_lane, iters); break;
        case 5: hipLaunchKernelGGL(pk_probe_kernel<5>, dim3(blocks), dim3(256), 0, s, in, bad_per_lane, iters); break;
        case 6: hipLaunchKernelGGL(pk_probe_kernel<6>, dim3(blocks), dim3(256), 0, s, in, bad_per_lane, iters); break;
        default: hipLaunchKernelGGL(pk_probe_kernel<4>, dim3(blocks), dim3(256), 0, s, in, bad_per_lane, iters); break;
    }
    return (int)hipGetLastError();
}
