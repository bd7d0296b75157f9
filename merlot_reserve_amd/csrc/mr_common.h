// Shared device/host helpers for libmreserve_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/mreserve_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define MR_LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

// thread-local error text (host)
void mr_set_error(const char* fmt, ...);

#define MR_CHECK_ARG(cond, ...)            \
    do {                                   \
        if (!(cond)) {                     \
            mr_set_error(__VA_ARGS__);     \
            return MR_EINVAL;              \
        }                                  \
    } while (0)

#define MR_CHECK_LAUNCH(name)                                                  \
    do {                                                                       \
        hipError_t e__ = hipGetLastError();                                    \
        if (e__ != hipSuccess) {                                               \
            mr_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return MR_ELAUNCH;                                                 \
        }                                                                      \
    } while (0)

__device__ __forceinline__ float bf16_to_f32(__bf16 v) { return (float)v; }
__device__ __forceinline__ __bf16 f32_to_bf16(float v) { return (__bf16)v; }  // v_cvt_pk_bf16_f32: RNE, NaN-preserving

__device__ __forceinline__ void unpack8(const u32x4& raw, float (&f)[8]) {
    bf16x8 v = __builtin_bit_cast(bf16x8, raw);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
}
__device__ __forceinline__ u32x4 pack8(const float (&f)[8]) {
    bf16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (__bf16)f[i];
    return __builtin_bit_cast(u32x4, v);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ float sigmoid1702(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * x)); }
__device__ __forceinline__ float gelu1702(float x) { return x * sigmoid1702(x); }
__device__ __forceinline__ float gelu1702_grad(float x) {
    float s = sigmoid1702(x);
    return s + 1.702f * x * s * (1.0f - s);
}
