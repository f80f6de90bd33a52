// Internal helpers shared by the gfx950 kernels of libreid_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/reid_hip.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8;   // 8 bf16 = 4 VGPRs (MFMA A/B fragment)
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef unsigned short bf16_t;   // 16-bit storage (bf16 or f16 by flavor)

void reid_set_error(const char* fmt, ...);

#define REID_CHECK_ARG(cond, ...)                     \
    do {                                              \
        if (!(cond)) {                                \
            reid_set_error(__VA_ARGS__);              \
            return REID_ERR_ARG;                      \
        }                                             \
    } while (0)

#define REID_CHECK_LAUNCH(name)                                                        \
    do {                                                                               \
        hipError_t e_ = hipGetLastError();                                             \
        if (e_ != hipSuccess) {                                                        \
            reid_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));     \
            return REID_ERR_LAUNCH;                                                    \
        }                                                                              \
    } while (0)

// ---- 16-bit operand format of this library flavor -------------------------------------------------------------
// The library is built twice from the same sources: libreid_hip.so (bf16 operands: 8 significant bits, fp32 range,
// no loss scaling) and libreid_hip_f16.so (-DREID_FLAVOR_F16: IEEE half operands, 11 significant bits, same MFMA
// rate).  Everything below this line is format-agnostic: `bf16_t` / `bf16x8` are just 16-bit storage types.
typedef _Float16 __attribute__((ext_vector_type(8))) half8_t;
#ifdef REID_FLAVOR_F16
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return __builtin_bit_cast(bf16_t, (_Float16)f); }
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, a), __builtin_bit_cast(half8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, a), __builtin_bit_cast(half8_t, b), c, 0, 0, 0);
}
#define REID_FLAVOR_ID 1
#define REID_T16_EPS 0.00048828125f      /* 2^-11 relative rounding error */
#else
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even, NaN preserved (plain cast lowers to v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
#define REID_FLAVOR_ID 0
#define REID_T16_EPS 0.00390625f         /* 2^-8 */
#endif

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
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

__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float dgelu_erf_f(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
    const float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}
__device__ __forceinline__ float quick_gelu_f(float x) { return x / (1.0f + __expf(-1.702f * x)); }
__device__ __forceinline__ float dquick_gelu_f(float x) {
    const float s = 1.0f / (1.0f + __expf(-1.702f * x));
    return s + x * 1.702f * s * (1.0f - s);
}
