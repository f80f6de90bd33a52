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

__attribute__((visibility("hidden"))) void reid_set_error(const char* fmt, ...);

// Experiment knobs (tile choice, early-exit builds, pipeline parameters): ONE cached table per process, filled from the
// environment (REID_<NAME>) on first use and changed afterwards only through reid_set_knob() -- no getenv() on the launch path.
enum reid_knob_id {
    KNOB_GEMM_TILE, KNOB_GEMM_DBG, KNOB_GEMM_GROUPM, KNOB_GEMM_EPI, KNOB_GEMM_STAGGER,
    KNOB_ATTN_DBG, KNOB_TN_BLOCKS, KNOB_TOPK_DBG, KNOB_TOPK_TILE, KNOB_STREAM_ROWS, KNOB_STREAM_GROUPS, KNOB_SDM_IMPL, KNOB_SKINNY_TILE, KNOB_GEMM_PERSIST,
    KNOB_ATTN_BWD, KNOB_LORA_IMPL, KNOB_GELU_IMPL, KNOB_HEAD_IMPL, KNOB_STREAM_FUSE, KNOB_TOPK_SCAN, KNOB_LN_IMPL, KNOB_COUNT
};
// (internal C++ symbols of the library: hidden, only the extern "C" entry points of include/reid_hip.h are exported)
__attribute__((visibility("hidden"))) int reid_knob(int id);
__attribute__((visibility("hidden"))) int reid_num_cus();   // compute units of the current device (cached)
// The *_DBG knobs switch kernels into timing-experiment modes that skip work (WRONG results).  They exist only in builds made with
// -DREID_EXPERIMENTS (tools/): in the shipped library REID_DBG(p) is the constant 0 -- the early exits are compiled out -- the
// REID_*_DBG environment variables are ignored and reid_set_knob() refuses those names.
#ifdef REID_EXPERIMENTS
#define REID_DBG(p) ((p).dbg)
#else
#define REID_DBG(p) 0
#endif

#define REID_CHECK_ARG(cond, ...)                     \
    do {                                              \
        if (!(cond)) {                                \
            reid_set_error(__VA_ARGS__);              \
            return REID_ERR_ARG;                      \
        }                                             \
    } while (0)

// Opt a kernel into more than 64 KiB of dynamic LDS.  The attribute is per device: remembered per (call site = kernel instance,
// device) in a bit mask, so a process that later uses a second GPU sets it there too; thread-safe (the call is idempotent, the mask
// atomic); the return code is checked.
static inline hipError_t reid_max_dyn_lds(const void* fn, int bytes, unsigned* done_mask) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 32 && ((__atomic_load_n(done_mask, __ATOMIC_ACQUIRE) >> dev) & 1u)) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess && dev >= 0 && dev < 32) __atomic_fetch_or(done_mask, 1u << dev, __ATOMIC_RELEASE);
    return e;
}
#define REID_MAX_LDS(kernel, bytes)                                                                              \
    do {                                                                                                         \
        static unsigned lds_done_ = 0;                                                                           \
        const hipError_t le_ = reid_max_dyn_lds((const void*)(kernel), (int)(bytes), &lds_done_);                \
        if (le_ != hipSuccess) {                                                                                 \
            reid_set_error("hipFuncSetAttribute(%s, dynamic LDS %d): %s", #kernel, (int)(bytes), hipGetErrorString(le_)); \
            return REID_ERR_LAUNCH;                                                                              \
        }                                                                                                        \
    } while (0)

#define REID_CHECK_HIP(call, what)                                                     \
    do {                                                                               \
        const hipError_t he_ = (call);                                                 \
        if (he_ != hipSuccess) {                                                       \
            reid_set_error("%s: %s", what, hipGetErrorString(he_));                    \
            return REID_ERR_LAUNCH;                                                    \
        }                                                                              \
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
// Range safety: MODE.FP16_OVFL (hardware mode bit 23, per wave) makes every conversion to IEEE half CLAMP a finite value that
// overflows to +-65504 instead of producing an infinity; true infinities and NaNs pass through unchanged, so a diverged run is still
// seen by the gradient sanitiser.  Set once at the entry of every kernel that stores 16-bit values.
#define REID_T16_ENTER() __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1)
#define REID_FLAVOR_ID 1
#define REID_T16_EPS 0.0009765625f       /* 2u = 2^-10: bound of |q~.g~ - q.g| for unit vectors rounded to f16 (u = 2^-11) */
#else
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even, NaN preserved (plain cast lowers to v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
#define REID_T16_ENTER() do { } while (0)      /* bf16 has fp32's exponent range */
#define REID_FLAVOR_ID 0
#define REID_T16_EPS 0.00390625f         /* 2u = 2^-8 for bf16 (u = 2^-9) */
#endif

// two floats -> one dword of the flavor's 16-bit format, converted as a 2-vector: ONE v_cvt_pk_bf16_f32 (the scalar-by-scalar form
// made hipcc pair the wrong halves and re-interleave them with v_and / v_lshl / two v_or_b32_sdwa per dword)
typedef float pk_f32x2 __attribute__((ext_vector_type(2)));
#ifdef REID_FLAVOR_F16
typedef _Float16 pk_t16x2 __attribute__((ext_vector_type(2)));
#else
typedef __bf16 pk_t16x2 __attribute__((ext_vector_type(2)));
#endif
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const pk_t16x2 r = __builtin_convertvector(pk_f32x2{lo, hi}, pk_t16x2);
    return __builtin_bit_cast(uint32_t, r);
}

// IEEE half whatever the flavor (REID_F16 tensors): round-to-nearest-even pair conversion (v_cvt_pk_f16_f32)
typedef _Float16 pk_h16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_f16x2(float lo, float hi) {
    const pk_h16x2 r = __builtin_convertvector(pk_f32x2{lo, hi}, pk_h16x2);
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ float f16_to_f32(unsigned short v) { return (float)__builtin_bit_cast(_Float16, v); }
// MODE.FP16_OVFL for the calling wave: conversions to half clamp finite overflow to +-65504 (what REID_T16_ENTER does in the f16 flavor)
#define REID_F16_SATURATE() __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1)

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

// Exact-erf GELU (nn.GELU() default, mer_lora.py:257) without libm's erff (~100 instructions with branches: it made
// the fc1 epilogue cost as much as the fc1 GEMM).  Normal CDF by Abramowitz-Stegun 7.1.26 on erfc (|abs error| <= 1.5e-7,
// and RELATIVE accuracy kept in the negative tail because erfc is evaluated directly, no 1 - x cancellation):
//   erfc(z) = t (a1 + t (a2 + t (a3 + t (a4 + t a5)))) exp(-z^2),  t = 1 / (1 + p z),  z = |x| / sqrt(2)
// exp(-z^2) = exp(-x^2/2) is also the Gaussian density needed by the derivative.
__device__ __forceinline__ void gauss_cdf_pdf(float x, float& cdf, float& pdf_unnorm) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    const float e = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170f);          // exp(-x^2/2) = 2^(-x^2 log2(e)/2)
    const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
    const float q = 0.5f * poly * e;                                                  // 0.5 erfc(|x|/sqrt2)
    cdf = x >= 0.f ? 1.0f - q : q;
    pdf_unnorm = e;
}
__device__ __forceinline__ float gelu_erf_f(float x) {
    float c, e;
    gauss_cdf_pdf(x, c, e);
    return x * c;
}
__device__ __forceinline__ void gelu_both_f(float x, float& g, float& dg) {      // one erfc / exp for value and derivative
    float c, e;
    gauss_cdf_pdf(x, c, e);
    g = x * c;
    dg = fmaf(x * 0.39894228040143268f, e, c);
}
// The same value / derivative for a PAIR of elements, written on 2-vectors so that the polynomial, the products and the sign
// handling compile to v_pk_fma_f32 / v_pk_mul_f32 (the lean 16-bit GEMM epilogues are bound by exactly this arithmetic: two waves per
// SIMD x 128 elements per lane).  cdf = 0.5 + copysign(0.5 - q, x) instead of a compare + select: absolute error <= 6e-8 more than
// gauss_cdf_pdf (the relative accuracy of the far negative tail is given up: fine for 16-bit outputs, which is where this is used).
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gelu_both_x2(f32x2_t x, f32x2_t& g, f32x2_t& dg) {
    const f32x2_t xs = (x * x) * -0.72134752044448170f;                              // -x^2 log2(e) / 2
    f32x2_t e; e.x = __builtin_amdgcn_exp2f(xs.x); e.y = __builtin_amdgcn_exp2f(xs.y);
    f32x2_t z; z.x = __builtin_fabsf(x.x); z.y = __builtin_fabsf(x.y);
    const f32x2_t den = z * (0.3275911f * 0.70710678118654752f) + 1.0f;
    f32x2_t t; t.x = __builtin_amdgcn_rcpf(den.x); t.y = __builtin_amdgcn_rcpf(den.y);
    f32x2_t poly = t * (0.5f * 1.061405429f) + (0.5f * -1.453152027f);                // 0.5 erfc: the 0.5 is folded into the coefficients
    poly = poly * t + (0.5f * 1.421413741f);
    poly = poly * t + (0.5f * -0.284496736f);
    poly = poly * t + (0.5f * 0.254829592f);
    const f32x2_t q = (poly * t) * e;
    const f32x2_t h = 0.5f - q;
    f32x2_t sh; sh.x = __builtin_copysignf(h.x, x.x); sh.y = __builtin_copysignf(h.y, x.y);
    const f32x2_t cdf = sh + 0.5f;
    g = x * cdf;
    dg = (x * 0.39894228040143268f) * e + cdf;
}
__device__ __forceinline__ float dgelu_erf_f(float x) {
    float c, e;
    gauss_cdf_pdf(x, c, e);
    return fmaf(x * 0.39894228040143268f, e, c);
}
__device__ __forceinline__ float quick_gelu_f(float x) { return x / (1.0f + __expf(-1.702f * x)); }
__device__ __forceinline__ float dquick_gelu_f(float x) {
    const float s = 1.0f / (1.0f + __expf(-1.702f * x));
    return s + x * 1.702f * s * (1.0f - s);
}
