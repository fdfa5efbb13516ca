// Shared device/host helpers for the iSegProbe gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/isegprobe_hip.h"

typedef unsigned short bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define ISP_LDS __attribute__((address_space(3)))
#define ISP_GLOBAL __attribute__((address_space(1)))

#define ISP_CHECK_ARG(cond) \
    do {                    \
        if (!(cond)) return ISP_ERR_INVALID; \
    } while (0)

static inline int isp_launch_status() {
    return hipGetLastError() == hipSuccess ? ISP_OK : ISP_ERR_LAUNCH;
}

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }

// f32 -> bf16, round-to-nearest-even; a plain cast lowers to v_cvt_pk_bf16_f32 on gfx950
// and keeps NaNs NaN (MI355X_MICROARCH "Correctness boundaries").
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}

__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
    return (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16);
}

// ---- IEEE half helpers (FeatUp-JBU stack, f16 form of the head's convolutions)
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2v_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2h(float lo, float hi) {  // round-to-nearest-even
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v_t{lo, hi}, f16x2_t));
}
__device__ __forceinline__ float h_lo(unsigned q) { return (float)__builtin_bit_cast(f16x2_t, q).x; }
__device__ __forceinline__ float h_hi(unsigned q) { return (float)__builtin_bit_cast(f16x2_t, q).y; }
template <bool OUT_BF16>
__device__ __forceinline__ unsigned pack2o(float lo, float hi) {
    if constexpr (OUT_BF16) return pack2bf(lo, hi);
    else return pack2h(lo, hi);
}
// Output stores of the half-precision inference streams (ViT trunk, LoftUp, head convolutions): half's largest finite
// value is 65504 and v_cvt_pk_f16_f32 overflows to inf beyond it, which the next LayerNorm / softmax turns into NaN.
// Saturate instead (v_med3_f32, one instruction per value); bf16 has fp32's range and needs nothing.
__device__ __forceinline__ unsigned pack2h_sat(float lo, float hi) {
    return pack2h(__builtin_amdgcn_fmed3f(lo, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(hi, -65504.f, 65504.f));
}
template <bool OUT_BF16>
__device__ __forceinline__ unsigned pack2o_sat(float lo, float hi) {
    if constexpr (OUT_BF16) return pack2bf(lo, hi);
    else return pack2h_sat(lo, hi);
}

// 16-byte async global -> LDS copy.  LDS destination = wave-uniform base + lane*16.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const ISP_GLOBAL void*)gsrc, (ISP_LDS void*)lds_wave_base, 16, 0, 0);
}

// Bijective XCD-aware block remap (blocks b and b+8 share an XCD under round-robin
// dispatch): gives each XCD a contiguous chunk of the tile grid so neighbouring tiles
// share that XCD's L2.  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (orig >> 3);
}

// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, i.e. fp32-level for a bf16-rounded result):
// ~12 VALU ops instead of libm erff's ~40 -- the GELU epilogue of the MLP GEMMs was costing more
// than their 6-step K loop.
__device__ __forceinline__ float fast_erf(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));  // v_rcp_f32 (1 ulp); __frcp_rn is a ~10-instruction IEEE divide
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float r = 1.0f - p * t * __expf(-ax * ax);
    return copysignf(r, x);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + fast_erf(x * 0.70710678118654752440f)); }

// GELU(x) = x * Phi(x) with Phi(x) ~ sigmoid(x * (c0 + c1 x^2 + c2 x^4)): max |error| 2.5e-5 over the real line (fit
// against the erf form, tools/fit_gelu.py), 1/160 of a bf16 ulp at 1.0 -- the result is rounded to bf16 right away.
// One v_exp + one v_rcp + 6 plain VALU per value instead of the erf polynomial's 14: this kernel's GELU has to fit into
// the issue slots the MFMAs leave free.  x^2 is clamped at 36: the quartic (c2 < 0) turns negative at |x| = 11.1, and
// beyond |x| = 6 the sigmoid is saturated to 2e-9 anyway.
__device__ __forceinline__ float gelu_sig5(float x) {
    constexpr float L2E = 1.4426950408889634f;
    const float x2 = fminf(x * x, 36.0f);
    float p = fmaf(-0.0007030391178699941f * L2E, x2, 0.07401132856622687f * L2E);
    p = fmaf(p, x2, 1.595015725363722f * L2E);
    const float e = __builtin_amdgcn_exp2f(-x * p);  // v_exp_f32
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}

// 16 zero bytes for out-of-image taps of the implicit-GEMM convs (LDS-DMA cannot
// predicate a lane, so out-of-range lanes read here instead).
static __device__ const uint4 g_isp_zero16[4] = {};
