// vrt_device_math.h -- device-side exp / erf variants for gfx950 (wave64, no MFMA: the path is
// VALU + transcendental bound).  Each variant names the reference approximation it stands for
// (paths relative to /root/reference/src).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/vrt_hip.h"

namespace vrtk {

// vrt/rt.h:18-20
constexpr float SQRT_2_PI = 0.7978845608028654f;
constexpr float INV_SQRT_2_PI = 1.f / SQRT_2_PI;
constexpr float SQRT_2 = 1.41421356237309504880f;

// A wave-uniform constant the compiler would keep in an SGPR, moved to a VGPR: on gfx950 a VALU instruction with an SGPR
// source issues in 4.4 cycles (3 waves per SIMD) where the same instruction on VGPRs, inline constants or a literal takes
// 2.8-3.2 (tools/ubench/valu.hip, profiles/r02_valu_ops.txt).  The empty asm is opaque to the compiler (no
// rematerialisation into an SGPR) but free of side effects (hoisted out of loops).
__device__ __forceinline__ float pin_vgpr(float c)
{
    asm("" : "+v"(c));
    return c;
}

// exp(x) to ~1 ulp on the quarter-rate v_exp_f32: 2^(x*log2e) with the product's rounding
// error and the low part of log2e folded back in as a first-order correction.
// Stands for expf (rt.h:32 default) and vcl_exp (approx.h:91-106), both ~1 ulp.
__device__ __forceinline__ float exp_accurate(float x)
{
    constexpr float L2E_HI = 1.44269504088896340736f;
    constexpr float L2E_LO = 1.925963033500011e-8f; // log2(e) - (float)log2(e)
    constexpr float LN2 = 0.6931471805599453f;
    const float t = x * L2E_HI;
    float e = __builtin_fmaf(x, L2E_HI, -t);
    e = __builtin_fmaf(x, L2E_LO, e);
    const float r = __builtin_amdgcn_exp2f(t);
    return __builtin_fmaf(r, e * LN2, r);
}

// vcl_exp flushes to 0 below -87.3 (vectormath_exp.h:393,447-456) -- never produces denormals.
__device__ __forceinline__ float exp_vcl(float x) { return (x < -87.3f) ? 0.f : exp_accurate(x); }

// approx.cpp:112-129 fast_exp (Schraudolph), NDEBUG unset => clamped; truncating float->u32.
__device__ __forceinline__ float exp_fast(float x)
{
    const float a = (float)(1 << 23) / 0.693147180559945309417f;
    const float b = (float)(1 << 23) * (127 - 0.043677448f);
    const float c = (float)(1 << 23);
    const float d = (float)(1 << 23) * 255;
    float v;
    {
#pragma clang fp contract(off)
        v = a * x + b;
    }
    if (v < c || v > d) v = (v < c) ? 0.f : d;
    return __uint_as_float((unsigned)v);
}

// cubic pieces ((c3*d + c2)*d + c1)*d + c0, d = x - lo on [lo, hi)
struct cubic_piece { float lo, hi, c3, c2, c1, c0; };

// approx.cpp:141-163 spline_exp
__device__ __forceinline__ float exp_spline(float x)
{
#pragma clang fp contract(off)
    static constexpr cubic_piece P[] = {
        { -9.0f, -8.0f, 2.0944866e-5f, 6.2834595e-5f, 0.0001198996f, 0.0001234098f },
        { -8.0f, -7.0f, 2.9318619e-5f, 0.00015079045f, 0.00033352466f, 0.00033546262f },
        { -7.0f, -6.0f, 9.210422e-5f, 0.0004271031f, 0.00091141823f, 0.000911882f },
        { -6.0f, -5.0f, 0.00022834886f, 0.0011121497f, 0.002450671f, 0.0024787523f },
        { -5.0f, -4.5f, 0.0006963741f, 0.0032012719f, 0.0067640925f, 0.006737947f },
        { -4.5f, -4.0f, 0.0015094817f, 0.0054654945f, 0.011097476f, 0.011108996f },
        { -4.0f, -3.5f, 0.0023322464f, 0.008963864f, 0.018312154f, 0.01831564f },
        { -3.5f, -3.0f, 0.0038776079f, 0.0147802755f, 0.030184224f, 0.030197384f },
        { -3.0f, -2.5f, 0.006420028f, 0.024410319f, 0.049779523f, 0.049787067f },
        { -2.5f, -2.0f, 0.010444719f, 0.040077396f, 0.08202338f, 0.082085f },
        { -2.0f, -1.75f, 0.017753968f, 0.06670835f, 0.13541625f, 0.13533528f },
        { -1.75f, -1.5f, 0.026580833f, 0.08664397f, 0.17375433f, 0.17377394f },
        { -1.5f, -1.25f, 0.03215266f, 0.11075847f, 0.22310494f, 0.22313017f },
        { -1.25f, -1.0f, 0.04326379f, 0.14320631f, 0.28659615f, 0.2865048f },
        { -1.0f, -0.75f, 0.04961379f, 0.18041666f, 0.36750188f, 0.36787945f },
        { -0.75f, -0.5f, 0.08547847f, 0.2445255f, 0.47373742f, 0.47236654f },
        { -0.5f, -0.25f, 0.02860214f, 0.2659771f, 0.60136306f, 0.60653067f },
        { -0.25f, 0.0f, 0.3395703f, 0.52065486f, 0.7980211f, 0.7788008f },
    };
    if (x <= -9.0f) return 0.0f;
    float r = 1.0f;
    bool done = false;
#pragma unroll
    for (int i = 0; i < 18; ++i) {
        if (!done && x < P[i].hi) {
            const float d = x - P[i].lo;
            r = ((P[i].c3 * d + P[i].c2) * d + P[i].c1) * d + P[i].c0;
            done = true;
        }
    }
    return r;
}

template <int EXP>
__device__ __forceinline__ float vexp(float x)
{
    if constexpr (EXP == VRT_EXP_VCL) return exp_vcl(x);
    else if constexpr (EXP == VRT_EXP_FAST) return exp_fast(x);
    else if constexpr (EXP == VRT_EXP_SPLINE) return exp_spline(x);
    else return exp_accurate(x);
}

// two floats in an aligned register pair: the operand of the packed fp32 instructions (v_pk_fma_f32, v_pk_mul_f32, v_pk_add_f32)
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

// ---- erf ------------------------------------------------------------------------------------

// Abramowitz-Stegun 7.1.27 (approx.cpp:90-110): sign * (1 - 1/(1 + a0 t + a1 t^2 + a2 t^3 + a3 t^4)^4): 4 fma +
// 2 mul + the 1-ulp v_rcp_f32 (the reference's SIMD path uses a 2^-14 estimate, its scalar path a divide) + one
// v_bfi for the sign.
__device__ __forceinline__ float erf_as(float x)
{
    const float t = __builtin_fabsf(x);
    float p = __builtin_fmaf(0.078108f, t, 0.000972f);
    p = __builtin_fmaf(p, t, 0.230389f);
    p = __builtin_fmaf(p, t, 0.278393f);
    p = __builtin_fmaf(p, t, 1.0f);
    const float p2 = p * p;
    return __builtin_copysignf(1.0f - __builtin_amdgcn_rcpf(p2 * p2), x);
}

// approx.cpp:9-24 spline_erf
__device__ __forceinline__ float erf_spline_piece(int i, float x)
{
#pragma clang fp contract(off)
    static constexpr cubic_piece P[] = {
        { -2.9f, -2.3f, 0.00019103826f, 0.00034386886f, 0.0002048055f, -0.9999589f },
        { -2.3f, -1.7f, 0.0039601973f, 0.007472224f, 0.0048944615f, -0.99885684f },
        { -1.7f, -1.1f, 0.043702256f, 0.08613629f, 0.061059568f, -0.98379046f },
        { -1.1f, -0.5f, 0.1663916f, 0.38564116f, 0.34412605f, -0.8802051f },
        { -0.5f, 0.1f, 0.066660866f, 0.50563073f, 0.8788892f, -0.5204999f },
        { 0.1f, 0.7f, -0.3536934f, -0.1310174f, 1.1036571f, 0.112462915f },
        { 0.7f, 1.3f, -0.2300452f, -0.5450987f, 0.6979875f, 0.6778012f },
        { 1.3f, 1.9f, 0.15578617f, -0.26468363f, 0.21211804f, 0.93400794f },
        { 1.9f, 2.5f, 0.12406375f, -0.041368887f, 0.028486524f, 0.9927904f },
        { 2.5f, 3.1f, 0.02131252f, -0.0030063519f, 0.0018613797f, 0.999593f },
    };
    const float d = x - P[i].lo;
    return ((P[i].c3 * d + P[i].c2) * d + P[i].c1) * d + P[i].c0;
}
__device__ __forceinline__ float erf_spline(float x)
{
    constexpr float HI[] = { -2.3f, -1.7f, -1.1f, -0.5f, 0.1f, 0.7f, 1.3f, 1.9f, 2.5f, 3.1f };
    if (x <= -2.9f) return -1.0f;
    float r = 1.0f;
    bool done = false;
#pragma unroll
    for (int i = 0; i < 10; ++i)
        if (!done && x < HI[i]) { r = erf_spline_piece(i, x); done = true; }
    return r;
}
// approx.cpp:45-55 spline_erf_mirror
__device__ __forceinline__ float erf_spline_mirror(float x)
{
    constexpr float HI[] = { -2.3f, -1.7f, -1.1f, -0.5f };
    const float inv_sign = -(float)((x >= 0) - (x < 0));
    x *= inv_sign;
    if (x <= -2.9f) return -inv_sign;
    float r = 0.f;
    bool done = false;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (!done && x < HI[i]) { r = erf_spline_piece(i, x); done = true; }
    if (!done) r = erf_spline_piece(4, x);
    return inv_sign * r;
}
// approx.cpp:71-80 taylor_erf
__device__ __forceinline__ float erf_taylor(float x)
{
#pragma clang fp contract(off)
    constexpr float ts[] = { 1.0f, -0.33333334f, 0.1f, -0.023809524f, 0.0046296297f, -0.00075757573f,
                             0.00010683761f, -1.3227514e-5f, 1.4589169e-6f, -1.4503853e-7f };
    if (x <= -2.f) return -1.f;
    if (x >= 2.f) return 1.f;
    float acc = ts[9];
#pragma unroll
    for (int i = 8; i >= 0; --i) acc = acc * x * x + ts[i];
    return (2.f * 0.564189583547756286948f) * acc * x;
}

template <int ERF>
__device__ __forceinline__ float verf(float x)
{
    if constexpr (ERF == VRT_ERF_AS) return erf_as(x);
    else if constexpr (ERF == VRT_ERF_SPLINE) return erf_spline(x);
    else if constexpr (ERF == VRT_ERF_SPLINE_MIRROR) return erf_spline_mirror(x);
    else if constexpr (ERF == VRT_ERF_TAYLOR) return erf_taylor(x);
    else return erff(x);
}
// |x| from which the variant returns EXACTLY +-1.0f (used to skip saturated terms without changing a bit):
//   A&S: 1 - rcp(p^4) rounds to 1 once p^4 > 2^25, i.e. |x| >= 5.46; erff: erfc(4.2) = 2.9e-9 << 2^-25;
//   the splines and the Taylor form clamp explicitly (approx.cpp:11,24,47,75-76).
// The erf of the hot loops, with its constants where the hardware reads them fastest.  On gfx950 a VALU instruction with
// an SGPR source operand issues in 4.4 cycles (3 waves per SIMD) where the same instruction on VGPRs, inline constants
// or a literal takes 2.8-3.2 (tools/ubench/valu.hip, profiles/r02_valu_ops.txt); the compiler keeps the four
// Abramowitz-Stegun coefficients in SGPRs (a VOP3 fma cannot carry a literal on this ISA), which costs the polynomial
// 6 cycles of its 17 per term.  Pinning them in VGPRs (an empty asm the compiler cannot see through) makes the four
// fmas full-rate (pin_vgpr above).  Same operations, same order, same results as erf_as().
template <int ERF>
struct ErfEval {
    __device__ __forceinline__ float operator()(float x) const { return verf<ERF>(x); }
};
template <>
struct ErfEval<VRT_ERF_AS> {
    float c3, c2, c1, c0;
    __device__ __forceinline__ ErfEval() : c3(pin_vgpr(0.078108f)), c2(pin_vgpr(0.000972f)), c1(pin_vgpr(0.230389f)), c0(pin_vgpr(0.278393f)) {}
    __device__ __forceinline__ float operator()(float x) const
    {
        const float t = __builtin_fabsf(x);
        float p = __builtin_fmaf(c3, t, c2);
        p = __builtin_fmaf(p, t, c1);
        p = __builtin_fmaf(p, t, c0);
        p = __builtin_fmaf(p, t, 1.0f);
        const float p2 = p * p;
        return __builtin_copysignf(1.0f - __builtin_amdgcn_rcpf(p2 * p2), x);
    }
    // R(x) = 1 / p(|x|)^4: Erf(x) = sign(x) (1 - R(x)).  Where the sign of x is known for a whole wave the callers form
    // E - Erf(x) without the sign transfer (a 3-source v_bfi_b32: 4.3 issue cycles, profiles/r02_valu_ops.txt).
    __device__ __forceinline__ float R(float x) const
    {
        const float t = __builtin_fabsf(x);
        float p = __builtin_fmaf(c3, t, c2);
        p = __builtin_fmaf(p, t, c1);
        p = __builtin_fmaf(p, t, c0);
        p = __builtin_fmaf(p, t, 1.0f);
        const float p2 = p * p;
        return __builtin_amdgcn_rcpf(p2 * p2);
    }
};

template <int ERF>
__host__ __device__ constexpr float erf_saturation()
{
    return ERF == VRT_ERF_AS ? 5.5f : ERF == VRT_ERF_SPLINE ? 3.1f : ERF == VRT_ERF_SPLINE_MIRROR ? 2.9f
         : ERF == VRT_ERF_TAYLOR ? 2.0f : 4.2f;
}

} // namespace vrtk
