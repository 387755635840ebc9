// ref_approx_driver.cpp -- thin extern "C" shim over the REAL reference functions
// (test infrastructure).  It defines nothing numeric itself: every call forwards to
// /root/reference/src/vrt/approx.{h,cpp}, compiled in place by oracle/Makefile.
#include <vrt/approx.h>
#include <cstddef>

using namespace vrt::approx;
typedef simd::Vec<simd::Float> vecf;

template <vecf (*F)(vecf)>
static void map_simd(const float *in, float *out, size_t n)
{
    // n is rounded up by the caller to a multiple of SIMD_FLOATS
    for (size_t i = 0; i < n; i += SIMD_FLOATS) simd::storeu(out + i, F(simd::loadu<SIMD_FLOATS * 4>(in + i)));
}
template <float (*F)(float)>
static void map_scalar(const float *in, float *out, size_t n)
{
    for (size_t i = 0; i < n; ++i) out[i] = F(in[i]);
}

extern "C" {
int ref_simd_floats() { return (int)SIMD_FLOATS; }
// scalar variants (approx.cpp)
void ref_as_erf(const float *in, float *out, size_t n) { map_scalar<abramowitz_stegun_erf>(in, out, n); }
void ref_spline_erf(const float *in, float *out, size_t n) { map_scalar<spline_erf>(in, out, n); }
void ref_spline_erf_mirror(const float *in, float *out, size_t n) { map_scalar<spline_erf_mirror>(in, out, n); }
void ref_taylor_erf(const float *in, float *out, size_t n) { map_scalar<taylor_erf>(in, out, n); }
void ref_fast_exp(const float *in, float *out, size_t n) { map_scalar<fast_exp>(in, out, n); }
void ref_spline_exp(const float *in, float *out, size_t n) { map_scalar<spline_exp>(in, out, n); }
// SIMD variants (approx.cpp, approx.h:91-127) -- the defaults simd::erf / simd::exp of the hot path
void ref_simd_as_erf(const float *in, float *out, size_t n) { map_simd<simd::erf>(in, out, n); }
void ref_simd_vcl_exp(const float *in, float *out, size_t n) { map_simd<simd::exp>(in, out, n); }
void ref_simd_spline_erf(const float *in, float *out, size_t n) { map_simd<simd_spline_erf>(in, out, n); }
void ref_simd_spline_erf_mirror(const float *in, float *out, size_t n) { map_simd<simd_spline_erf_mirror>(in, out, n); }
void ref_simd_taylor_erf(const float *in, float *out, size_t n) { map_simd<simd_taylor_erf>(in, out, n); }
void ref_simd_fast_exp(const float *in, float *out, size_t n) { map_simd<simd_fast_exp>(in, out, n); }
void ref_simd_spline_exp(const float *in, float *out, size_t n) { map_simd<simd_spline_exp>(in, out, n); }
}
