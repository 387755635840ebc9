/*
 * vrt_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Scalar plain-C restatement of the reference's hot path.  See vrt_oracle.h for
 * the pinning status.  Build with -ffp-contract=off and WITHOUT -ffast-math so
 * the arithmetic written here is the arithmetic executed; fused operations the
 * reference gets from its SIMD libraries are written as explicit fmaf().
 *
 * Reference paths are relative to /root/reference/src.
 */
#include "vrt_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* vrt/rt.h:18-20 */
static const float SQRT_2_PI = 0.7978845608028654f;
#define INV_SQRT_2_PI (1.f / SQRT_2_PI)
static const float SQRT_2 = 1.41421356237309504880f;

/* ------------------------------------------------------------------------- */
/* approximations                                                            */
/* ------------------------------------------------------------------------- */

/* approx.cpp:5 SIGN(x) = (x >= 0) - (x < 0) */
static inline float sign_of(float x) { return (float)((x >= 0) - (x < 0)); }

/* approx.cpp:90-99: erf(x) ~ sign * (1 - 1/(1 + a0 t + a1 t^2 + a2 t^3 + a3 t^4)^4), t = |x|
 * (Abramowitz & Stegun 7.1.27; the code's a1 = 0.230389 is authoritative). */
float oracle_as_erf(float x)
{
    const float sg = sign_of(x);
    const float t = x * sg;
    const float a0 = 0.278393f, a1 = 0.230389f, a2 = 0.000972f, a3 = 0.078108f;
    const float den = (((a3 * t + a2) * t + a1) * t + a0) * t + 1;
    const float den2 = den * den;
    const float val = 1 - 1 / (den2 * den2);
    return val * sg;
}

/* Cubic pieces "((c3*d + c2)*d + c1)*d + c0, d = x - lo" on [lo, hi). */
typedef struct { float lo, hi, c3, c2, c1, c0; } cubic_piece;

static float eval_piece(const cubic_piece *p, float x)
{
    const float d = x - p->lo;
    return ((p->c3 * d + p->c2) * d + p->c1) * d + p->c0;
}

/* approx.cpp:9-24: supports -2.9:0.6:3.1 */
static const cubic_piece ERF_PIECES[] = {
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
#define N_ERF_PIECES (sizeof(ERF_PIECES) / sizeof(ERF_PIECES[0]))

float oracle_spline_erf(float x)
{
    if (x <= -2.9f) return -1.0f;
    for (size_t i = 0; i < N_ERF_PIECES; ++i)
        if (x < ERF_PIECES[i].hi) return eval_piece(&ERF_PIECES[i], x);
    return 1.0f;
}

/* approx.cpp:45-55: evaluate the negative half on -|x| and mirror. */
float oracle_spline_erf_mirror(float x)
{
    const float inv_sign = -sign_of(x);
    x *= inv_sign; /* -|x| */
    if (x <= -2.9f) return -inv_sign;
    for (size_t i = 0; i < 4; ++i)
        if (x < ERF_PIECES[i].hi) return inv_sign * eval_piece(&ERF_PIECES[i], x);
    return inv_sign * eval_piece(&ERF_PIECES[4], x);
}

/* approx.cpp:71-80: odd Taylor series, terms up to x^19, clamped at |x| >= 2 */
float oracle_taylor_erf(float x)
{
    static const float ts[] = { 1.0f, -0.33333334f, 0.1f, -0.023809524f, 0.0046296297f, -0.00075757573f,
                                0.00010683761f, -1.3227514e-5f, 1.4589169e-6f, -1.4503853e-7f };
    if (x <= -2.f) return -1.f;
    if (x >= 2.f) return 1.f;
    float acc = ts[9];
    for (int i = 8; i >= 0; --i) acc = acc * x * x + ts[i];
    const float two_inv_sqrtpi = 2.f * 0.564189583547756286948f;
    return two_inv_sqrtpi * acc * x;
}

/* include/vectorclass/vectormath_exp.h:373-458, exp_f<VTYPE,0,0> for one lane:
 * r = round(x*log2e); Cody-Waite reduction with fused nmul_add (FMA builds);
 * degree-5 Taylor polynomial in Estrin form (vectormath_common.h:206-213);
 * scale by 2^r through the exponent field (vectormath_exp.h:84-92);
 * 0 / inf outside |x| < 87.3. */
float oracle_vcl_exp(float x0)
{
    const float P0 = 1.f / 2.f, P1 = 1.f / 6.f, P2 = 1.f / 24.f, P3 = 1.f / 120.f, P4 = 1.f / 720.f,
                P5 = 1.f / 5040.f;
    const float ln2f_hi = 0.693359375f, ln2f_lo = -2.12194440e-4f;
    const float max_x = 87.3f;
    if (isnan(x0)) return x0;
    if (!(fabsf(x0) < max_x)) return (x0 < 0) ? 0.f : INFINITY;
    const float r = nearbyintf(x0 * (float)1.44269504088896340736);
    float x = fmaf(-r, ln2f_hi, x0);
    x = fmaf(-r, ln2f_lo, x);
    const float x2 = x * x;
    const float x4 = x2 * x2;
    float z = fmaf(fmaf(P3, x, P2), x2, fmaf(fmaf(P5, x, P4), x4, fmaf(P1, x, P0)));
    z = fmaf(z, x2, x);
    union { float f; uint32_t u; } n2;
    n2.f = r + (127.0f + 8388608.0f);
    n2.u <<= 23;
    return (z + 1.0f) * n2.f;
}

/* approx.cpp:112-129 (Schraudolph); NDEBUG unset => out-of-range inputs clamped. */
float oracle_fast_exp(float x)
{
    const float a = (float)(1 << 23) / 0.693147180559945309417f;
    const float b = (float)(1 << 23) * (127 - 0.043677448f);
    const float c = (float)(1 << 23);
    const float d = (float)(1 << 23) * 255;
    x = a * x + b;
    if (x < c || x > d) x = (x < c) ? 0.f : d;
    union { float f; uint32_t u; } v;
    v.u = (uint32_t)x;
    return v.f;
}

/* approx.cpp:141-163 */
static const cubic_piece EXP_PIECES[] = {
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
#define N_EXP_PIECES (sizeof(EXP_PIECES) / sizeof(EXP_PIECES[0]))

float oracle_spline_exp(float x)
{
    if (x <= -9.0f) return 0.0f;
    for (size_t i = 0; i < N_EXP_PIECES; ++i)
        if (x < EXP_PIECES[i].hi) return eval_piece(&EXP_PIECES[i], x);
    return 1.0f;
}

float oracle_exp(int kind, float x)
{
    switch (kind) {
    case ORACLE_EXP_VCL: return oracle_vcl_exp(x);
    case ORACLE_EXP_FAST: return oracle_fast_exp(x);
    case ORACLE_EXP_SPLINE: return oracle_spline_exp(x);
    default: return expf(x);
    }
}

float oracle_erf(int kind, float x)
{
    switch (kind) {
    case ORACLE_ERF_AS: return oracle_as_erf(x);
    case ORACLE_ERF_SPLINE: return oracle_spline_erf(x);
    case ORACLE_ERF_SPLINE_MIRROR: return oracle_spline_erf_mirror(x);
    case ORACLE_ERF_TAYLOR: return oracle_taylor_erf(x);
    default: return erff(x);
    }
}

/* ------------------------------------------------------------------------- */
/* vec4 helpers: vrt/types.h:19-82 -- dot/sqnorm/normalize include w          */
/* ------------------------------------------------------------------------- */
static inline ovec4 v4(const float p[4]) { ovec4 r = { p[0], p[1], p[2], p[3] }; return r; }
static inline ovec4 v4sub(ovec4 a, ovec4 b) { ovec4 r = { a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w }; return r; }
static inline ovec4 v4add(ovec4 a, ovec4 b) { ovec4 r = { a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w }; return r; }
static inline ovec4 v4scale(ovec4 a, float l) { ovec4 r = { a.x * l, a.y * l, a.z * l, a.w * l }; return r; }
static inline float v4dot(ovec4 a, ovec4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
static inline ovec4 v4normalize(ovec4 a)
{
    const float norm = sqrtf(v4dot(a, a));
    ovec4 r = { a.x / norm, a.y / norm, a.z / norm, a.w / norm };
    return r;
}

/* vrt/types.h:204-208 gaussian_t::pdf */
static inline float pdf(const ogaussian *g, ovec4 x, int exp_kind)
{
    const ovec4 d = v4sub(x, g->mu);
    return g->magnitude * oracle_exp(exp_kind, -(v4dot(d, d)) / (2 * g->sigma * g->sigma));
}

/* ------------------------------------------------------------------------- */
/* transmittance / radiance                                                  */
/* ------------------------------------------------------------------------- */

/* vrt/rt.h:32-54 */
static float transmittance(ovec4 o, ovec4 n, float s, const ogaussian *g, size_t ng, int exp_kind,
                           int erf_kind)
{
    float T = 0.f;
    for (size_t q = 0; q < ng; ++q) {
        const ogaussian *gq = &g[q];
        const ovec4 oc = v4sub(gq->mu, o);
        const float mu_bar = v4dot(oc, n);
        const float oc_sqnorm = v4dot(oc, oc);
        const float mb2 = mu_bar * mu_bar;
        const float inv_2_sigma2 = 1.f / (2.f * gq->sigma * gq->sigma);
        const float c_bar = gq->magnitude * oracle_exp(exp_kind, -((oc_sqnorm - mb2) * inv_2_sigma2));
        const float sqrt_2_sig = SQRT_2 * gq->sigma;
        const float mu_bar_n = mu_bar / sqrt_2_sig;
        const float s_n = s / sqrt_2_sig;
        const float erf1 = oracle_erf(erf_kind, -mu_bar_n);
        const float erf2 = oracle_erf(erf_kind, s_n - mu_bar_n);
        T += gq->sigma * c_bar * INV_SQRT_2_PI * (erf1 - erf2);
    }
    return oracle_exp(exp_kind, T);
}

float oracle_transmittance(const float o[4], const float n[4], float s, const ogaussian *g, size_t ng,
                           int exp_kind, int erf_kind)
{
    return transmittance(v4(o), v4(n), s, g, ng, exp_kind, erf_kind);
}

/* vrt/rt.cpp:8-17 */
float oracle_transmittance_step(const float o[4], const float n[4], float s, float delta,
                                const ogaussian *g, size_t ng)
{
    float T = 0.f;
    for (float t = 0; t <= s; t += delta)
        for (size_t q = 0; q < ng; ++q)
            T += delta * pdf(&g[q], v4add(v4(o), v4scale(v4(n), t)), ORACLE_EXP_LIBM);
    return oracle_fast_exp(-T);
}

/* vrt/rt.cpp:19-27 */
float oracle_density(const float pt[4], const ogaussian *g, size_t ng)
{
    float D = 0.f;
    for (size_t q = 0; q < ng; ++q) D += pdf(&g[q], v4(pt), ORACLE_EXP_LIBM);
    return D;
}

/* vrt/rt.h:146-164 (scalar) == rt.h:205-223 (one SIMD lane) */
static ovec4 radiance(ovec4 o, ovec4 n, const ogaussian *g, size_t ng, int exp_kind, int erf_kind)
{
    ovec4 L_hat = { 0.f, 0.f, 0.f, 0.f };
    for (size_t i = 0; i < ng; ++i) {
        const ogaussian *gq = &g[i];
        const float lambda_q = gq->sigma;
        float inner = 0.f;
        for (int k = -4; k <= 0; ++k) {
            const float s = v4dot(v4sub(gq->mu, o), n) + k * lambda_q;
            const float T = transmittance(o, n, s, g, ng, exp_kind, erf_kind);
            inner += pdf(gq, v4add(o, v4scale(n, s)), exp_kind) * T * lambda_q;
        }
        L_hat = v4add(L_hat, v4scale(gq->albedo, inner));
    }
    return L_hat;
}

void oracle_radiance(const float o[4], const float n[4], const ogaussian *g, size_t ng, int exp_kind,
                     int erf_kind, float out[4])
{
    const ovec4 r = radiance(v4(o), v4(n), g, ng, exp_kind, erf_kind);
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}

/* ------------------------------------------------------------------------- */
/* scene producers                                                           */
/* ------------------------------------------------------------------------- */

/* volumetric-ray-tracer/main.cpp:194-205 */
size_t oracle_grid_scene(unsigned grid_dim_in, ogaussian *out)
{
    const uint8_t grid_dim = (uint8_t)grid_dim_in; /* main.cpp:196 */
    size_t n = 0;
    for (uint8_t i = 0; i < grid_dim; ++i)
        for (uint8_t j = 0; j < grid_dim; ++j) {
            ogaussian g;
            const float q = (i * grid_dim + j) / (float)(grid_dim * grid_dim);
            g.albedo.x = 1.f - q; g.albedo.y = 0.f; g.albedo.z = 0.f + q; g.albedo.w = 1.f;
            g.mu.x = -1.f + 1.f / grid_dim + i * 1.f / (grid_dim / 2.f);
            g.mu.y = -1.f + 1.f / grid_dim + j * 1.f / (grid_dim / 2.f);
            g.mu.z = 1.f; g.mu.w = 0.f;
            g.sigma = 1.f / (2 * grid_dim);
            g.magnitude = 1.f;
            out[n++] = g;
        }
    return n;
}

/* vrt/gaussians-from-file.cpp:7-44: every "v x y z" line becomes a Gaussian;
 * sigma by vertex count, albedo = normalize(v)*0.5 + (0.5,0.5,0.5,1). */
long oracle_read_obj(const char *path, ogaussian *out, size_t cap)
{
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    char line[1024];
    long n = 0;
    /* pass 1: count (sigma depends on the total) */
    while (fgets(line, sizeof line, f))
        if (line[0] == 'v' && (line[1] == ' ' || line[1] == '\t')) ++n;
    if (!out) { fclose(f); return n; }
    const float sig = (n < 300) ? 0.3f : (n < 1000) ? 0.15f : 0.05f;
    rewind(f);
    long k = 0;
    while (fgets(line, sizeof line, f)) {
        if (!(line[0] == 'v' && (line[1] == ' ' || line[1] == '\t'))) continue;
        if ((size_t)k >= cap) break;
        char *p = line + 1;
        float v[3] = { 0, 0, 0 };
        for (int c = 0; c < 3; ++c) v[c] = (float)strtod(p, &p);
        ovec4 pt = { v[0], v[1], v[2], 0.0f };
        ovec4 c = v4normalize(pt);
        ogaussian g;
        const ovec4 half = { 0.5f, 0.5f, 0.5f, 1.0f };
        g.albedo = v4add(v4scale(c, 0.5f), half);
        g.mu = pt;
        g.sigma = sig;
        g.magnitude = 1.0f;
        out[k++] = g;
    }
    fclose(f);
    return k;
}

/* ------------------------------------------------------------------------- */
/* glm restatement (the reference links glm, un-vendored; algorithms as       */
/* published in glm 0.9.9/1.0: func_geometric.inl, matrix_transform.inl,      */
/* func_matrix.inl).  Column-major: m[col*4 + row].                           */
/* ------------------------------------------------------------------------- */
static inline float g_radians(float deg) { return deg * 0.01745329251994329576923690768489f; }
static inline float dot3(const float a[3], const float b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void cross3(const float a[3], const float b[3], float r[3])
{
    r[0] = a[1] * b[2] - b[1] * a[2];
    r[1] = a[2] * b[0] - b[2] * a[0];
    r[2] = a[0] * b[1] - b[0] * a[1];
}
static inline void normalize3(const float a[3], float r[3])
{
    const float inv = 1.f / sqrtf(dot3(a, a));
    r[0] = a[0] * inv; r[1] = a[1] * inv; r[2] = a[2] * inv;
}

/* glm::lookAtRH */
static void look_at_rh(const float eye[3], const float center[3], const float up[3], float m[16])
{
    float d[3] = { center[0] - eye[0], center[1] - eye[1], center[2] - eye[2] }, f[3], s[3], sc[3], u[3];
    normalize3(d, f);
    cross3(f, up, sc);
    normalize3(sc, s);
    cross3(s, f, u);
    memset(m, 0, 16 * sizeof(float));
    m[0] = s[0]; m[4] = s[1]; m[8] = s[2];
    m[1] = u[0]; m[5] = u[1]; m[9] = u[2];
    m[2] = -f[0]; m[6] = -f[1]; m[10] = -f[2];
    m[12] = -dot3(s, eye); m[13] = -dot3(u, eye); m[14] = dot3(f, eye);
    m[15] = 1.f;
}

/* glm::translate(m, v): m[3] = m[0]*v.x + m[1]*v.y + m[2]*v.z + m[3] */
static void translate(float m[16], const float v[3])
{
    for (int r = 0; r < 4; ++r) m[12 + r] = m[0 + r] * v[0] + m[4 + r] * v[1] + m[8 + r] * v[2] + m[12 + r];
}

/* glm mat4 * vec4: (m0*v0 + m1*v1) + (m2*v2 + m3*v3) */
static void mat_vec(const float m[16], const float v[4], float r[4])
{
    for (int i = 0; i < 4; ++i) r[i] = (m[i] * v[0] + m[4 + i] * v[1]) + (m[8 + i] * v[2] + m[12 + i] * v[3]);
}

/* glm::inverse(mat4) -- cofactor expansion (func_matrix.inl compute_inverse<4,4>) */
static void inverse4(const float a[16], float out[16])
{
#define M(c, r) a[(c) * 4 + (r)]
    const float c00 = M(2, 2) * M(3, 3) - M(3, 2) * M(2, 3), c02 = M(1, 2) * M(3, 3) - M(3, 2) * M(1, 3),
                c03 = M(1, 2) * M(2, 3) - M(2, 2) * M(1, 3);
    const float c04 = M(2, 1) * M(3, 3) - M(3, 1) * M(2, 3), c06 = M(1, 1) * M(3, 3) - M(3, 1) * M(1, 3),
                c07 = M(1, 1) * M(2, 3) - M(2, 1) * M(1, 3);
    const float c08 = M(2, 1) * M(3, 2) - M(3, 1) * M(2, 2), c10 = M(1, 1) * M(3, 2) - M(3, 1) * M(1, 2),
                c11 = M(1, 1) * M(2, 2) - M(2, 1) * M(1, 2);
    const float c12 = M(2, 0) * M(3, 3) - M(3, 0) * M(2, 3), c14 = M(1, 0) * M(3, 3) - M(3, 0) * M(1, 3),
                c15 = M(1, 0) * M(2, 3) - M(2, 0) * M(1, 3);
    const float c16 = M(2, 0) * M(3, 2) - M(3, 0) * M(2, 2), c18 = M(1, 0) * M(3, 2) - M(3, 0) * M(1, 2),
                c19 = M(1, 0) * M(2, 2) - M(2, 0) * M(1, 2);
    const float c20 = M(2, 0) * M(3, 1) - M(3, 0) * M(2, 1), c22 = M(1, 0) * M(3, 1) - M(3, 0) * M(1, 1),
                c23 = M(1, 0) * M(2, 1) - M(2, 0) * M(1, 1);
    const float f0[4] = { c00, c00, c02, c03 }, f1[4] = { c04, c04, c06, c07 }, f2[4] = { c08, c08, c10, c11 };
    const float f3[4] = { c12, c12, c14, c15 }, f4[4] = { c16, c16, c18, c19 }, f5[4] = { c20, c20, c22, c23 };
    const float v0[4] = { M(1, 0), M(0, 0), M(0, 0), M(0, 0) }, v1[4] = { M(1, 1), M(0, 1), M(0, 1), M(0, 1) };
    const float v2[4] = { M(1, 2), M(0, 2), M(0, 2), M(0, 2) }, v3[4] = { M(1, 3), M(0, 3), M(0, 3), M(0, 3) };
    const float sa[4] = { +1, -1, +1, -1 }, sb[4] = { -1, +1, -1, +1 };
    float inv[16];
    for (int i = 0; i < 4; ++i) {
        inv[0 * 4 + i] = (v1[i] * f0[i] - v2[i] * f1[i] + v3[i] * f2[i]) * sa[i];
        inv[1 * 4 + i] = (v0[i] * f0[i] - v2[i] * f3[i] + v3[i] * f4[i]) * sb[i];
        inv[2 * 4 + i] = (v0[i] * f1[i] - v1[i] * f3[i] + v3[i] * f5[i]) * sa[i];
        inv[3 * 4 + i] = (v0[i] * f2[i] - v1[i] * f4[i] + v2[i] * f5[i]) * sb[i];
    }
    const float d0 = M(0, 0) * inv[0], d1 = M(0, 1) * inv[4], d2 = M(0, 2) * inv[8], d3 = M(0, 3) * inv[12];
    const float one_over_det = 1.f / ((d0 + d1) + (d2 + d3));
    for (int i = 0; i < 16; ++i) out[i] = inv[i] * one_over_det;
#undef M
}

/* ------------------------------------------------------------------------- */
/* camera                                                                    */
/* ------------------------------------------------------------------------- */

/* vrt/camera.cpp:52 */
static void camera_update_view(ocamera *c)
{
    float center[3] = { c->position[0] + c->front[0], c->position[1] + c->front[1], c->position[2] + c->front[2] };
    look_at_rh(c->position, center, c->up, c->view);
    const float t[3] = { c->focal_length * c->front[0], c->focal_length * c->front[1], c->focal_length * c->front[2] };
    translate(c->view, t);
}

/* vrt/camera.cpp:7-23 */
void oracle_camera_turn(ocamera *c, float yaw, float pitch)
{
    float p = pitch;
    p = (p > 89.f) ? 89.f : p;
    p = (p < -89.f) ? -89.f : p;
    const float f[3] = { cosf(g_radians(yaw)) * cosf(g_radians(p)), sinf(g_radians(p)),
                         sinf(g_radians(yaw)) * cosf(g_radians(p)) };
    float t[3];
    normalize3(f, c->front);
    cross3(c->front, c->world_up, t);
    normalize3(t, c->right);
    cross3(c->right, c->front, t);
    normalize3(t, c->up);
    camera_update_view(c);
}

/* vrt/camera.cpp:25-36 */
void oracle_camera_init(ocamera *c, const float position[3], const float up[3], const float front[3],
                        float yaw, float pitch, uint64_t w, uint64_t h, float focal_length)
{
    memset(c, 0, sizeof *c);
    memcpy(c->position, position, 3 * sizeof(float));
    memcpy(c->up, up, 3 * sizeof(float));
    memcpy(c->world_up, up, 3 * sizeof(float));
    memcpy(c->front, front, 3 * sizeof(float));
    c->focal_length = focal_length;
    c->w = w;
    c->h = h;
    oracle_camera_turn(c, yaw, pitch);
}

/* vrt/camera.cpp:60-69 */
void oracle_camera_plane(const ocamera *c, float *xs, float *ys, float *zs)
{
    float inv[16];
    inverse4(c->view, inv);
    for (uint64_t i = 0; i < c->h; ++i)
        for (uint64_t j = 0; j < c->w; ++j) {
            const float v[4] = { -1.f + j / (c->w / 2.f), -1.f + i / (c->h / 2.f), 0.f, 1.f };
            float pt[4];
            mat_vec(inv, v, pt);
            xs[i * c->w + j] = pt[0];
            ys[i * c->w + j] = pt[1];
            zs[i * c->w + j] = pt[2];
        }
}

/* volumetric-ray-tracer/main.cpp:330-334 (and 252-255 for the initial rotation):
 * position = rotate(I, radians(deg), +Y) * (position, 1); angle -= deg; turn(angle, 0). */
void oracle_orbit_step(ocamera *c, float *angle, float deg)
{
    const float a = g_radians(deg);
    const float cs = cosf(a), sn = sinf(a);
    const float one_minus_c = 1.f - cs;
    /* glm::rotate with axis (0,1,0): temp = (1-c)*axis */
    const float ax[3] = { 0.f, 1.f, 0.f };
    const float tmp[3] = { one_minus_c * ax[0], one_minus_c * ax[1], one_minus_c * ax[2] };
    float R[16];
    memset(R, 0, sizeof R);
    R[0] = cs + tmp[0] * ax[0];          R[1] = tmp[0] * ax[1] + sn * ax[2];  R[2] = tmp[0] * ax[2] - sn * ax[1];
    R[4] = tmp[1] * ax[0] - sn * ax[2];  R[5] = cs + tmp[1] * ax[1];          R[6] = tmp[1] * ax[2] + sn * ax[0];
    R[8] = tmp[2] * ax[0] + sn * ax[1];  R[9] = tmp[2] * ax[1] - sn * ax[0];  R[10] = cs + tmp[2] * ax[2];
    R[15] = 1.f;
    const float p[4] = { c->position[0], c->position[1], c->position[2], 1.f };
    float q[4];
    mat_vec(R, p, q);
    c->position[0] = q[0]; c->position[1] = q[1]; c->position[2] = q[2];
    *angle -= deg;
    oracle_camera_turn(c, *angle, 0.f);
}

/* ------------------------------------------------------------------------- */
/* tiling: vrt/rt.cpp:29-69                                                  */
/* ------------------------------------------------------------------------- */
size_t oracle_tile_gaussians(float tw, float th, const ogaussian *g, size_t ng, const float view[16],
                             uint64_t *tiles_w, uint64_t *tiles_h, uint32_t **offsets_out,
                             uint32_t **indices_out)
{
    float *pmx = malloc(sizeof(float) * (ng + 1)), *pmy = malloc(sizeof(float) * (ng + 1)),
          *psig = malloc(sizeof(float) * (ng + 1));
    uint32_t *idxs = malloc(sizeof(uint32_t) * (ng + 1));
    size_t np = 0;
    for (size_t i = 0; i < ng; ++i) {
        const float v[4] = { g[i].mu.x, g[i].mu.y, g[i].mu.z, 1.f };
        float proj[4];
        mat_vec(view, v, proj);
        if (proj[2] < 1.f) continue;
        const float sigma = g[i].sigma / proj[2];
        if (sigma < 1e-5f) continue;
        pmx[np] = proj[0] / proj[2];
        pmy[np] = proj[1] / proj[2];
        psig[np] = sigma;
        idxs[np++] = (uint32_t)i;
    }
    /* types.h:280 */
    *tiles_w = (uint64_t)ceilf(2.f / tw);
    *tiles_h = (uint64_t)ceilf(2.f / th);
    size_t ntiles_loop = 0;
    for (float y = -1.f + th / 2; y < 1.f; y += th)
        for (float x = -1.f + tw / 2; x < 1.f; x += tw) ++ntiles_loop;
    const size_t ntiles_decl = (size_t)(*tiles_w * *tiles_h);
    const size_t ntiles = ntiles_loop > ntiles_decl ? ntiles_loop : ntiles_decl;
    uint32_t *offsets = calloc(ntiles + 1, sizeof(uint32_t));
    size_t cap = 1024, cnt = 0;
    uint32_t *indices = malloc(cap * sizeof(uint32_t));
    size_t t = 0;
    for (float y = -1.f + th / 2; y < 1.f; y += th)
        for (float x = -1.f + tw / 2; x < 1.f; x += tw) {
            offsets[t] = (uint32_t)cnt;
            for (size_t i = 0; i < np; ++i) {
                const float px = fabsf(x - pmx[i]), py = fabsf(y - pmy[i]);
                if (px <= fabsf(x) + tw / 2 + 3.3f * psig[i] && py <= fabsf(y) + th / 2 + 3.3f * psig[i]) {
                    if (cnt == cap) { cap *= 2; indices = realloc(indices, cap * sizeof(uint32_t)); }
                    indices[cnt++] = idxs[i];
                }
            }
            ++t;
        }
    for (; t <= ntiles; ++t) offsets[t] = (uint32_t)cnt;
    free(pmx); free(pmy); free(psig); free(idxs);
    *offsets_out = offsets;
    *indices_out = indices;
    return ntiles_loop;
}

void oracle_free(void *p) { free(p); }

/* ------------------------------------------------------------------------- */
/* render drivers                                                            */
/* ------------------------------------------------------------------------- */

/* rt.h:239-243 (trunc, opaque) / rt.h:329-333 (round, opaque) / rt.h:373-377 (round, computed alpha) */
uint32_t oracle_pack_pixel(const float c[4], int flags)
{
    uint32_t A, R, G, B;
    if (flags & ORACLE_PACK_ROUND) {
        R = (uint32_t)lrintf(fminf(c[0], 1.0f) * 255.f);
        G = (uint32_t)lrintf(fminf(c[1], 1.0f) * 255.f);
        B = (uint32_t)lrintf(fminf(c[2], 1.0f) * 255.f);
    } else {
        R = (uint32_t)(fminf(c[0], 1.0f) * 255);
        G = (uint32_t)(fminf(c[1], 1.0f) * 255);
        B = (uint32_t)(fminf(c[2], 1.0f) * 255);
    }
    if (flags & ORACLE_ALPHA_COMPUTED)
        A = (uint32_t)lrintf(fminf(1.f, c[3]) * 255.f) << 24;
    else
        A = 0xFF000000u;
    return A | R << 16 | G << 8 | B;
}

/* rt.h:231-237: dir = plane[i] - origin (w = 0 - origin.w), normalize, Radiance */
static ovec4 shade(size_t i, const float *xs, const float *ys, const float *zs, ovec4 origin,
                   const ogaussian *g, size_t ng, int exp_kind, int erf_kind)
{
    ovec4 p = { xs[i], ys[i], zs[i], 0.f };
    ovec4 dir = v4normalize(v4sub(p, origin));
    return radiance(origin, dir, g, ng, exp_kind, erf_kind);
}

/* rt.h:227-247, rt.h:315-337 */
void oracle_render_image(uint32_t w, uint32_t h, uint32_t *image, float *radiance_out, const float *xs,
                         const float *ys, const float *zs, const float origin[4], const ogaussian *g,
                         size_t ng, int exp_kind, int erf_kind, int pack_flags, const uint32_t *pixels,
                         size_t npix, int threads)
{
    const size_t total = pixels ? npix : (size_t)w * h;
    const ovec4 o = v4(origin);
    (void)threads;
#pragma omp parallel for schedule(dynamic, 16) num_threads(threads > 0 ? threads : 1)
    for (size_t q = 0; q < total; ++q) {
        const size_t i = pixels ? pixels[q] : q;
        const ovec4 c = shade(i, xs, ys, zs, o, g, ng, exp_kind, erf_kind);
        const float cf[4] = { c.x, c.y, c.z, c.w };
        if (radiance_out) memcpy(radiance_out + 4 * q, cf, sizeof cf);
        if (image) image[i] = oracle_pack_pixel(cf, pack_flags);
    }
}

/* rt.h:251-310, rt.h:344-404.  Tile geometry: tile_width = (u64)(width*tw/2.f) (rt.h:348),
 * raster index i = tx*tile_width + lx + (tile_width*tiles_w)*(ly + ty*tile_height) (rt.h:364-365). */
void oracle_render_image_tiled(uint32_t w, uint32_t h, uint32_t *image, float *radiance_out,
                               const float *xs, const float *ys, const float *zs, const float origin[4],
                               const ogaussian *g, size_t ng, float tw, float th, uint64_t tiles_w,
                               uint64_t tiles_h, const uint32_t *offsets, const uint32_t *indices,
                               int exp_kind, int erf_kind, int pack_flags, const uint32_t *pixels,
                               size_t npix, int threads)
{
    const uint64_t tile_width = (uint64_t)(w * tw / 2.f);
    const uint64_t tile_height = (uint64_t)(h * th / 2.f);
    const uint64_t stride = tile_width * tiles_w;
    const size_t total = pixels ? npix : (size_t)(stride * tile_height * tiles_h);
    const ovec4 o = v4(origin);
    (void)ng; (void)threads;
    /* per-tile AoS copies, like tiles_t (types.h:272-287) */
    const size_t ntiles = (size_t)(tiles_w * tiles_h);
    ogaussian **sets = malloc(ntiles * sizeof(*sets));
    for (size_t t = 0; t < ntiles; ++t) {
        const uint32_t cnt = offsets[t + 1] - offsets[t];
        sets[t] = malloc((cnt ? cnt : 1) * sizeof(ogaussian));
        for (uint32_t k = 0; k < cnt; ++k) sets[t][k] = g[indices[offsets[t] + k]];
    }
#pragma omp parallel for schedule(dynamic, 16) num_threads(threads > 0 ? threads : 1)
    for (size_t q = 0; q < total; ++q) {
        const size_t i = pixels ? pixels[q] : q;
        const uint64_t row = i / stride, col = i % stride;
        const uint64_t tx = col / tile_width, ty = row / tile_height;
        if (ty >= tiles_h || tile_width == 0 || i >= (size_t)w * h) { /* a listed index outside the tile grid: not a pixel the reference renders */
            const float z[4] = { 0.f, 0.f, 0.f, 0.f };
            if (radiance_out) memcpy(radiance_out + 4 * q, z, sizeof z);
            continue;
        }
        const size_t t = (size_t)(ty * tiles_w + tx);
        const ovec4 c = shade(i, xs, ys, zs, o, sets[t], offsets[t + 1] - offsets[t], exp_kind, erf_kind);
        const float cf[4] = { c.x, c.y, c.z, c.w };
        if (radiance_out) memcpy(radiance_out + 4 * q, cf, sizeof cf);
        if (image) image[i] = oracle_pack_pixel(cf, pack_flags);
    }
    for (size_t t = 0; t < ntiles; ++t) free(sets[t]);
    free(sets);
}
