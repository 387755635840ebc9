// vrt_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the volumetric Gaussian ray tracer.
// No MFMA: the path is VALU + quarter-rate transcendental bound (one v_rcp_f32 per Abramowitz-Stegun
// erf term); Gaussian parameters reach the inner loops through LDS rows or wave-uniform scalar loads.
//
// Reference semantics (paths relative to /root/reference/src):
//   L(ray) = sum_i albedo_i * sum_{k=-4..0} pdf_i(o + n s_ik) * T(s_ik) * sigma_i,
//            s_ik = (mu_i - o).n + k sigma_i                               (vrt/rt.h:205-223)
//   T(s)   = Exp( sum_j sigma_j cbar_j K (Erf(-mubar_j/(sqrt2 sigma_j)) - Erf((s - mubar_j)/(sqrt2 sigma_j))) )
//            cbar_j = mag_j Exp(-(|oc_j|^2 - mubar_j^2)/(2 sigma_j^2)), K = 1/0.79788456  (vrt/rt.h:102-127)
// evaluated by the reference with an O(5 N^2) double loop per ray that recomputes cbar_j, mubar_j
// and Erf(-m_j) for every (i, k, j).  Here:
//   * culling in four levels -- reference tile (rt.cpp:29-69) ^ tile cone, 32x32-px cell cone, 8x8-px block
//     cone (all conservative, `cone_keeps`), then the exact per-ray criterion sigma*mag*exp(-x) >= cull_eps;
//   * hoisting -- A_j = K sigma_j cbar_j, m_j = mubar_j r_j and E_j = Erf(-m_j) depend on the ray but not on
//     the sample point: T(s_ik) = Exp(sum_j A_j (E_j - Erf(s_ik r_j - m_j))), summed per term like the reference;
//   * register blocking -- EC emitters x 5 samples = 5*EC running sums per lane while the absorbers stream by;
//     (A_j, m_j, E_j) are recomputed per (ray, j, chunk) and amortised over the 5*EC terms, so nothing per-ray
//     is ever stored;
//   * two shading kernels -- one wavefront per 8x8 block with per-ray candidate lists (sparse scenes), one
//     16-wave workgroup per block with depth-sorted candidates and exact saturation skipping (dense scenes).
// DESIGN.md section 4 has the table of kernels and their measured costs.
#include <hip/hip_fp16.h>
#include "vrt_kernels.h"
#include "vrt_device_math.h"

namespace vrtk {

// Uniform (scalar) 16-byte load: constant address space + a wave-uniform index => s_load_dwordx4.
typedef float vf4 __attribute__((ext_vector_type(4)));
typedef const vf4 __attribute__((address_space(4))) *cf4ptr;
__device__ __forceinline__ float4 uload(const float4 *base, uint32_t idx)
{
    const vf4 v = ((cf4ptr)(const void *)base)[idx];
    return make_float4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ float wave_min(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
    return v;
}

__device__ __forceinline__ uint32_t pack_pixel(float r, float g, float b, float a, int flags)
{
    uint32_t R, G, B, A;
    if (flags & VRT_PACK_ROUND) {
        R = (uint32_t)__float2int_rn(fminf(r, 1.f) * 255.f);
        G = (uint32_t)__float2int_rn(fminf(g, 1.f) * 255.f);
        B = (uint32_t)__float2int_rn(fminf(b, 1.f) * 255.f);
    } else {
        R = (uint32_t)(fminf(r, 1.f) * 255.f);
        G = (uint32_t)(fminf(g, 1.f) * 255.f);
        B = (uint32_t)(fminf(b, 1.f) * 255.f);
    }
    if (flags & VRT_ALPHA_COMPUTED) A = ((uint32_t)__float2int_rn(fminf(1.f, a) * 255.f)) << 24;
    else A = 0xFF000000u;
    return A | (R << 16) | (G << 8) | B;
}

// ---------------------------------------------------------------------------------------------
// Shading core shared by the image kernel (uniform origin: oc comes from the per-frame table)
// and the arbitrary-ray kernel (per-lane origin: oc = mu - o).
// ---------------------------------------------------------------------------------------------
struct LaneRay { float nx, ny, nz, ox, oy, oz; };

// The reference forms cbar_j from |oc|^2 - mubar^2 (rt.h:110-116): a difference of two numbers of size
// |oc|^2 ~ 25 whose result is ~sigma^2, so its fp32 rounding noise (a few 1e-6 absolute, times
// 1/(2 sigma^2) up to ~1e4) is far above 1 ulp of the result and shows up in the image at the
// 1e-4 level for small sigma.  Parity therefore needs the reference's operations in the reference's
// order, unfused -- not a "more accurate" formula.  These helpers pin that order (vec4f_t::dot,
// types.h:54-57: ((x*x' + y*y') + z*z') + w*w', the w terms being exactly 0 here).
__device__ __forceinline__ float dot3_ref(float ax, float ay, float az, float bx, float by, float bz)
{
#pragma clang fp contract(off)
    return ((ax * bx + ay * by) + az * bz);
}
__device__ __forceinline__ float sub_ref(float a, float b)
{
#pragma clang fp contract(off)
    return a - b;
}
__device__ __forceinline__ float add_ref(float a, float b)
{
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ float mul_ref(float a, float b)
{
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ float madd_ref(float a, float b, float c) // a*b + c, two roundings
{
#pragma clang fp contract(off)
    return a * b + c;
}

template <bool UNIFORM_ORIGIN>
__device__ __forceinline__ void ray_gaussian(const SceneTables &S, uint32_t idx, const LaneRay &ray, float &mubar,
                                             float &d2)
{
    if constexpr (UNIFORM_ORIGIN) {
        const float4 a = uload(S.gA, idx); // (oc, |oc|^2) from prep_frame_kernel, reference order
        mubar = dot3_ref(a.x, a.y, a.z, ray.nx, ray.ny, ray.nz);
        d2 = sub_ref(a.w, mul_ref(mubar, mubar));
    } else {
        const float4 m = uload(S.mu_sig, idx);
        const float cx = m.x - ray.ox, cy = m.y - ray.oy, cz = m.z - ray.oz;
        mubar = dot3_ref(cx, cy, cz, ray.nx, ray.ny, ray.nz);
        d2 = sub_ref(dot3_ref(cx, cy, cz, cx, cy, cz), mul_ref(mubar, mubar));
    }
}

// One emission sample: pdf * T * sigma = q Exp(-x_pdf) Exp(acc).  For the accurate Exp variants the two
// exponentials are merged into one (Exp(a)Exp(b) = Exp(a+b) to ~1e-7 relative); the approximating variants
// (fast_exp, spline_exp) are not multiplicative and keep the reference's two calls.
template <int EXP>
__device__ __forceinline__ float emission_term(float q, float x_pdf, float acc)
{
    if constexpr (EXP == VRT_EXP_VCL || EXP == VRT_EXP_LIBM) return q * vexp<EXP>(acc - x_pdf);
    else return q * vexp<EXP>(-x_pdf) * vexp<EXP>(acc);
}

// list: wave-uniform index list (LDS or global, read through a flat pointer); n entries.
// Emitter chunks i_start, i_start + i_step, ... (default: all of them) -- a workgroup can deal them to its waves.
template <int EXP, int ERF, int EC, bool UNIFORM_ORIGIN>
__device__ __forceinline__ void shade_list(const SceneTables &S, const uint32_t *list, uint32_t n, const LaneRay &ray,
                                           float &Lr, float &Lg, float &Lb, float &La, uint32_t i_start = 0,
                                           uint32_t i_step = EC)
{
    Lr = Lg = Lb = La = 0.f;
    if (n == 0) return;
    const ErfEval<ERF> erf;

    for (uint32_t i0 = i_start; i0 < n; i0 += i_step) {
        // emitter chunk set-up
        float e_mubar[EC];
        float e_sigma[EC]; // wave-uniform
        uint32_t e_idx[EC];
#pragma unroll
        for (int e = 0; e < EC; ++e) {
            const uint32_t jj = (i0 + e < n) ? (i0 + e) : i0; // pad the tail with a duplicate; masked below
            e_idx[e] = __builtin_amdgcn_readfirstlane(list[jj]);
            float d2;
            ray_gaussian<UNIFORM_ORIGIN>(S, e_idx[e], ray, e_mubar[e], d2);
            e_sigma[e] = uload(S.gD, e_idx[e]).x;
        }
        float acc[EC][5];
#pragma unroll
        for (int e = 0; e < EC; ++e)
#pragma unroll
            for (int k = 0; k < 5; ++k) acc[e][k] = 0.f;

        // absorber stream: 5*EC erf terms per (ray, j)
        for (uint32_t j = 0; j < n; ++j) {
            const uint32_t idx = __builtin_amdgcn_readfirstlane(list[j]);
            float mubar, d2;
            ray_gaussian<UNIFORM_ORIGIN>(S, idx, ray, mubar, d2);
            const float4 b = uload(S.gB, idx);
            const float A = b.z * vexp<EXP>(-(d2 * b.y));
            const float m = mubar * b.x;
            const float E = erf(-m); // Erf(-mubar_j / (sqrt2 sigma_j)), rt.h:122
#pragma unroll
            for (int e = 0; e < EC; ++e) {
                const float base = __builtin_fmaf(e_mubar[e], b.x, -m); // (mubar_i - mubar_j) r_j
                const float step = e_sigma[e] * b.x;                     // sigma_i r_j
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    const float x = __builtin_fmaf((float)(k - 4), step, base);
                    // rt.h:124: T += sigma cbar K (erf1 - erf2).  Summing the per-term DIFFERENCE like the
                    // reference (not C - sum A erf2) keeps the running sum small in optically thick scenes,
                    // where saturated pairs cancel exactly.
                    acc[e][k] = __builtin_fmaf(A, E - erf(x), acc[e][k]);
                }
            }
        }

        // emission: pdf_i(o + n s_ik) = mag_i Exp(-|o + n s_ik - mu_i|^2 / (2 sigma_i^2)), formed from the
        // sample POINT like the reference (rt.h:216-218, types.h:204-208) -- not from d2_i + k^2 sigma^2,
        // whose |oc|^2 - mubar^2 carries the cancellation noise described above.
#pragma unroll
        for (int e = 0; e < EC; ++e) {
            if (i0 + e < n) {
                const float4 ms = uload(S.mu_sig, e_idx[e]);
                const float4 bq = uload(S.gB, e_idx[e]);
                const float q = uload(S.gD, e_idx[e]).y; // sigma * mag
                float inner = 0.f;
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    const float sk = madd_ref((float)(k - 4), ms.w, e_mubar[e]);      // s = mubar_i + k sigma_i
                    const float px = sub_ref(madd_ref(ray.nx, sk, ray.ox), ms.x);     // (o + n s) - mu
                    const float py = sub_ref(madd_ref(ray.ny, sk, ray.oy), ms.y);
                    const float pz = sub_ref(madd_ref(ray.nz, sk, ray.oz), ms.z);
                    const float dd = dot3_ref(px, py, pz, px, py, pz);
                    inner += emission_term<EXP>(q, dd * bq.y, acc[e][k]);
                }
                const float4 alb = uload(S.gC, e_idx[e]);
                Lr = __builtin_fmaf(alb.x, inner, Lr);
                Lg = __builtin_fmaf(alb.y, inner, Lg);
                Lb = __builtin_fmaf(alb.z, inner, Lb);
                La = __builtin_fmaf(alb.w, inner, La);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Fast path of the image kernel: the block's candidates sit in LDS (index + the two parameter rows the
// inner loop needs) and every lane walks ITS OWN list of them -- the candidates whose sigma*mag*exp(-x)
// reaches cull_eps on that lane's ray.  In sparse scenes (sigma of a pixel or two) a ray meets a third of
// its block's candidates, and the pair loop is quadratic in the list length.  All lanes run the loops to
// the longest lane list; a lane past the end of its list adds exact zeros (A = 0).
// ---------------------------------------------------------------------------------------------
// One chunk of EC emitters (list positions i0 .. i0+EC-1 of every lane) against the lane's whole list.
template <int EXP, int ERF, int EC>
__device__ __forceinline__ void shade_chunk(const float4 *s_A, const float4 *s_B, const float4 *s_M, const float4 *s_C,
                                            const float *s_q, const uint8_t *s_lane /*[k*64 + lane]*/, uint32_t nl,
                                            uint32_t nmax, uint32_t lane, const LaneRay &ray, uint32_t i0, float &Lr,
                                            float &Lg, float &Lb, float &La)
{
    const ErfEval<ERF> erf;
    float e_mubar[EC], e_sigma[EC];
    uint32_t e_li[EC];
#pragma unroll
    for (int e = 0; e < EC; ++e) {
        const bool ve = i0 + e < nl;
        e_li[e] = ve ? s_lane[(i0 + e) * 64 + lane] : 0u;
        const float4 a = s_A[e_li[e]];
        e_mubar[e] = dot3_ref(a.x, a.y, a.z, ray.nx, ray.ny, ray.nz);
        e_sigma[e] = s_M[e_li[e]].w;
    }
    float acc[EC][5];
#pragma unroll
    for (int e = 0; e < EC; ++e)
#pragma unroll
        for (int k = 0; k < 5; ++k) acc[e][k] = 0.f;

    // absorber stream over this lane's list; next entry's LDS rows are fetched one iteration ahead
    uint32_t lj = nl ? s_lane[lane] : 0u;
    float4 a = s_A[lj], b = s_B[lj];
    for (uint32_t j = 0; j < nmax; ++j) {
        const float4 ca = a, cb = b;
        const bool vj = j < nl;
        if (j + 1 < nmax) {
            lj = (j + 1 < nl) ? s_lane[(j + 1) * 64 + lane] : 0u;
            a = s_A[lj]; b = s_B[lj];
        }
        const float mubar = dot3_ref(ca.x, ca.y, ca.z, ray.nx, ray.ny, ray.nz);
        const float d2 = sub_ref(ca.w, mul_ref(mubar, mubar));
        const float A = vj ? cb.z * vexp<EXP>(-(d2 * cb.y)) : 0.f;
        const float m = mubar * cb.x;
        const float E = erf(-m);
#pragma unroll
        for (int e = 0; e < EC; ++e) {
            const float base = __builtin_fmaf(e_mubar[e], cb.x, -m);
            const float step = e_sigma[e] * cb.x;
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const float x = __builtin_fmaf((float)(k - 4), step, base);
                acc[e][k] = __builtin_fmaf(A, E - erf(x), acc[e][k]);
            }
        }
    }

    // emission (see shade_list)
#pragma unroll
    for (int e = 0; e < EC; ++e) {
        if (i0 + e < nl) {
            const float4 ms = s_M[e_li[e]];
            const float inv2s2 = s_B[e_li[e]].y;
            const float q = s_q[e_li[e]];
            float inner = 0.f;
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const float sk = madd_ref((float)(k - 4), ms.w, e_mubar[e]);
                const float px = sub_ref(madd_ref(ray.nx, sk, ray.ox), ms.x);
                const float py = sub_ref(madd_ref(ray.ny, sk, ray.oy), ms.y);
                const float pz = sub_ref(madd_ref(ray.nz, sk, ray.oz), ms.z);
                const float dd = dot3_ref(px, py, pz, px, py, pz);
                inner += emission_term<EXP>(q, dd * inv2s2, acc[e][k]);
            }
            const float4 alb = s_C[e_li[e]];
            Lr = __builtin_fmaf(alb.x, inner, Lr);
            Lg = __builtin_fmaf(alb.y, inner, Lg);
            Lb = __builtin_fmaf(alb.z, inner, Lb);
            La = __builtin_fmaf(alb.w, inner, La);
        }
    }
}

// Chunks of EC emitters, then one chunk of exactly the remaining 1..EC-1: the pair loop costs nmax^2, not
// nmax * (nmax rounded up to a multiple of EC).
template <int EXP, int ERF, int EC>
__device__ __forceinline__ void shade_lanes(const float4 *s_A, const float4 *s_B, const float4 *s_M, const float4 *s_C,
                                            const float *s_q, const uint8_t *s_lane /*[k*64 + lane]*/, uint32_t nl,
                                            uint32_t nmax, uint32_t lane, const LaneRay &ray, float &Lr, float &Lg,
                                            float &Lb, float &La, uint32_t i_start = 0, uint32_t i_step = EC)
{
    Lr = Lg = Lb = La = 0.f;
    for (uint32_t i0 = i_start; i0 < nmax; i0 += i_step) {
        const uint32_t rem = nmax - i0;
        if (rem >= (uint32_t)EC) shade_chunk<EXP, ERF, EC>(s_A, s_B, s_M, s_C, s_q, s_lane, nl, nmax, lane, ray, i0, Lr, Lg, Lb, La);
        else if (EC > 3 && rem == 3) shade_chunk<EXP, ERF, 3>(s_A, s_B, s_M, s_C, s_q, s_lane, nl, nmax, lane, ray, i0, Lr, Lg, Lb, La);
        else if (EC > 2 && rem == 2) shade_chunk<EXP, ERF, 2>(s_A, s_B, s_M, s_C, s_q, s_lane, nl, nmax, lane, ray, i0, Lr, Lg, Lb, La);
        else shade_chunk<EXP, ERF, 1>(s_A, s_B, s_M, s_C, s_q, s_lane, nl, nmax, lane, ray, i0, Lr, Lg, Lb, La);
    }
}

// The same with balanced chunks of at most ECMAX emitters: ceil(nmax / ECMAX) chunks of nearly equal size, so that a list
// of 5 is ONE pass over the absorbers (not 4 + 1) and a list of 9 is 5 + 4 (not 4 + 4 + 1).  Emitters and absorbers are
// visited in the same order as before: the sums are bit-identical.
#ifndef VRT_RENDER_ECMAX
#define VRT_RENDER_ECMAX 4
#endif
template <int EXP, int ERF, int ECMAX>
__device__ __forceinline__ void shade_lanes_balanced(const float4 *s_A, const float4 *s_B, const float4 *s_M, const float4 *s_C,
                                                     const float *s_q, const uint8_t *s_lane, uint32_t nl, uint32_t nmax, uint32_t lane,
                                                     const LaneRay &ray, float &Lr, float &Lg, float &Lb, float &La)
{
    Lr = Lg = Lb = La = 0.f;
    uint32_t chunks = (nmax + ECMAX - 1) / ECMAX;
    for (uint32_t i0 = 0; i0 < nmax; --chunks) {
        const uint32_t size = (nmax - i0 + chunks - 1) / chunks;
        if (ECMAX >= 6 && size == 6) shade_chunk<EXP, ERF, (ECMAX >= 6 ? 6 : 1)>(s_A, s_B, s_M, s_C, s_q, s_lane, nl, nmax, lane, ray, i0, Lr, Lg, Lb, La);
        else if (ECMAX >= 5 && size == 5) shade_chunk<EXP, ERF, (ECMAX >= 5 ? 5 : 1)>(s_A, s_B, s_M, s_C, s_q, s_lane, nl, nmax, lane, ray, i0, Lr, Lg, Lb, La);
        else if (size == 4) shade_chunk<EXP, ERF, 4>(s_A, s_B, s_M, s_C, s_q, s_lane, nl, nmax, lane, ray, i0, Lr, Lg, Lb, La);
        else if (size == 3) shade_chunk<EXP, ERF, 3>(s_A, s_B, s_M, s_C, s_q, s_lane, nl, nmax, lane, ray, i0, Lr, Lg, Lb, La);
        else if (size == 2) shade_chunk<EXP, ERF, 2>(s_A, s_B, s_M, s_C, s_q, s_lane, nl, nmax, lane, ray, i0, Lr, Lg, Lb, La);
        else shade_chunk<EXP, ERF, 1>(s_A, s_B, s_M, s_C, s_q, s_lane, nl, nmax, lane, ray, i0, Lr, Lg, Lb, La);
        i0 += size;
    }
}

// ---------------------------------------------------------------------------------------------
// (ray, emitter) PAIRS as lanes (round 3).  shade_chunk above gives every lane its ray and runs up to six of the ray's
// emitters side by side: all lanes loop to the block's longest list, so a ray with 3 entries beside one with 6 idles through
// half of the emitter slots AND half of the absorber iterations (64 % useful lanes on `-g 64 -w 2048`).  Here the emitters of all
// 64 rays are packed densely: pair p = (ray, e-th entry of its list), lane = pair, NPASS = ceil(pairs / 64) passes held side
// by side in registers.  Per absorber slot j every lane first computes, AS A RAY, its j-th absorber's (A, m, E, r) -- once per
// (ray, absorber), as before -- and puts them into a 1-KB LDS row, from which each pass fetches the values of its pair's ray with
// one 16-byte read (by ds_bpermute, four per pass and slot, the LDS instructions ate the gain: VALU -11 %, wave cycles +2 %); then
// the five terms of the pair.  Emitter slots are no longer padded; absorber slots still are (a pair whose ray has no j-th absorber
// adds A = 0).  The inner sums go through LDS back to the ray's lane, which adds them in list order.
// LDS: the tail of s_lane (lists are at most PAIR_PL = 8 long on this path): pair -> ray map, the rays' first pair, the inner sums.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t PAIR_PL = 8, PAIR_NPASS_MAX = 6;
static_assert(PL * 64 >= PAIR_PL * 64 + 512 + 256 + PAIR_NPASS_MAX * 64 * 4, "s_lane holds the pair path's scratch behind the lists");
template <int EXP, int ERF, int NPASS>
__device__ __forceinline__ void shade_pairs(const float4 *s_A, const float4 *s_B, const float4 *s_M, const float4 *s_C, const float *s_q,
                                            float4 *s_pre /* [2][64] */, uint8_t *s_lane, uint32_t nl, uint32_t nmax, uint32_t first_pair, uint32_t n_pairs, uint32_t lane,
                                            const LaneRay &ray, float &Lr, float &Lg, float &Lb, float &La)
{
    const ErfEval<ERF> erf;
    uint8_t *s_map = s_lane + PAIR_PL * 64;                                   // [<= 384] ray of pair p
    uint32_t *s_first = reinterpret_cast<uint32_t *>(s_lane + PAIR_PL * 64 + 512);  // [64] first pair of ray l
    float *s_inner = reinterpret_cast<float *>(s_lane + PAIR_PL * 64 + 512 + 256);  // [<= 384] inner sum of pair p
    for (uint32_t e = 0; e < nmax; ++e)
        if (e < nl) s_map[first_pair + e] = (uint8_t)lane;
    s_first[lane] = first_pair;
    __syncthreads();
    // the pairs of this lane, one per pass
    uint32_t p_ray[NPASS], p_li[NPASS];
    float e_mubar[NPASS], e_sigma[NPASS], acc[NPASS][5];
#pragma unroll
    for (int q = 0; q < NPASS; ++q) {
        const uint32_t p = (uint32_t)q * 64u + lane;
        const bool vp = p < n_pairs;
        p_ray[q] = vp ? s_map[p] : 0u;
        const uint32_t e = vp ? p - s_first[p_ray[q]] : 0u;
        p_li[q] = vp ? s_lane[e * 64 + p_ray[q]] : 0u;
        const float nx = __shfl(ray.nx, (int)p_ray[q], 64), ny = __shfl(ray.ny, (int)p_ray[q], 64), nz = __shfl(ray.nz, (int)p_ray[q], 64);
        const float4 a = s_A[p_li[q]];
        e_mubar[q] = dot3_ref(a.x, a.y, a.z, nx, ny, nz);
        e_sigma[q] = s_M[p_li[q]].w;
#pragma unroll
        for (int k = 0; k < 5; ++k) acc[q][k] = 0.f;
    }
    // absorber slots.  The rays' values of slot j + 1 are computed while the passes work on slot j (two rows of LDS; one
    // wavefront: its LDS instructions execute in order, so a row written before it is read needs no barrier, only the
    // compiler kept from reordering the two)
    auto ray_values = [&](uint32_t j) {
        const bool vj = j < nl;
        const uint32_t lj = vj ? s_lane[j * 64 + lane] : 0u;
        const float4 ca = s_A[lj], cb = s_B[lj];
        const float mubar = dot3_ref(ca.x, ca.y, ca.z, ray.nx, ray.ny, ray.nz);
        const float d2 = sub_ref(ca.w, mul_ref(mubar, mubar));
        const float A = vj ? cb.z * vexp<EXP>(-(d2 * cb.y)) : 0.f;
        const float m = mubar * cb.x;
        return make_float4(A, m, erf(-m), cb.x);
    };
    s_pre[lane] = ray_values(0);
    for (uint32_t j = 0; j < nmax; ++j) {
        __builtin_amdgcn_wave_barrier();
        const float4 *row = s_pre + (j & 1u) * 64u;
        float4 nxt = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j + 1 < nmax) nxt = ray_values(j + 1);
#pragma unroll
        for (int q = 0; q < NPASS; ++q) {
            const float4 v = row[p_ray[q]];
            const float base = __builtin_fmaf(e_mubar[q], v.w, -v.y);
            const float step = e_sigma[q] * v.w;
#pragma unroll
            for (int k = 0; k < 5; ++k) acc[q][k] = __builtin_fmaf(v.x, v.z - erf(__builtin_fmaf((float)(k - 4), step, base)), acc[q][k]);
        }
        __builtin_amdgcn_wave_barrier();
        if (j + 1 < nmax) s_pre[((j + 1) & 1u) * 64u + lane] = nxt;
    }
    // emission (see shade_list), per pair
#pragma unroll
    for (int q = 0; q < NPASS; ++q) {
        const uint32_t p = (uint32_t)q * 64u + lane;
        const float nx = __shfl(ray.nx, (int)p_ray[q], 64), ny = __shfl(ray.ny, (int)p_ray[q], 64), nz = __shfl(ray.nz, (int)p_ray[q], 64);
        const float4 ms = s_M[p_li[q]];
        const float inv2s2 = s_B[p_li[q]].y;
        const float qq = s_q[p_li[q]];
        float inner = 0.f;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const float sk = madd_ref((float)(k - 4), ms.w, e_mubar[q]);
            const float px = sub_ref(madd_ref(nx, sk, ray.ox), ms.x);
            const float py = sub_ref(madd_ref(ny, sk, ray.oy), ms.y);
            const float pz = sub_ref(madd_ref(nz, sk, ray.oz), ms.z);
            const float dd = dot3_ref(px, py, pz, px, py, pz);
            inner += emission_term<EXP>(qq, dd * inv2s2, acc[q][k]);
        }
        if (p < n_pairs) s_inner[p] = inner;
    }
    __syncthreads();
    // back on the ray's lane: its emitters in list order
    Lr = Lg = Lb = La = 0.f;
    for (uint32_t e = 0; e < nmax; ++e) {
        if (e < nl) {
            const float inner = s_inner[first_pair + e];
            const float4 alb = s_C[s_lane[e * 64 + lane]];
            Lr = __builtin_fmaf(alb.x, inner, Lr);
            Lg = __builtin_fmaf(alb.y, inner, Lg);
            Lb = __builtin_fmaf(alb.z, inner, Lb);
            La = __builtin_fmaf(alb.w, inner, La);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Rays and cones
// ---------------------------------------------------------------------------------------------
// (column, row) of raster index pix; 32-bit division whenever the index fits (a 64-bit divide is a ~150-instruction
// software routine on this hardware)
__device__ __forceinline__ void col_row(const RayGen &R, uint64_t pix, uint32_t &jcol, uint32_t &irow)
{
    if (pix <= 0xFFFFFFFFull) {
        const uint32_t p = (uint32_t)pix;
        irow = p / R.width; jcol = p - irow * R.width;
    } else {
        irow = (uint32_t)(pix / R.width); jcol = (uint32_t)(pix % R.width);
    }
}

// World-space ray through raster pixel `pix` (rt.h:362-371).
__device__ __forceinline__ LaneRay pixel_ray(const RayGen &R, uint64_t pix)
{
    float px, py, pz;
    if (R.xs) {
        px = R.xs[pix]; py = R.ys[pix]; pz = R.zs[pix];
    } else if (R.view_mode) {
        // camera.cpp:60-69 with glm's mat4 * vec4: (m0 x + m1 y) + (m2 0 + m3 1), unfused -- the reference's plane point
        uint32_t jcol, irow;
        col_row(R, pix, jcol, irow);
        const float x = add_ref(-1.f, (float)jcol / R.half_w), y = add_ref(-1.f, (float)irow / R.half_h);
        px = add_ref(add_ref(mul_ref(R.m0[0], x), mul_ref(R.m1[0], y)), R.m3[0]);
        py = add_ref(add_ref(mul_ref(R.m0[1], x), mul_ref(R.m1[1], y)), R.m3[1]);
        pz = add_ref(add_ref(mul_ref(R.m0[2], x), mul_ref(R.m1[2], y)), R.m3[2]);
    } else {
        // closed form of camera.cpp:52,60-69: plane = pos + x right + y up - focal front
        uint32_t jcol, irow;
        col_row(R, pix, jcol, irow);
        const float x = -1.f + (float)jcol * R.inv_half_w;
        const float y = -1.f + (float)irow * R.inv_half_h;
        px = R.pos[0] + x * R.right[0] + y * R.up[0] - R.focal * R.front[0];
        py = R.pos[1] + x * R.right[1] + y * R.up[1] - R.focal * R.front[1];
        pz = R.pos[2] + x * R.right[2] + y * R.up[2] - R.focal * R.front[2];
    }
    LaneRay ray;
    ray.ox = R.origin[0]; ray.oy = R.origin[1]; ray.oz = R.origin[2];
    // rt.h:366-371 + vec4f_t::normalize (types.h:75-82): IEEE sqrt and divides, unfused dot.
    // (sqrtf and '/' are correctly rounded in HIP's default mode; __fsqrt_rn is NOT -- it maps to the
    // 1-ulp v_sqrt_f32, and a 1-ulp change of n is amplified by the cancellation noise above)
    const float dx = px - ray.ox, dy = py - ray.oy, dz = pz - ray.oz;
    const float norm = __builtin_sqrtf(dot3_ref(dx, dy, dz, dx, dy, dz));
    ray.nx = dx / norm; ray.ny = dy / norm; ray.nz = dz / norm;
    return ray;
}

// A bundle of rays from one origin inside the cone (axis c, half angle theta).  For a Gaussian at
// oc = mu - o the distance to any line of the bundle is >= |oc| sin(phi - theta), phi = angle(oc, axis line)
// = dperp cos(theta) - |oc.c| sin(theta); the Gaussian can be dropped for the whole bundle when even that
// best case gives d^2/(2 sigma^2) > cull_x, i.e. sigma*mag*exp(-..) < cull_eps (or Exp underflows to 0).
struct Cone { float cx, cy, cz, cos_t, sin_t; };
// cosine and sine of the angle between unit vectors n and c; the sine from the cross product (1 - cos^2 has
// no digits left for the milliradian cones of a pixel block)
__device__ __forceinline__ void cos_sin(float nx, float ny, float nz, float cx, float cy, float cz, float &co, float &si)
{
    co = nx * cx + ny * cy + nz * cz;
    const float ux = ny * cz - nz * cy, uy = nz * cx - nx * cz, uz = nx * cy - ny * cx;
    si = __builtin_amdgcn_sqrtf(ux * ux + uy * uy + uz * uz);
}
__device__ __forceinline__ Cone make_cone(float cx, float cy, float cz, float min_cos, float max_sin)
{
    Cone k;
    k.cx = cx; k.cy = cy; k.cz = cz;
    k.sin_t = max_sin * 1.001f + 1e-6f;          // conservative: never over-estimate
    k.cos_t = fminf(min_cos, 1.f) * 0.9999f;     // the distance to the cone
    return k;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
// Ray through a pixel for CONE construction only: same geometry as pixel_ray, fast reciprocal square root
// (shading rays need the reference's exactly rounded normalisation; a cone bound does not).
__device__ __forceinline__ LaneRay cone_ray(const RayGen &R, uint64_t pix)
{
    float px, py, pz;
    if (R.xs) {
        px = R.xs[pix]; py = R.ys[pix]; pz = R.zs[pix];
    } else if (R.view_mode) {
        uint32_t jcol, irow;
        col_row(R, pix, jcol, irow);
        const float x = -1.f + (float)jcol * R.inv_half_w, y = -1.f + (float)irow * R.inv_half_h;
        px = R.m0[0] * x + R.m1[0] * y + R.m3[0];
        py = R.m0[1] * x + R.m1[1] * y + R.m3[1];
        pz = R.m0[2] * x + R.m1[2] * y + R.m3[2];
    } else {
        uint32_t jcol, irow;
        col_row(R, pix, jcol, irow);
        const float x = -1.f + (float)jcol * R.inv_half_w, y = -1.f + (float)irow * R.inv_half_h;
        px = R.pos[0] + x * R.right[0] + y * R.up[0] - R.focal * R.front[0];
        py = R.pos[1] + x * R.right[1] + y * R.up[1] - R.focal * R.front[1];
        pz = R.pos[2] + x * R.right[2] + y * R.up[2] - R.focal * R.front[2];
    }
    LaneRay ray;
    ray.ox = R.origin[0]; ray.oy = R.origin[1]; ray.oz = R.origin[2];
    const float dx = px - ray.ox, dy = py - ray.oy, dz = pz - ray.oz;
    const float inv = __builtin_amdgcn_rsqf(dx * dx + dy * dy + dz * dz);
    ray.nx = dx * inv; ray.ny = dy * inv; ray.nz = dz * inv;
    return ray;
}
// cone of the pixel rectangle [x0,x1] x [y0,y1] (image coordinates via `at`): axis = centre ray, angle = the
// farthest corner ray (pinhole rays: the farthest ray of a rectangle on the image plane is a corner ray);
// lanes 0..3 take a corner each
template <typename At>
__device__ __forceinline__ Cone rect_cone(At at, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint32_t lane)
{
    const LaneRay c = at((x0 + x1 + 1) / 2, (y0 + y1 + 1) / 2);
    const LaneRay k = at((lane & 1) ? x1 : x0, (lane & 2) ? y1 : y0);
    float co, si;
    cos_sin(k.nx, k.ny, k.nz, c.nx, c.ny, c.nz, co, si);
    Cone cone = make_cone(c.nx, c.ny, c.nz, wave_min(co), wave_max(si));
    cone.sin_t += 1e-4f; // the centre pixel is up to half a pixel off the rectangle's centre
    return cone;
}
__device__ __forceinline__ bool cone_keeps(const Cone &k, float4 a /*oc,|oc|^2*/, float4 bq /*r,1/2s^2,qK,cull_x*/)
{
    const float tc = a.x * k.cx + a.y * k.cy + a.z * k.cz;
    const float dperp = __builtin_amdgcn_sqrtf(fmaxf(0.f, a.w - tc * tc));
    const float dmin = fmaxf(0.f, dperp * k.cos_t - fabsf(tc) * k.sin_t);
    const float xmin = dmin * dmin * bq.y;
    return !(xmin * 0.999f - 1e-3f > bq.w);
}

// Level-wise thresholds (TileLists::cull_ref_n): the slack of a level that n candidates enter, and a candidate's cull_x with
// it applied -- unless that sits at the Exp floor ("keep unless the contribution is exactly 0") or is -inf (sigma*mag = 0).
__device__ __forceinline__ float level_slack(float cull_ref_n, uint32_t n) { return cull_ref_n > 0.f ? __logf(cull_ref_n / (float)max(n, 1u)) : 0.f; }
__device__ __forceinline__ float slack_cull_x(float cull_x, float slack, float floor_x) { return cull_x < floor_x ? cull_x - slack : cull_x; }

// ---------------------------------------------------------------------------------------------
// Image kernel: persistent one-wave workgroups; a work item is one 8x8 pixel block (64 rays, lane = ray) of a
// 32x32 pixel cell with a non-empty candidate list.  A wave's first block is static (item = wave); frames with
// more blocks than waves hand the rest out through eight work counters (CellGrid::rq).  Empty cells are cleared
// by the fused list kernel; when that did not run for this target (unfused lists, re-render) they are cleared here.
// ---------------------------------------------------------------------------------------------
struct BlockPos { uint32_t lt, t, pxt, pyt; bool inside; };
// where lane `lane` of block `bi` of `cell` writes: raster, compact shard [lt][tile_h][tile_w], or sparse shard (cell-major)
__device__ __forceinline__ uint64_t out_index(const TileLists &T, const CellGrid &C, const RenderTarget &O, uint32_t cell,
                                              uint32_t bi, uint32_t lane, const struct BlockPos &p, uint64_t pix, uint32_t n_active);
__device__ __forceinline__ BlockPos block_of(const TileLists &T, const CellGrid &C, const RenderTarget &O, uint32_t cell,
                                             uint32_t bi, uint32_t lane)
{
    BlockPos p;
    const uint32_t cpt = C.cells_x * C.cells_y;
    p.lt = cell / cpt;
    const uint32_t ci = cell % cpt;
    p.t = O.tile_map ? O.tile_map[p.lt] : p.lt;
    const uint32_t bxi = (ci % C.cells_x) * (CELL / BLOCK_W) + (bi & 3), byi = (ci / C.cells_x) * (CELL / BLOCK_H) + (bi >> 2);
    p.inside = bxi * BLOCK_W < T.tile_w && byi * BLOCK_H < T.tile_h; // wave-uniform
    p.pxt = bxi * BLOCK_W + (lane & 7);
    p.pyt = byi * BLOCK_H + (lane >> 3);
    return p;
}

// occupancy experiment knob (csrc/Makefile EXTRA=-DVRT_RENDER_WPE=4): cap the registers for N waves per SIMD
#ifdef VRT_RENDER_WPE
#define VRT_RENDER_ATTR __attribute__((amdgpu_waves_per_eu(VRT_RENDER_WPE, VRT_RENDER_WPE)))
#else
#define VRT_RENDER_ATTR
#endif
__device__ __forceinline__ uint64_t out_index(const TileLists &T, const CellGrid &C, const RenderTarget &O, uint32_t cell,
                                              uint32_t bi, uint32_t lane, const BlockPos &p, uint64_t pix, uint32_t n_active)
{
    if (O.sparse) {
        const uint32_t s = C.slot[cell];
        const uint32_t slot = (s & 0x7FFFFFFFu) + ((s >> 31) ? n_active : 0u);
        return (uint64_t)slot * (CELL * CELL) + ((bi >> 2) * BLOCK_H + (lane >> 3)) * CELL + (bi & 3) * BLOCK_W + (lane & 7);
    }
    return O.compact ? ((uint64_t)p.lt * T.tile_h + p.pyt) * T.tile_w + p.pxt : pix;
}

// exactly `size` emitters (list positions i0 .. i0+size-1 of every lane), size 0..4
template <int EXP, int ERF>
__device__ __forceinline__ void shade_range(const float4 *s_A, const float4 *s_B, const float4 *s_M, const float4 *s_C,
                                            const float *s_q, const uint8_t *s_lane, uint32_t nl, uint32_t nmax, uint32_t lane,
                                            const LaneRay &ray, uint32_t i0, uint32_t size, float &Lr, float &Lg, float &Lb, float &La)
{
    Lr = Lg = Lb = La = 0.f;
    if (size == 4) shade_chunk<EXP, ERF, 4>(s_A, s_B, s_M, s_C, s_q, s_lane, nl, nmax, lane, ray, i0, Lr, Lg, Lb, La);
    else if (size == 3) shade_chunk<EXP, ERF, 3>(s_A, s_B, s_M, s_C, s_q, s_lane, nl, nmax, lane, ray, i0, Lr, Lg, Lb, La);
    else if (size == 2) shade_chunk<EXP, ERF, 2>(s_A, s_B, s_M, s_C, s_q, s_lane, nl, nmax, lane, ray, i0, Lr, Lg, Lb, La);
    else if (size == 1) shade_chunk<EXP, ERF, 1>(s_A, s_B, s_M, s_C, s_q, s_lane, nl, nmax, lane, ray, i0, Lr, Lg, Lb, La);
}

// Budgeted ray-level cull (round 3).  The level-wise thresholds above are worst-case counting: a level that n candidates enter drops
// below eps * 1365 / n, as if all n sat just under it.  A ray's own list knows better: what it lost is the SUM of sigma*mag*exp(-x) over
// what it dropped, and on a grid scene one or two of a ray's five entries carry 1e-7 .. 1e-6 while the worst case reserves room for
// dozens.  So after the unconditional pass the lane looks at the entries it kept (e_k = sigma*mag*exp(-x_k) in units of the tile
// level's eps; the pass leaves ln e_k = cull_x - x_k in LDS, as fp16 rounded up) and drops the smallest ones as long as their sum stays inside `budget` (CellGrid::prune_budget = kappa * 1365 eps:
// 3 * that is what the ray's radiance can change by): smallest first, exactly -- entry k goes iff the sum of all
// entries not larger than it fits.  Entries whose threshold sits at the Exp floor (cull_eps = 0, huge magnitudes) are never dropped.
// A function of the block's survivors alone, so every path that must give identical bits still does.  `-g 64 -w 2048`: per-ray lists
// 3.9 -> 2.5, the block's longest 5.4 -> 3.6; the pair loops are quadratic in that.
constexpr uint32_t PRUNE_PL = 16; // lists up to this long are pruned; s_t holds ln(e_k) of their entries (fp16, rounded up)
template <int N>
__device__ __forceinline__ uint32_t prune_list(const __half *s_t, uint8_t *s_lane, uint32_t nl, uint32_t lane, float budget)
{
    float e[N];
#pragma unroll
    for (int k = 0; k < N; ++k) e[k] = (uint32_t)k < nl ? __expf(__half2float(s_t[k * 64 + lane])) : INFINITY;
    float least = e[0];
#pragma unroll
    for (int k = 1; k < N; ++k) least = fminf(least, e[k]);
    if (__ballot(least <= budget) == 0ull) return nl; // nothing in this block is small enough
    uint32_t drop = 0;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        float below = 0.f; // the sum of everything not larger than entry k, itself included
#pragma unroll
        for (int l = 0; l < N; ++l) below += e[l] <= e[k] ? e[l] : 0.f;
        if (below <= budget) drop |= 1u << k;
    }
    uint32_t w = 0;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        if ((uint32_t)k < nl) {
            const uint8_t v = s_lane[k * 64 + lane];
            if (!((drop >> k) & 1u)) { s_lane[w * 64 + lane] = v; ++w; }
        }
    }
    return w;
}

// NW = waves per block.  NW = 1: one wavefront shades a block on its own.  NW = 2: the two waves of a workgroup hold the
// same 64 rays, share ONE block cull and ONE set of per-ray lists through LDS and take half of the emitters each; their
// partial radiances are added in wave order.  A frame of `-g 64 -w 2048` is ~2500 equally heavy blocks for 1024 SIMDs,
// three resident waves each: with whole blocks as the unit half of the SIMDs carry three heavy blocks and the rest two
// (27 us against a balanced 21 us, VRT_HIP_TIMELINE); with half blocks pulled from the work queues the unit is half
// as long and the per-SIMD sums even out.
template <int EXP, int ERF, int EC, int NW, bool CLAIM = false>
__device__ __forceinline__ void render_body(const SceneTables &S, const TileLists &T, const CellGrid &C, const RayGen &R, const RenderTarget &O)
{
    // every row a kept candidate needs later (absorber: A, B; emitter: mu/sigma, albedo, sigma*mag) is fetched in the one
    // round trip of the block cull: the shading loops then run out of LDS only
    __shared__ float4 s_A[PCAP], s_B[PCAP], s_M[PCAP], s_C[PCAP];
    __shared__ float s_q[PCAP];
    __shared__ uint8_t s_lane[PL * 64];
    __shared__ __half s_t[NW == 1 ? PRUNE_PL * 64 : 1]; // ln(sigma*mag*exp(-x) / eps) of the first PRUNE_PL entries of every lane's list (prune_list)
#ifdef VRT_PAIR_LANES
    __shared__ float4 s_pre[NW == 1 ? 128 : 1]; // pair lanes: the rays' (A, m, E, r) of the current and the next absorber slot
#endif
    __shared__ float4 s_L[NW > 1 ? 64 : 1];
    __shared__ uint32_t s_cnt[2], s_item;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6, wave = blockIdx.x, G = gridDim.x;
    const bool first = threadIdx.x == 0;
    const uint64_t npix = (uint64_t)R.width * R.height;
    // A block's way to its candidates is a chain of dependent loads (queue entry -> list -> parameter rows), and at the start of a launch
    // every wave walks it at the same time, with nothing to hide it behind.  Two links are taken out: an entry of the active queue carries
    // its cell's list length in its top byte (ACTIVE_COUNT_SHIFT), and the entry of a wave's static first block is fetched together with
    // the counters it is checked against (stale beyond n_active: used only below it).
    const uint32_t spec_entry = C.n_cells ? C.active[min(wave >> 4, C.n_cells - 1u)] : 0u;
    const uint32_t n_active = *C.n_active, n_dense_cells = *C.n_dense;
    if (C.feedback && wave == 0 && first) {
        __hip_atomic_store(&C.feedback[0], n_dense_cells, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const uint32_t n_light = C.light_threshold ? *C.n_light : 0u; // cells filed from the back of the queue: shaded last
    const uint32_t n_shade = (n_active + n_light) * 16u; // the dense cells belong to the 16-waves-per-block kernel behind this one
    if (C.feedback && wave == 0 && first) __hip_atomic_store(&C.feedback[1], n_shade, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); // for the host's choice of the CLAIM variant
    if (O.sparse_hdr && wave == 0 && first) { // sparse shard header; the counts are final: the list kernel is done
        O.sparse_hdr[0] = n_active + n_dense_cells; O.sparse_hdr[1] = O.sparse_cap;
        O.sparse_hdr[2] = C.cells_x * C.cells_y; O.sparse_hdr[3] = 0;
    }

    // ---- clear the cells nothing can reach (4 B per ray: the only HBM traffic of most of the frame) ----
    const uint32_t zero_px = (O.pack_flags & VRT_ALPHA_COMPUTED) ? 0u : 0xFF000000u;
    // (only when the list kernel of this frame did not do it: unfused lists, or a re-render from unchanged lists)
    for (uint32_t cell = wave; cell < C.n_cells && !O.cleared && wv == 0; cell += G) { // one whole cell per item: 16 x (2 rows of 32 px)
        if (C.count[cell] != 0u) continue;
        const uint32_t cpt = C.cells_x * C.cells_y;
        const uint32_t lt = cell / cpt, ci = cell % cpt;
        const uint32_t t = O.tile_map ? O.tile_map[lt] : lt;
        const uint32_t tx = t % T.tiles_w, ty = t / T.tiles_w;
        const uint32_t pxt = (ci % C.cells_x) * CELL + (lane & 31);
#pragma unroll 4
        for (uint32_t pass = 0; pass < CELL / 2; ++pass) {
            const uint32_t pyt = (ci / C.cells_x) * CELL + pass * 2 + (lane >> 5);
            const uint64_t pix = (uint64_t)(tx * T.tile_w + pxt) + (uint64_t)T.stride * (ty * T.tile_h + pyt);
            if (pxt < T.tile_w && pyt < T.tile_h && pix < npix) {
                const uint64_t out = O.compact ? ((uint64_t)lt * T.tile_h + pyt) * T.tile_w + pxt : pix;
                if (O.image) O.image[out] = zero_px;
                if (O.radiance) O.radiance[out] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }

    // ---- shade ----
    if (wave == 0 && threadIdx.x < RQ_N) C.rq_next[threadIdx.x * RQ_STRIDE] = 0;
    const uint32_t n_dyn = n_shade > G ? n_shade - G : 0u;
    uint32_t rq_tries = 0, rq_dead = 0; // rq_dead: queues this wave has seen run out (bit l = the l-th from its own)
    // next block: blocks cost between ~1 and ~30 units (the pair loops are quadratic in the per-ray list length), so
    // after its static first block a workgroup pulls more from the queues, its own first, until all are empty
    auto next_item = [&]() -> uint32_t {
        if constexpr (NW == 1) {
            // all RQ_N counters are looked at in ONE round trip (lane l reads the l-th queue from the wave's own on): a wave that is done
            // leaves after one load instead of after RQ_N dependent ones -- at the end of a launch that was 5 us of every wave's exit
            while (true) {
                const uint32_t q = (wave + lane) % RQ_N;
                const uint32_t per = n_dyn > q ? (n_dyn - q + RQ_N - 1) / RQ_N : 0u;
                bool have = false;
                if (lane < RQ_N && per && !((rq_dead >> lane) & 1u)) have = __hip_atomic_load(C.rq + q * RQ_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < per;
                const unsigned long long mask = __ballot(have);
                if (!mask) return 0xFFFFFFFFu;
                const uint32_t l = (uint32_t)__builtin_ctzll(mask);
                const uint32_t ql = (wave + l) % RQ_N, perl = (n_dyn - ql + RQ_N - 1) / RQ_N;
                uint32_t m = 0xFFFFFFFFu;
                if (first) m = atomicAdd(C.rq + ql * RQ_STRIDE, 1u);
                m = __builtin_amdgcn_readfirstlane(m);
                if (m < perl) return G + ql + RQ_N * m;
                rq_dead |= 1u << l; // lost the race for its last entry: that queue is empty for good, so at most RQ_N rounds
            }
        } else {
            while (rq_tries < RQ_N) {
                const uint32_t q = (wave + rq_tries) % RQ_N;
                const uint32_t per = n_dyn > q ? (n_dyn - q + RQ_N - 1) / RQ_N : 0u;
                uint32_t m = 0xFFFFFFFFu;
                if (first && per) {
                    uint32_t *ctr = C.rq + q * RQ_STRIDE;
                    if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < per) m = atomicAdd(ctr, 1u);
                }
                // both waves take the same item: through LDS
                if (first) s_item = m;
                __syncthreads();
                m = s_item;
                __syncthreads();
                if (m < per) return G + q + RQ_N * m;
                ++rq_tries;
            }
            return 0xFFFFFFFFu;
        }
    };
    auto write_block = [&](float Lr, float Lg, float Lb, float La, bool valid, uint64_t out) {
        if (valid) {
            if (O.image) O.image[out] = pack_pixel(Lr, Lg, Lb, La, O.pack_flags);
            if (O.radiance) O.radiance[out] = make_float4(Lr, Lg, Lb, La);
        }
    };
    // The next entry of the wave's own queue is claimed BEFORE the block is shaded: the atomic's round trip (device scope: microseconds
    // when thousands of waves ask) passes behind the pair loops instead of in front of the next block's chain of loads.  A claimed entry
    // is always worked off by the wave that claimed it.  (Issued after the cull's loads have been consumed: returns are in order.)
    const uint32_t q_own = wave % RQ_N, per_own = n_dyn > q_own ? (n_dyn - q_own + RQ_N - 1) / RQ_N : 0u;
    // (only where most claims succeed: with few entries beyond the static ones thousands of failing claims would queue up on 8 counters)
    // (a kernel variant of its own, CLAIM: with the claim compiled in, the kernel that never claims -- the headline frame has no queue entry
    // at all -- ran 6 % longer: 10 more VGPRs, 20 more spilled SGPRs; the host picks the variant from what earlier frames reported)
    const bool claim_early = CLAIM && C.claim_early > 0 && per_own && (uint64_t)n_dyn * (uint32_t)C.claim_early >= G;
    uint32_t claim = 0xFFFFFFFFu;
    bool claimed = false;
    auto advance = [&]() -> uint32_t {
        if constexpr (NW == 1 && CLAIM) {
            if (claimed) {
                claimed = false;
                const uint32_t m = __builtin_amdgcn_readfirstlane(claim);
                if (m < per_own) return G + q_own + RQ_N * m;
                rq_dead |= 1u; // the own queue is empty for good
            }
        }
        return next_item();
    };
    for (uint32_t item = wave; item < n_shade; item = advance()) {
        const unsigned long long tl0 = O.timeline ? wall_clock64() : 0ull; // diagnostics (VRT_HIP_TIMELINE runs only)
        const uint32_t ci = item >> 4;
        const uint32_t entry = (item == wave && ci < n_active) ? spec_entry : C.active[ci < n_active ? ci : C.n_cells - 1u - (ci - n_active)];
        const uint32_t cell = entry & ACTIVE_CELL_MASK;
        const uint32_t bi = item & 15u;
        const BlockPos p = block_of(T, C, O, cell, bi, lane);
        if (!p.inside) continue;
        const uint32_t tx = p.t % T.tiles_w, ty = p.t / T.tiles_w;
        bool valid = p.pxt < T.tile_w && p.pyt < T.tile_h;
        const uint32_t pxc = min(p.pxt, T.tile_w - 1), pyc = min(p.pyt, T.tile_h - 1);
        uint64_t pix = (uint64_t)(tx * T.tile_w + pxc) + (uint64_t)T.stride * (ty * T.tile_h + pyc);
        if (pix >= npix) { valid = false; pix = npix - 1; }
        const uint64_t out = out_index(T, C, O, cell, bi, lane, p, pix, n_active);

        // the cell's candidate list (or, if it overflowed its slot, the tile's)
        uint32_t n_list = entry >> ACTIVE_COUNT_SHIFT;
        if (n_list == 255u) n_list = C.count[cell]; // a list length that does not fit the byte
        const uint32_t *list = C.indices + (size_t)cell * C.cstride;
        if (n_list == 0xFFFFFFFFu) { n_list = T.count[p.t]; list = T.indices + T.start[p.t]; }

        LaneRay ray = pixel_ray(R, pix); // NW = 2: both waves hold the same 64 rays
        // the origin is wave-uniform (SGPRs): as a VGPR operand the 15 adds per emitter of the emission issue at full rate
        ray.ox = pin_vgpr(ray.ox); ray.oy = pin_vgpr(ray.oy); ray.oz = pin_vgpr(ray.oz);

        // ---- block cone: axis = mean of the four centre rays, angle = farthest lane ----
        float cx = __shfl(ray.nx, 27, 64) + __shfl(ray.nx, 28, 64) + __shfl(ray.nx, 35, 64) + __shfl(ray.nx, 36, 64);
        float cy = __shfl(ray.ny, 27, 64) + __shfl(ray.ny, 28, 64) + __shfl(ray.ny, 35, 64) + __shfl(ray.ny, 36, 64);
        float cz = __shfl(ray.nz, 27, 64) + __shfl(ray.nz, 28, 64) + __shfl(ray.nz, 35, 64) + __shfl(ray.nz, 36, 64);
        {
            const float inv = __builtin_amdgcn_rsqf(cx * cx + cy * cy + cz * cz);
            cx *= inv; cy *= inv; cz *= inv;
        }
        float co, si;
        cos_sin(ray.nx, ray.ny, ray.nz, cx, cy, cz, co, si);
        const Cone cone = make_cone(cx, cy, cz, wave_min(co), wave_max(si));

        // ---- block cull over the cell's list (ballot compaction, order preserving; NW = 2: 128 entries per pass, the
        //      second wave's survivors behind the first's) ----
        __syncthreads(); // previous item's LDS reads are done
        const float slack = level_slack(T.cull_ref_n, n_list);
        uint32_t cnt = 0;
        for (uint32_t base = 0; base < n_list; base += 64 * NW) {
            const uint32_t k = base + wv * 64 + lane;
            bool keep = false;
            float4 a, bq, ms, alb;
            float q;
            if (k < n_list) {
                const uint32_t idx = list[k];
                a = S.gA[idx]; bq = S.gB[idx]; ms = S.mu_sig[idx]; alb = S.gC[idx]; q = S.gD[idx].y;
                keep = cone_keeps(cone, a, make_float4(bq.x, bq.y, bq.z, slack_cull_x(bq.w, slack, T.floor_x)));
            }
            const unsigned long long mask = __ballot(keep);
            uint32_t before = 0, pass_total = (uint32_t)__popcll(mask);
            if constexpr (NW == 2) {
                if (lane == 0) s_cnt[wv] = pass_total;
                __syncthreads();
                before = wv ? s_cnt[0] : 0u;
                pass_total = s_cnt[0] + s_cnt[1];
            }
            const uint32_t pos = cnt + before + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
            if (keep && pos < PCAP) { s_A[pos] = a; s_B[pos] = bq; s_M[pos] = ms; s_C[pos] = alb; s_q[pos] = q; }
            cnt += pass_total;
            if constexpr (NW == 2) __syncthreads(); // s_cnt is rewritten by the next pass
        }
        __syncthreads();

        const unsigned long long tl1 = O.timeline ? wall_clock64() : 0ull;
        // ---- lane cull: this ray's own candidates (exact per-ray criterion x > cull_x, no margin needed).  NW = 2: both
        //      waves count (each needs nl), the first one files the list ----
        uint32_t nl = 0;
        bool fast = cnt <= PCAP;
        if (fast) {
            const float slack_r = level_slack(T.cull_ref_n, cnt); // ray level: the block's survivors enter
            for (uint32_t j = 0; j < cnt; ++j) {
                const float4 a = s_A[j], bq = s_B[j];
                const float mubar = dot3_ref(a.x, a.y, a.z, ray.nx, ray.ny, ray.nz);
                const float x = sub_ref(a.w, mul_ref(mubar, mubar)) * bq.y;
                if (!(x > slack_cull_x(bq.w, slack_r, T.floor_x))) {
                    if (nl < PL && wv == 0) s_lane[nl * 64 + lane] = (uint8_t)j;
                    if constexpr (NW == 1) {
                        if (nl < PRUNE_PL) { // fp16 rounds to nearest within 2^-11: the bias keeps the stored value above the true one
                            const float t = bq.w - x;
                            s_t[nl * 64 + lane] = __float2half(bq.w < T.floor_x ? t + 0.001f * fabsf(t) + 1e-4f : INFINITY);
                        }
                    }
                    ++nl;
                }
            }
            fast = __ballot(nl > PL) == 0ull;
        }
        __syncthreads();
        if (!fast) {
            // Per-ray lists that outgrow LDS: the block goes to the 16-waves-per-block kernel, which runs after this one
            // (the host leaves that launch out only when a frame of exactly this state has reported that nothing is
            // handed over and no cell is dense).  Which kernel shades a block is a function of the block alone (the two
            // kernels sum in different orders), so the image does not depend on launch heuristics: the host only
            // chooses how LARGE the dense launch is (vrt_hip_api.cpp, render_common).
            if (first) C.overflow[atomicAdd(C.n_overflow, 1u)] = (cell << 4) | bi;
            continue;
        }
        if (O.stats && first) {
            atomicAdd(&O.stats[0], (unsigned long long)cnt);
            atomicAdd(&O.stats[1], (unsigned long long)n_list);
            atomicAdd(&O.stats[5], 1ull);
        }
        uint32_t nmax = nl;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) nmax = max(nmax, (uint32_t)__shfl_xor((int)nmax, off, 64));
        if constexpr (NW == 1) {
            if (C.prune_budget > 0.f && nmax <= PRUNE_PL) {
                switch (nmax) { // exactly as many entries as the block's longest list has, while that is cheap
                case 0: break;
                case 1: nl = prune_list<1>(s_t, s_lane, nl, lane, C.prune_budget); break;
                case 2: nl = prune_list<2>(s_t, s_lane, nl, lane, C.prune_budget); break;
                case 3: nl = prune_list<3>(s_t, s_lane, nl, lane, C.prune_budget); break;
                case 4: nl = prune_list<4>(s_t, s_lane, nl, lane, C.prune_budget); break;
                case 5: nl = prune_list<5>(s_t, s_lane, nl, lane, C.prune_budget); break;
                case 6: nl = prune_list<6>(s_t, s_lane, nl, lane, C.prune_budget); break;
                case 7: nl = prune_list<7>(s_t, s_lane, nl, lane, C.prune_budget); break;
                case 8: nl = prune_list<8>(s_t, s_lane, nl, lane, C.prune_budget); break;
                case 9: case 10: case 11: case 12: nl = prune_list<12>(s_t, s_lane, nl, lane, C.prune_budget); break;
                default: nl = prune_list<16>(s_t, s_lane, nl, lane, C.prune_budget); break;
                }
                nmax = nl;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) nmax = max(nmax, (uint32_t)__shfl_xor((int)nmax, off, 64));
            }
        }
        if (O.stats && wv == 0) {
            unsigned long long tot = nl;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor((int)tot, off, 64);
            unsigned long long sq = (unsigned long long)nl * nl; // this ray's (emitter, absorber) pairs: 5 erf terms each
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) sq += (unsigned long long)(uint32_t)__shfl_xor((int)sq, off, 64);
            if (lane == 0) { atomicAdd(&O.stats[3], tot); atomicAdd(&O.stats[4], (unsigned long long)nmax); atomicAdd(&O.stats[12], sq); }
        }
        const unsigned long long tl2 = O.timeline ? wall_clock64() : 0ull;
        if constexpr (NW == 1 && CLAIM) {
            if (claim_early && !(rq_dead & 1u)) {
                claimed = true;
                if (first) claim = atomicAdd(C.rq + q_own * RQ_STRIDE, 1u);
            }
        }
        float Lr, Lg, Lb, La;
        if constexpr (NW == 1) {
#ifdef VRT_PAIR_LANES // experiment (profiles/r03_experiments.md): make LANES='-DVRT_RENDER_ECMAX=6 -DVRT_RENDER_WPE=3 -DVRT_PAIR_LANES', VRT_HIP_PAIR_LANES=1|2
            // short lists (sparse scenes): (ray, emitter) pairs as lanes; a function of the block alone (its list lengths)
            uint32_t incl = nl;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t up = (uint32_t)__shfl_up((int)incl, off, 64);
                if (lane >= (uint32_t)off) incl += up;
            }
            const uint32_t n_pairs = (uint32_t)__shfl((int)incl, 63, 64), npass = (n_pairs + 63u) / 64u;
            // instruction counts of the two layouts for this block (per lane; absorber set-up 30-35, five terms 60-66, an emission 160):
            // ray per lane -- balanced chunks of at most 6 emitters, every chunk walks all nmax absorber slots; pairs -- npass passes
            const uint32_t chunks = (nmax + VRT_RENDER_ECMAX - 1) / VRT_RENDER_ECMAX;
            const uint32_t cost_rays = nmax * (30u * chunks + 60u * nmax) + 160u * nmax;
            const uint32_t cost_pairs = nmax * (35u + 66u * npass) + 160u * npass + 150u;
            if (C.pair_lanes && nmax <= PAIR_PL && npass >= 1u && npass <= PAIR_NPASS_MAX && (C.pair_lanes > 1 || cost_pairs * 20u < cost_rays * 19u)) {
                switch (npass) {
                case 1: shade_pairs<EXP, ERF, 1>(s_A, s_B, s_M, s_C, s_q, s_pre, s_lane, nl, nmax, incl - nl, n_pairs, lane, ray, Lr, Lg, Lb, La); break;
                case 2: shade_pairs<EXP, ERF, 2>(s_A, s_B, s_M, s_C, s_q, s_pre, s_lane, nl, nmax, incl - nl, n_pairs, lane, ray, Lr, Lg, Lb, La); break;
                case 3: shade_pairs<EXP, ERF, 3>(s_A, s_B, s_M, s_C, s_q, s_pre, s_lane, nl, nmax, incl - nl, n_pairs, lane, ray, Lr, Lg, Lb, La); break;
                case 4: shade_pairs<EXP, ERF, 4>(s_A, s_B, s_M, s_C, s_q, s_pre, s_lane, nl, nmax, incl - nl, n_pairs, lane, ray, Lr, Lg, Lb, La); break;
                case 5: shade_pairs<EXP, ERF, 5>(s_A, s_B, s_M, s_C, s_q, s_pre, s_lane, nl, nmax, incl - nl, n_pairs, lane, ray, Lr, Lg, Lb, La); break;
                default: shade_pairs<EXP, ERF, 6>(s_A, s_B, s_M, s_C, s_q, s_pre, s_lane, nl, nmax, incl - nl, n_pairs, lane, ray, Lr, Lg, Lb, La); break;
                }
            } else
#endif
            if constexpr (VRT_RENDER_ECMAX > 4) shade_lanes_balanced<EXP, ERF, VRT_RENDER_ECMAX>(s_A, s_B, s_M, s_C, s_q, s_lane, nl, nmax, lane, ray, Lr, Lg, Lb, La);
            else shade_lanes<EXP, ERF, EC>(s_A, s_B, s_M, s_C, s_q, s_lane, nl, nmax, lane, ray, Lr, Lg, Lb, La);
            write_block(Lr, Lg, Lb, La, valid, out);
        } else {
            // emitters: up to 2*EC of them are cut in two halves, one chunk per wave; longer lists alternate chunks of EC
            if (nmax <= 2u * EC) {
                const uint32_t h = (nmax + 1) / 2;
                shade_range<EXP, ERF>(s_A, s_B, s_M, s_C, s_q, s_lane, nl, nmax, lane, ray, wv ? h : 0u, wv ? nmax - h : h, Lr, Lg, Lb, La);
            } else {
                shade_lanes<EXP, ERF, EC>(s_A, s_B, s_M, s_C, s_q, s_lane, nl, nmax, lane, ray, Lr, Lg, Lb, La, wv * EC, 2 * EC);
            }
            if (wv == 1) s_L[lane] = make_float4(Lr, Lg, Lb, La);
            __syncthreads();
            if (wv == 0) {
                const float4 o2 = s_L[lane];
                write_block(Lr + o2.x, Lg + o2.y, Lb + o2.z, La + o2.w, valid, out);
            }
        }
        if (O.timeline && first) {
            unsigned long long *tl = O.timeline + 5 * (size_t)item;
            tl[0] = tl0; tl[1] = tl1; tl[2] = tl2; tl[3] = wall_clock64();
            const uint32_t hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_REG_HW_ID, 32 bits
            const uint32_t xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)); // HW_REG_XCC_ID
            tl[4] = ((unsigned long long)nmax << 48) | ((unsigned long long)(xcc & 0xFFFFu) << 32) | hw;
        }
    }
}

template <int EXP, int ERF, int EC, int NW, bool CLAIM = false>
__global__ __launch_bounds__(64 * NW) VRT_RENDER_ATTR void render_kernel(SceneTables S, TileLists T, CellGrid C, RayGen R, RenderTarget O)
{
    render_body<EXP, ERF, EC, NW, CLAIM>(S, T, C, R, O);
}
// several frames per launch: blockIdx.y is the frame (FrameArgs)
template <int EXP, int ERF, int EC, bool CLAIM = false>
__global__ __launch_bounds__(64) VRT_RENDER_ATTR void render_batch_kernel(const FrameArgs *__restrict__ frames)
{
    const FrameArgs &a = frames[blockIdx.y];
    render_body<EXP, ERF, EC, 1, CLAIM>(a.S, a.T, a.C, a.R, a.O);
}

// This file is compiled twice (csrc/Makefile): once for everything except the one-wave image kernel, and once with
// -DVRT_TU_LANES for that kernel alone under -mllvm -amdgpu-sched-strategy=max-ilp.  The default scheduler chains the
// 20 independent erf terms of an absorber one after the other through two registers to save VGPRs, so a wave that is
// alone on its SIMD (the tail of the kernel) crawls; max-ilp interleaves them (145 VGPRs, three waves per SIMD, which is
// what the persistent grid uses anyway).  The 16-wave dense kernel has the thread-level parallelism and keeps the default.
#if !defined(VRT_TU_LANES) && !defined(VRT_TU_TABLE)
// ---------------------------------------------------------------------------------------------
// Dense blocks (hundreds of candidates per 8x8 block: sigma of many pixels, rays of a block see the
// same Gaussians).  One 16-wave workgroup per block: the block's candidates are culled cooperatively
// into LDS once, the EMITTERS are dealt to the 16 waves (wave w takes chunks w, w+16, ...), every
// wave streams all absorbers for its emitters out of LDS (wave-uniform broadcast reads), and the 16
// partial radiances are summed in wave order -- deterministic, no float atomics.  Blocks are pulled
// from a queue with one atomic per block (a block is >= 1e5 instructions; the counter is cold).
// ---------------------------------------------------------------------------------------------
template <int EXP, int ERF, int EC, int DW, bool SKIP = true>
__device__ __forceinline__ void render_dense_body(const SceneTables &S, const TileLists &T, const CellGrid &C, const RayGen &R,
                                                  const RenderTarget &O)
{
    constexpr int DCAP = vrtk::DCAP;
    // One block of LDS carved by hand: the two rows the absorber loop reads sit in the first 64 KB, where a DS
    // instruction's 16-bit offset field reaches them (arrays placed beyond cost a VALU address add per read: +4 %).
    struct Lds {
        float4 A[DCAP], B[DCAP];          // sorted by depth
        float4 L[DW][64];
        uint32_t idx0[DCAP], idx[DCAP];   // "0": in list order; idx: sorted
        float key[DCAP];
    };
    __shared__ Lds lds;
    float4(&s_A)[DCAP] = lds.A;
    float4(&s_B)[DCAP] = lds.B;
    float4(&s_L)[DW][64] = lds.L;
    uint32_t(&s_idx0)[DCAP] = lds.idx0;
    uint32_t(&s_idx)[DCAP] = lds.idx;
    float(&s_key)[DCAP] = lds.key;
    __shared__ uint32_t s_wave_cnt[DW];
    __shared__ uint32_t s_item;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t npix = (uint64_t)R.width * R.height;
    constexpr float SAT = erf_saturation<ERF>();
    constexpr float SAT_M = SAT + 1e-3f; // the range bounds are re-associated forms of the arguments: keep a margin
    const ErfEval<ERF> erf;
    const uint32_t n_dense16 = *C.n_dense * 16u, n_items = n_dense16 + *C.n_overflow;
    if (C.feedback && blockIdx.x == 0 && tid == 0) { // launch feedback: how much this frame had for this kernel
        __hip_atomic_store(&C.feedback[2], n_items, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&C.feedback[3], C.frame_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const uint32_t *dense_queue = C.dense_is_sorted ? C.dense_sorted : C.dense;
    uint32_t *scratch = C.scratch + (size_t)blockIdx.x * C.cstride;
    const unsigned long long t_start = O.stats ? wall_clock64() : 0ull;
    // what the saturation tests decide, per (emitter chunk, absorber) visit of this wave (wave-uniform: scalar adds
    // beside the vector work; written out only when statistics are on)
    uint32_t n_visit_full = 0, n_visit_zero = 0, n_visit_common = 0;

    for (;;) {
        __syncthreads(); // everyone is done with the previous item's LDS
        if (tid == 0) s_item = atomicAdd(C.dense_next, 1u);
        __syncthreads();
        const uint32_t item = s_item;
        if (item >= n_items) {
            if (O.stats && lane == 0) {
                atomicAdd(&O.stats[13], (unsigned long long)n_visit_full); atomicAdd(&O.stats[14], (unsigned long long)n_visit_zero);
                atomicAdd(&O.stats[15], (unsigned long long)n_visit_common);
            }
            if (O.stats && tid == 0) { // workgroup timeline: how long the queue kept this workgroup busy
                const unsigned long long t_end = wall_clock64();
                atomicMin(&O.stats[8], t_start); atomicMax(&O.stats[9], t_end);
                atomicAdd(&O.stats[10], t_end - t_start); atomicAdd(&O.stats[11], 1ull);
            }
            break;
        }
        uint32_t cell, bi;
        if (item < n_dense16) { cell = dense_queue[item >> 4]; bi = item & 15u; }
        else { const uint32_t packed = C.overflow[item - n_dense16]; cell = packed >> 4; bi = packed & 15u; }
        const BlockPos p = block_of(T, C, O, cell, bi, lane);
        if (!p.inside) continue;
        const uint32_t tx = p.t % T.tiles_w, ty = p.t / T.tiles_w;
        bool valid = p.pxt < T.tile_w && p.pyt < T.tile_h;
        const uint32_t pxc = min(p.pxt, T.tile_w - 1), pyc = min(p.pyt, T.tile_h - 1);
        uint64_t pix = (uint64_t)(tx * T.tile_w + pxc) + (uint64_t)T.stride * (ty * T.tile_h + pyc);
        if (pix >= npix) { valid = false; pix = npix - 1; }
        const uint32_t n_active_cells = *C.n_active;
        const uint64_t out = out_index(T, C, O, cell, bi, lane, p, pix, n_active_cells);
        if (O.sparse && item < n_dense16 && bi == 0 && tid == 0) // a dense cell's key (the active cells' are filed by the list kernel)
            O.keys[n_active_cells + (C.slot[cell] & 0x7FFFFFFFu)] = p.t * (C.cells_x * C.cells_y) + cell % (C.cells_x * C.cells_y);

        uint32_t n_list = C.count[cell];
        const uint32_t *list = C.indices + (size_t)cell * C.cstride;
        if (n_list == 0xFFFFFFFFu) { n_list = T.count[p.t]; list = T.indices + T.start[p.t]; }

        const LaneRay ray = pixel_ray(R, pix); // every wave holds the same 64 rays
        float cx = __shfl(ray.nx, 27, 64) + __shfl(ray.nx, 28, 64) + __shfl(ray.nx, 35, 64) + __shfl(ray.nx, 36, 64);
        float cy = __shfl(ray.ny, 27, 64) + __shfl(ray.ny, 28, 64) + __shfl(ray.ny, 35, 64) + __shfl(ray.ny, 36, 64);
        float cz = __shfl(ray.nz, 27, 64) + __shfl(ray.nz, 28, 64) + __shfl(ray.nz, 35, 64) + __shfl(ray.nz, 36, 64);
        {
            const float inv = __builtin_amdgcn_rsqf(cx * cx + cy * cy + cz * cz);
            cx *= inv; cy *= inv; cz *= inv;
        }
        float co, si;
        cos_sin(ray.nx, ray.ny, ray.nz, cx, cy, cz, co, si);
        const Cone cone = make_cone(cx, cy, cz, wave_min(co), wave_max(si));

        // ---- cooperative block cull, order preserving across the 16 waves ----
        uint32_t cnt = 0;
        for (uint32_t base = 0; base < n_list; base += DW * 64) {
            const uint32_t k = base + tid;
            bool keep = false;
            uint32_t idx = 0;
            float4 a, bq;
            if (k < n_list) {
                idx = list[k];
                a = S.gA[idx]; bq = S.gB[idx];
                bq.w = slack_cull_x(bq.w, level_slack(T.cull_ref_n, n_list), T.floor_x);
                keep = cone_keeps(cone, a, bq);
            }
            const unsigned long long mask = __ballot(keep);
            if (lane == 0) s_wave_cnt[wave] = (uint32_t)__popcll(mask);
            __syncthreads();
            uint32_t before = 0, chunk = 0;
#pragma unroll
            for (uint32_t wv = 0; wv < DW; ++wv) {
                const uint32_t c = s_wave_cnt[wv];
                before += (wv < wave) ? c : 0;
                chunk += c;
            }
            const uint32_t pos = cnt + before + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
            if (keep && pos < DCAP) {
                s_idx0[pos] = idx;
                s_key[pos] = a.x * cone.cx + a.y * cone.cy + a.z * cone.cz; // depth along the block's axis
            }
            // the survivors also go to this workgroup's slot of global scratch: should they outgrow LDS, the fallback
            // below streams THEM (cnt^2 pairs) and not the whole cell list (n_list^2)
            if (keep && pos < C.cstride) scratch[pos] = idx;
            cnt += chunk;
            __syncthreads();
        }
        // ---- sort the candidates by depth (rank sort: every thread ranks one candidate against all keys) so
        //      that the emitters of a chunk are neighbours in depth and whole absorbers saturate for them ----
        if (cnt <= DCAP) {
            for (uint32_t i = tid; i < cnt; i += DW * 64) {
                const float ki = s_key[i];
                uint32_t r = 0;
                for (uint32_t k = 0; k < cnt; ++k) {
                    const float kk = s_key[k];
                    r += (kk < ki || (kk == ki && k < i)) ? 1u : 0u;
                }
                const uint32_t idx = s_idx0[i];
                s_idx[r] = idx; s_A[r] = S.gA[idx]; s_B[r] = S.gB[idx]; // rows come back from L2 (read a moment ago)
            }
        }
        __syncthreads();
        if (O.stats && tid == 0) {
            atomicAdd(&O.stats[0], (unsigned long long)(cnt <= C.cstride ? cnt : n_list));
            atomicAdd(&O.stats[1], (unsigned long long)n_list);
            if (cnt > DCAP) atomicAdd(&O.stats[2], 1ull);
            atomicAdd(&O.stats[6], 1ull);
        }

        float Lr = 0.f, Lg = 0.f, Lb = 0.f, La = 0.f;
        if (cnt > DCAP) {
            // does not fit LDS: every wave streams the block's survivors (written above by this workgroup: visible to
            // all its waves after the barrier + fence) for its share of the emitters
            __threadfence_block();
            if (cnt <= C.cstride) shade_list<EXP, ERF, 4, true>(S, scratch, cnt, ray, Lr, Lg, Lb, La, wave * 4, DW * 4);
            else shade_list<EXP, ERF, 4, true>(S, list, n_list, ray, Lr, Lg, Lb, La, wave * 4, DW * 4);
        } else {
            for (uint32_t i0 = wave * EC; i0 < cnt; i0 += DW * EC) {
                float e_mubar[EC], e_sigma[EC];
                uint32_t e_idx[EC];
#pragma unroll
                for (int e = 0; e < EC; ++e) {
                    const uint32_t ii = (i0 + e < cnt) ? (i0 + e) : i0;
                    e_idx[e] = __builtin_amdgcn_readfirstlane(s_idx[ii]);
                    const float4 a = s_A[ii];
                    e_mubar[e] = dot3_ref(a.x, a.y, a.z, ray.nx, ray.ny, ray.nz);
                    e_sigma[e] = uload(S.gD, e_idx[e]).x;
                }
                float acc[EC][5];
#pragma unroll
                for (int e = 0; e < EC; ++e)
#pragma unroll
                    for (int k = 0; k < 5; ++k) acc[e][k] = 0.f;

                // Saturation: Erf(x) is EXACTLY +-1 in fp32 for |x| >= SAT.  An absorber j in front of the camera
                // (m_j >= SAT => E_j = -1) whose Erf argument is <= -SAT at every sample of every emitter of the chunk
                // on every ray of the block adds exactly A_j*(-1 - -1) = 0: skipped before even forming A_j.  One whose
                // arguments are all >= SAT adds exactly -2 A_j to all 5*EC sums: one fma into `common`.
                float common = 0.f;
                float s_max = -INFINITY, s_min = INFINITY; // this ray's sample range over the chunk's emitters
#pragma unroll
                for (int e = 0; e < EC; ++e) {
                    s_max = fmaxf(s_max, e_mubar[e]);
                    s_min = fminf(s_min, __builtin_fmaf(-4.f, e_sigma[e], e_mubar[e]));
                }
                float4 a = s_A[0], b = s_B[0];
                for (uint32_t j = 0; j < cnt; ++j) {
                    const float4 ca = a, cb = b;
                    if (j + 1 < cnt) { a = s_A[j + 1]; b = s_B[j + 1]; }
                    const float mubar = dot3_ref(ca.x, ca.y, ca.z, ray.nx, ray.ny, ray.nz);
                    const float m = mubar * cb.x;
                    // argument range over the chunk's samples on this ray: [(s_min - mubar_j) r_j, (s_max - mubar_j) r_j]
                    const float hi = __builtin_fmaf(s_max, cb.x, -m), lo = __builtin_fmaf(s_min, cb.x, -m);
                    const bool front = m >= SAT;
                    if (SKIP && __all(front && hi <= -SAT_M)) { ++n_visit_zero; continue; }
                    const float d2 = sub_ref(ca.w, mul_ref(mubar, mubar));
                    const float A = cb.z * vexp<EXP>(-(d2 * cb.y));
                    if (SKIP && __all(front && lo >= SAT_M)) { common = __builtin_fmaf(A, -2.f, common); ++n_visit_common; continue; }
                    ++n_visit_full;
                    const float E = erf(-m);
#pragma unroll
                    for (int e = 0; e < EC; ++e) {
                        const float base = __builtin_fmaf(e_mubar[e], cb.x, -m);
                        const float step = e_sigma[e] * cb.x;
#pragma unroll
                        for (int k = 0; k < 5; ++k) {
                            const float x = __builtin_fmaf((float)(k - 4), step, base);
                            acc[e][k] = __builtin_fmaf(A, E - erf(x), acc[e][k]);
                        }
                    }
                }
#pragma unroll
                for (int e = 0; e < EC; ++e) {
                    if (i0 + e < cnt) {
                        const float4 ms = uload(S.mu_sig, e_idx[e]);
                        const float inv2s2 = uload(S.gB, e_idx[e]).y;
                        const float q = uload(S.gD, e_idx[e]).y;
                        float inner = 0.f;
#pragma unroll
                        for (int k = 0; k < 5; ++k) {
                            const float sk = madd_ref((float)(k - 4), ms.w, e_mubar[e]);
                            const float px = sub_ref(madd_ref(ray.nx, sk, ray.ox), ms.x);
                            const float py = sub_ref(madd_ref(ray.ny, sk, ray.oy), ms.y);
                            const float pz = sub_ref(madd_ref(ray.nz, sk, ray.oz), ms.z);
                            const float dd = dot3_ref(px, py, pz, px, py, pz);
                            inner += emission_term<EXP>(q, dd * inv2s2, acc[e][k] + common);
                        }
                        const float4 alb = uload(S.gC, e_idx[e]);
                        Lr = __builtin_fmaf(alb.x, inner, Lr);
                        Lg = __builtin_fmaf(alb.y, inner, Lg);
                        Lb = __builtin_fmaf(alb.z, inner, Lb);
                        La = __builtin_fmaf(alb.w, inner, La);
                    }
                }
            }
        }
        // ---- sum the waves' partial radiances in wave order ----
        s_L[wave][lane] = make_float4(Lr, Lg, Lb, La);
        __syncthreads();
        if (wave == 0 && valid) {
            float4 sum = s_L[0][lane];
#pragma unroll
            for (int w = 1; w < DW; ++w) {
                const float4 v = s_L[w][lane];
                sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
            }
            if (O.image) O.image[out] = pack_pixel(sum.x, sum.y, sum.z, sum.w, O.pack_flags);
            if (O.radiance) O.radiance[out] = sum;
        }
    }
}
template <int EXP, int ERF, int EC, int DW, bool SKIP = true>
__global__ __launch_bounds__(DW * 64, 4) void render_dense_kernel(SceneTables S, TileLists T, CellGrid C, RayGen R,
                                                                RenderTarget O)
{
    render_dense_body<EXP, ERF, EC, DW, SKIP>(S, T, C, R, O);
}
template <int EXP, int ERF, int EC, int DW, bool SKIP = true>
__global__ __launch_bounds__(DW * 64, 4) void render_dense_batch_kernel(const FrameArgs *__restrict__ frames)
{
    const FrameArgs &a = frames[blockIdx.y];
    render_dense_body<EXP, ERF, EC, DW, SKIP>(a.S, a.T, a.C, a.R, a.O);
}

template <int EXP, int ERF, int EC, int DW, bool SKIP = true>
__global__ __launch_bounds__(DW * 64, 4) void render_dense_batch2_kernel(const FrameArgs *__restrict__ frames)
{
    const FrameArgs &a = frames[blockIdx.y];
    render_dense_body<EXP, ERF, EC, DW, SKIP>(a.S, a.T, a.C2, a.R, a.O);
}

// Queue order of the dense kernel: cells by descending candidate count (a block costs ~ count^2), so that the
// blocks still running when the queue empties are the cheapest ones.  Counting sort, one workgroup.
__device__ __forceinline__ void order_dense_body(const CellGrid &C)
{
    __shared__ uint32_t s_hist[1024], s_scan[1024];
    const uint32_t n = *C.n_dense, tid = threadIdx.x;
    s_hist[tid] = 0;
    __syncthreads();
    for (uint32_t i = tid; i < n; i += 1024) atomicAdd(&s_hist[1023u - (min(C.count[C.dense[i]], 4095u) >> 2)], 1u);
    __syncthreads();
    // exclusive prefix over the buckets (bucket 0 = longest lists)
    uint32_t v = s_hist[tid];
    s_scan[tid] = v;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
        const uint32_t add = tid >= off ? s_scan[tid - off] : 0u;
        __syncthreads();
        s_scan[tid] += add;
        __syncthreads();
    }
    s_hist[tid] = s_scan[tid] - v;
    __syncthreads();
    for (uint32_t i = tid; i < n; i += 1024) {
        const uint32_t cell = C.dense[i];
        C.dense_sorted[atomicAdd(&s_hist[1023u - (min(C.count[cell], 4095u) >> 2)], 1u)] = cell;
    }
}

__global__ __launch_bounds__(1024) void order_dense_kernel(CellGrid C) { order_dense_body(C); }
__global__ __launch_bounds__(1024) void order_dense_batch_kernel(const FrameArgs *__restrict__ frames)
{
    const FrameArgs &a = frames[blockIdx.x];
    if (a.do_order) order_dense_body(a.C);
}
void launch_order_dense(const CellGrid &c, hipStream_t st)
{
    hipLaunchKernelGGL(order_dense_kernel, dim3(1), dim3(1024), 0, st, c);
}
void launch_order_dense_batch(const FrameArgs *d_frames, const FrameArgs *h_frames, uint32_t nframes, hipStream_t st)
{
    bool any = false;
    for (uint32_t f = 0; f < nframes; ++f) any = any || h_frames[f].do_order;
    if (any) hipLaunchKernelGGL(order_dense_batch_kernel, dim3(nframes), dim3(1024), 0, st, d_frames);
}

#endif // main translation unit

#ifdef VRT_TU_TABLE
// ---------------------------------------------------------------------------------------------
// Table mode (default; vrt_hip_set_table_step(0) = the exact kernels only): dense blocks through a per-ray TABLE of the
// transmittance exponent.  Along one ray  X(s) = sum_j A_j (E_j - Erf(s r_j - m_j))  is ONE function of s, and the radiance
// needs it at 5 n points (five samples per emitter).  The exact kernel evaluates every one of them term by term: 5 n^2 erf
// terms per ray.  Here X is evaluated at G equidistant nodes of the ray's sample range (n G terms) and the 5 n samples are
// read off by 4-point Lagrange interpolation.
//
// Error control.  The Abramowitz-Stegun erf has a jump of 0.586 in its second derivative at 0 (it is an odd extension of a
// rational function), so X has a kink at every mubar_j.  For a unit erf tabulated with node spacing u (in units of 1/r_j) the
// 4-point interpolant is off by at most 0.0212 u^2 where the stencil contains the kink -- less, by a known factor w <= 1, depending
// on where in the stencil it lies (TB_W0 below) -- and by at most 0.36 u^4 where it does not (tools/table_error_study.py, all phases,
// u <= 0.3).  So for a sample s in node interval g
//     |dX(s)| <= 0.0212 u^2 * K(g) + 0.36 u^4 * S_all,   K(g) = sum of w_j |A_j| over the kinks in intervals g-1 .. g+1,  S_all = sum_j |A_j|,
// and a ray's radiance moves by at most  sum_ik |albedo_i|max * |term_ik| * |dX(s_ik)|  (term_ik = the emission sample).
// The kernel accumulates exactly this sum per ray (K from a per-ray pass over the A_j, kept as one byte per node) and
// keeps a block only if every ray stays below the budget (CellGrid::table_budget, default 2.5e-5: with the cull thresholds'
// 2.5e-5 the frame's worst case is 5e-5, half the 1e-4 tolerance).  A block that fails is tried once more at 0.6 of the
// spacing and then handed to the exact kernel (second queue), as are blocks with more than 2048 survivors or a sample range
// of more than 8 x 376 nodes.  The bound is a worst case (every kink at its worst phase,
// all errors aligned): measured deviations are 20-30 x smaller.
//
// Work split: 16 waves hold the same 64 rays (lane = ray).  Wave w owns the contiguous nodes [w NT, (w+1) NT): an absorber
// whose erf is saturated (exactly -1 or +1, erf_saturation<>) over that whole range on all 64 rays costs one add.  The
// per-(ray, absorber) set-up (A_j, m_j: a dot product, an Exp) is made ONCE per block, 16 absorbers at a time into LDS
// (wave w stages absorber w of the chunk), instead of by every wave for its own nodes.
// ---------------------------------------------------------------------------------------------
constexpr int TB_TC = 2048, TB_GMAX = 384, TB_CH = 16, TB_DW = 16;
static_assert(TB_CH == TB_DW, "one staged absorber per wave and chunk");
// Per-kink error bound of the 4-point interpolant, in units of u^2 |A_j| (tools/table_error_study.py verifies the constants
// for u <= 0.3, all phases): a kink at offset theta in [0, 1) of its node interval moves the interpolant by at most
//   TB_W0 min(1, 0.28 + 2.58 |theta - 1/2|) u^2 in that interval, TB_W0 theta^2 u^2 in the interval to its right,
//   TB_W0 (1 - theta)^2 u^2 in the one to its left, and by at most TB_COUT u^4 anywhere else.
constexpr float TB_W0 = 0.0212f, TB_COUT = 0.36f;
struct TableLds {
    uint32_t idx[TB_TC];                       //  8 KB: the block's survivors, in list order (their rows come from the tables by
                                               //        wave-uniform loads: every pass below needs a row once per wave)
    union {                                    // 96 KB: X at node g of lane l; before that the kink weights per interval (fixed point);
        float tab[TB_GMAX][64];                //        after the emission pass the partial error sums
        uint32_t hist[TB_GMAX][64];
    };
    uint8_t s3[TB_GMAX][64];                   // 24 KB: kink weight of interval g / S_all in 1/255, rounded up
    union {                                    // 24 KB
        struct { float A[2][TB_CH][64], M[2][TB_CH][64], E[2][TB_CH][64]; } st;   // staged set-up, double-buffered
        float4 L[TB_DW][64];                                     // partial radiances
        float red[4][TB_DW][64];                                 // partial sums of the range and histogram passes
    };
    float st_r[2][TB_CH];                      // r_j of the staged absorbers
};

// one pass over the survivors for the NT nodes [g0, g0 + NT) of this wave; tab[g] = sum_j A_j (E_j - Erf(x_gj)), summed per term
// like the exact kernels (C - sum A_j Erf would round at the magnitude of sum |A_j|: 5e-5 of radiance for 1500 wide Gaussians,
// tests/fuzz_parity.py seed 3 case 6)
template <int EXP, int ERF, int NT>
__device__ __forceinline__ void table_nodes(const SceneTables &S, TableLds &lds, uint32_t cnt, const LaneRay &ray,
                                            float s_first /* node g0 of this lane */, float h, uint32_t g0, uint32_t wave,
                                            uint32_t lane, uint32_t &n_skip)
{
    constexpr float SAT_M = erf_saturation<ERF>() + 1e-3f;
    const ErfEval<ERF> erf;
    float acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = 0.f;
    float common = 0.f;
    // Staging: wave w computes (A, m, E) of absorber w of a chunk for its 64 rays.  The absorber's two parameter rows come by
    // wave-uniform loads issued one chunk AHEAD of their use (fetch(c + 2) before the node loop of chunk c): their latency --
    // the longest single wait of a small block -- hides behind the erf terms.
    float4 pa = make_float4(0.f, 0.f, 0.f, 0.f), pb = pa;
    bool pv = false;
    auto fetch = [&](uint32_t chunk) {
        const uint32_t j = chunk * TB_CH + wave;
        pv = j < cnt; // wave-uniform
        if (pv) {
            const uint32_t idx = __builtin_amdgcn_readfirstlane(lds.idx[j]);
            pa = uload(S.gA, idx); pb = uload(S.gB, idx);
        }
    };
    auto stage = [&](uint32_t chunk) { // from the rows fetched last
        const uint32_t b = chunk & 1u;
        float A = 0.f, m = 0.f, r = 0.f, E = 0.f;
        if (pv) {
            const float mubar = dot3_ref(pa.x, pa.y, pa.z, ray.nx, ray.ny, ray.nz);
            const float d2 = sub_ref(pa.w, mul_ref(mubar, mubar));
            A = pb.z * vexp<EXP>(-(d2 * pb.y));
            m = mubar * pb.x;
            E = erf(-m);
            r = pb.x;
        }
        lds.st.A[b][wave][lane] = A; lds.st.M[b][wave][lane] = m; lds.st.E[b][wave][lane] = E;
        if (lane == 0) lds.st_r[b][wave] = r;
    };
    const uint32_t chunks = (cnt + TB_CH - 1) / TB_CH;
    fetch(0);
    stage(0);
    if (chunks > 1) fetch(1);
    __syncthreads();
    for (uint32_t c = 0; c < chunks; ++c) {
        if (c + 1 < chunks) {
            stage(c + 1);
            if (c + 2 < chunks) fetch(c + 2);
        }
        const uint32_t b = c & 1u, nj = min((uint32_t)TB_CH, cnt - c * TB_CH);
        // the next absorber's staged values are requested one iteration ahead (LDS latency behind the erf terms)
        float nA = lds.st.A[b][0][lane], nM = lds.st.M[b][0][lane], nE = lds.st.E[b][0][lane], nR = lds.st_r[b][0];
        for (uint32_t jj = 0; jj < nj; ++jj) {
            const float A = nA, m = nM, E = nE, r = nR;
            if (jj + 1 < nj) { nA = lds.st.A[b][jj + 1][lane]; nM = lds.st.M[b][jj + 1][lane]; nE = lds.st.E[b][jj + 1][lane]; nR = lds.st_r[b][jj + 1]; }
            const float hr = h * r;
            const float x0 = __builtin_fmaf(s_first, r, -m), x1 = __builtin_fmaf((float)(NT - 1), hr, x0);
            // Four wave-uniform questions about the argument range [x0, x1] of this wave's nodes on all rays, asked together (one
            // after the other each would wait for its own vector compare to reach the scalar unit: a quarter of a small block's
            // node loop): saturated -- Erf = -1 (the absorber lies behind the nodes) or +1 (in front): one fma; or of ONE sign (all
            // but the absorbers whose kink lies inside the range): Erf = +-(1 - R), so E - Erf = (E - 1) + R or (E + 1) - R -- ten
            // instructions per term instead of twelve, no sign transfer (v_bfi_b32: 4.3 issue cycles)
            const bool sat_lo = __all(x1 <= -SAT_M), sat_hi = __all(x0 >= SAT_M);
            const bool all_pos = ERF == VRT_ERF_AS && __all(x0 >= 0.f), all_neg = ERF == VRT_ERF_AS && __all(x1 <= 0.f);
            if (sat_lo | sat_hi) {
                common = __builtin_fmaf(A, sat_lo ? E + 1.f : E - 1.f, common);
                ++n_skip;
            } else if (all_pos) {
                if constexpr (ERF == VRT_ERF_AS) {
                    const float Em1 = E - 1.f;
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[t] = __builtin_fmaf(A, Em1 + erf.R(__builtin_fmaf((float)t, hr, x0)), acc[t]);
                }
            } else if (all_neg) {
                if constexpr (ERF == VRT_ERF_AS) {
                    const float Ep1 = E + 1.f;
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[t] = __builtin_fmaf(A, Ep1 - erf.R(__builtin_fmaf((float)t, hr, x0)), acc[t]);
                }
            } else {
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_fmaf(A, E - erf(__builtin_fmaf((float)t, hr, x0)), acc[t]);
            }
        }
        __syncthreads(); // chunk c+1 is staged, and everyone is done with buffer b (chunk c+2 goes there)
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
        if (g0 + t < (uint32_t)TB_GMAX) lds.tab[g0 + t][lane] = acc[t] + common;
}

template <int EXP, int ERF>
__device__ __forceinline__ void render_table_body(const SceneTables &S, const TileLists &T, const CellGrid &C, const RayGen &R,
                                                  const RenderTarget &O)
{
    constexpr int DW = TB_DW, TC = TB_TC;
    __shared__ TableLds lds;
    __shared__ uint32_t s_wave_cnt[DW];
    __shared__ float s_rmax[DW];
    __shared__ uint32_t s_item, s_flag;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t npix = (uint64_t)R.width * R.height;
    const uint32_t n_dense16 = *C.n_dense * 16u, n_items = n_dense16 + *C.n_overflow;
    if (C.feedback && blockIdx.x == 0 && tid == 0) {
        __hip_atomic_store(&C.feedback[2], n_items, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&C.feedback[3], C.frame_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const uint32_t *dense_queue = C.dense_is_sorted ? C.dense_sorted : C.dense;
    uint32_t n_skip = 0; // (absorber, wave) visits the saturation test settled with one add (statistics)

    for (;;) {
        __syncthreads(); // everyone is done with the previous item's LDS
        if (tid == 0) s_item = atomicAdd(C.dense_next, 1u);
        __syncthreads();
        const uint32_t item = s_item;
        if (item >= n_items) break;
        // phase clock (statistics runs only): thread 0 adds the time since the last stamp to stats[24 + phase]
        unsigned long long t_last = (O.stats && tid == 0) ? wall_clock64() : 0ull;
        auto stamp = [&](int phase) {
            if (O.stats && tid == 0) { const unsigned long long t = wall_clock64(); atomicAdd(&O.stats[24 + phase], t - t_last); t_last = t; }
        };
        uint32_t cell, bi;
        if (item < n_dense16) { cell = dense_queue[item >> 4]; bi = item & 15u; }
        else { const uint32_t packed = C.overflow[item - n_dense16]; cell = packed >> 4; bi = packed & 15u; }
        const BlockPos p = block_of(T, C, O, cell, bi, lane);
        if (!p.inside) continue;
        const uint32_t tx = p.t % T.tiles_w, ty = p.t / T.tiles_w;
        bool valid = p.pxt < T.tile_w && p.pyt < T.tile_h;
        const uint32_t pxc = min(p.pxt, T.tile_w - 1), pyc = min(p.pyt, T.tile_h - 1);
        uint64_t pix = (uint64_t)(tx * T.tile_w + pxc) + (uint64_t)T.stride * (ty * T.tile_h + pyc);
        if (pix >= npix) { valid = false; pix = npix - 1; }
        const uint32_t n_active_cells = *C.n_active;
        const uint64_t out = out_index(T, C, O, cell, bi, lane, p, pix, n_active_cells);
        if (O.sparse && item < n_dense16 && bi == 0 && tid == 0) // a dense cell's key (the active cells' are filed by the list kernel)
            O.keys[n_active_cells + (C.slot[cell] & 0x7FFFFFFFu)] = p.t * (C.cells_x * C.cells_y) + cell % (C.cells_x * C.cells_y);

        uint32_t n_list = C.count[cell];
        const uint32_t *list = C.indices + (size_t)cell * C.cstride;
        if (n_list == 0xFFFFFFFFu) { n_list = T.count[p.t]; list = T.indices + T.start[p.t]; }

        const LaneRay ray = pixel_ray(R, pix); // every wave holds the same 64 rays
        float cx = __shfl(ray.nx, 27, 64) + __shfl(ray.nx, 28, 64) + __shfl(ray.nx, 35, 64) + __shfl(ray.nx, 36, 64);
        float cy = __shfl(ray.ny, 27, 64) + __shfl(ray.ny, 28, 64) + __shfl(ray.ny, 35, 64) + __shfl(ray.ny, 36, 64);
        float cz = __shfl(ray.nz, 27, 64) + __shfl(ray.nz, 28, 64) + __shfl(ray.nz, 35, 64) + __shfl(ray.nz, 36, 64);
        {
            const float inv = __builtin_amdgcn_rsqf(cx * cx + cy * cy + cz * cz);
            cx *= inv; cy *= inv; cz *= inv;
        }
        float co, si;
        cos_sin(ray.nx, ray.ny, ray.nz, cx, cy, cz, co, si);
        const Cone cone = make_cone(cx, cy, cz, wave_min(co), wave_max(si));
        stamp(0);

        // ---- cooperative block cull, order preserving across the 16 waves (as in the exact kernel) ----
        uint32_t cnt = 0;
        for (uint32_t base = 0; base < n_list; base += DW * 64) {
            const uint32_t k = base + tid;
            bool keep = false;
            uint32_t idx = 0;
            if (k < n_list) {
                idx = list[k];
                float4 bq = S.gB[idx];
                bq.w = slack_cull_x(bq.w, level_slack(T.cull_ref_n, n_list), T.floor_x);
                keep = cone_keeps(cone, S.gA[idx], bq);
            }
            const unsigned long long mask = __ballot(keep);
            if (lane == 0) s_wave_cnt[wave] = (uint32_t)__popcll(mask);
            __syncthreads();
            uint32_t before = 0, chunk = 0;
#pragma unroll
            for (uint32_t wv = 0; wv < DW; ++wv) {
                const uint32_t c = s_wave_cnt[wv];
                before += (wv < wave) ? c : 0;
                chunk += c;
            }
            const uint32_t pos = cnt + before + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
            if (keep && pos < TC) lds.idx[pos] = idx;
            cnt += chunk;
            __syncthreads();
        }

        stamp(1);
        if (cnt == 0) { // nothing reaches this block (the rim of a dense cell): background
            if (wave == 0 && valid) {
                if (O.image) O.image[out] = pack_pixel(0.f, 0.f, 0.f, 0.f, O.pack_flags);
                if (O.radiance) O.radiance[out] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (O.stats && tid == 0) { atomicAdd(&O.stats[1], (unsigned long long)n_list); atomicAdd(&O.stats[6], 1ull); atomicAdd(&O.stats[7], 1ull); atomicAdd(&O.stats[21], 1ull); }
            continue;
        }
        // ---- every ray's sample range (wave w looks at survivors w, w + 16, ...), the block's node spacing ----
        bool ok = cnt <= (uint32_t)TC;
        float s_lo = INFINITY, s_hi = -INFINITY, r_max = 0.f;
        if (ok) {
            // (rows by wave-uniform loads, the next iteration's requested before this one's arithmetic: here and in the passes below)
            float4 na = make_float4(0.f, 0.f, 0.f, 0.f), nb = na;
            if (wave < cnt) { const uint32_t i0_ = __builtin_amdgcn_readfirstlane(lds.idx[wave]); na = uload(S.gA, i0_); nb = uload(S.gB, i0_); }
            for (uint32_t j = wave; j < cnt; j += DW) {
                const float4 a = na;
                const float r = nb.x;
                if (j + DW < cnt) { const uint32_t in_ = __builtin_amdgcn_readfirstlane(lds.idx[j + DW]); na = uload(S.gA, in_); nb = uload(S.gB, in_); }
                const float mubar = dot3_ref(a.x, a.y, a.z, ray.nx, ray.ny, ray.nz);
                s_hi = fmaxf(s_hi, mubar);
                s_lo = fminf(s_lo, mubar - 2.8285f / r); // mubar - 4 sigma, sigma = 1/(sqrt2 r), rounded outwards
                r_max = fmaxf(r_max, r);
            }
            lds.red[0][wave][lane] = s_hi; lds.red[1][wave][lane] = s_lo;
            if (lane == 0) s_rmax[wave] = r_max;
            __syncthreads();
#pragma unroll
            for (int w = 0; w < DW; ++w) {
                s_hi = fmaxf(s_hi, lds.red[0][w][lane]); s_lo = fminf(s_lo, lds.red[1][w][lane]);
                r_max = fmaxf(r_max, s_rmax[w]);
            }
            __syncthreads();
        }
        stamp(2);
        const float range = wave_max(s_hi - s_lo); // the block's longest sample range
        const float h_req = C.table_hx / r_max;    // requested spacing: table_hx in units of 1/r of the narrowest Gaussian
        ok = ok && range >= 0.f && h_req > 0.f && range < INFINITY; // (false for NaN)
        bool done = false;
        float h_target = h_req;
        for (int attempt = 0; ok && !done; ++attempt) {
            // Every ray has its own grid of Gtot nodes from its first sample on (two nodes of margin at either end, so that
            // every sample has its four neighbours), all with the block's spacing.  The table holds TB_GMAX nodes: a deeper
            // range is worked off in segments of SL intervals (+ the margins); a sample belongs to the segment its interval
            // lies in.  Segments in empty space cost next to nothing: every absorber is saturated there.  The waves evaluate
            // NT nodes each, NT from a short menu: the spacing is then REDUCED until the segments fill 16 NT nodes exactly.
            float h = 0.f, u = 0.f, lo = 0.f, inv_h = 0.f;
            uint32_t nseg = 0, SL = 0, G = 0, NTsel = 0, Gtot = 0;
            auto plan = [&](float ht) -> bool {
                const float need = ceilf(range / ht); // intervals the samples span
                if (!(need < 8.f * (float)(TB_GMAX - 8))) return false;
                nseg = max(1u, ((uint32_t)need + (uint32_t)(TB_GMAX - 8) - 1u) / (uint32_t)(TB_GMAX - 8));
                const uint32_t sl_need = max(1u, ((uint32_t)need + nseg - 1u) / nseg);
                const uint32_t nt = (sl_need + 8u + DW - 1) / DW;
                NTsel = nt <= 4 ? 4 : nt <= 6 ? 6 : nt <= 8 ? 8 : nt <= 12 ? 12 : nt <= 16 ? 16 : nt <= 20 ? 20 : 24;
                // intervals per segment; nodes in the table: the segment's intervals, two nodes before them, and up to six behind
                // the last one (the samples' intervals start at 2 and end at Gtot - 4 <= nseg SL + 2, whose stencil ends at nseg SL + 4)
                G = NTsel * DW; SL = G - 8u;
                h = fminf(ht, range / (float)(nseg * SL) * 1.00001f);
                if (!(h > 0.f)) h = ht; // range == 0: one sample point per ray
                u = h * r_max;
                Gtot = nseg * SL + 6u;
                lo = s_lo - 2.f * h; inv_h = 1.f / h;
                return u <= 0.3f;
            };
            if (!plan(h_target)) { ok = false; break; }

            float S_all = 0.f;
            // ---- kink pass for one segment: wave w takes absorbers w, w + 16, ...: the weight of the kink of j (TB_W0 units,
            //      fixed point, rounded up; integer adds: the order of the atomics does not matter) into the interval of mubar_j
            //      and its two neighbours; with `sums` also S_all = sum |A_j| ----
            auto kink_pass = [&](uint32_t seg, bool sums) {
                const float node0 = (float)(seg * SL) - 2.f; // the segment's first node on the ray's grid
                for (uint32_t g = wave; g < G; g += DW) lds.hist[g][lane] = 0u;
                __syncthreads();
                float s_part = 0.f;
                float4 na = make_float4(0.f, 0.f, 0.f, 0.f), nb = na;
                if (wave < cnt) { const uint32_t i0_ = __builtin_amdgcn_readfirstlane(lds.idx[wave]); na = uload(S.gA, i0_); nb = uload(S.gB, i0_); }
                for (uint32_t j = wave; j < cnt; j += DW) {
                    const float4 ca = na, cb = nb;
                    if (j + DW < cnt) { const uint32_t in_ = __builtin_amdgcn_readfirstlane(lds.idx[j + DW]); na = uload(S.gA, in_); nb = uload(S.gB, in_); }
                    const float mubar = dot3_ref(ca.x, ca.y, ca.z, ray.nx, ray.ny, ray.nz);
                    const float d2 = sub_ref(ca.w, mul_ref(mubar, mubar));
                    const float A = cb.z * vexp<EXP>(-(d2 * cb.y));
                    if (sums) s_part += fabsf(A);
                    const float pos = (mubar - lo) * inv_h;
                    const float gb = floorf(pos);
                    const float th = fminf(fmaxf(pos - gb, 0.f), 1.f);
                    const float a16 = fminf(fabsf(A), 60.f) * 65536.f;
                    const float gl = gb - node0; // interval of the kink in this segment's table
                    if (gl >= 0.f && gl < (float)G) atomicAdd(&lds.hist[(uint32_t)gl][lane], (uint32_t)ceilf(a16 * fminf(1.f, 0.28f + 2.58f * fabsf(th - 0.5f))));
                    if (gl + 1.f >= 0.f && gl + 1.f < (float)G) atomicAdd(&lds.hist[(uint32_t)(gl + 1.f)][lane], (uint32_t)ceilf(a16 * th * th));
                    if (gl - 1.f >= 0.f && gl - 1.f < (float)G) atomicAdd(&lds.hist[(uint32_t)(gl - 1.f)][lane], (uint32_t)ceilf(a16 * (1.f - th) * (1.f - th)));
                }
                if (sums) lds.red[3][wave][lane] = s_part;
                __syncthreads();
                if (sums) {
                    S_all = 0.f;
#pragma unroll
                    for (int w = 0; w < DW; ++w) S_all += lds.red[3][w][lane];
                }
            };

            // ---- first attempt: how much coarser than requested may the nodes be?  An ESTIMATE of the bound the emission pass
            //      will find, from the kink weights at the requested spacing: emission of interval g ~ 1.4 D_g T_g with
            //      D_g = K_g / 1.35 the absorber mass of the interval and T_g = exp(-2 sum of the mass before it); the estimate
            //      scales with the spacing like kappa^3 (kink part) and kappa^4 (smooth part).  It only picks the spacing: the
            //      bound itself is checked below, and a block that fails it is redone at 0.6 of the spacing. ----
            bool have_kinks = false; // the kink weights of segment 0 at the final spacing are in LDS
            bool have_sums = false;  // S_all is known (it does not depend on the spacing)
            if (attempt == 0 && C.table_adapt > 1.f) {
                float P = 0.f, pin = 0.f, pout = 0.f;
                for (uint32_t seg = 0; seg < nseg; ++seg) {
                    kink_pass(seg, seg == 0);
                    const uint32_t ga = seg ? 2u : 0u, gz = min(G, SL + 2u); // the segment's own intervals
                    const uint32_t wa = min(gz, ga + wave * NTsel), wz = min(gz, wa + NTsel);
                    float mass = 0.f;
                    for (uint32_t g = wa; g < wz; ++g) mass += (float)lds.hist[g][lane];
                    lds.red[0][wave][lane] = mass * (1.f / (1.35f * 65536.f));
                    __syncthreads();
                    float before = P, total = 0.f;
#pragma unroll
                    for (int w = 0; w < DW; ++w) {
                        const float mw = lds.red[0][w][lane];
                        before += (uint32_t)w < wave ? mw : 0.f;
                        total += mw;
                    }
                    for (uint32_t g = wa; g < wz; ++g) {
                        const float Kg = (float)lds.hist[g][lane] * (1.f / 65536.f), D = Kg * (1.f / 1.35f);
                        const float e = 1.4f * D * __expf(-2.f * before);
                        pin = __builtin_fmaf(e, Kg, pin); pout += e;
                        before += D;
                    }
                    P += total;
                    __syncthreads(); // red[0] is rewritten by the next segment
                }
                have_sums = true;
                lds.red[0][wave][lane] = pin; lds.red[1][wave][lane] = pout;
                __syncthreads();
                float pin_t = 0.f, pout_t = 0.f;
#pragma unroll
                for (int w = 0; w < DW; ++w) { pin_t += lds.red[0][w][lane]; pout_t += lds.red[1][w][lane]; }
                const float est_in = 1.01f * TB_W0 * u * u * pin_t, est_out = 1.01f * TB_COUT * (u * u) * (u * u) * S_all * pout_t;
                float kappa = 1.f;
                const float room = C.table_room * C.table_budget;
#pragma unroll
                for (int c = 0; c < 5; ++c) {
                    const float k = c == 0 ? 3.f : c == 1 ? 2.5f : c == 2 ? 2.f : c == 3 ? 1.6f : 1.3f;
                    if (kappa == 1.f && k <= C.table_adapt && k * u <= 0.3f && k * k * k * (est_in + k * est_out) <= room) kappa = k;
                }
                kappa = wave_min(valid ? kappa : 3.f); // the same in every wave: they hold the same rays
                __syncthreads();
                const float h_before = h;
                const uint32_t nodes_before = nseg * NTsel, nseg_before = nseg;
                if (kappa > 1.f && (!plan(h * kappa) || nseg * NTsel >= nodes_before)) { // the menu has no smaller table: as requested
                    if (!plan(h_target)) { ok = false; break; }
                }
                have_kinks = h == h_before && nseg_before == 1u; // segment 0's weights at this spacing are still in LDS
                if (O.stats && tid == 0 && h != h_before) atomicAdd(&O.stats[20], 1ull);
            }
            stamp(3);
            const uint32_t g0 = wave * NTsel;
            float Lr = 0.f, Lg = 0.f, Lb = 0.f, La = 0.f, b_in = 0.f, b_out = 0.f;

            for (uint32_t seg = 0; seg < nseg; ++seg) {
                const float node0 = (float)(seg * SL) - 2.f; // the segment's first node on the ray's grid
                if (!(seg == 0 && have_kinks)) kink_pass(seg, seg == 0 && !have_sums);
                const float s3_scale = 255.f / (fmaxf(S_all, 1e-30f) * 65536.f);
                for (uint32_t g = wave; g < G; g += DW)
                    lds.s3[g][lane] = (uint8_t)fminf(floorf((float)lds.hist[g][lane] * s3_scale) + 1.f, 255.f);
                __syncthreads(); // the weights are read: their memory becomes the table; the partial sums' memory the staging buffers
                stamp(4);

                // ---- table: wave w evaluates the nodes [w NT, (w+1) NT) of the segment against all survivors ----
                const float s_first = __builtin_fmaf(node0 + (float)g0, h, lo);
                switch (NTsel) {
                case 4: table_nodes<EXP, ERF, 4>(S, lds, cnt, ray, s_first, h, g0, wave, lane, n_skip); break;
                case 6: table_nodes<EXP, ERF, 6>(S, lds, cnt, ray, s_first, h, g0, wave, lane, n_skip); break;
                case 8: table_nodes<EXP, ERF, 8>(S, lds, cnt, ray, s_first, h, g0, wave, lane, n_skip); break;
                case 12: table_nodes<EXP, ERF, 12>(S, lds, cnt, ray, s_first, h, g0, wave, lane, n_skip); break;
                case 16: table_nodes<EXP, ERF, 16>(S, lds, cnt, ray, s_first, h, g0, wave, lane, n_skip); break;
                case 20: table_nodes<EXP, ERF, 20>(S, lds, cnt, ray, s_first, h, g0, wave, lane, n_skip); break;
                default: table_nodes<EXP, ERF, 24>(S, lds, cnt, ray, s_first, h, g0, wave, lane, n_skip); break;
                }
                __syncthreads();
                stamp(5);

                // ---- emission: the emitters are dealt to the waves; X(s_ik) by 4-point Lagrange interpolation for the samples
                //      of this segment; the error bound is accumulated beside the radiance ----
                const float seg_lo = (float)(seg * SL), seg_hi = seg + 1 == nseg ? INFINITY : (float)((seg + 1) * SL);
                // (emitter i of wave w: w, w + 16, ...; the five rows of the NEXT emitter are requested before this one's samples)
                struct Rows { float4 a, ms, alb; float inv2s2, q; } nx = {};
                auto fetch_rows = [&](uint32_t i) {
                    const uint32_t idx = __builtin_amdgcn_readfirstlane(lds.idx[i]);
                    nx.a = uload(S.gA, idx); nx.ms = uload(S.mu_sig, idx); nx.alb = uload(S.gC, idx);
                    nx.inv2s2 = uload(S.gB, idx).y; nx.q = uload(S.gD, idx).y;
                };
                if (wave < cnt) fetch_rows(wave);
                for (uint32_t i = wave; i < cnt; i += DW) {
                    {
                        const Rows cur = nx;
                        if (i + DW < cnt) fetch_rows(i + DW);
                        const float4 a = cur.a, ms = cur.ms, alb = cur.alb;
                        const float inv2s2 = cur.inv2s2, q = cur.q;
                        const float e_mubar = dot3_ref(a.x, a.y, a.z, ray.nx, ray.ny, ray.nz);
                        float inner = 0.f, inner_abs = 0.f, inner_s3 = 0.f;
#pragma unroll
                        for (int k = 0; k < 5; ++k) {
                            const float sk = madd_ref((float)(k - 4), ms.w, e_mubar);
                            const float uu = (sk - lo) * inv_h;
                            const float gi = fminf(fmaxf(floorf(uu), 2.f), (float)(Gtot - 4)); // interval on the ray's grid
                            const bool mine = nseg == 1 || (gi >= seg_lo && gi < seg_hi);
                            const float gfl = mine ? gi - node0 : 2.f; // ... and in the table
                            const float t = uu - (gfl + node0);
                            const uint32_t g = (uint32_t)gfl;
                            const float px = sub_ref(madd_ref(ray.nx, sk, ray.ox), ms.x);
                            const float py = sub_ref(madd_ref(ray.ny, sk, ray.oy), ms.y);
                            const float pz = sub_ref(madd_ref(ray.nz, sk, ray.oz), ms.z);
                            const float dd = dot3_ref(px, py, pz, px, py, pz);
                            const float tm1 = t - 1.f, tm2 = t - 2.f, tp1 = t + 1.f;
                            const float w0 = t * tm1 * tm2 * (-1.f / 6.f), w1 = tp1 * tm1 * tm2 * 0.5f;
                            const float w2 = tp1 * t * tm2 * -0.5f, w3 = tp1 * t * tm1 * (1.f / 6.f);
                            const float X = w0 * lds.tab[g - 1][lane] + w1 * lds.tab[g][lane] + w2 * lds.tab[g + 1][lane] + w3 * lds.tab[g + 2][lane];
                            const float term = mine ? emission_term<EXP>(q, dd * inv2s2, X) : 0.f;
                            inner += term;
                            inner_abs += fabsf(term);
                            inner_s3 = __builtin_fmaf(fabsf(term), (float)lds.s3[g][lane], inner_s3);
                        }
                        Lr = __builtin_fmaf(alb.x, inner, Lr);
                        Lg = __builtin_fmaf(alb.y, inner, Lg);
                        Lb = __builtin_fmaf(alb.z, inner, Lb);
                        La = __builtin_fmaf(alb.w, inner, La);
                        const float amax = fmaxf(fmaxf(fabsf(alb.x), fabsf(alb.y)), fmaxf(fabsf(alb.z), fabsf(alb.w)));
                        b_in = __builtin_fmaf(amax, inner_s3, b_in);
                        b_out = __builtin_fmaf(amax, inner_abs, b_out);
                    }
                }
                __syncthreads(); // nobody reads the staging buffers or the table any more
                stamp(6);
            }
            lds.L[wave][lane] = make_float4(Lr, Lg, Lb, La);
            float2 *bparts = reinterpret_cast<float2 *>(&lds.tab[0][0]); // [DW][64]
            bparts[wave * 64 + lane] = make_float2(b_in, b_out);
            __syncthreads();
            if (wave == 0) {
                float4 sum = lds.L[0][lane];
                float2 bs = bparts[lane];
#pragma unroll
                for (int w = 1; w < DW; ++w) {
                    const float4 v = lds.L[w][lane];
                    sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
                    const float2 bv = bparts[w * 64 + lane];
                    bs.x += bv.x; bs.y += bv.y;
                }
                // worst-case change of this ray's radiance (header comment); e^dX - 1 <= 1.01 dX for the dX in question
                const float e_in = TB_W0 * u * u, e_out = TB_COUT * (u * u) * (u * u);
                const float bound = 1.01f * S_all * (e_in * (1.f / 255.f) * bs.x + e_out * bs.y);
                // (false for NaN; S_all beyond the fixed-point range of the weights: no bound)
                const bool good = !valid || (bound <= C.table_budget && S_all < 60.f);
                const bool all_good = __all(good);
                if (all_good && valid) {
                    if (O.image) O.image[out] = pack_pixel(sum.x, sum.y, sum.z, sum.w, O.pack_flags);
                    if (O.radiance) O.radiance[out] = sum;
                }
                if (lane == 0) s_flag = all_good ? 1u : 0u;
            }
            __syncthreads();
            stamp(7);
            done = s_flag != 0u;
            if (!done) {
                if (attempt >= 1) { ok = false; break; }
                h_target = 0.6f * h;
            } else if (O.stats && tid == 0) {
                atomicAdd(&O.stats[0], (unsigned long long)cnt);
                atomicAdd(&O.stats[1], (unsigned long long)n_list);
                atomicAdd(&O.stats[6], 1ull);
                atomicAdd(&O.stats[7], 1ull);
                atomicAdd(&O.stats[16], (unsigned long long)Gtot);
                if (attempt) atomicAdd(&O.stats[17], 1ull);
            }
        }
        if (!ok) { // wave-uniform and the same in every wave
            if (tid == 0) {
                C.overflow2[atomicAdd(C.n_overflow2, 1u)] = (cell << 4) | bi;
                if (O.stats) atomicAdd(&O.stats[19], 1ull);
            }
            continue;
        }
    }
    if (O.stats && lane == 0) atomicAdd(&O.stats[18], (unsigned long long)n_skip);
}

template <int EXP, int ERF>
__global__ __launch_bounds__(1024) void render_table_kernel(SceneTables S, TileLists T, CellGrid C, RayGen R, RenderTarget O)
{
    render_table_body<EXP, ERF>(S, T, C, R, O);
}
// several frames per launch: blockIdx.y is the frame; C2 is the frame's second queue (what the table kernel declines)
template <int EXP, int ERF>
__global__ __launch_bounds__(1024) void render_table_batch_kernel(const FrameArgs *__restrict__ frames)
{
    const FrameArgs &a = frames[blockIdx.y];
    render_table_body<EXP, ERF>(a.S, a.T, a.C, a.R, a.O);
}
#endif // VRT_TU_TABLE

#ifdef VRT_TU_LANES
template <int EXP, int ERF>
static void launch_render_t(const SceneTables &s, const TileLists &t, const CellGrid &c, const RayGen &r,
                            const RenderTarget &o, uint32_t grid, int nw, hipStream_t st)
{
    if (grid == 0) return;
    if (nw == 2) hipLaunchKernelGGL((render_kernel<EXP, ERF, 4, 2>), dim3(grid), dim3(128), 0, st, s, t, c, r, o);
    else if (c.claim_early > 0 && EXP == VRT_EXP_VCL && ERF == VRT_ERF_AS) // frames with many more blocks than waves (the default pair only: compile time)
        hipLaunchKernelGGL((render_kernel<VRT_EXP_VCL, VRT_ERF_AS, 4, 1, true>), dim3(grid), dim3(64), 0, st, s, t, c, r, o);
    else hipLaunchKernelGGL((render_kernel<EXP, ERF, 4, 1>), dim3(grid), dim3(64), 0, st, s, t, c, r, o);
}

template <int EXP, int ERF>
static void launch_render_batch_t(const FrameArgs *d_frames, uint32_t nframes, uint32_t grid, bool claim, hipStream_t st)
{
    if (grid == 0 || nframes == 0) return;
    if (claim && EXP == VRT_EXP_VCL && ERF == VRT_ERF_AS) hipLaunchKernelGGL((render_batch_kernel<VRT_EXP_VCL, VRT_ERF_AS, 4, true>), dim3(grid, nframes), dim3(64), 0, st, d_frames);
    else hipLaunchKernelGGL((render_batch_kernel<EXP, ERF, 4>), dim3(grid, nframes), dim3(64), 0, st, d_frames);
}

#endif // VRT_TU_LANES

#define VRT_DISPATCH_EXP_ERF(FN, ...)                                                              \
    switch (exp_kind * 8 + erf_kind) {                                                             \
    case VRT_EXP_LIBM * 8 + VRT_ERF_LIBM: FN<VRT_EXP_LIBM, VRT_ERF_LIBM>(__VA_ARGS__); break;      \
    case VRT_EXP_LIBM * 8 + VRT_ERF_AS: FN<VRT_EXP_LIBM, VRT_ERF_AS>(__VA_ARGS__); break;          \
    case VRT_EXP_VCL * 8 + VRT_ERF_LIBM: FN<VRT_EXP_VCL, VRT_ERF_LIBM>(__VA_ARGS__); break;        \
    case VRT_EXP_VCL * 8 + VRT_ERF_AS: FN<VRT_EXP_VCL, VRT_ERF_AS>(__VA_ARGS__); break;            \
    case VRT_EXP_FAST * 8 + VRT_ERF_AS: FN<VRT_EXP_FAST, VRT_ERF_AS>(__VA_ARGS__); break;          \
    case VRT_EXP_SPLINE * 8 + VRT_ERF_AS: FN<VRT_EXP_SPLINE, VRT_ERF_AS>(__VA_ARGS__); break;      \
    case VRT_EXP_VCL * 8 + VRT_ERF_SPLINE: FN<VRT_EXP_VCL, VRT_ERF_SPLINE>(__VA_ARGS__); break;    \
    case VRT_EXP_VCL * 8 + VRT_ERF_SPLINE_MIRROR: FN<VRT_EXP_VCL, VRT_ERF_SPLINE_MIRROR>(__VA_ARGS__); break; \
    case VRT_EXP_VCL * 8 + VRT_ERF_TAYLOR: FN<VRT_EXP_VCL, VRT_ERF_TAYLOR>(__VA_ARGS__); break;    \
    default: FN<VRT_EXP_VCL, VRT_ERF_AS>(__VA_ARGS__); break;                                      \
    }

#ifdef VRT_TU_LANES
void launch_render(const SceneTables &s, const TileLists &t, const CellGrid &c, const RayGen &r, const RenderTarget &o,
                   uint32_t grid, int nw, int exp_kind, int erf_kind, hipStream_t st)
{
    VRT_DISPATCH_EXP_ERF(launch_render_t, s, t, c, r, o, grid, nw, st);
}
void launch_render_batch(const FrameArgs *d_frames, uint32_t nframes, uint32_t grid, bool claim, int exp_kind, int erf_kind, hipStream_t st)
{
    VRT_DISPATCH_EXP_ERF(launch_render_batch_t, d_frames, nframes, grid, claim, st);
}
#elif defined(VRT_TU_TABLE)
// The table kernel's error bound is that of the Abramowitz-Stegun erf (its kink) or of a smoother one (libm); the Exp must
// be an accurate one (Exp(a)Exp(b) = Exp(a + b)): four pairs are instantiated, the host keeps every other pair exact.
#define VRT_DISPATCH_TABLE(FN, ...)                                                                \
    switch (exp_kind * 8 + erf_kind) {                                                             \
    case VRT_EXP_LIBM * 8 + VRT_ERF_LIBM: FN<VRT_EXP_LIBM, VRT_ERF_LIBM>(__VA_ARGS__); break;      \
    case VRT_EXP_LIBM * 8 + VRT_ERF_AS: FN<VRT_EXP_LIBM, VRT_ERF_AS>(__VA_ARGS__); break;          \
    case VRT_EXP_VCL * 8 + VRT_ERF_LIBM: FN<VRT_EXP_VCL, VRT_ERF_LIBM>(__VA_ARGS__); break;        \
    default: FN<VRT_EXP_VCL, VRT_ERF_AS>(__VA_ARGS__); break;                                      \
    }
template <int EXP, int ERF>
static void launch_render_table_t(const SceneTables &s, const TileLists &t, const CellGrid &c, const RayGen &r,
                                  const RenderTarget &o, uint32_t grid, hipStream_t st)
{
    if (grid == 0) return;
    hipLaunchKernelGGL((render_table_kernel<EXP, ERF>), dim3(grid), dim3(1024), 0, st, s, t, c, r, o);
}
void launch_render_table(const SceneTables &s, const TileLists &t, const CellGrid &c, const RayGen &r,
                         const RenderTarget &o, uint32_t grid, int exp_kind, int erf_kind, hipStream_t st)
{
    VRT_DISPATCH_TABLE(launch_render_table_t, s, t, c, r, o, grid, st);
}
template <int EXP, int ERF>
static void launch_render_table_only_batch_t(const FrameArgs *d_frames, uint32_t nframes, uint32_t grid, hipStream_t st)
{
    if (nframes && grid) hipLaunchKernelGGL((render_table_batch_kernel<EXP, ERF>), dim3(grid, nframes), dim3(1024), 0, st, d_frames);
}
void launch_render_table_only_batch(const FrameArgs *d_frames, uint32_t nframes, uint32_t grid, int exp_kind, int erf_kind, hipStream_t st)
{
    VRT_DISPATCH_TABLE(launch_render_table_only_batch_t, d_frames, nframes, grid, st);
}
#else

template <int EXP, int ERF>
static void launch_render_dense_t(const SceneTables &s, const TileLists &t, const CellGrid &c, const RayGen &r,
                                  const RenderTarget &o, uint32_t grid, int dw, hipStream_t st)
{
    if (grid == 0) return;
    if (dw == 17) hipLaunchKernelGGL((render_dense_kernel<EXP, ERF, 6, 16, false>), dim3(grid), dim3(1024), 0, st, s, t, c, r, o);
    else if (dw == 16) hipLaunchKernelGGL((render_dense_kernel<EXP, ERF, 6, 16>), dim3(grid), dim3(1024), 0, st, s, t, c, r, o);
    else if (dw == 8) hipLaunchKernelGGL((render_dense_kernel<EXP, ERF, 6, 8>), dim3(grid), dim3(512), 0, st, s, t, c, r, o);
    else hipLaunchKernelGGL((render_dense_kernel<EXP, ERF, 6, 4>), dim3(grid), dim3(256), 0, st, s, t, c, r, o);
}
void launch_render_dense(const SceneTables &s, const TileLists &t, const CellGrid &c, const RayGen &r,
                         const RenderTarget &o, uint32_t grid, int dw, int exp_kind, int erf_kind, hipStream_t st)
{
    VRT_DISPATCH_EXP_ERF(launch_render_dense_t, s, t, c, r, o, grid, dw, st);
}
template <int EXP, int ERF>
static void launch_render_dense_batch_t(const FrameArgs *d_frames, uint32_t nframes, uint32_t grid, int dw, hipStream_t st)
{
    if (grid == 0 || nframes == 0) return;
    const dim3 g(grid, nframes);
    if (dw == 17) hipLaunchKernelGGL((render_dense_batch_kernel<EXP, ERF, 6, 16, false>), g, dim3(1024), 0, st, d_frames);
    else if (dw == 16) hipLaunchKernelGGL((render_dense_batch_kernel<EXP, ERF, 6, 16>), g, dim3(1024), 0, st, d_frames);
    else if (dw == 8) hipLaunchKernelGGL((render_dense_batch_kernel<EXP, ERF, 6, 8>), g, dim3(512), 0, st, d_frames);
    else hipLaunchKernelGGL((render_dense_batch_kernel<EXP, ERF, 6, 4>), g, dim3(256), 0, st, d_frames);
}
void launch_render_dense_batch(const FrameArgs *d_frames, uint32_t nframes, uint32_t grid, int dw, int exp_kind, int erf_kind,
                               hipStream_t st)
{
    VRT_DISPATCH_EXP_ERF(launch_render_dense_batch_t, d_frames, nframes, grid, dw, st);
}

void launch_render_table_only_batch(const FrameArgs *d_frames, uint32_t nframes, uint32_t grid, int exp_kind, int erf_kind, hipStream_t st); // table TU
template <int EXP, int ERF>
static void launch_render_table_batch_t(const FrameArgs *d_frames, uint32_t nframes, uint32_t grid2, int dw, hipStream_t st)
{
    if (!nframes || !grid2) return;
    const dim3 g(grid2, nframes);
    if (dw == 17) hipLaunchKernelGGL((render_dense_batch2_kernel<EXP, ERF, 6, 16, false>), g, dim3(1024), 0, st, d_frames);
    else if (dw == 16) hipLaunchKernelGGL((render_dense_batch2_kernel<EXP, ERF, 6, 16>), g, dim3(1024), 0, st, d_frames);
    else if (dw == 8) hipLaunchKernelGGL((render_dense_batch2_kernel<EXP, ERF, 6, 8>), g, dim3(512), 0, st, d_frames);
    else hipLaunchKernelGGL((render_dense_batch2_kernel<EXP, ERF, 6, 4>), g, dim3(256), 0, st, d_frames);
}
// table kernel over the frames' dense queues, then the exact kernel over what it declined (FrameArgs::C2)
void launch_render_table_batch(const FrameArgs *d_frames, uint32_t nframes, uint32_t grid, uint32_t grid2, int dw, int exp_kind, int erf_kind,
                               hipStream_t st)
{
    launch_render_table_only_batch(d_frames, nframes, grid, exp_kind, erf_kind, st);
    VRT_DISPATCH_EXP_ERF(launch_render_table_batch_t, d_frames, nframes, grid2, dw, st);
}

// ---------------------------------------------------------------------------------------------
// Scene tables
// ---------------------------------------------------------------------------------------------
__global__ void build_static_kernel(uint32_t n, const float *mu_x, const float *mu_y, const float *mu_z,
                                    const float *ar, const float *ag, const float *ab, const float *aa,
                                    const float *sigma, const float *mag, float cull_eps, float exp_floor_x,
                                    float4 *mu_sig, float4 *gB, float4 *gC, float4 *gD)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float s = sigma[i], m = mag[i];
    mu_sig[i] = make_float4(mu_x[i], mu_y[i], mu_z[i], s);
    const float q = s * m;
    // cull_x: drop when d^2/(2 sigma^2) > ln(|q|/eps); never keep what Exp flushes to zero anyway
    float cull_x = exp_floor_x;
    if (q == 0.f) cull_x = -INFINITY;
    else if (cull_eps > 0.f) cull_x = fminf(cull_x, logf(fabsf(q) / cull_eps));
    gB[i] = make_float4(1.f / (SQRT_2 * s), 1.f / (2.f * s * s), q * INV_SQRT_2_PI, cull_x);
    gC[i] = make_float4(ar[i], ag[i], ab[i], aa ? aa[i] : 1.f);
    gD[i] = make_float4(s, q, m, 0.f);
}

void launch_build_static(uint32_t n, const float *mu_x, const float *mu_y, const float *mu_z, const float *ar,
                         const float *ag, const float *ab, const float *aa, const float *sigma, const float *mag,
                         float cull_eps, float exp_floor_x, float4 *mu_sig, float4 *gB, float4 *gC, float4 *gD,
                         hipStream_t st)
{
    if (!n) return;
    hipLaunchKernelGGL(build_static_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, mu_x, mu_y, mu_z, ar, ag, ab,
                       aa, sigma, mag, cull_eps, exp_floor_x, mu_sig, gB, gC, gD);
}

__global__ void prep_frame_kernel(uint32_t n, const float4 *mu_sig, float4 *gA, float ox, float oy, float oz)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 m = mu_sig[i];
    const float cx = m.x - ox, cy = m.y - oy, cz = m.z - oz;
    gA[i] = make_float4(cx, cy, cz, dot3_ref(cx, cy, cz, cx, cy, cz)); // vec4f_t::sqnorm order (types.h:69-72)
}

__global__ void prep_frame_batch_kernel(const FrameArgs *__restrict__ frames)
{
    const FrameArgs &a = frames[blockIdx.y];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (!a.do_prep || i >= a.S.n) return;
    const float4 m = a.S.mu_sig[i];
    const float cx = m.x - a.prep_origin[0], cy = m.y - a.prep_origin[1], cz = m.z - a.prep_origin[2];
    a.prep_gA[i] = make_float4(cx, cy, cz, dot3_ref(cx, cy, cz, cx, cy, cz));
}
void launch_prep_frame(const SceneTables &s, float4 *gA_out, const float origin[3], hipStream_t st)
{
    if (!s.n) return;
    hipLaunchKernelGGL(prep_frame_kernel, dim3((s.n + 255) / 256), dim3(256), 0, st, s.n, s.mu_sig, gA_out, origin[0],
                       origin[1], origin[2]);
}

__global__ void iota_kernel(uint32_t *p, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = i;
}
void launch_iota(uint32_t *p, uint32_t n, hipStream_t st)
{
    if (n) hipLaunchKernelGGL(iota_kernel, dim3((n + 255) / 256), dim3(256), 0, st, p, n);
}

// ---------------------------------------------------------------------------------------------
// Per-tile Gaussian lists.  One 1024-thread workgroup per reference tile builds, in one pass,
//   (a) the reference's tile set: vrt/rt.cpp:29-69 on device (or takes a caller-made list), and
//   (b) optionally ("refine") drops from it every Gaussian that is below cull_eps for the whole tile
//       (cone through the tile's corner rays), so that the 8x8-block cull of the render kernel
//       scans tens of candidates instead of the reference's ~1600.
// The reference-set arithmetic is kept unfused and in the reference's order so that inclusion
// decisions (a "<=" on floats) reproduce the host algorithm; order of indices is preserved.
// ---------------------------------------------------------------------------------------------
// With F.enabled the same workgroup goes on to the second level: the tile's surviving candidates stay in LDS
// (index + the two parameter rows the cone test reads) and each of its 16 waves filters them for the tile's
// 32x32-pixel cells, files every non-empty cell as active or dense (at most two atomics per TILE) and clears the
// pixels of empty cells on the spot -- no second kernel, no global round trip, no idle clear phase later.
// cone of a whole reference tile, from the centre and the four corner pixels (pinhole rays: the farthest ray of a rectangle
// on the image plane from its centre ray is a corner ray)
__device__ __forceinline__ Cone tile_cone(const BinArgs &P, uint32_t tx, uint32_t ty, uint32_t lane)
{
    const uint64_t npix0 = (uint64_t)P.R.width * P.R.height;
    auto at = [&](uint32_t x, uint32_t y) {
        uint64_t pix = (uint64_t)(tx * P.tile_w + x) + (uint64_t)P.stride * (ty * P.tile_h + y);
        if (pix >= npix0) pix = npix0 - 1;
        return cone_ray(P.R, pix);
    };
    return rect_cone(at, 0, 0, P.tile_w - 1, P.tile_h - 1, lane);
}
// cone of cell ci of a tile (32x32 px, clipped to the tile), as the second level builds it
__device__ __forceinline__ Cone cell_cone(const BinArgs &P, uint32_t tx, uint32_t ty, uint32_t ci, uint32_t cells_x, uint32_t lane)
{
    const uint64_t npix = (uint64_t)P.R.width * P.R.height;
    const uint32_t x0 = (ci % cells_x) * CELL, y0 = (ci / cells_x) * CELL;
    const uint32_t x1 = min(x0 + CELL, P.tile_w) - 1, y1 = min(y0 + CELL, P.tile_h) - 1;
    auto at = [&](uint32_t x, uint32_t y) {
        uint64_t pix = (uint64_t)(tx * P.tile_w + x) + (uint64_t)P.stride * (ty * P.tile_h + y);
        if (pix >= npix) pix = npix - 1;
        return cone_ray(P.R, pix);
    };
    return rect_cone(at, x0, y0, x1, y1, lane);
}
// A row of the cone table: (axis, tag), (cos, sin, tag, -).  The tag (BinArgs::cone_gen) says which camera the row was made for; it sits in
// BOTH halves, so a reader that catches a row between the writer's two stores sees two different tags and takes the row for missing.
__device__ __forceinline__ void cone_row_write(float4 *table, size_t row, const Cone &k, uint32_t gen)
{
    table[2 * row] = make_float4(k.cx, k.cy, k.cz, __uint_as_float(gen));
    table[2 * row + 1] = make_float4(k.cos_t, k.sin_t, __uint_as_float(gen), 0.f);
}
__device__ __forceinline__ bool cone_row_read(const float4 *table, size_t row, uint32_t gen, Cone &k)
{
    const float4 c0 = table[2 * row], c1 = table[2 * row + 1];
    if (__float_as_uint(c0.w) != gen || __float_as_uint(c1.z) != gen) return false;
    k.cx = c0.x; k.cy = c0.y; k.cz = c0.z; k.cos_t = c1.x; k.sin_t = c1.y;
    return true;
}
// One wave per cone: per tile id its own cone (slot 0) and the cones of its cells (slots 1 .. cells per tile) -- what the
// list kernel's workgroups build (and file) themselves when they do not find them; frames of a batch get them from this one launch.
__global__ __launch_bounds__(256) void tile_cones_batch_kernel(const FrameArgs *__restrict__ frames)
{
    const FrameArgs &a = frames[blockIdx.y];
    if (!a.do_cones) return;
    const BinArgs &P = a.bin;
    const uint32_t per = 1 + a.cones_cx * a.cones_cy;
    const uint32_t k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (k >= a.cones_tiles * per) return;
    const uint32_t t = k / per, c = k % per;
    const Cone cn = c ? cell_cone(P, t % P.tiles_w, t / P.tiles_w, c - 1, a.cones_cx, lane) : tile_cone(P, t % P.tiles_w, t / P.tiles_w, lane);
    if (lane == 0) cone_row_write(a.cones_out, k, cn, P.cone_gen);
}
void launch_frame_setup_batch(const FrameArgs *d_frames, const FrameArgs *h_frames, uint32_t nframes, hipStream_t st)
{
    uint32_t n_prep = 0, n_cones = 0;
    for (uint32_t f = 0; f < nframes; ++f) {
        if (h_frames[f].do_prep) n_prep = std::max(n_prep, h_frames[f].S.n);
        if (h_frames[f].do_cones) n_cones = std::max(n_cones, h_frames[f].cones_tiles * (1 + h_frames[f].cones_cx * h_frames[f].cones_cy));
    }
    if (n_prep) hipLaunchKernelGGL(prep_frame_batch_kernel, dim3((n_prep + 255) / 256, nframes), dim3(256), 0, st, d_frames);
    if (n_cones) hipLaunchKernelGGL(tile_cones_batch_kernel, dim3((n_cones + 3) / 4, nframes), dim3(256), 0, st, d_frames);
}

// ---------------------------------------------------------------------------------------------
// Chunk table (round 3): every 64 consecutive Gaussians get a bounding sphere whose radius includes the members' reach -- the distance
// beyond which cone_keeps drops them: x = d^2 / (2 sigma^2) with 0.999 x - 1e-3 > cull_x.  cone_keeps' lower bound of the distance
// between a point and the rays of a cone is 1-Lipschitz in the point, so a cone farther than the radius from the chunk's centre keeps
// none of its members: the tile level then tests N / 64 spheres and the members of the chunks that are left instead of all N Gaussians
// (256 tiles x 4096 x 32 B = 33 MB of L2 reads per `-g 64 -w 2048` frame before: the level ran at L2 bandwidth).  Index order is kept (chunks in
// order, members in order), so the lists are the same lists.  Depends on the scene and cull_eps only: built with the static tables.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void build_chunks_kernel(uint32_t n, const float4 *__restrict__ mu_sig, const float4 *__restrict__ gB, float4 *__restrict__ chunks)
{
    const uint32_t lane = threadIdx.x & 63u, chunk = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (chunk * 64u >= n) return;
    const uint32_t i = chunk * 64u + lane;
    const bool valid = i < n;
    const float4 p = valid ? mu_sig[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 bq = valid ? gB[i] : make_float4(0.f, 1.f, 0.f, -INFINITY);
    const float lo_x = wave_min(valid ? p.x : INFINITY), hi_x = wave_max(valid ? p.x : -INFINITY);
    const float lo_y = wave_min(valid ? p.y : INFINITY), hi_y = wave_max(valid ? p.y : -INFINITY);
    const float lo_z = wave_min(valid ? p.z : INFINITY), hi_z = wave_max(valid ? p.z : -INFINITY);
    const float mx = 0.5f * (lo_x + hi_x), my = 0.5f * (lo_y + hi_y), mz = 0.5f * (lo_z + hi_z);
    const float dx = p.x - mx, dy = p.y - my, dz = p.z - mz;
    const float xr = bq.w + 1e-3f; // kept iff 0.999 d^2 bq.y - 1e-3 <= cull_x
    const float reach = xr > 0.f ? sqrtf(xr / (0.999f * bq.y)) : 0.f;
    float rho = wave_max(valid ? sqrtf(dx * dx + dy * dy + dz * dz) + reach : 0.f);
    rho = rho * 1.0001f + 1e-6f * (1.f + fabsf(mx) + fabsf(my) + fabsf(mz));
    if (lane == 0) chunks[chunk] = make_float4(mx, my, mz, rho);
}
void launch_build_chunks(uint32_t n, const float4 *mu_sig, const float4 *gB, float4 *chunks, hipStream_t st)
{
    const uint32_t nch = (n + 63u) / 64u;
    if (nch) hipLaunchKernelGGL(build_chunks_kernel, dim3((nch + 3u) / 4u), dim3(256), 0, st, n, mu_sig, gB, chunks);
}
__device__ __forceinline__ bool chunk_keeps(const Cone &k, float4 ch, float ox, float oy, float oz)
{
    const float ax = ch.x - ox, ay = ch.y - oy, az = ch.z - oz;
    const float d2 = ax * ax + ay * ay + az * az;
    const float tc = ax * k.cx + ay * k.cy + az * k.cz;
    const float dperp = __builtin_amdgcn_sqrtf(fmaxf(0.f, d2 - tc * tc - 8e-6f * d2)); // the cancellation's rounding, on the keeping side
    const float dmin = fmaxf(0.f, dperp * k.cos_t - fabsf(tc) * k.sin_t);
    return !(dmin * 0.9999f > ch.w);
}

template <bool FROM_LIST, bool CHUNKS>
__device__ __forceinline__ void build_tile_lists_body(const BinArgs &P, const FuseArgs &F)
{
    static_assert(!(FROM_LIST && CHUNKS), "chunks are runs of consecutive indices: device binning only");
    __shared__ uint32_t s_wave_cnt[64];
    __shared__ uint32_t s_idx[TCAP];
    __shared__ float4 s_A[TCAP], s_B[TCAP];
    __shared__ uint32_t s_flag[MAX_FUSED_CELLS], s_inact[MAX_FUSED_CELLS];
    __shared__ uint32_t s_base[4];
    __shared__ uint32_t s_chunk[CHUNKS ? CH_CAP : 1];
    const uint32_t lt = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t t = F.tile_map ? F.tile_map[lt] : lt;
    const uint32_t tx = t % P.tiles_w, ty = t / P.tiles_w;
    // the queue counters are cleared by the PREVIOUS frame's last kernel or a memset when fusing (this kernel
    // adds to them); the unfused pipeline clears them here for the cell kernel that follows
    if (P.zero8 && !F.enabled && lt == 0 && tid < 8) P.zero8[tid] = 0;
    if (P.next_zero8 && lt == 0 && tid < 8) P.next_zero8[tid] = 0;
    unsigned long long *tl = (F.timeline && tid == 0) ? F.timeline + 8 * (size_t)lt : nullptr;
    if (tl) tl[0] = wall_clock64();
    const float org_x = P.R.origin[0], org_y = P.R.origin[1], org_z = P.R.origin[2];
    // the per-origin row of a Gaussian: centre - origin and its squared norm in vec4f_t::sqnorm order (types.h:69-72) -- prep_frame_kernel's arithmetic
    auto rel = [&](const float4 &m) {
        const float cx = m.x - org_x, cy = m.y - org_y, cz = m.z - org_z;
        return make_float4(cx, cy, cz, dot3_ref(cx, cy, cz, cx, cy, cz));
    };

    float x = 0.f, y = 0.f, ax = 0.f, ay = 0.f;
    uint32_t n_in;
    const uint32_t *in_list = nullptr;
    if constexpr (FROM_LIST) {
        n_in = P.in_count[t];
        in_list = P.in_indices + P.in_start[t];
    } else {
#pragma clang fp contract(off)
        n_in = P.n;
        x = P.xc[tx]; y = P.yc[ty];
        ax = fabsf(x) + P.tw / 2; ay = fabsf(y) + P.th / 2; // rt.cpp:58-59 (left-to-right sums)
    }
    uint32_t *out = P.out_indices + P.out_start[t];
    uint32_t total = 0;
    // Four candidates per thread and pass: their rows are requested together (one memory round trip instead of four)
    // and the order-preserving compaction needs one barrier pair per 4096 candidates.  Order = index order:
    // (sub-pass u, wave, lane) lexicographic.
    bool keep[4];
    uint32_t idx[4];
    float4 gb[4], gm[4]; // gm: centre and sigma -- the per-origin row is computed from it (rel), and the reference's tile test needs it anyway
    // chunked: the candidates are the members of the chunks that passed the chunk test, one chunk per (sub-pass, wave) -- the same
    // lexicographic (sub-pass, wave, lane) order as below, which is index order again
    bool chunked = CHUNKS && P.refine && P.chunks != nullptr;
    uint32_t n_slots = 0;
    auto fetch = [&](uint32_t base) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (CHUNKS && chunked) {
                const uint32_t slot = base / 64u + u * 16u + wave;
                idx[u] = slot < n_slots ? s_chunk[slot] * 64u + lane : 0xFFFFFFFFu;
                keep[u] = idx[u] < n_in;
                if (!keep[u]) idx[u] = 0u;
            } else {
                const uint32_t k = base + u * 1024 + tid;
                keep[u] = k < n_in;
                idx[u] = keep[u] ? (FROM_LIST ? in_list[k] : k) : 0u;
            }
            if (keep[u] && (P.refine || F.enabled)) gb[u] = P.gB[idx[u]];
            if (keep[u] && (P.refine || F.enabled || !FROM_LIST)) gm[u] = P.mu_sig[idx[u]];
        }
    };
    if (!CHUNKS || !chunked) fetch(0); // in flight while the cone is set up
    // (b) tile cone from the centre and the four corner pixels of the tile (pinhole rays: the
    // farthest ray of a rectangle on the image plane from its centre ray is a corner ray)
    Cone cone = {};
    if (P.refine) {
        bool have = false;
        const size_t row = (size_t)t * (1 + P.cones_cells);
        if (P.tile_cones && P.cones_known) have = cone_row_read(P.tile_cones, row, P.cone_gen, cone); // filed by an earlier frame with this camera (or by tile_cones_kernel)
        if (!have) {
            cone = tile_cone(P, tx, ty, lane);
            if (P.tile_cones && tid == 0) cone_row_write(P.tile_cones, row, cone, P.cone_gen);
        }
    }

    if constexpr (CHUNKS) {
        if (chunked) { // ---- chunk test: one sphere per thread and pass, order-preserving compaction of the chunk ids ----
            const uint32_t nch = (n_in + 63u) / 64u;
            const float ox = P.R.origin[0], oy = P.R.origin[1], oz = P.R.origin[2];
            for (uint32_t cb = 0; cb < nch; cb += 1024) {
                const uint32_t c = cb + tid;
                const bool kc = c < nch && chunk_keeps(cone, P.chunks[c], ox, oy, oz);
                const unsigned long long m = __ballot(kc);
                if (lane == 0) s_wave_cnt[wave] = (uint32_t)__popcll(m);
                __syncthreads();
                const uint32_t v = lane < 16 ? s_wave_cnt[lane] : 0u;
                uint32_t incl = v;
#pragma unroll
                for (int off = 1; off < 16; off <<= 1) {
                    const uint32_t up = (uint32_t)__shfl_up((int)incl, off, 64);
                    if (lane >= (uint32_t)off) incl += up;
                }
                const uint32_t before = (uint32_t)__shfl((int)(incl - v), (int)wave, 64);
                if (kc) {
                    const uint32_t pos = n_slots + before + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
                    if (pos < CH_CAP) s_chunk[pos] = c;
                }
                n_slots += (uint32_t)__shfl((int)incl, 15, 64);
                __syncthreads();
            }
            if (n_slots > CH_CAP) chunked = false; // (wave-uniform) too many chunks for LDS: every Gaussian is a candidate
            fetch(0);
        }
    }
    if (tl) tl[1] = wall_clock64();
    const uint32_t n_cand = (CHUNKS && chunked) ? n_slots * 64u : n_in;
    for (uint32_t base = 0; base < n_cand; base += 4096) {
        if (base) fetch(base);
        unsigned long long mask[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            // The result is (reference tile test) AND (cone test).  The cone test goes first: it is the cheaper
            // one (no divisions) and drops >95 % of the pairs in sparse scenes.
            if (keep[u] && P.refine) keep[u] = cone_keeps(cone, rel(gm[u]), gb[u]);
            if constexpr (!FROM_LIST) {
                if (keep[u]) {
#pragma clang fp contract(off)
                    keep[u] = false;
                    const float4 g = gm[u];
                    // glm mat4*vec4: (m0*v0 + m1*v1) + (m2*v2 + m3*v3), v = (mu, 1)   (rt.cpp:37)
                    const float vx = (P.V.m[0] * g.x + P.V.m[4] * g.y) + (P.V.m[8] * g.z + P.V.m[12] * 1.f);
                    const float vy = (P.V.m[1] * g.x + P.V.m[5] * g.y) + (P.V.m[9] * g.z + P.V.m[13] * 1.f);
                    const float vz = (P.V.m[2] * g.x + P.V.m[6] * g.y) + (P.V.m[10] * g.z + P.V.m[14] * 1.f);
                    if (!(vz < 1.f)) {                       // rt.cpp:38
                        const float sig = g.w / vz;          // rt.cpp:40
                        if (!(sig < 1e-5f)) {                // rt.cpp:41
                            const float dx = fabsf(x - vx / vz), dy = fabsf(y - vy / vz);
                            const float s33 = 3.3f * sig;
                            keep[u] = (dx <= ax + s33) && (dy <= ay + s33); // rt.cpp:58-59
                        }
                    }
                }
            }
            mask[u] = __ballot(keep[u]);
            if (lane == 0) s_wave_cnt[u * 16 + wave] = (uint32_t)__popcll(mask[u]);
        }
        __syncthreads();
        // exclusive prefix over the 64 (sub-pass, wave) counts: one count per lane, wave-level scan
        const uint32_t v = s_wave_cnt[lane];
        uint32_t incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, off, 64);
            if (lane >= (uint32_t)off) incl += up;
        }
        const uint32_t excl = incl - v;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t before = (uint32_t)__shfl((int)excl, u * 16 + (int)wave, 64);
            if (keep[u]) {
                const uint32_t pos = total + before + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask[u] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask[u], 0));
                out[pos] = idx[u];
                if (F.enabled && pos < TCAP) { s_idx[pos] = idx[u]; s_A[pos] = rel(gm[u]); s_B[pos] = gb[u]; }
            }
        }
        total += (uint32_t)__shfl((int)incl, 63, 64);
        __syncthreads();
    }
    const float slack = level_slack(P.cull_ref_n, total); // cell level: the tile's work list enters
    if (tid == 0) P.out_count[t] = total;
    if (tl) tl[2] = wall_clock64();
    // the per-origin table for the render kernels of this frame (this kernel reads none of it): behind everything the frame waits for
    auto write_prep = [&]() {
        if (P.prep_gA)
            for (uint32_t i = blockIdx.x * 1024u + tid; i < P.n; i += gridDim.x * 1024u) P.prep_gA[i] = rel(P.mu_sig[i]);
    };
    if (!F.enabled) { write_prep(); return; }

    // ---------------- second level, fused ----------------
    const CellGrid &C = F.C;
    const uint32_t cpt = C.cells_x * C.cells_y;
    const uint64_t npix = (uint64_t)P.R.width * P.R.height;
    for (uint32_t ci = wave; ci < cpt; ci += 16) {
        const uint32_t cell = lt * cpt + ci;
        uint32_t ctotal = 0;
        if (total > TCAP) {
            ctotal = 0xFFFFFFFFu; // the tile's list did not fit LDS: its cells use the tile list itself
        } else if (total) {
            Cone cc = {};
            if (P.refine) {
                bool have = false;
                const size_t row = (size_t)t * (1 + cpt) + 1 + ci;
                const bool tabled = P.tile_cones && P.cones_cells == cpt;
                if (tabled && P.cones_known) have = cone_row_read(P.tile_cones, row, P.cone_gen, cc);
                if (!have) {
                    cc = cell_cone(P, tx, ty, ci, C.cells_x, lane);
                    if (tabled && lane == 0) cone_row_write(P.tile_cones, row, cc, P.cone_gen);
                }
            }
            uint32_t *cout = C.indices + (size_t)cell * C.cstride;
            for (uint32_t base = 0; base < total; base += 64) {
                const uint32_t k = base + lane;
                bool keep = k < total;
                if (keep && P.refine) {
                    float4 bq = s_B[k];
                    bq.w = slack_cull_x(bq.w, slack, P.floor_x);
                    keep = cone_keeps(cc, s_A[k], bq);
                }
                const unsigned long long mask = __ballot(keep);
                const uint32_t pos = ctotal + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
                if (keep && pos < C.cstride) cout[pos] = s_idx[k];
                ctotal += (uint32_t)__popcll(mask);
            }
            if (ctotal > C.cstride) ctotal = 0xFFFFFFFFu;
        }
        if (lane == 0) {
            C.count[cell] = ctotal;
            s_flag[ci] = (ctotal ? (ctotal > C.dense_threshold ? 3u : (ctotal <= C.light_threshold ? 5u : 1u)) : 0u) | (min(ctotal, 255u) << 8);
        }
    }
    __syncthreads();
    if (tl) tl[3] = wall_clock64();
    if (wave == 0) { // cpt <= 64: one flag per lane
        const uint32_t flag = lane < cpt ? s_flag[lane] : 2u;
        const uint32_t mine = flag & 0xFFu, packed_count = (flag >> 8) << ACTIVE_COUNT_SHIFT; // the list length rides in the queue entry
        const unsigned long long m_act = __ballot(mine == 1u), m_dense = __ballot(mine == 3u);
        const unsigned long long m_light = __ballot(mine == 5u);
        // Retained frame buffer (RenderTarget::stamp): the buffer still holds this context's previous frame, so an empty cell
        // needs its 4 KB of background only if it was lit then; a lit cell notes the frame it was lit in.
        bool clear_me = mine == 0u;
        if (F.O.stamp && lane < cpt) {
            uint32_t *stamp = F.O.stamp + (size_t)t * cpt + lane;
            if (mine == 0u) clear_me = *stamp == F.O.stamp_seq - 1u;
            else *stamp = F.O.stamp_seq;
        }
        const unsigned long long m_empty = __ballot(clear_me);
        const uint32_t na = (uint32_t)__popcll(m_act), nd = (uint32_t)__popcll(m_dense), nl = (uint32_t)__popcll(m_light);
        // empty cells are not queued (count == 0 says it): most tiles of a sparse frame then add to no counter at all
        uint32_t base_a = 0, base_d = 0, base_l = 0;
        if (lane == 0) {
            if (na) base_a = atomicAdd(C.n_active, na);
            if (nd) base_d = atomicAdd(C.n_dense, nd);
            if (nl) base_l = atomicAdd(C.n_light, nl);
            s_base[3] = (uint32_t)__popcll(m_empty);
        }
        base_a = (uint32_t)__shfl((int)base_a, 0, 64); base_d = (uint32_t)__shfl((int)base_d, 0, 64);
        base_l = (uint32_t)__shfl((int)base_l, 0, 64);
        const unsigned long long below = (1ull << lane) - 1ull;
        const uint32_t cell = lt * cpt + lane;
        if (mine == 1u) {
            const uint32_t pos = base_a + (uint32_t)__popcll(m_act & below);
            C.active[pos] = cell | packed_count;
            if (C.slot) C.slot[cell] = pos;
            if (F.O.sparse) F.O.keys[pos] = t * cpt + lane;
        } else if (mine == 5u) { // light: from the back of the active queue (raster frames only: no slot, no key)
            C.active[C.n_cells - 1u - (base_l + (uint32_t)__popcll(m_light & below))] = cell | packed_count;
        } else if (mine == 3u) {
            const uint32_t pos = base_d + (uint32_t)__popcll(m_dense & below);
            C.dense[pos] = cell;
            if (C.slot) C.slot[cell] = pos | 0x80000000u;
        } else if (clear_me) {
            s_inact[(uint32_t)__popcll(m_empty & below)] = lane;
        }
    }
    __syncthreads();

    // ---- clear the cells nothing can reach: 4 B per ray, most of the frame's HBM traffic.  All 1024 threads,
    //      16-byte stores (4 pixels per lane, 512 B per row segment) when the geometry is 4-pixel aligned ----
    if (tl) tl[4] = wall_clock64();
    if (!F.do_clear) { write_prep(); return; }
    const uint32_t zero_px = (F.O.pack_flags & VRT_ALPHA_COMPUTED) ? 0u : 0xFF000000u;
    const bool wide = F.O.image && !F.O.radiance && (P.tile_w % 4 == 0) && (P.stride % 4 == 0) &&
                      ((uintptr_t)F.O.image % 16 == 0) && (!F.O.compact || (P.tile_w * P.tile_h) % 4 == 0);
    const uint32_t n_inact = s_base[3];
    if (wide) {
        for (uint32_t it = tid; it < n_inact * (CELL * CELL / 4); it += 1024) { // one 4-pixel quad per item
            const uint32_t ci = s_inact[it / (CELL * CELL / 4)], q = it % (CELL * CELL / 4);
            const uint32_t pxt = (ci % C.cells_x) * CELL + (q % (CELL / 4)) * 4, pyt = (ci / C.cells_x) * CELL + q / (CELL / 4);
            const uint64_t pix = (uint64_t)(tx * P.tile_w + pxt) + (uint64_t)P.stride * (ty * P.tile_h + pyt);
            if (pxt < P.tile_w && pyt < P.tile_h && pix + 3 < npix) {
                const uint64_t o = F.O.compact ? ((uint64_t)lt * P.tile_h + pyt) * P.tile_w + pxt : pix;
                *reinterpret_cast<uint4 *>(F.O.image + o) = make_uint4(zero_px, zero_px, zero_px, zero_px);
            }
        }
    } else {
        for (uint32_t it = tid; it < n_inact * (CELL * CELL); it += 1024) {
            const uint32_t ci = s_inact[it / (CELL * CELL)], q = it % (CELL * CELL);
            const uint32_t pxt = (ci % C.cells_x) * CELL + q % CELL, pyt = (ci / C.cells_x) * CELL + q / CELL;
            const uint64_t pix = (uint64_t)(tx * P.tile_w + pxt) + (uint64_t)P.stride * (ty * P.tile_h + pyt);
            if (pxt < P.tile_w && pyt < P.tile_h && pix < npix) {
                const uint64_t o = F.O.compact ? ((uint64_t)lt * P.tile_h + pyt) * P.tile_w + pxt : pix;
                if (F.O.image) F.O.image[o] = zero_px;
                if (F.O.radiance) F.O.radiance[o] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
    if (tl) tl[5] = wall_clock64();
    write_prep();
}
template <bool FROM_LIST, bool CHUNKS = false>
__global__ __launch_bounds__(1024) void build_tile_lists_kernel(BinArgs P, FuseArgs F)
{
    build_tile_lists_body<FROM_LIST, CHUNKS>(P, F);
}
template <bool FROM_LIST, bool CHUNKS = false>
__global__ __launch_bounds__(1024) void build_tile_lists_batch_kernel(const FrameArgs *__restrict__ frames)
{
    const FrameArgs &a = frames[blockIdx.y];
    build_tile_lists_body<FROM_LIST, CHUNKS>(a.bin, a.fuse);
}

void launch_build_tile_lists(const BinArgs &a, const FuseArgs &f, bool from_list, uint32_t ntiles, hipStream_t st)
{
    if (!ntiles) return;
    if (from_list) hipLaunchKernelGGL(build_tile_lists_kernel<true>, dim3(ntiles), dim3(1024), 0, st, a, f);
    else if (a.chunks && a.refine) hipLaunchKernelGGL((build_tile_lists_kernel<false, true>), dim3(ntiles), dim3(1024), 0, st, a, f);
    else hipLaunchKernelGGL(build_tile_lists_kernel<false>, dim3(ntiles), dim3(1024), 0, st, a, f);
}
void launch_build_tile_lists_batch(const FrameArgs *d_frames, uint32_t nframes, bool from_list, bool chunks, uint32_t ntiles, hipStream_t st)
{
    if (!ntiles || !nframes) return;
    if (from_list) hipLaunchKernelGGL(build_tile_lists_batch_kernel<true>, dim3(ntiles, nframes), dim3(1024), 0, st, d_frames);
    else if (chunks) hipLaunchKernelGGL((build_tile_lists_batch_kernel<false, true>), dim3(ntiles, nframes), dim3(1024), 0, st, d_frames);
    else hipLaunchKernelGGL(build_tile_lists_batch_kernel<false>, dim3(ntiles, nframes), dim3(1024), 0, st, d_frames);
}

// ---------------------------------------------------------------------------------------------
// Second level: one wavefront per 32x32 pixel cell filters its tile's list with the cell's cone.
// 16 cells share a 1024-thread workgroup so that filing the non-empty cells as active or dense costs
// at most two atomics per 16 cells -- one hot counter word serialises at ~11 ns per returning atomic,
// which 4096 single-cell atomics would turn into the longest kernel.  Empty cells: count == 0, no queue.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void build_cell_lists_kernel(SceneTables S, TileLists T, CellGrid C, RayGen R,
                                                                 const uint32_t *tile_map, uint32_t n_cells, int refine,
                                                                 uint32_t *keys /* sparse shard keys, nullable */)
{
    __shared__ uint32_t s_flag[16];  // 0 = empty, 1 = active (sparse), 3 = active (dense), 2 = no such cell
    __shared__ uint32_t s_base[3];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t cell = blockIdx.x * 16 + wave;
    uint32_t total = 0;
    if (cell < n_cells) {
        const uint32_t cpt = C.cells_x * C.cells_y;
        const uint32_t lt = cell / cpt, ci = cell % cpt;
        const uint32_t t = tile_map ? tile_map[lt] : lt;
        const uint32_t n_in = T.count[t];
        if (n_in) {
            const uint32_t tx = t % T.tiles_w, ty = t / T.tiles_w;
            const uint32_t x0 = (ci % C.cells_x) * CELL, y0 = (ci / C.cells_x) * CELL;
            const uint32_t x1 = min(x0 + CELL, T.tile_w) - 1, y1 = min(y0 + CELL, T.tile_h) - 1;
            Cone cone = {};
            if (refine) {
                const uint64_t npix = (uint64_t)R.width * R.height;
                auto at = [&](uint32_t x, uint32_t y) {
                    uint64_t pix = (uint64_t)(tx * T.tile_w + x) + (uint64_t)T.stride * (ty * T.tile_h + y);
                    if (pix >= npix) pix = npix - 1;
                    return cone_ray(R, pix);
                };
                cone = rect_cone(at, x0, y0, x1, y1, lane);
            }
            const uint32_t *in_list = T.indices + T.start[t];
            uint32_t *out = C.indices + (size_t)cell * C.cstride;
            for (uint32_t base = 0; base < n_in; base += 64) {
                const uint32_t k = base + lane;
                bool keep = false;
                uint32_t idx = 0;
                if (k < n_in) {
                    idx = in_list[k];
                    if (refine) {
                        float4 bq = S.gB[idx];
                        bq.w = slack_cull_x(bq.w, level_slack(T.cull_ref_n, n_in), T.floor_x);
                        keep = cone_keeps(cone, S.gA[idx], bq);
                    } else {
                        keep = true;
                    }
                }
                const unsigned long long mask = __ballot(keep);
                const uint32_t pos = total + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
                if (keep && pos < C.cstride) out[pos] = idx;
                total += (uint32_t)__popcll(mask);
            }
        }
        if (lane == 0) C.count[cell] = total > C.cstride ? 0xFFFFFFFFu : total;
    }
    if (lane == 0) s_flag[wave] = cell < n_cells ? (total ? (total > C.dense_threshold ? 3u : 1u) : 0u) : 2u;
    const uint32_t packed_count = min(total, 255u) << ACTIVE_COUNT_SHIFT;
    __syncthreads();
    if (tid == 0) {
        uint32_t na = 0, nd = 0;
        for (int w = 0; w < 16; ++w) { na += s_flag[w] == 1u; nd += s_flag[w] == 3u; }
        s_base[0] = na ? atomicAdd(C.n_active, na) : 0u;
        s_base[2] = nd ? atomicAdd(C.n_dense, nd) : 0u;
    }
    __syncthreads();
    if (lane == 0 && cell < n_cells) {
        uint32_t before = 0;
        const uint32_t mine = s_flag[wave];
        for (uint32_t w = 0; w < wave; ++w) before += s_flag[w] == mine;
        if (mine == 1u) {
            C.active[s_base[0] + before] = cell | packed_count;
            if (C.slot) C.slot[cell] = s_base[0] + before;
            if (keys) {
                const uint32_t cpt = C.cells_x * C.cells_y, lt = cell / cpt;
                keys[s_base[0] + before] = (tile_map ? tile_map[lt] : lt) * cpt + cell % cpt;
            }
        }
        else if (mine == 3u) { C.dense[s_base[2] + before] = cell; if (C.slot) C.slot[cell] = (s_base[2] + before) | 0x80000000u; }
    }
}

void launch_build_cell_lists(const SceneTables &s, const TileLists &t, const CellGrid &c, const RayGen &r,
                             const uint32_t *tile_map, uint32_t n_cells, int refine, uint32_t *keys, hipStream_t st)
{
    if (!n_cells) return;
    hipLaunchKernelGGL(build_cell_lists_kernel, dim3((n_cells + 15) / 16), dim3(1024), 0, st, s, t, c, r, tile_map,
                       n_cells, refine, keys);
}

// scatter rank-major shard buffers [rank](stride rank_stride)[slot][tile_h][tile_w] into the raster image (rt.h:388-399)
__global__ void assemble_kernel(const uint32_t *gathered, uint32_t *image, const uint32_t *tile_of_slot, TileLists T,
                                uint32_t width, uint32_t height, uint32_t slots_per_rank, uint64_t rank_stride)
{
    const uint32_t slot = blockIdx.y;
    const uint32_t t = tile_of_slot[slot];
    if (t == 0xFFFFFFFFu) return;
    const uint32_t *src = gathered + (slot / slots_per_rank) * rank_stride + (size_t)(slot % slots_per_rank) * (T.tile_w * T.tile_h);
    const uint32_t tx = t % T.tiles_w, ty = t / T.tiles_w;
    const uint32_t per_tile = T.tile_w * T.tile_h;
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < per_tile; p += gridDim.x * blockDim.x) {
        const uint32_t lx = p % T.tile_w, ly = p / T.tile_w;
        const uint64_t pix = (uint64_t)(tx * T.tile_w + lx) + (uint64_t)T.stride * (ty * T.tile_h + ly);
        if (pix < (uint64_t)width * height) image[pix] = src[p];
    }
}

void launch_assemble(const uint32_t *gathered, uint32_t *image, const uint32_t *tile_of_slot, uint32_t slots_per_rank,
                     uint32_t world, uint64_t rank_stride, const TileLists &t, uint32_t width, uint32_t height, hipStream_t st)
{
    const uint32_t n_slots = slots_per_rank * world;
    if (!n_slots) return;
    const uint32_t per_tile = t.tile_w * t.tile_h;
    const uint32_t gx = min((per_tile + 255u) / 256u, 64u);
    hipLaunchKernelGGL(assemble_kernel, dim3(gx ? gx : 1, n_slots), dim3(256), 0, st, gathered, image, tile_of_slot, t,
                       width, height, slots_per_rank, rank_stride);
}

// Frame assembly from sparse shards (multi-GPU): one workgroup per (shard, slot).  The shards may live in another
// GPU's memory (peer access over xGMI): they are read once, 16 B per lane, and only the stored cells travel.
// `stamp` (nullable): per cell of the FRAME (key order), the sequence number of the last assembly that stored it -- see
// clear_stale_cells_kernel.
__device__ __forceinline__ void scatter_sparse_body(const uint32_t *sh, uint32_t max_cells, uint32_t *image, const TileLists &T,
                                                    uint32_t cells_x, uint32_t cells_y, uint32_t width, uint32_t height,
                                                    uint32_t *stamp, uint32_t seq)
{
    const uint32_t slot = blockIdx.x, n = sh[0], cap = sh[1];
    if (cap != max_cells || sh[2] != cells_x * cells_y) return; // not a shard of this job's geometry: touch nothing
    if (slot >= n || slot >= max_cells) return;
    const uint32_t cpt = cells_x * cells_y;
    const uint32_t key = sh[SPARSE_HDR_WORDS + slot];
    const uint32_t t = key / cpt, ci = key % cpt;
    if (t >= T.tiles_w * T.tiles_h) return;
    const uint32_t tx = t % T.tiles_w, ty = t / T.tiles_w;
    if (stamp && threadIdx.x == 0) stamp[key] = seq;
    const uint4 *src = reinterpret_cast<const uint4 *>(sh + sparse_pixel_offset(cap) + (size_t)slot * (CELL * CELL));
    const uint64_t npix = (uint64_t)width * height;
    for (uint32_t q = threadIdx.x; q < CELL * CELL / 4; q += blockDim.x) { // one 4-pixel quad per lane and pass
        const uint4 v = src[q];
        const uint32_t cx = (q % (CELL / 4)) * 4, cy = q / (CELL / 4);
        const uint32_t pxt = (ci % cells_x) * CELL + cx, pyt = (ci / cells_x) * CELL + cy;
        if (pyt >= T.tile_h) continue;
        const uint64_t pix = (uint64_t)(tx * T.tile_w + pxt) + (uint64_t)T.stride * (ty * T.tile_h + pyt);
        const uint32_t px[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k)
            if (pxt + k < T.tile_w && pix + k < npix) image[pix + k] = px[k];
    }
}

__global__ __launch_bounds__(256) void scatter_sparse_kernel(ShardPtrs shards, uint32_t max_cells, uint32_t *image, TileLists T,
                                                             uint32_t cells_x, uint32_t cells_y, uint32_t width, uint32_t height,
                                                             uint32_t *stamp, uint32_t seq)
{
    scatter_sparse_body(shards.p[blockIdx.y], max_cells, image, T, cells_x, cells_y, width, height, stamp, seq);
}
// several frames per launch (blockIdx.z): frame f's shard s starts frame_stride words behind frame f-1's
__global__ __launch_bounds__(256) void scatter_sparse_batch_kernel(ShardPtrs shards, size_t frame_stride, AssemblyFrames F,
                                                                   uint32_t max_cells, TileLists T, uint32_t cells_x, uint32_t cells_y,
                                                                   uint32_t width, uint32_t height)
{
    const uint32_t f = blockIdx.z;
    scatter_sparse_body(shards.p[blockIdx.y] + f * frame_stride, max_cells, F.image[f], T, cells_x, cells_y, width, height,
                        F.stamp[f], F.seq[f]);
}

// Retained frames: an image buffer that still holds the previous assembly needs no background fill -- only the cells that
// were stored last time and are not stored now go back to background.  One wave per cell of the frame: stamp == seq - 1
// means "stored by the previous assembly, not by this one" (the scatter kernel of this assembly ran before this kernel).
__device__ __forceinline__ void clear_stale_cells_body(const uint32_t *stamp, uint32_t seq, uint32_t n_cells, uint32_t *image,
                                                       const TileLists &T, uint32_t cells_x, uint32_t cells_y, uint32_t width,
                                                       uint32_t height, uint32_t background)
{
    // one cell per LANE to look at (a frame has thousands of cells and a handful of stale ones), the wave then clears the
    // stale ones of its 64 one after the other
    const uint32_t lane = threadIdx.x & 63, key0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
    if (!stamp || key0 >= n_cells) return;
    const uint32_t mine = key0 + lane;
    unsigned long long stale = __ballot(mine < n_cells && stamp[mine] == seq - 1);
    const uint32_t cpt = cells_x * cells_y;
    const uint64_t npix = (uint64_t)width * height;
    while (stale) {
        const uint32_t key = key0 + (uint32_t)__builtin_ctzll(stale);
        stale &= stale - 1;
        const uint32_t t = key / cpt, ci = key % cpt;
        const uint32_t tx = t % T.tiles_w, ty = t / T.tiles_w;
        for (uint32_t q = lane; q < CELL * CELL; q += 64) {
            const uint32_t pxt = (ci % cells_x) * CELL + q % CELL, pyt = (ci / cells_x) * CELL + q / CELL;
            const uint64_t pix = (uint64_t)(tx * T.tile_w + pxt) + (uint64_t)T.stride * (ty * T.tile_h + pyt);
            if (pxt < T.tile_w && pyt < T.tile_h && pix < npix) image[pix] = background;
        }
    }
}
__global__ __launch_bounds__(256) void clear_stale_cells_kernel(const uint32_t *stamp, uint32_t seq, uint32_t n_cells, uint32_t *image,
                                                                TileLists T, uint32_t cells_x, uint32_t cells_y, uint32_t width,
                                                                uint32_t height, uint32_t background)
{
    clear_stale_cells_body(stamp, seq, n_cells, image, T, cells_x, cells_y, width, height, background);
}
// blockIdx.y = frame; frames whose buffer got the full fill this time carry clear[f] = 0
__global__ __launch_bounds__(256) void clear_stale_cells_batch_kernel(AssemblyFrames F, uint32_t n_cells, TileLists T, uint32_t cells_x,
                                                                      uint32_t cells_y, uint32_t width, uint32_t height,
                                                                      uint32_t background)
{
    const uint32_t f = blockIdx.y;
    if (!F.clear[f]) return;
    clear_stale_cells_body(F.stamp[f], F.seq[f], n_cells, F.image[f], T, cells_x, cells_y, width, height, background);
}

void launch_scatter_sparse(const ShardPtrs &shards, int nshards, uint32_t max_cells, uint32_t *image, const TileLists &t,
                           uint32_t cells_x, uint32_t cells_y, uint32_t width, uint32_t height, uint32_t *stamp, uint32_t seq,
                           hipStream_t st)
{
    if (nshards <= 0 || !max_cells) return;
    hipLaunchKernelGGL(scatter_sparse_kernel, dim3(max_cells, (uint32_t)nshards), dim3(256), 0, st, shards, max_cells, image, t,
                       cells_x, cells_y, width, height, stamp, seq);
}
void launch_assemble_sparse_batch(const ShardPtrs &shards, int nshards, size_t frame_stride, const AssemblyFrames &frames, int nframes,
                                  uint32_t max_cells, uint32_t n_cells, const TileLists &t, uint32_t cells_x, uint32_t cells_y,
                                  uint32_t width, uint32_t height, uint32_t background, hipStream_t st)
{
    if (nshards <= 0 || nframes <= 0 || !max_cells) return;
    // a frame stride shorter than a whole shard (a gathered prefix) cannot hold more cells than fit in it: no workgroups for
    // slots that did not travel
    uint32_t slots = max_cells;
    const size_t hdr = sparse_pixel_offset(max_cells);
    if (frame_stride > hdr && frame_stride < hdr + (size_t)max_cells * (CELL * CELL))
        slots = (uint32_t)std::max<size_t>(1, (frame_stride - hdr) / (CELL * CELL));
    hipLaunchKernelGGL(scatter_sparse_batch_kernel, dim3(slots, (uint32_t)nshards, (uint32_t)nframes), dim3(256), 0, st, shards,
                       frame_stride, frames, max_cells, t, cells_x, cells_y, width, height);
    bool any = false;
    for (int f = 0; f < nframes; ++f) any = any || frames.clear[f];
    if (any && n_cells)
        hipLaunchKernelGGL(clear_stale_cells_batch_kernel, dim3((n_cells + 255) / 256, (uint32_t)nframes), dim3(256), 0, st, frames, n_cells, t,
                           cells_x, cells_y, width, height, background);
}
void launch_clear_stale_cells(const uint32_t *stamp, uint32_t seq, uint32_t n_cells, uint32_t *image, const TileLists &t,
                              uint32_t cells_x, uint32_t cells_y, uint32_t width, uint32_t height, uint32_t background, hipStream_t st)
{
    if (!n_cells) return;
    hipLaunchKernelGGL(clear_stale_cells_kernel, dim3((n_cells + 255) / 256), dim3(256), 0, st, stamp, seq, n_cells, image, t, cells_x,
                       cells_y, width, height, background);
}

// ---------------------------------------------------------------------------------------------
// Point queries (API parity with rt.h:32-54, rt.cpp:8-27, rt.h:146-223); not performance paths.
// ---------------------------------------------------------------------------------------------
template <int EXP, int ERF>
__global__ void transmittance_kernel(SceneTables S, float ox, float oy, float oz, float nx, float ny, float nz,
                                     const float *s_in, size_t ns, float *T_out)
{
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ns) return;
    const float s = s_in[k];
    float T = 0.f;
    for (uint32_t q = 0; q < S.n; ++q) { // rt.h:36-52, same operations in the same order, unfused (see dot3_ref)
        const float4 g = S.mu_sig[q];
        const float mag = S.gD[q].z;
        const float cx = sub_ref(g.x, ox), cy = sub_ref(g.y, oy), cz = sub_ref(g.z, oz);
        const float mu_bar = dot3_ref(cx, cy, cz, nx, ny, nz);
        const float oc_sq = dot3_ref(cx, cy, cz, cx, cy, cz);
        const float inv_2_sigma2 = 1.f / mul_ref(mul_ref(2.f, g.w), g.w);
        const float c_bar = mul_ref(mag, vexp<EXP>(-mul_ref(sub_ref(oc_sq, mul_ref(mu_bar, mu_bar)), inv_2_sigma2)));
        const float sqrt_2_sig = mul_ref(SQRT_2, g.w);
        const float mu_bar_n = mu_bar / sqrt_2_sig;
        const float s_n = s / sqrt_2_sig;
        const float term = mul_ref(mul_ref(mul_ref(g.w, c_bar), INV_SQRT_2_PI), sub_ref(verf<ERF>(-mu_bar_n), verf<ERF>(sub_ref(s_n, mu_bar_n))));
        T = add_ref(T, term);
    }
    T_out[k] = vexp<EXP>(T);
}
// The same per ray: ray k has its own origin, direction and sample point -- broadcast_transmittance (rt.h:102-127),
// lane = ray.  The arithmetic is transmittance_kernel's (exact divides; the reference's rcp14 estimates are not
// reproduced).
template <int EXP, int ERF>
__global__ void transmittance_rays_kernel(SceneTables S, const float *origins, const float *dirs, const float *s_in,
                                          size_t nrays, float *T_out)
{
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nrays) return;
    const float ox = origins[3 * k], oy = origins[3 * k + 1], oz = origins[3 * k + 2];
    const float nx = dirs[3 * k], ny = dirs[3 * k + 1], nz = dirs[3 * k + 2];
    const float s = s_in[k];
    float T = 0.f;
    for (uint32_t q = 0; q < S.n; ++q) {
        const float4 g = S.mu_sig[q];
        const float mag = S.gD[q].z;
        const float cx = sub_ref(g.x, ox), cy = sub_ref(g.y, oy), cz = sub_ref(g.z, oz);
        const float mu_bar = dot3_ref(cx, cy, cz, nx, ny, nz);
        const float oc_sq = dot3_ref(cx, cy, cz, cx, cy, cz);
        const float inv_2_sigma2 = 1.f / mul_ref(mul_ref(2.f, g.w), g.w);
        const float c_bar = mul_ref(mag, vexp<EXP>(-mul_ref(sub_ref(oc_sq, mul_ref(mu_bar, mu_bar)), inv_2_sigma2)));
        const float sqrt_2_sig = mul_ref(SQRT_2, g.w);
        const float mu_bar_n = mu_bar / sqrt_2_sig;
        const float s_n = s / sqrt_2_sig;
        const float term = mul_ref(mul_ref(mul_ref(g.w, c_bar), INV_SQRT_2_PI), sub_ref(verf<ERF>(-mu_bar_n), verf<ERF>(sub_ref(s_n, mu_bar_n))));
        T = add_ref(T, term);
    }
    T_out[k] = vexp<EXP>(T);
}
template <int EXP, int ERF>
static void launch_transmittance_rays_t(const SceneTables &s, const float *d_o, const float *d_n, const float *d_s, size_t nrays,
                                        float *d_T, hipStream_t st)
{
    hipLaunchKernelGGL((transmittance_rays_kernel<EXP, ERF>), dim3((uint32_t)((nrays + 63) / 64)), dim3(64), 0, st, s, d_o,
                       d_n, d_s, nrays, d_T);
}
void launch_transmittance_rays(const SceneTables &s, const float *d_o, const float *d_n, const float *d_s, size_t nrays,
                               float *d_T, int exp_kind, int erf_kind, hipStream_t st)
{
    if (!nrays) return;
    VRT_DISPATCH_EXP_ERF(launch_transmittance_rays_t, s, d_o, d_n, d_s, nrays, d_T, st);
}

template <int EXP, int ERF>
static void launch_transmittance_t(const SceneTables &s, const float o[3], const float n[3], const float *d_s, size_t ns,
                                   float *d_T, hipStream_t st)
{
    hipLaunchKernelGGL((transmittance_kernel<EXP, ERF>), dim3((uint32_t)((ns + 63) / 64)), dim3(64), 0, st, s, o[0],
                       o[1], o[2], n[0], n[1], n[2], d_s, ns, d_T);
}
void launch_transmittance(const SceneTables &s, const float o[3], const float n[3], const float *d_s, size_t ns,
                          float *d_T, int exp_kind, int erf_kind, hipStream_t st)
{
    if (!ns) return;
    VRT_DISPATCH_EXP_ERF(launch_transmittance_t, s, o, n, d_s, ns, d_T, st);
}

// rt.cpp:8-17: Riemann sum with step delta, fast_exp of the negated sum
__global__ void transmittance_step_kernel(SceneTables S, float ox, float oy, float oz, float nx, float ny, float nz,
                                          const float *s_in, size_t ns, float delta, float *T_out)
{
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ns) return;
    const float s = s_in[k];
    float T = 0.f;
    for (float t = 0; t <= s; t += delta)
        for (uint32_t q = 0; q < S.n; ++q) {
            const float4 g = S.mu_sig[q];
            const float dx = ox + nx * t - g.x, dy = oy + ny * t - g.y, dz = oz + nz * t - g.z;
            T += delta * (S.gD[q].z * exp_accurate(-(dx * dx + dy * dy + dz * dz) / (2 * g.w * g.w)));
        }
    T_out[k] = exp_fast(-T);
}
void launch_transmittance_step(const SceneTables &s, const float o[3], const float n[3], const float *d_s, size_t ns,
                               float delta, float *d_T, hipStream_t st)
{
    if (!ns) return;
    hipLaunchKernelGGL(transmittance_step_kernel, dim3((uint32_t)((ns + 63) / 64)), dim3(64), 0, st, s, o[0], o[1],
                       o[2], n[0], n[1], n[2], d_s, ns, delta, d_T);
}

// rt.cpp:19-27
__global__ void density_kernel(SceneTables S, const float *pts, size_t npts, float *D)
{
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= npts) return;
    const float x = pts[3 * k], y = pts[3 * k + 1], z = pts[3 * k + 2];
    float acc = 0.f;
    for (uint32_t q = 0; q < S.n; ++q) {
        const float4 g = S.mu_sig[q];
        const float dx = x - g.x, dy = y - g.y, dz = z - g.z;
        acc += S.gD[q].z * exp_accurate(-(dx * dx + dy * dy + dz * dz) / (2 * g.w * g.w));
    }
    D[k] = acc;
}
void launch_density(const SceneTables &s, const float *d_pts, size_t npts, float *d_D, hipStream_t st)
{
    if (!npts) return;
    hipLaunchKernelGGL(density_kernel, dim3((uint32_t)((npts + 63) / 64)), dim3(64), 0, st, s, d_pts, npts, d_D);
}

// arbitrary rays: lane = ray, every Gaussian of the scene, per-lane origin
template <int EXP, int ERF>
__global__ __launch_bounds__(64) void radiance_kernel(SceneTables S, const float *origins, const float *dirs,
                                                       size_t nrays, const uint32_t *iota, float4 *out)
{
    const size_t r = (size_t)blockIdx.x * 64 + threadIdx.x;
    const size_t rc = r < nrays ? r : nrays - 1;
    LaneRay ray;
    ray.ox = origins[3 * rc]; ray.oy = origins[3 * rc + 1]; ray.oz = origins[3 * rc + 2];
    ray.nx = dirs[3 * rc]; ray.ny = dirs[3 * rc + 1]; ray.nz = dirs[3 * rc + 2];
    float Lr, Lg, Lb, La;
    shade_list<EXP, ERF, 4, false>(S, iota, S.n, ray, Lr, Lg, Lb, La);
    if (r < nrays) out[r] = make_float4(Lr, Lg, Lb, La);
}
template <int EXP, int ERF>
static void launch_radiance_t(const SceneTables &s, const float *d_origins, const float *d_dirs, size_t nrays,
                              const uint32_t *iota, float4 *d_out, hipStream_t st)
{
    hipLaunchKernelGGL((radiance_kernel<EXP, ERF>), dim3((uint32_t)((nrays + 63) / 64)), dim3(64), 0, st, s, d_origins,
                       d_dirs, nrays, iota, d_out);
}
void launch_radiance(const SceneTables &s, const float *d_origins, const float *d_dirs, size_t nrays,
                     const uint32_t *iota, float4 *d_out, int exp_kind, int erf_kind, hipStream_t st)
{
    if (!nrays) return;
    VRT_DISPATCH_EXP_ERF(launch_radiance_t, s, d_origins, d_dirs, nrays, iota, d_out, st);
}

template <int K>
__global__ void eval_erf_kernel(const float *x, size_t n, float *y)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = verf<K>(x[i]);
}
template <int K>
__global__ void eval_exp_kernel(const float *x, size_t n, float *y)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = vexp<K>(x[i]);
}
void launch_eval_erf(int kind, const float *x, size_t n, float *y, hipStream_t st)
{
    if (!n) return;
    const dim3 g((uint32_t)((n + 255) / 256)), b(256);
    switch (kind) {
    case VRT_ERF_AS: hipLaunchKernelGGL(eval_erf_kernel<VRT_ERF_AS>, g, b, 0, st, x, n, y); break;
    case VRT_ERF_SPLINE: hipLaunchKernelGGL(eval_erf_kernel<VRT_ERF_SPLINE>, g, b, 0, st, x, n, y); break;
    case VRT_ERF_SPLINE_MIRROR: hipLaunchKernelGGL(eval_erf_kernel<VRT_ERF_SPLINE_MIRROR>, g, b, 0, st, x, n, y); break;
    case VRT_ERF_TAYLOR: hipLaunchKernelGGL(eval_erf_kernel<VRT_ERF_TAYLOR>, g, b, 0, st, x, n, y); break;
    default: hipLaunchKernelGGL(eval_erf_kernel<VRT_ERF_LIBM>, g, b, 0, st, x, n, y); break;
    }
}
void launch_eval_exp(int kind, const float *x, size_t n, float *y, hipStream_t st)
{
    if (!n) return;
    const dim3 g((uint32_t)((n + 255) / 256)), b(256);
    switch (kind) {
    case VRT_EXP_VCL: hipLaunchKernelGGL(eval_exp_kernel<VRT_EXP_VCL>, g, b, 0, st, x, n, y); break;
    case VRT_EXP_FAST: hipLaunchKernelGGL(eval_exp_kernel<VRT_EXP_FAST>, g, b, 0, st, x, n, y); break;
    case VRT_EXP_SPLINE: hipLaunchKernelGGL(eval_exp_kernel<VRT_EXP_SPLINE>, g, b, 0, st, x, n, y); break;
    default: hipLaunchKernelGGL(eval_exp_kernel<VRT_EXP_LIBM>, g, b, 0, st, x, n, y); break;
    }
}

#endif // !VRT_TU_LANES
} // namespace vrtk
