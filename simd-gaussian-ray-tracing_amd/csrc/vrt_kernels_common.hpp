// vrt_kernels_common.hpp -- device code shared by the three kernel translation units of libvrt_hip.so (csrc/Makefile):
//   vrt_block_kernel.hip   the one-wave "block kernel" (sparse scenes), scheduled for instruction-level parallelism
//   vrt_table_kernel.hip   the table kernel (default path of dense blocks)
//   vrt_kernels.hip        everything else: exact dense kernel, list kernels, scene tables, frame assembly, point queries
// Hand-written for gfx950 (CDNA4, wave64).  No MFMA: the path is VALU + quarter-rate transcendental bound (one v_rcp_f32 per
// Abramowitz-Stegun erf term); Gaussian parameters reach the inner loops through LDS rows or wave-uniform scalar loads.
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
//   * three shading kernels -- one wavefront per 8x8 block with per-ray candidate lists (sparse scenes); one
//     16-wave workgroup per block with a per-ray table of the transmittance exponent (dense blocks, bounded error)
//     or with depth-sorted candidates and exact saturation skipping (what the table kernel declines).
// DESIGN.md section 4 has the table of kernels and their measured costs.
#pragma once
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

// Reductions over the 64 lanes of a FULL wave (every call site is wave-uniform code), on the DPP path of the vector unit: four steps inside
// the rows of 16 lanes (quad permutes, row_half_mirror, row_mirror), two row broadcasts, one v_readlane_b32 of lane 63.  __shfl_xor is
// six ds_bpermute_b32 one after the other -- an LDS round trip each, ~1,400 cycles per reduction on the critical path of a block
// (36 of them per block of the block kernel before round 4).  min / max are order-independent: same bits as before.
template <typename Op>
__device__ __forceinline__ int wave_reduce_bits(int v, Op op)
{
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false));  // quad_perm [1, 0, 3, 2]
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false));  // quad_perm [2, 3, 0, 1]
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false)); // row_half_mirror
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false)); // row_mirror: every lane holds its row's result
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false)); // row_bcast:15 into rows 1 and 3
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x143, 0xc, 0xf, false)); // row_bcast:31 into rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
}
// The ds_bpermute_b32 forms (rounds 1-3).  Kept for the exact dense kernel's per-block set-up: with the DPP forms there -- same resource
// usage, instruction-for-instruction the same hot loops -- render_dense_kernel ran 12 % slower (teapot 2048^2 22.8 -> 25.6 ms, monkey
// 4096^2 69.0 -> 78.3; profiles/r04_experiments.md); 24 round trips per block are nothing in blocks of milliseconds.
__device__ __forceinline__ float wave_min_bpermute(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ float wave_max_bpermute(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ float wave_min(float v)
{
    return __int_as_float(wave_reduce_bits(__float_as_int(v), [](int a, int b) { return __float_as_int(fminf(__int_as_float(a), __int_as_float(b))); }));
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
    return (uint32_t)wave_reduce_bits((int)v, [](int a, int b) { return (int)max((uint32_t)a, (uint32_t)b); });
}
// inclusive prefix sum over the 64 lanes of a full wave on the DPP path: Hillis-Steele inside the rows of 16 (row_shr 1, 2, 4, 8 with zeros
// shifted in), then the row totals by two row broadcasts -- six v_add_u32 with a DPP source instead of six ds_bpermute_b32 in a row
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t x)
{
    int v = (int)x;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true); // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true); // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true); // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true); // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false); // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false); // row_bcast:31 into rows 2 and 3
    return (uint32_t)v;
}
// lane `l` (wave-uniform at run time, e.g. the wave's number) of a full wave's register
__device__ __forceinline__ uint32_t lane_value_u32(uint32_t v, uint32_t l)
{
    return (uint32_t)__builtin_amdgcn_readlane((int)v, __builtin_amdgcn_readfirstlane((int)l));
}
// lane `l` (a constant) of a full wave's register, through the scalar unit (__shfl is a ds_bpermute_b32)
__device__ __forceinline__ float lane_value(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

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
    return __int_as_float(wave_reduce_bits(__float_as_int(v), [](int a, int b) { return __float_as_int(fmaxf(__int_as_float(a), __int_as_float(b))); }));
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

// Kernel arguments through the kernarg segment pointer (round 4).  The render kernels take five argument structures (~140 dwords), the
// list kernel two (~150).  As ordinary by-value parameters the compiler loads all of them into SGPRs at entry, keeps them live through
// the kernel's loops and spills what does not fit to VGPR lanes -- the block kernel had 121 spilled SGPRs and ~370 v_readlane_b32 in its
// per-block code, each a VALU issue slot.  Read through the segment pointer they are scalar loads at the place of use (the scalar cache
// holds the few hundred bytes): 9 spilled SGPRs, 16 v_readlane_b32.  The frame-batch variants, whose arguments sit behind a pointer
// anyway, never spilled.  The by-value parameter stays in the signature: it is what puts the bytes into the segment.
struct RenderArgs { SceneTables S; TileLists T; CellGrid C; RayGen R; RenderTarget O; };
template <typename Args>
__device__ __forceinline__ const Args &kernel_args()
{
    return *(const Args *)__builtin_amdgcn_kernarg_segment_ptr(); // (a C-style cast: it leaves the constant address space)
}

// ---------------------------------------------------------------------------------------------
// Work items of the shading kernels: an 8x8 pixel block (64 rays, lane = ray) of a 32x32 pixel cell
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

} // namespace vrtk
