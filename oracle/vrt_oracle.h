/*
 * vrt_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference's volumetric Gaussian ray-tracing path
 * (Sebastian-Dawid/simd-gaussian-ray-tracing, src/vrt/).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the shipped HIP path never links or calls it.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - erf/exp approximations: PINNED against the real reference approx.cpp
 *     compiled in place (oracle/_ref) and against tests/golden/approx_ref.npz.
 *   - OBJ loader, camera, tile_gaussians, radiance / transmittance (A&S erf, VCL
 *     exp), rounding pack with computed alpha, PNG byte order: PINNED against the
 *     one image in the reference tree that the reference renderer produced,
 *     thesis/images/teapot.png (fixture tests/golden/thesis/teapot.png).  Its
 *     command line, found by search (tools/thesis_png_fit.py), is
 *     `-f test-objects/teapot.obj -w 1024 --focal-length 1.7`; this restatement
 *     and the SIMD port (vrt_cpu_simd.*) reproduce the PNG within ONE u8 step on
 *     every sampled channel value (tests/test_reference_golden.py) -- the size
 *     of the reference's own rcp-estimate noise.
 *   - NOT pinned by a reference output: the scalar-mode variants (truncating
 *     pack, opaque alpha, expf/erff: rt.h:227-310), the alternative
 *     approximations inside a render, transmittance_step / density, and the
 *     orbit step.  These are held to the reference test's own property (analytic
 *     transmittance == numeric integral, tests/transmittance.cpp:24-31) and to
 *     hand-derived known answers (tests/golden/tile_counts.json).  rt.h itself
 *     cannot be compiled here (it needs glm, which the image lacks), and
 *     thesis/images/cube.png could not be matched to a command line.
 *
 * Every function cites the reference file:line it follows
 * (paths relative to /root/reference/src).
 */
#ifndef VRT_ORACLE_H
#define VRT_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* vrt/types.h:19-22 (vec4f_t) and vrt/types.h:195-200 (gaussian_t, 40 bytes). */
typedef struct { float x, y, z, w; } ovec4;
typedef struct { ovec4 albedo; ovec4 mu; float sigma; float magnitude; } ogaussian;

/* Exp / Erf selectors: the reference passes these as template arguments
 * (vrt/rt.h:32,61,102; defaults vrt/approx.h:110-127). */
enum { ORACLE_EXP_LIBM = 0, ORACLE_EXP_VCL = 1, ORACLE_EXP_FAST = 2, ORACLE_EXP_SPLINE = 3 };
enum { ORACLE_ERF_LIBM = 0, ORACLE_ERF_AS = 1, ORACLE_ERF_SPLINE = 2, ORACLE_ERF_SPLINE_MIRROR = 3,
       ORACLE_ERF_TAYLOR = 4 };

/* Pixel packing conventions (vrt/rt.h:239-243 vs 329-333 vs 373-377). */
enum { ORACLE_PACK_TRUNC = 0,        /* (u32)(min(c,1)*255), scalar modes            */
       ORACLE_PACK_ROUND = 1 };      /* cvtps_epi32 round-to-nearest-even, SIMD modes */
enum { ORACLE_ALPHA_OPAQUE = 0,      /* A = 0xFF                                      */
       ORACLE_ALPHA_COMPUTED = 2 };  /* A = round(min(1,color.w)*255), tiled SIMD     */

/* ---- approximations (vrt/approx.cpp) ------------------------------------ */
float oracle_as_erf(float x);            /* approx.cpp:90-99   abramowitz_stegun_erf */
float oracle_spline_erf(float x);        /* approx.cpp:9-24    */
float oracle_spline_erf_mirror(float x); /* approx.cpp:45-55   */
float oracle_taylor_erf(float x);        /* approx.cpp:75-80   */
float oracle_vcl_exp(float x);           /* include/vectorclass/vectormath_exp.h:373-458 (exp_f<.,0,0>) */
float oracle_fast_exp(float x);          /* approx.cpp:119-129 (NDEBUG unset: clamped) */
float oracle_spline_exp(float x);        /* approx.cpp:141-163 */
float oracle_exp(int kind, float x);
float oracle_erf(int kind, float x);

/* ---- transmittance / radiance (vrt/rt.h, vrt/rt.cpp) --------------------- */
/* rt.h:32-54 */
float oracle_transmittance(const float o[4], const float n[4], float s,
                           const ogaussian *g, size_t ng, int exp_kind, int erf_kind);
/* rt.cpp:8-17 (uses fast_exp like the reference) */
float oracle_transmittance_step(const float o[4], const float n[4], float s, float delta,
                                const ogaussian *g, size_t ng);
/* rt.cpp:19-27 */
float oracle_density(const float pt[4], const ogaussian *g, size_t ng);
/* rt.h:146-164 with Tr = transmittance<Exp,Erf>; pdf uses `exp_kind` too
 * (the SIMD twin broadcast_radiance, rt.h:205-223, uses the same Exp for both). */
void oracle_radiance(const float o[4], const float n[4], const ogaussian *g, size_t ng,
                     int exp_kind, int erf_kind, float out[4]);

/* ---- scene producers ------------------------------------------------------ */
/* volumetric-ray-tracer/main.cpp:194-205; grid_dim is truncated to u8 there. */
size_t oracle_grid_scene(unsigned grid_dim, ogaussian *out /* grid_dim^2 */);
/* vrt/gaussians-from-file.cpp:7-44; returns count, or -1 on I/O error.
 * Call with out == NULL to get the count. */
long oracle_read_obj(const char *path, ogaussian *out, size_t cap);

/* ---- camera (vrt/camera.cpp) ---------------------------------------------- */
typedef struct {
    float position[3], front[3], up[3], world_up[3], right[3];
    float view[16];      /* column-major like glm::mat4 */
    float focal_length;
    uint64_t w, h;
} ocamera;
/* camera.cpp:25-36 constructor + turn */
void oracle_camera_init(ocamera *c, const float position[3], const float up[3], const float front[3],
                        float yaw, float pitch, uint64_t w, uint64_t h, float focal_length);
/* camera.cpp:7-23 (also recomputes the view matrix like update(), camera.cpp:52) */
void oracle_camera_turn(ocamera *c, float yaw, float pitch);
/* camera.cpp:60-69: fills w*h plane points. */
void oracle_camera_plane(const ocamera *c, float *xs, float *ys, float *zs);
/* main.cpp:330-334: rotate position about +Y by `deg` degrees (glm::rotate). */
void oracle_orbit_step(ocamera *c, float *angle, float deg);

/* ---- tiling (vrt/rt.cpp:29-69, vrt/types.h:272-287) ----------------------- */
/* Produces per-tile index lists (input order preserved).  offsets has
 * (*tiles_w * *tiles_h + 1) entries.  The caller frees *offsets and *indices
 * with oracle_free.  Returns number of tiles actually produced by the float
 * loops (may differ from tiles_w*tiles_h for odd tile counts, rt.cpp:47-49). */
size_t oracle_tile_gaussians(float tw, float th, const ogaussian *g, size_t ng, const float view[16],
                             uint64_t *tiles_w, uint64_t *tiles_h,
                             uint32_t **offsets, uint32_t **indices);
void oracle_free(void *p);

/* ---- render drivers (vrt/rt.h:227-404) ------------------------------------ */
/* Untiled: rt.h:227-247 (scalar, TRUNC|OPAQUE) and rt.h:315-337 (SIMD pixels,
 * ROUND|OPAQUE).  `pixels`/`npix`: optional sparse subset of raster indices
 * (NULL = all).  `radiance_out` (nullable) receives 4 floats per evaluated
 * pixel (dense: index = pixel; sparse: index = position in `pixels`).
 * `image` (nullable) is written at the pixel's raster index. */
void oracle_render_image(uint32_t w, uint32_t h, uint32_t *image, float *radiance_out,
                         const float *xs, const float *ys, const float *zs, const float origin[4],
                         const ogaussian *g, size_t ng, int exp_kind, int erf_kind, int pack_flags,
                         const uint32_t *pixels, size_t npix, int threads);
/* Tiled: rt.h:251-310 (scalar, TRUNC|OPAQUE) and rt.h:344-404 (SIMD pixels,
 * ROUND|COMPUTED alpha).  Tile geometry as rt.h:348-349, 364-365. */
void oracle_render_image_tiled(uint32_t w, uint32_t h, uint32_t *image, float *radiance_out,
                               const float *xs, const float *ys, const float *zs, const float origin[4],
                               const ogaussian *g, size_t ng,
                               float tw, float th, uint64_t tiles_w, uint64_t tiles_h,
                               const uint32_t *offsets, const uint32_t *indices,
                               int exp_kind, int erf_kind, int pack_flags,
                               const uint32_t *pixels, size_t npix, int threads);
/* rt.h:239-243 / 373-377 */
uint32_t oracle_pack_pixel(const float color[4], int pack_flags);

/* ---- CPU baseline: own SIMD port of mode 8 (vrt_cpu_simd.c) --------------- */
/* Port of simd_render_image tiled (rt.h:344-404 -> 205-223 -> 102-127): SIMD
 * over pixels, one task per tile on `threads` workers, doing the reference's
 * full per-(i,k,j) work (1 exp, 2 erf, 3 rcp).  `tile_subset` (nullable) limits
 * the run to the listed tile ids and `max_rows` to the first rows of each tile
 * (bounded samples for bench.py); returns the number of (ray,i,k,j) inner terms
 * executed.  simd_width_out receives 16 (AVX-512), 8 (AVX2) or 1. */
uint64_t oracle_simd_render_tiled(uint32_t w, uint32_t h, uint32_t *image,
                                  const float *xs, const float *ys, const float *zs, const float origin[4],
                                  const ogaussian *g, size_t ng,
                                  float tw, float th, uint64_t tiles_w, uint64_t tiles_h,
                                  const uint32_t *offsets, const uint32_t *indices,
                                  const uint32_t *tile_subset, size_t ntile_subset,
                                  int threads, uint64_t max_rows /* 0 = whole tiles */, int *simd_width_out);

#ifdef __cplusplus
}
#endif
#endif
