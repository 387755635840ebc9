/*
 * vrt_cpu_simd.c -- CPU BASELINE PORT dispatcher (test/bench infrastructure, NOT product code).
 *
 * One task per image tile on an OpenMP team, like the reference's thread_pool_t use in
 * simd_render_image (vrt/rt.h:355-386); per-tile work in vrt_cpu_simd.inc.  Picks the
 * widest SIMD unit the host CPU has at run time (the library is built in one container
 * and executed on another machine).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "vrt_oracle.h"

typedef uint64_t (*tile_fn)(uint32_t *, const float *, const float *, const float *, const float *,
                            const ogaussian *, uint32_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t);
uint64_t vrt_cpu_render_tile_w16(uint32_t *, const float *, const float *, const float *, const float *,
                                 const ogaussian *, uint32_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t);
uint64_t vrt_cpu_render_tile_w8(uint32_t *, const float *, const float *, const float *, const float *,
                                const ogaussian *, uint32_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t);

uint64_t oracle_simd_render_tiled(uint32_t w, uint32_t h, uint32_t *image, const float *xs, const float *ys,
                                  const float *zs, const float origin[4], const ogaussian *g, size_t ng,
                                  float tw, float th, uint64_t tiles_w, uint64_t tiles_h,
                                  const uint32_t *offsets, const uint32_t *indices,
                                  const uint32_t *tile_subset, size_t ntile_subset, int threads,
                                  uint64_t max_rows, int *simd_width_out)
{
    (void)ng;
    tile_fn fn = NULL;
    int width = 1;
    __builtin_cpu_init();
    if (__builtin_cpu_supports("avx512f")) { fn = vrt_cpu_render_tile_w16; width = 16; }
    else if (__builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma")) { fn = vrt_cpu_render_tile_w8; width = 8; }
    if (simd_width_out) *simd_width_out = width;
    if (!fn) return 0; /* caller falls back to the scalar oracle */

    const uint64_t tile_width = (uint64_t)(w * tw / 2.f);   /* rt.h:348 */
    const uint64_t tile_height = (uint64_t)(h * th / 2.f);  /* rt.h:349 */
    const uint64_t stride = tile_width * tiles_w;           /* rt.h:364-365 */
    const size_t ntiles = tile_subset ? ntile_subset : (size_t)(tiles_w * tiles_h);
    uint64_t terms = 0;
    (void)threads;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : terms) num_threads(threads > 0 ? threads : 1)
    for (size_t q = 0; q < ntiles; ++q) {
        const size_t t = tile_subset ? tile_subset[q] : q;
        const uint32_t cnt = offsets[t + 1] - offsets[t];
        ogaussian *set = (ogaussian *)malloc((cnt ? cnt : 1) * sizeof(ogaussian)); /* rt.h:360: per-task AoS copy */
        for (uint32_t k = 0; k < cnt; ++k) set[k] = g[indices[offsets[t] + k]];
        terms += fn(image, xs, ys, zs, origin, set, cnt, t % tiles_w, t / tiles_w, tile_width, tile_height, stride, max_rows);
        free(set);
    }
    return terms;
}
