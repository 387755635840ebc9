/* AVX-512 instantiation of the CPU baseline port (test/bench infrastructure; see vrt_cpu_simd.inc). */
#include <immintrin.h>
#include <stdint.h>
#include "vrt_oracle.h"
#define VRT_W 16
#include "vrt_cpu_simd.inc"
uint64_t vrt_cpu_render_tile_w16(uint32_t *image, const float *xs, const float *ys, const float *zs,
                                 const float origin[4], const ogaussian *g, uint32_t ng, uint64_t tx, uint64_t ty,
                                 uint64_t tile_width, uint64_t tile_height, uint64_t stride, uint64_t max_rows)
{
    return render_tile_w16(image, xs, ys, zs, origin, g, ng, tx, ty, tile_width, tile_height, stride, max_rows);
}
