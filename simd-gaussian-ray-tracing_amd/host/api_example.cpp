// api_example.cpp -- the reference's tests/transmittance.cpp and the render calls of its main.cpp, written
// against include/vrt/vrt.hpp (compile check of the vrt:: mirror; run on a GPU box to see numbers).
#include <cstdio>

#include "../../include/vrt/vrt.hpp"

using namespace vrt;

int main()
{
    // tests/transmittance.cpp:9-31
    const std::vector<gaussian_t> _gaussians = {
        gaussian_t{ { 0.f, 1.f, 0.f, .1f }, { .3f, .3f, .5f }, 0.1f, 2.f },
        gaussian_t{ { 0.f, 0.f, 1.f, .7f }, { -.3f, -.3f, 0.f }, 0.4f, .7f },
        gaussian_t{ { 1.f, 0.f, 0.f, 1.f }, { 0.f, 0.f, 2.f }, .75f, 1.f } };
    gaussians_t gaussians{ _gaussians };
    const vec4f_t origin = { 0.f, 0.f, -5.f };
    vec4f_t dir = { 0.f, 0.f, 1.f };
    std::printf("s, T, T_s, err, D\n");
    for (f32 k = -6.f; k <= 6; k += 1.f) {
        f32 s = (gaussians.gaussians[2].mu - origin).dot(dir) + k * gaussians.gaussians[2].sigma;
        f32 T = transmittance(origin, dir, s, gaussians);
        f32 T_s = transmittance_step(origin, dir, s, gaussians.gaussians[2].sigma, gaussians.gaussians);
        f32 D = density(origin + dir * s, gaussians.gaussians);
        std::printf("%g, %g, %g, %g, %g\n", s, T, T_s, std::abs(T - T_s), D);
    }
    const vec4f_t L = radiance<exp_kind::vcl, erf_kind::abramowitz_stegun>(origin, dir, gaussians);
    std::printf("radiance %g %g %g %g\n", L.x, L.y, L.z, L.w);

    // main.cpp:247-296, mode 8 and mode 5
    const u32 w = 64, h = 64;
    camera_t cam({ 0.f, 0.f, -4.f }, { 0.f, 1.f, 0.f }, { 0.f, 0.f, 1.f }, -90.f, 0.f, w, h, 1.f);
    std::vector<u32> image(w * h);
    bool running = true;
    const vec4f_t o{ cam.position[0], cam.position[1], cam.position[2] };
    tiles_t tiles = tile_gaussians(2.f / 16, 2.f / 16, _gaussians, cam.view_matrix);
    bool res = simd_render_image(w, h, image.data(), cam, o, tiles, running, 1);
    res |= render_image<exp_kind::libm, erf_kind::libm>(w, h, image.data(), cam, o, tiles, running, 1);
    res |= simd_render_image(w, h, image.data(), cam, o, gaussians, running);
    std::printf("aborted: %d, centre pixel %08x, tile 0 holds %u gaussians\n", (int)res, image[(h / 2) * w + w / 2], tiles.counts[0]);

    // rt.h:102-127, 205-223: the W-wide entry points of modes 3/4/7/8 -- lane l looks at pixel l of the middle row
    simd_vec4f_t so = simd_vec4f_t::from_vec4f_t(o), sn;
    simd::Vec<simd::Float> ss;
    for (u64 l = 0; l < SIMD_FLOATS; ++l) {
        const u64 pix = (h / 2) * w + l % w;
        vec4f_t d = vec4f_t{ cam.projection_plane.xs[pix], cam.projection_plane.ys[pix], cam.projection_plane.zs[pix] } - o;
        d.normalize();
        sn.set_lane(l, d);
        ss[l] = (_gaussians[2].mu - o).dot(d) - 0.03f * l;
    }
    const simd::Vec<simd::Float> bT = broadcast_transmittance(so, sn, ss, gaussians);
    const simd_vec4f_t bL = broadcast_radiance(so, sn, gaussians);
    for (u64 l = 0; l < SIMD_FLOATS; l += 21) {
        const f32 T1 = transmittance<exp_kind::vcl, erf_kind::abramowitz_stegun>(o, sn.lane(l), ss[l], gaussians);
        const vec4f_t L1 = simd_radiance(o, sn.lane(l), gaussians);
        std::printf("lane %llu: broadcast T %.9g single T %.9g | broadcast L %.9g %.9g %.9g %.9g single L %.9g %.9g %.9g %.9g\n",
                    (unsigned long long)l, bT[l], T1, bL.x[l], bL.y[l], bL.z[l], bL.w[l], L1.x, L1.y, L1.z, L1.w);
    }

    // tiles_t is a value like the reference's: tiled -> untiled -> another scene's tiles -> the first tiles again
    // must give the first image again (img-error.cpp:34-43 renders several variants from one tiles_t in one process)
    std::vector<u32> first(w * h), again(w * h), untiled(w * h), other(w * h);
    simd_render_image(w, h, first.data(), cam, o, tiles, running, 1);
    simd_render_image(w, h, untiled.data(), cam, o, gaussians, running);
    const std::vector<gaussian_t> one = { _gaussians[0] };
    tiles_t tiles_one = tile_gaussians(2.f / 4, 2.f / 4, one, cam.view_matrix);
    simd_render_image(w, h, other.data(), cam, o, tiles_one, running, 1);
    simd_render_image(w, h, again.data(), cam, o, tiles, running, 1);
    u64 diff_again = 0, diff_untiled = 0, diff_other = 0;
    for (u64 i = 0; i < (u64)w * h; ++i) {
        diff_again += first[i] != again[i]; diff_untiled += first[i] != untiled[i]; diff_other += first[i] != other[i];
    }
    std::printf("tiles_t reuse: %llu pixels differ from the first tiled image (untiled: %llu, other scene: %llu)\n",
                (unsigned long long)diff_again, (unsigned long long)diff_untiled, (unsigned long long)diff_other);
    return 0;
}
