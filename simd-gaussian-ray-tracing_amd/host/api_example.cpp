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
    return 0;
}
