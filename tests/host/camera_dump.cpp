// Test helper: builds the CLI camera (main.cpp:247-255) through vrt::camera_t, takes `steps` orbit steps of `deg`
// degrees (main.cpp:329-334) and writes position, view matrix and the three plane arrays as raw float32 to stdout.
// tests/test_abi_and_host.py compares the bytes with the oracle's camera.
#include <cstdio>
#include <cstdlib>

#include "vrt/vrt.hpp"

int main(int argc, char **argv)
{
    if (argc < 6) return 2;
    const u64 w = strtoul(argv[1], nullptr, 10), h = strtoul(argv[2], nullptr, 10);
    const f32 rot = strtof(argv[3], nullptr), deg = strtof(argv[5], nullptr);
    const int steps = atoi(argv[4]);
    vrt::camera_t cam({ 0.f, 0.f, -4.f }, { 0.f, 1.f, 0.f }, { 0.f, 0.f, 1.f }, -90.f, 0.f, w, h, 1.f);
    f32 angle = -90.f;
    cam.orbit(rot);
    angle -= rot;
    cam.turn(angle, 0.f);
    for (int k = 0; k < steps; ++k) {
        cam.orbit(deg);
        angle -= deg;
        cam.turn(angle, 0.f);
    }
    fwrite(cam.position.data(), 4, 3, stdout);
    fwrite(cam.view_matrix.data(), 4, 16, stdout);
    fwrite(cam.projection_plane.xs.data(), 4, w * h, stdout);
    fwrite(cam.projection_plane.ys.data(), 4, w * h, stdout);
    fwrite(cam.projection_plane.zs.data(), 4, w * h, stdout);
    return 0;
}
