// vrt.hpp -- C++ host-side mirror of the reference's vrt:: interface for the hot path, implemented
// on the C ABI of libvrt_hip.so (include/vrt_hip.h).  Same names, argument meaning and return values as
//   src/vrt/types.h   vec4f_t, gaussian_t, gaussian_vec_t, gaussians_t, tiles_t
//   src/vrt/camera.h  camera_t, camera_create_info_t
//   src/vrt/rt.h      transmittance, radiance, render_image (2 overloads), simd_render_image (2 overloads),
//                     tile_gaussians, transmittance_step, density
//   src/vrt/gaussians-from-file.h  read_from_obj
// so that the reference's callers (volumetric-ray-tracer/main.cpp:257-296, tests/transmittance.cpp,
// tests/img-error.cpp) compile against it with two mechanical changes:
//   * the Exp/Erf template arguments are tags (vrt::exp_kind / vrt::erf_kind) instead of function
//     pointers -- device code cannot take a host function pointer;
//   * glm::mat4 is replaced by the 16-float column-major array camera_t::view_matrix holds.
// There is no CPU path here: every function forwards to the GPU and throws vrt::hip_error when the
// library reports a failure (the reference _exit(1)s on its assertion failures, definitions.h:23-30).
#pragma once

#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../vrt_hip.h"

typedef uint8_t u8;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int8_t i8;
typedef int32_t i32;
typedef float f32;

namespace vrt {

struct hip_error : std::runtime_error { using std::runtime_error::runtime_error; };

// ---- template tags for the reference's Exp / Erf arguments (rt.h:32,61,102; approx.h) ------------
enum class exp_kind : int { libm = VRT_EXP_LIBM, vcl = VRT_EXP_VCL, fast = VRT_EXP_FAST, spline = VRT_EXP_SPLINE };
enum class erf_kind : int { libm = VRT_ERF_LIBM, abramowitz_stegun = VRT_ERF_AS, spline = VRT_ERF_SPLINE,
                            spline_mirror = VRT_ERF_SPLINE_MIRROR, taylor = VRT_ERF_TAYLOR };

// ---- types.h:19-113 ---------------------------------------------------------------------------------
struct vec4f_t {
    f32 x, y, z, w = 0.f;
    vec4f_t operator+(const vec4f_t &o) const { return { x + o.x, y + o.y, z + o.z, w + o.w }; }
    vec4f_t operator-(const vec4f_t &o) const { return { x - o.x, y - o.y, z - o.z, w - o.w }; }
    vec4f_t operator*(const f32 &l) const { return { x * l, y * l, z * l, w * l }; }
    vec4f_t operator/(const vec4f_t &o) const { return { x / o.x, y / o.y, z / o.z, w / o.w }; }
    f32 dot(const vec4f_t &o) const { return x * o.x + y * o.y + z * o.z + w * o.w; }
    f32 sqnorm() const { return dot(*this); }
    void normalize() { const f32 n = std::sqrt(dot(*this)); x /= n; y /= n; z /= n; w /= n; }
};

// ---- types.h:195-229 (40 bytes, the layout vrt_hip_set_gaussians_aos takes) -------------------------
struct gaussian_t {
    vec4f_t albedo;
    vec4f_t mu;
    f32 sigma;
    f32 magnitude;
};
static_assert(sizeof(gaussian_t) == 40, "gaussian_t must match the reference's AoS layout");

// ---- types.h:232-264: SoA mirror (padded to (n/W+1)*W with sigma 1, magnitude 0; W = 16) --------------
struct gaussian_vec_t {
    struct { std::vector<f32> r, g, b; } albedo;
    struct { std::vector<f32> x, y, z; } mu;
    std::vector<f32> sigma, magnitude;
    u64 size = 0;
    void load_gaussians(const std::vector<gaussian_t> &g)
    {
        for (u64 i = 0; i < size; ++i) {
            const bool pad = i >= g.size();
            mu.x[i] = pad ? 0.f : g[i].mu.x; mu.y[i] = pad ? 0.f : g[i].mu.y; mu.z[i] = pad ? 0.f : g[i].mu.z;
            albedo.r[i] = pad ? 0.f : g[i].albedo.x; albedo.g[i] = pad ? 0.f : g[i].albedo.y; albedo.b[i] = pad ? 0.f : g[i].albedo.z;
            sigma[i] = pad ? 1.f : g[i].sigma; magnitude[i] = pad ? 0.f : g[i].magnitude;
        }
    }
    static gaussian_vec_t *from_gaussians(const std::vector<gaussian_t> &g)
    {
        gaussian_vec_t *v = new gaussian_vec_t();
        v->size = (g.size() / 16 + 1) * 16;
        for (auto *a : { &v->albedo.r, &v->albedo.g, &v->albedo.b, &v->mu.x, &v->mu.y, &v->mu.z, &v->sigma, &v->magnitude }) a->resize(v->size);
        v->load_gaussians(g);
        return v;
    }
};

// ---- types.h:266-270 ------------------------------------------------------------------------------------
struct gaussians_t {
    std::vector<gaussian_t> gaussians;
    gaussian_vec_t *soa_gaussians = nullptr;
};

// ---- types.h:272-287.  The per-tile sets live on the device; `counts` mirrors gaussians[t].size() ---------
struct tiles_t {
    f32 tw, th;
    u64 w, h;
    std::vector<u32> counts; // per tile, row-major
    u64 generation;          // device-side tile set this object names (see tile_gaussians)
};

// ---- camera.h:7-44, camera.cpp:7-79 (own 3-vector maths; view_matrix is column-major like glm::mat4) ------
struct camera_create_info_t {
    std::array<f32, 3> position{ 0.f, 0.f, 0.f }, up{ 0.f, 1.f, 0.f }, front{ 0.f, 0.f, 1.f };
    f32 yaw = -90.f, pitch = 0.f;
    u64 width = 256, height = 256;
    f32 focal_length = 1.f;
};

struct camera_t {
    std::array<f32, 3> position, front, up, world_up, right;
    std::array<f32, 16> view_matrix;
    f32 focal_length;
    u64 w, h;
    struct { std::vector<f32> xs, ys, zs; } projection_plane;

    camera_t(std::array<f32, 3> position_, std::array<f32, 3> up_ = { 0.f, 1.f, 0.f }, std::array<f32, 3> front_ = { 0.f, 0.f, 1.f },
             f32 yaw = -90.f, f32 pitch = 0.f, u64 width = 256, u64 height = 256, f32 focal = 1.f)
        : position(position_), front(front_), up(up_), world_up(up_), focal_length(focal), w(width), h(height)
    {
        turn(yaw, pitch);
    }
    explicit camera_t(const camera_create_info_t &ci)
        : camera_t(ci.position, ci.up, ci.front, ci.yaw, ci.pitch, ci.width, ci.height, ci.focal_length) {}

    // camera.cpp:7-23
    void turn(const f32 yaw, const f32 pitch, const bool constrain = true)
    {
        f32 p = pitch;
        if (constrain) { p = p > 89.f ? 89.f : p; p = p < -89.f ? -89.f : p; }
        const f32 ry = radians(yaw), rp = radians(p);
        front = normalize3({ std::cos(ry) * std::cos(rp), std::sin(rp), std::sin(ry) * std::cos(rp) });
        right = normalize3(cross(front, world_up));
        up = normalize3(cross(right, front));
        update();
    }
    // camera.cpp:50-71: view = translate(lookAt(pos, pos+front, up), focal*front); plane = inverse(view)*(x,y,0,1),
    // evaluated through its closed form plane = pos + x*right + y*up - focal*front
    void update()
    {
        std::array<f32, 16> &m = view_matrix;
        m.fill(0.f);
        m[0] = right[0]; m[4] = right[1]; m[8] = right[2];
        m[1] = up[0]; m[5] = up[1]; m[9] = up[2];
        m[2] = -front[0]; m[6] = -front[1]; m[10] = -front[2];
        m[12] = -dot3(right, position); m[13] = -dot3(up, position); m[14] = dot3(front, position); m[15] = 1.f;
        const f32 t[3] = { focal_length * front[0], focal_length * front[1], focal_length * front[2] };
        for (int r = 0; r < 4; ++r) m[12 + r] = m[r] * t[0] + m[4 + r] * t[1] + m[8 + r] * t[2] + m[12 + r];
        projection_plane.xs.resize(w * h); projection_plane.ys.resize(w * h); projection_plane.zs.resize(w * h);
        for (u64 i = 0; i < h; ++i)
            for (u64 j = 0; j < w; ++j) {
                const f32 x = -1.f + j / (w / 2.f), y = -1.f + i / (h / 2.f);
                projection_plane.xs[i * w + j] = position[0] + x * right[0] + y * up[0] - focal_length * front[0];
                projection_plane.ys[i * w + j] = position[1] + x * right[1] + y * up[1] - focal_length * front[1];
                projection_plane.zs[i * w + j] = position[2] + x * right[2] + y * up[2] - focal_length * front[2];
            }
    }
    // main.cpp:330-334: position = rotate(I, radians(deg), +Y) * position
    void orbit(const f32 deg)
    {
        const f32 a = radians(deg), c = std::cos(a), s = std::sin(a);
        const f32 x = position[0], z = position[2];
        position[0] = c * x + s * z;
        position[2] = -s * x + c * z;
    }

    static f32 radians(f32 d) { return d * 0.01745329251994329576923690768489f; }
    static f32 dot3(const std::array<f32, 3> &a, const std::array<f32, 3> &b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
    static std::array<f32, 3> cross(const std::array<f32, 3> &a, const std::array<f32, 3> &b)
    {
        return { a[1] * b[2] - b[1] * a[2], a[2] * b[0] - b[2] * a[0], a[0] * b[1] - b[0] * a[1] };
    }
    static std::array<f32, 3> normalize3(const std::array<f32, 3> &a)
    {
        const f32 inv = 1.f / std::sqrt(dot3(a, a));
        return { a[0] * inv, a[1] * inv, a[2] * inv };
    }
};

// ---- the process-wide device context behind the free functions -------------------------------------------------
namespace detail {
struct device_t {
    vrt_hip_ctx *ctx = nullptr;
    const void *scene_key = nullptr; // identity of the last uploaded Gaussian vector
    size_t scene_n = 0;
    u64 scene_hash = 0;
    u64 tiles_generation = 0;
    const f32 *plane_key = nullptr;
    u64 plane_w = 0, plane_h = 0;
    f32 cull_eps = 1e-9f;

    static device_t &get()
    {
        static device_t d;
        if (!d.ctx) {
            const char *dev = std::getenv("VRT_HIP_DEVICE");
            if (vrt_hip_create(dev ? std::atoi(dev) : 0, &d.ctx) != VRT_HIP_OK)
                throw hip_error(std::string("vrt_hip_create: ") + vrt_hip_last_error(nullptr));
            if (const char *e = std::getenv("VRT_HIP_CULL_EPS")) d.cull_eps = (f32)std::atof(e);
        }
        return d;
    }
    void check(int rc, const char *what) const
    {
        if (rc != VRT_HIP_OK) throw hip_error(std::string(what) + ": " + vrt_hip_last_error(ctx));
    }
    static u64 hash(const std::vector<gaussian_t> &g)
    {
        u64 h = 1469598103934665603ull;
        const unsigned char *p = reinterpret_cast<const unsigned char *>(g.data());
        for (size_t i = 0; i < g.size() * sizeof(gaussian_t); ++i) { h ^= p[i]; h *= 1099511628211ull; }
        return h;
    }
    void upload_scene(const std::vector<gaussian_t> &g)
    {
        const u64 h = hash(g);
        if (scene_n == g.size() && scene_hash == h && scene_key) return;
        check(vrt_hip_set_gaussians_aos(ctx, g.size(), g.data()), "vrt_hip_set_gaussians_aos");
        scene_key = g.data(); scene_n = g.size(); scene_hash = h;
    }
    void upload_plane(const camera_t &cam)
    {
        check(vrt_hip_set_plane(ctx, (u32)cam.w, (u32)cam.h, cam.projection_plane.xs.data(), cam.projection_plane.ys.data(),
                                cam.projection_plane.zs.data()), "vrt_hip_set_plane");
    }
    void options(exp_kind e, erf_kind r) { check(vrt_hip_set_options(ctx, (int)e, (int)r, cull_eps), "vrt_hip_set_options"); }
};
} // namespace detail

// ---- rt.h:140 / rt.cpp:29-69.  The reference passes glm::mat4; here the camera's 16 floats.  The tile sets stay
//      on the device (index lists); the returned object carries their sizes and names the device-side set. -----
inline tiles_t tile_gaussians(const f32 tw, const f32 th, const std::vector<gaussian_t> &gaussians, const std::array<f32, 16> &view)
{
    auto &d = detail::device_t::get();
    d.upload_scene(gaussians);
    d.check(vrt_hip_tile_gaussians(d.ctx, tw, th, view.data()), "vrt_hip_tile_gaussians");
    tiles_t t{ tw, th, 0, 0, {}, ++d.tiles_generation };
    d.check(vrt_hip_get_tile_counts(d.ctx, nullptr, 0, &t.w, &t.h), "vrt_hip_get_tile_counts");
    t.counts.resize(t.w * t.h);
    d.check(vrt_hip_get_tile_counts(d.ctx, t.counts.data(), t.counts.size(), nullptr, nullptr), "vrt_hip_get_tile_counts");
    return t;
}

namespace detail {
inline bool render(u32 width, u32 height, u32 *image, const camera_t &cam, const vec4f_t &origin, int pack, exp_kind e, erf_kind r,
                   const bool &running)
{
    if (!running) return true;
    auto &d = device_t::get();
    if (cam.w != width || cam.h != height) throw hip_error("render: camera size differs from image size");
    d.upload_plane(cam);
    d.options(e, r);
    const f32 o[3] = { origin.x, origin.y, origin.z };
    d.check(vrt_hip_render(d.ctx, o, pack, image, nullptr), "vrt_hip_render");
    return !running; // the reference returns true when the viewer asked to stop (rt.h:244, 308, 334, 402)
}
} // namespace detail

// ---- rt.h:227-247: scalar render, untiled: truncating pack, opaque alpha ------------------------------------------
template <exp_kind Exp = exp_kind::libm, erf_kind Erf = erf_kind::libm>
bool render_image(const u32 width, const u32 height, u32 *image, const camera_t &cam, const vec4f_t &origin,
                  const gaussians_t &gaussians, const bool &running = true)
{
    auto &d = detail::device_t::get();
    d.upload_scene(gaussians.gaussians);
    d.check(vrt_hip_clear_tiles(d.ctx), "vrt_hip_clear_tiles");
    return detail::render(width, height, image, cam, origin, VRT_PACK_TRUNC | VRT_ALPHA_OPAQUE, Exp, Erf, running);
}
// ---- rt.h:251-310: scalar render, tiled (tc = thread count: the GPU grid replaces the pool) -----------------------
template <exp_kind Exp = exp_kind::libm, erf_kind Erf = erf_kind::libm>
bool render_image(const u32 width, const u32 height, u32 *image, const camera_t &cam, const vec4f_t &origin,
                  const tiles_t &tiles, const bool &running, const u64 /*tc*/)
{
    auto &d = detail::device_t::get();
    if (tiles.generation != d.tiles_generation) throw hip_error("render_image: stale tiles_t (call tile_gaussians again)");
    return detail::render(width, height, image, cam, origin, VRT_PACK_TRUNC | VRT_ALPHA_OPAQUE, Exp, Erf, running);
}
// ---- rt.h:315-337: SIMD-over-pixels render, untiled: rounding pack, opaque alpha ----------------------------------
template <exp_kind Exp = exp_kind::vcl, erf_kind Erf = erf_kind::abramowitz_stegun>
bool simd_render_image(const u32 width, const u32 height, u32 *image, const camera_t &cam, const vec4f_t &origin,
                       const gaussians_t &gaussians, const bool &running = true)
{
    auto &d = detail::device_t::get();
    d.upload_scene(gaussians.gaussians);
    d.check(vrt_hip_clear_tiles(d.ctx), "vrt_hip_clear_tiles");
    return detail::render(width, height, image, cam, origin, VRT_PACK_ROUND | VRT_ALPHA_OPAQUE, Exp, Erf, running);
}
// ---- rt.h:344-404: SIMD-over-pixels render, tiled (the CLI default, mode 8): rounding pack, computed alpha --------
template <exp_kind Exp = exp_kind::vcl, erf_kind Erf = erf_kind::abramowitz_stegun>
bool simd_render_image(const u32 width, const u32 height, u32 *image, const camera_t &cam, const vec4f_t origin,
                       const tiles_t &tiles, const bool &running, const u64 /*tc*/)
{
    auto &d = detail::device_t::get();
    if (tiles.generation != d.tiles_generation) throw hip_error("simd_render_image: stale tiles_t (call tile_gaussians again)");
    return detail::render(width, height, image, cam, origin, VRT_PACK_ROUND | VRT_ALPHA_COMPUTED, Exp, Erf, running);
}

// ---- rt.h:32-54 (and its SIMD twins rt.h:61-127, which compute the same function) -----------------------------------
template <exp_kind Exp = exp_kind::libm, erf_kind Erf = erf_kind::libm>
f32 transmittance(const vec4f_t o, const vec4f_t n, const f32 s, const gaussians_t &gaussians)
{
    auto &d = detail::device_t::get();
    d.upload_scene(gaussians.gaussians);
    d.options(Exp, Erf);
    const f32 oo[3] = { o.x, o.y, o.z }, nn[3] = { n.x, n.y, n.z };
    f32 T = 0.f;
    d.check(vrt_hip_transmittance(d.ctx, oo, nn, &s, 1, &T), "vrt_hip_transmittance");
    return T;
}
template <exp_kind Exp = exp_kind::vcl, erf_kind Erf = erf_kind::abramowitz_stegun>
f32 simd_transmittance(const vec4f_t o, const vec4f_t n, const f32 s, const gaussians_t &g) { return transmittance<Exp, Erf>(o, n, s, g); }

// ---- rt.h:146-164 / 166-199 / 205-223: L_hat incl. w = sum albedo.w * inner ------------------------------------------
template <exp_kind Exp = exp_kind::libm, erf_kind Erf = erf_kind::libm>
vec4f_t radiance(const vec4f_t o, const vec4f_t n, const gaussians_t &gaussians)
{
    auto &d = detail::device_t::get();
    d.upload_scene(gaussians.gaussians);
    d.options(Exp, Erf);
    const f32 oo[3] = { o.x, o.y, o.z }, nn[3] = { n.x, n.y, n.z };
    f32 out[4];
    d.check(vrt_hip_radiance(d.ctx, 1, oo, nn, out), "vrt_hip_radiance");
    return { out[0], out[1], out[2], out[3] };
}
template <exp_kind Exp = exp_kind::vcl, erf_kind Erf = erf_kind::abramowitz_stegun>
vec4f_t simd_radiance(const vec4f_t o, const vec4f_t n, const gaussians_t &g) { return radiance<Exp, Erf>(o, n, g); }

// ---- rt.cpp:8-27 -----------------------------------------------------------------------------------------------------
inline f32 transmittance_step(const vec4f_t o, const vec4f_t n, const f32 s, const f32 delta, const std::vector<gaussian_t> gaussians)
{
    auto &d = detail::device_t::get();
    d.upload_scene(gaussians);
    const f32 oo[3] = { o.x, o.y, o.z }, nn[3] = { n.x, n.y, n.z };
    f32 T = 0.f;
    d.check(vrt_hip_transmittance_step(d.ctx, oo, nn, &s, 1, delta, &T), "vrt_hip_transmittance_step");
    return T;
}
inline f32 density(const vec4f_t pt, const std::vector<gaussian_t> gaussians)
{
    auto &d = detail::device_t::get();
    d.upload_scene(gaussians);
    const f32 p[3] = { pt.x, pt.y, pt.z };
    f32 D = 0.f;
    d.check(vrt_hip_density(d.ctx, 1, p, &D), "vrt_hip_density");
    return D;
}

} // namespace vrt

// ---- gaussians-from-file.h:5 / .cpp:7-44: every "v x y z" line of an OBJ file becomes a Gaussian -----------------------
inline std::vector<vrt::gaussian_t> read_from_obj(const char *const filename)
{
    FILE *f = std::fopen(filename, "r");
    if (!f) {
        std::fprintf(stderr, "[ ERROR ]\tread_from_obj: cannot open %s\n", filename);
        std::exit(EXIT_FAILURE); // the reference exits too (gaussians-from-file.cpp:16)
    }
    std::vector<std::array<f32, 3>> verts;
    char line[1024];
    while (std::fgets(line, sizeof line, f)) {
        if (line[0] != 'v' || (line[1] != ' ' && line[1] != '\t')) continue;
        char *p = line + 1;
        std::array<f32, 3> v{};
        for (int c = 0; c < 3; ++c) v[c] = (f32)std::strtod(p, &p);
        verts.push_back(v);
    }
    std::fclose(f);
    const f32 sig = verts.size() < 300 ? 0.3f : verts.size() < 1000 ? 0.15f : 0.05f;
    std::vector<vrt::gaussian_t> g;
    g.reserve(verts.size());
    for (const auto &v : verts) {
        vrt::vec4f_t pt{ v[0], v[1], v[2], 0.0f };
        vrt::vec4f_t c = pt;
        c.normalize();
        g.push_back(vrt::gaussian_t{ c * 0.5f + vrt::vec4f_t{ 0.5f, 0.5f, 0.5f, 1.0f }, pt, sig, 1.0f });
    }
    return g;
}
