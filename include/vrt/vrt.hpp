// vrt.hpp -- C++ host-side mirror of the reference's vrt:: interface for the hot path, implemented
// on the C ABI of libvrt_hip.so (include/vrt_hip.h).  Same names, argument meaning and return values as
//   src/vrt/types.h   vec4f_t, simd_vec4f_t, gaussian_t, simd_gaussian_t, gaussian_vec_t, gaussians_t, tiles_t
//   src/vrt/camera.h  camera_t, camera_create_info_t
//   src/vrt/rt.h      transmittance, simd_transmittance, broadcast_transmittance, radiance, simd_radiance,
//                     broadcast_radiance, render_image (2 overloads), simd_render_image (2 overloads),
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
#include <memory>
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

// ---- types.h:115-193, 289-306: the W-wide twins.  The reference's W is the host's SIMD width (16 on AVX-512, 8 on
//      AVX2); here a lane is a ray of a wave64, so W = 64 and simd::Vec<simd::Float> is a plain array of 64 floats.
//      They exist so that callers of broadcast_transmittance / broadcast_radiance (rt.h:102-103, 205-206) compile:
//      the arithmetic on them happens on the GPU, one lane per ray.  (simd_gaussian_t::pdf has no host evaluator: the
//      density is evaluated inside the kernels, types.h:299-302 <-> emission_term in csrc/vrt_kernels.hip.) ---------
constexpr u64 SIMD_FLOATS = 64;
namespace simd {
struct Float {};
template <typename T> struct Vec;
template <> struct Vec<Float> {
    std::array<f32, SIMD_FLOATS> v;
    f32 &operator[](size_t i) { return v[i]; }
    const f32 &operator[](size_t i) const { return v[i]; }
};
template <typename T> inline Vec<T> set1(f32 x) { Vec<T> r; r.v.fill(x); return r; }
inline void store(f32 *p, const Vec<Float> &a) { std::memcpy(p, a.v.data(), sizeof a.v); }
inline Vec<Float> load(const f32 *p) { Vec<Float> r; std::memcpy(r.v.data(), p, sizeof r.v); return r; }
} // namespace simd

struct simd_vec4f_t {
    simd::Vec<simd::Float> x, y, z, w = simd::set1<simd::Float>(0.f);
    static simd_vec4f_t from_vec4f_t(const vec4f_t &o) // types.h:183-192
    {
        return { simd::set1<simd::Float>(o.x), simd::set1<simd::Float>(o.y), simd::set1<simd::Float>(o.z), simd::set1<simd::Float>(o.w) };
    }
    vec4f_t lane(size_t i) const { return { x[i], y[i], z[i], w[i] }; }
    void set_lane(size_t i, const vec4f_t &o) { x[i] = o.x; y[i] = o.y; z[i] = o.z; w[i] = o.w; }
};

struct simd_gaussian_t {
    simd_vec4f_t albedo, mu;
    simd::Vec<simd::Float> sigma, magnitude;
    static simd_gaussian_t from_gaussian_t(const gaussian_t &g) // types.cpp:113-121
    {
        return { simd_vec4f_t::from_vec4f_t(g.albedo), simd_vec4f_t::from_vec4f_t(g.mu), simd::set1<simd::Float>(g.sigma),
                 simd::set1<simd::Float>(g.magnitude) };
    }
};

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

// ---- types.h:272-287.  The reference's tiles_t OWNS a copy of every tile's Gaussians; here the per-tile sets live on
//      the device as index lists, and the object keeps what is needed to rebuild them (the scene it was made from, the
//      tile size and the view matrix): a tiles_t stays valid whatever the process renders in between -- an untiled
//      render, another scene, another tiling -- the tiled overloads re-bin when the device no longer holds THIS set. ---
struct tiles_t {
    f32 tw, th;
    u64 w, h;
    std::vector<u32> counts; // per tile, row-major: gaussians[t].gaussians.size() of the reference
    std::shared_ptr<const std::vector<gaussian_t>> scene;
    std::array<f32, 16> view;
    mutable u64 generation;  // device state (detail::device_t::generation) that holds this tile set
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

    // camera.cpp:7-23.  The arithmetic is the library's (vrt_hip_camera_*, csrc/vrt_host_camera.cpp): glm's lookAt ->
    // translate -> inverse -> mat4*vec4 in glm's order of operations, compiled once without contraction or fast-math,
    // so the view matrix and the plane points do not depend on the flags THIS header is compiled with -- a last-bit
    // difference in a ray shows up as up to 5e-4 of radiance for small sigma.
    void turn(const f32 yaw, const f32 pitch, const bool constrain = true)
    {
        vrt_hip_camera c = to_c();
        vrt_hip_camera_turn(&c, yaw, pitch, constrain ? 1 : 0);
        from_c(c);
        planes(c);
    }
    // camera.cpp:50-71: view = translate(lookAt(pos, pos + front, up), focal * front); plane = inverse(view) * (x, y, 0, 1)
    void update()
    {
        vrt_hip_camera c = to_c();
        vrt_hip_camera_refresh(&c); // front / right / up as they are (camera_t::update does not re-derive them)
        from_c(c);
        planes(c);
    }
    // main.cpp:252, 330: position = rotate(I, radians(deg), +Y) * position (follow with turn(angle - deg, 0))
    void orbit(const f32 deg)
    {
        vrt_hip_camera c = to_c();
        vrt_hip_camera_orbit(&c, deg);
        for (int i = 0; i < 3; ++i) position[i] = c.position[i];
    }

private:
    vrt_hip_camera to_c() const
    {
        vrt_hip_camera c;
        std::memset(&c, 0, sizeof c);
        for (int i = 0; i < 3; ++i) {
            c.position[i] = position[i]; c.front[i] = front[i]; c.up[i] = up[i]; c.world_up[i] = world_up[i]; c.right[i] = right[i];
        }
        for (int i = 0; i < 16; ++i) c.view[i] = view_matrix[i];
        c.focal_length = focal_length; c.w = w; c.h = h;
        return c;
    }
    void from_c(const vrt_hip_camera &c)
    {
        for (int i = 0; i < 3; ++i) { front[i] = c.front[i]; up[i] = c.up[i]; right[i] = c.right[i]; }
        for (int i = 0; i < 16; ++i) view_matrix[i] = c.view[i];
    }
    void planes(const vrt_hip_camera &c)
    {
        projection_plane.xs.resize(w * h); projection_plane.ys.resize(w * h); projection_plane.zs.resize(w * h);
        vrt_hip_camera_plane(&c, projection_plane.xs.data(), projection_plane.ys.data(), projection_plane.zs.data());
    }
};

// ---- the process-wide device context behind the free functions -------------------------------------------------
namespace detail {
struct device_t {
    vrt_hip_ctx *ctx = nullptr;
    const void *scene_key = nullptr; // identity of the last uploaded Gaussian vector
    size_t scene_n = 0;
    u64 scene_hash = 0;
    u64 generation = 0;   // bumped whenever the device's scene or tile sets change: names what the device holds
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
        ++generation; // tile sets of the old scene are gone
    }
    void upload_plane(const camera_t &cam)
    {
        check(vrt_hip_set_plane(ctx, (u32)cam.w, (u32)cam.h, cam.projection_plane.xs.data(), cam.projection_plane.ys.data(),
                                cam.projection_plane.zs.data()), "vrt_hip_set_plane");
    }
    void untiled()
    {
        check(vrt_hip_clear_tiles(ctx), "vrt_hip_clear_tiles");
        ++generation;
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
    tiles_t t{ tw, th, 0, 0, {}, std::make_shared<const std::vector<gaussian_t>>(gaussians), view, ++d.generation };
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
// make the device hold `tiles` (its scene and its tile sets) before a tiled render
inline void bind_tiles(const tiles_t &tiles)
{
    auto &d = device_t::get();
    if (tiles.generation == d.generation) return;
    if (!tiles.scene) throw hip_error("tiles_t was not made by vrt::tile_gaussians");
    d.upload_scene(*tiles.scene);
    d.check(vrt_hip_tile_gaussians_device(d.ctx, tiles.tw, tiles.th, tiles.view.data(), nullptr), "vrt_hip_tile_gaussians");
    tiles.generation = ++d.generation;
}
} // namespace detail

// ---- rt.h:227-247: scalar render, untiled: truncating pack, opaque alpha ------------------------------------------
template <exp_kind Exp = exp_kind::libm, erf_kind Erf = erf_kind::libm>
bool render_image(const u32 width, const u32 height, u32 *image, const camera_t &cam, const vec4f_t &origin,
                  const gaussians_t &gaussians, const bool &running = true)
{
    auto &d = detail::device_t::get();
    d.upload_scene(gaussians.gaussians);
    d.untiled();
    return detail::render(width, height, image, cam, origin, VRT_PACK_TRUNC | VRT_ALPHA_OPAQUE, Exp, Erf, running);
}
// ---- rt.h:251-310: scalar render, tiled (tc = thread count: the GPU grid replaces the pool) -----------------------
template <exp_kind Exp = exp_kind::libm, erf_kind Erf = erf_kind::libm>
bool render_image(const u32 width, const u32 height, u32 *image, const camera_t &cam, const vec4f_t &origin,
                  const tiles_t &tiles, const bool &running, const u64 /*tc*/)
{
    detail::bind_tiles(tiles);
    return detail::render(width, height, image, cam, origin, VRT_PACK_TRUNC | VRT_ALPHA_OPAQUE, Exp, Erf, running);
}
// ---- rt.h:315-337: SIMD-over-pixels render, untiled: rounding pack, opaque alpha ----------------------------------
template <exp_kind Exp = exp_kind::vcl, erf_kind Erf = erf_kind::abramowitz_stegun>
bool simd_render_image(const u32 width, const u32 height, u32 *image, const camera_t &cam, const vec4f_t &origin,
                       const gaussians_t &gaussians, const bool &running = true)
{
    auto &d = detail::device_t::get();
    d.upload_scene(gaussians.gaussians);
    d.untiled();
    return detail::render(width, height, image, cam, origin, VRT_PACK_ROUND | VRT_ALPHA_OPAQUE, Exp, Erf, running);
}
// ---- rt.h:344-404: SIMD-over-pixels render, tiled (the CLI default, mode 8): rounding pack, computed alpha --------
template <exp_kind Exp = exp_kind::vcl, erf_kind Erf = erf_kind::abramowitz_stegun>
bool simd_render_image(const u32 width, const u32 height, u32 *image, const camera_t &cam, const vec4f_t origin,
                       const tiles_t &tiles, const bool &running, const u64 /*tc*/)
{
    detail::bind_tiles(tiles);
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
// ---- rt.h:102-127: SIMD_FLOATS rays at once, each with its own origin, direction and sample point -------------------
template <exp_kind Exp = exp_kind::vcl, erf_kind Erf = erf_kind::abramowitz_stegun>
simd::Vec<simd::Float> broadcast_transmittance(const simd_vec4f_t &o, const simd_vec4f_t &n, const simd::Vec<simd::Float> &s,
                                               const gaussians_t &gaussians)
{
    auto &d = detail::device_t::get();
    d.upload_scene(gaussians.gaussians);
    d.options(Exp, Erf);
    f32 oo[3 * SIMD_FLOATS], nn[3 * SIMD_FLOATS];
    for (u64 l = 0; l < SIMD_FLOATS; ++l) {
        oo[3 * l] = o.x[l]; oo[3 * l + 1] = o.y[l]; oo[3 * l + 2] = o.z[l];
        nn[3 * l] = n.x[l]; nn[3 * l + 1] = n.y[l]; nn[3 * l + 2] = n.z[l];
    }
    simd::Vec<simd::Float> T;
    d.check(vrt_hip_transmittance_rays(d.ctx, SIMD_FLOATS, oo, nn, s.v.data(), T.v.data()), "vrt_hip_transmittance_rays");
    return T;
}

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
// ---- rt.h:205-223: SIMD_FLOATS rays at once (lane = ray); the kernel the tiled renderer runs per pixel vector --------
template <exp_kind Exp = exp_kind::vcl, erf_kind Erf = erf_kind::abramowitz_stegun>
simd_vec4f_t broadcast_radiance(const simd_vec4f_t o, const simd_vec4f_t n, const gaussians_t &gaussians)
{
    auto &d = detail::device_t::get();
    d.upload_scene(gaussians.gaussians);
    d.options(Exp, Erf);
    f32 oo[3 * SIMD_FLOATS], nn[3 * SIMD_FLOATS], out[4 * SIMD_FLOATS];
    for (u64 l = 0; l < SIMD_FLOATS; ++l) {
        oo[3 * l] = o.x[l]; oo[3 * l + 1] = o.y[l]; oo[3 * l + 2] = o.z[l];
        nn[3 * l] = n.x[l]; nn[3 * l + 1] = n.y[l]; nn[3 * l + 2] = n.z[l];
    }
    d.check(vrt_hip_radiance(d.ctx, SIMD_FLOATS, oo, nn, out), "vrt_hip_radiance");
    simd_vec4f_t L;
    for (u64 l = 0; l < SIMD_FLOATS; ++l) L.set_lane(l, { out[4 * l], out[4 * l + 1], out[4 * l + 2], out[4 * l + 3] });
    return L;
}

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
