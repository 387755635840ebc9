// vrt_host_camera.cpp -- host half of the ray source: the reference's yaw/pitch camera (src/vrt/camera.h:20-44,
// camera.cpp:7-71) and the orbit step of its frame loop (volumetric-ray-tracer/main.cpp:252-255, 330-334), in glm's
// order of operations.  Pure host code, no HIP: callable without a GPU.
//
// Why this lives in the library and not in a header: the reference's |oc|^2 - mubar^2 amplifies a last-bit
// difference of a ray into up to 5e-4 of radiance for sigma < 0.1, so the view matrix and the
// projection-plane points have to be the reference's to the last bit -- lookAtRH -> translate -> inverse -> mat4*vec4
// exactly as glm evaluates them, every product and sum rounded on its own.  A header would inherit the caller's
// floating-point flags (-ffast-math, -march=native contraction); this translation unit is built with
// -ffp-contract=off and without fast-math (csrc/Makefile), once, for every caller: include/vrt/vrt.hpp
// (vrt::camera_t), the CLI, the ctypes binding (scene.Camera) and vrt_hip_set_camera_view below the ABI.
//
// glm is an un-vendored dependency of the reference (glm 0.9.9 / 1.0 series); the algorithms restated here are its
// published ones: func_geometric.inl (cross, normalize = v * inversesqrt(dot)), ext/matrix_transform.inl (lookAtRH,
// translate, rotate), detail/func_matrix.inl (compute_inverse<4,4>), detail/type_mat4x4.inl (mat4 * vec4).
#include <cmath>
#include <cstring>

#include "../../include/vrt_hip.h"

namespace {

inline float radians(float deg) { return deg * 0.01745329251994329576923690768489f; }
inline float dot(const float a[3], const float b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline void cross(const float x[3], const float y[3], float r[3])
{
    r[0] = x[1] * y[2] - y[1] * x[2];
    r[1] = x[2] * y[0] - y[2] * x[0];
    r[2] = x[0] * y[1] - y[0] * x[1];
}
inline void normalize(const float v[3], float r[3])
{
    const float inv = 1.f / std::sqrt(dot(v, v)); // glm::inversesqrt(x) = 1 / sqrt(x)
    r[0] = v[0] * inv; r[1] = v[1] * inv; r[2] = v[2] * inv;
}

// column-major 4x4: element (column c, row r) at [4*c + r]
void look_at_rh(const float eye[3], const float center[3], const float up[3], float m[16])
{
    const float dir[3] = { center[0] - eye[0], center[1] - eye[1], center[2] - eye[2] };
    float f[3], s[3], u[3], fxup[3];
    normalize(dir, f);
    cross(f, up, fxup);
    normalize(fxup, s);
    cross(s, f, u);
    for (int i = 0; i < 16; ++i) m[i] = 0.f;
    for (int c = 0; c < 3; ++c) { m[4 * c + 0] = s[c]; m[4 * c + 1] = u[c]; m[4 * c + 2] = -f[c]; }
    m[12] = -dot(s, eye); m[13] = -dot(u, eye); m[14] = dot(f, eye); m[15] = 1.f;
}

// glm::translate(m, v): column 3 <- m0*v.x + m1*v.y + m2*v.z + m3, left to right
void translate(float m[16], const float v[3])
{
    for (int r = 0; r < 4; ++r) m[12 + r] = m[r] * v[0] + m[4 + r] * v[1] + m[8 + r] * v[2] + m[12 + r];
}

// glm mat4 * vec4: (m0*v.x + m1*v.y) + (m2*v.z + m3*v.w)
inline void mul(const float m[16], const float v[4], float out[4])
{
    for (int r = 0; r < 4; ++r) out[r] = (m[r] * v[0] + m[4 + r] * v[1]) + (m[8 + r] * v[2] + m[12 + r] * v[3]);
}

void refresh_view(vrt_hip_camera *c)
{
    const float center[3] = { c->position[0] + c->front[0], c->position[1] + c->front[1], c->position[2] + c->front[2] };
    look_at_rh(c->position, center, c->up, c->view);
    const float t[3] = { c->focal_length * c->front[0], c->focal_length * c->front[1], c->focal_length * c->front[2] };
    translate(c->view, t); // camera.cpp:52
}

} // namespace

extern "C" {

// glm::inverse(mat4): cofactors of 2x2 sub-determinants, sign pattern, 1/det from the first row
void vrt_hip_mat4_inverse(const float a[16], float out[16])
{
    auto M = [&](int c, int r) { return a[4 * c + r]; };
    const float c00 = M(2, 2) * M(3, 3) - M(3, 2) * M(2, 3), c02 = M(1, 2) * M(3, 3) - M(3, 2) * M(1, 3),
                c03 = M(1, 2) * M(2, 3) - M(2, 2) * M(1, 3);
    const float c04 = M(2, 1) * M(3, 3) - M(3, 1) * M(2, 3), c06 = M(1, 1) * M(3, 3) - M(3, 1) * M(1, 3),
                c07 = M(1, 1) * M(2, 3) - M(2, 1) * M(1, 3);
    const float c08 = M(2, 1) * M(3, 2) - M(3, 1) * M(2, 2), c10 = M(1, 1) * M(3, 2) - M(3, 1) * M(1, 2),
                c11 = M(1, 1) * M(2, 2) - M(2, 1) * M(1, 2);
    const float c12 = M(2, 0) * M(3, 3) - M(3, 0) * M(2, 3), c14 = M(1, 0) * M(3, 3) - M(3, 0) * M(1, 3),
                c15 = M(1, 0) * M(2, 3) - M(2, 0) * M(1, 3);
    const float c16 = M(2, 0) * M(3, 2) - M(3, 0) * M(2, 2), c18 = M(1, 0) * M(3, 2) - M(3, 0) * M(1, 2),
                c19 = M(1, 0) * M(2, 2) - M(2, 0) * M(1, 2);
    const float c20 = M(2, 0) * M(3, 1) - M(3, 0) * M(2, 1), c22 = M(1, 0) * M(3, 1) - M(3, 0) * M(1, 1),
                c23 = M(1, 0) * M(2, 1) - M(2, 0) * M(1, 1);
    const float f0[4] = { c00, c00, c02, c03 }, f1[4] = { c04, c04, c06, c07 }, f2[4] = { c08, c08, c10, c11 };
    const float f3[4] = { c12, c12, c14, c15 }, f4[4] = { c16, c16, c18, c19 }, f5[4] = { c20, c20, c22, c23 };
    const float v0[4] = { M(1, 0), M(0, 0), M(0, 0), M(0, 0) }, v1[4] = { M(1, 1), M(0, 1), M(0, 1), M(0, 1) };
    const float v2[4] = { M(1, 2), M(0, 2), M(0, 2), M(0, 2) }, v3[4] = { M(1, 3), M(0, 3), M(0, 3), M(0, 3) };
    const float sa[4] = { +1, -1, +1, -1 }, sb[4] = { -1, +1, -1, +1 };
    float inv[16];
    for (int i = 0; i < 4; ++i) {
        inv[0 + i] = (v1[i] * f0[i] - v2[i] * f1[i] + v3[i] * f2[i]) * sa[i];
        inv[4 + i] = (v0[i] * f0[i] - v2[i] * f3[i] + v3[i] * f4[i]) * sb[i];
        inv[8 + i] = (v0[i] * f1[i] - v1[i] * f3[i] + v3[i] * f5[i]) * sa[i];
        inv[12 + i] = (v0[i] * f2[i] - v1[i] * f4[i] + v2[i] * f5[i]) * sb[i];
    }
    const float d0 = M(0, 0) * inv[0], d1 = M(0, 1) * inv[4], d2 = M(0, 2) * inv[8], d3 = M(0, 3) * inv[12];
    const float one_over_det = 1.f / ((d0 + d1) + (d2 + d3));
    for (int i = 0; i < 16; ++i) out[i] = inv[i] * one_over_det;
}

void vrt_hip_camera_refresh(vrt_hip_camera *c) // camera.cpp:52
{
    if (c) refresh_view(c);
}

void vrt_hip_camera_turn(vrt_hip_camera *c, float yaw, float pitch, int constrain) // camera.cpp:7-23
{
    if (!c) return;
    float p = pitch;
    if (constrain) { p = (p > 89.f) ? 89.f : p; p = (p < -89.f) ? -89.f : p; }
    const float dir[3] = { std::cos(radians(yaw)) * std::cos(radians(p)), std::sin(radians(p)),
                           std::sin(radians(yaw)) * std::cos(radians(p)) };
    float t[3];
    normalize(dir, c->front);
    cross(c->front, c->world_up, t);
    normalize(t, c->right);
    cross(c->right, c->front, t);
    normalize(t, c->up);
    refresh_view(c);
}

void vrt_hip_camera_init(vrt_hip_camera *c, const float position[3], const float up[3], const float front[3], float yaw,
                         float pitch, uint64_t w, uint64_t h, float focal_length) // camera.cpp:25-36
{
    if (!c || !position || !up || !front) return;
    std::memset(c, 0, sizeof *c);
    for (int i = 0; i < 3; ++i) { c->position[i] = position[i]; c->up[i] = c->world_up[i] = up[i]; c->front[i] = front[i]; }
    c->focal_length = focal_length;
    c->w = w; c->h = h;
    vrt_hip_camera_turn(c, yaw, pitch, 1);
}

void vrt_hip_camera_plane(const vrt_hip_camera *c, float *xs, float *ys, float *zs) // camera.cpp:60-69
{
    if (!c || !xs || !ys || !zs) return;
    float inv[16];
    vrt_hip_mat4_inverse(c->view, inv); // the reference inverts per pixel; the matrix is the same every time
    for (uint64_t i = 0; i < c->h; ++i)
        for (uint64_t j = 0; j < c->w; ++j) {
            const float v[4] = { -1.f + j / (c->w / 2.f), -1.f + i / (c->h / 2.f), 0.f, 1.f };
            float pt[4];
            mul(inv, v, pt);
            xs[i * c->w + j] = pt[0]; ys[i * c->w + j] = pt[1]; zs[i * c->w + j] = pt[2];
        }
}

// main.cpp:252, 330: position = vec3(rotate(mat4(1), radians(deg), (0,1,0)) * vec4(position, 1)).  glm::rotate builds
// the axis-angle matrix from c, s and (1 - c) * axis and multiplies it into the identity (exact), so the product below
// is the reference's.  The caller then does `angle -= deg; cam.turn(angle, 0)` (main.cpp:254-255, 332-333).
void vrt_hip_camera_orbit(vrt_hip_camera *c, float deg)
{
    if (!c) return;
    const float a = radians(deg), co = std::cos(a), si = std::sin(a);
    const float axis[3] = { 0.f, 1.f, 0.f };
    const float temp[3] = { (1.f - co) * axis[0], (1.f - co) * axis[1], (1.f - co) * axis[2] };
    float R[16];
    for (int i = 0; i < 16; ++i) R[i] = 0.f;
    R[0] = co + temp[0] * axis[0];           R[1] = temp[0] * axis[1] + si * axis[2];  R[2] = temp[0] * axis[2] - si * axis[1];
    R[4] = temp[1] * axis[0] - si * axis[2]; R[5] = co + temp[1] * axis[1];            R[6] = temp[1] * axis[2] + si * axis[0];
    R[8] = temp[2] * axis[0] + si * axis[1]; R[9] = temp[2] * axis[1] - si * axis[0];  R[10] = co + temp[2] * axis[2];
    R[15] = 1.f;
    const float p[4] = { c->position[0], c->position[1], c->position[2], 1.f };
    float q[4];
    mul(R, p, q);
    c->position[0] = q[0]; c->position[1] = q[1]; c->position[2] = q[2];
}

} // extern "C"
