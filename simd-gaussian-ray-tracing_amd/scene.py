"""Host-side producers either side of the hot path (product code, numpy float32):
scene synthesis, OBJ vertices -> Gaussians, the yaw/pitch camera and its view matrix.

Mirrors (reference paths relative to /root/reference/src):
  grid_scene      volumetric-ray-tracer/main.cpp:194-205
  read_obj        vrt/gaussians-from-file.cpp:7-44
  Camera          vrt/camera.cpp:7-71 (turn / update), orbit step main.cpp:330-334
"""
import ctypes as C

import numpy as np

GAUSSIAN = np.dtype([("albedo", np.float32, 4), ("mu", np.float32, 4), ("sigma", np.float32),
                     ("magnitude", np.float32)])
f32 = np.float32
_f32p = C.POINTER(C.c_float)


def grid_scene(grid_dim):
    """dim x dim Gaussians on the z = 1 plane; index order i-major; grid_dim truncated to u8 (main.cpp:196)."""
    d = int(grid_dim) & 0xFF
    g = np.zeros(d * d, GAUSSIAN)
    i, j = np.meshgrid(np.arange(d), np.arange(d), indexing="ij")
    i = i.ravel().astype(f32)
    j = j.ravel().astype(f32)
    q = (i * f32(d) + j) / f32(d * d)
    g["albedo"][:, 0] = f32(1) - q
    g["albedo"][:, 2] = f32(0) + q
    g["albedo"][:, 3] = 1
    inv = f32(1) / f32(d)
    half = f32(d) / f32(2)
    g["mu"][:, 0] = f32(-1) + inv + i * f32(1) / half
    g["mu"][:, 1] = f32(-1) + inv + j * f32(1) / half
    g["mu"][:, 2] = 1
    g["sigma"] = f32(1) / f32(2 * d)
    g["magnitude"] = 1
    return g


def read_obj(path):
    """Every 'v x y z' line becomes a Gaussian: sigma by vertex count (<300: .3, <1000: .15, else .05),
    albedo = normalize(v)*0.5 + (0.5, 0.5, 0.5, 1), magnitude 1."""
    verts = []
    with open(path, "r") as fh:
        for line in fh:
            if line[:2] in ("v ", "v\t"):
                p = line.split()
                verts.append((float(p[1]), float(p[2]), float(p[3])))
    v = np.asarray(verts, np.float64).astype(f32).reshape(-1, 3)
    n = len(v)
    sig = 0.3 if n < 300 else (0.15 if n < 1000 else 0.05)
    g = np.zeros(n, GAUSSIAN)
    norm = np.sqrt((v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1] + v[:, 2] * v[:, 2]).astype(f32)).astype(f32)
    c = (v / norm[:, None]).astype(f32)
    g["albedo"][:, :3] = c * f32(0.5) + f32(0.5)
    g["albedo"][:, 3] = 1.0   # (0 * 0.5 + 1.0)
    g["mu"][:, :3] = v
    g["sigma"] = sig
    g["magnitude"] = 1
    return g


class _CCamera(C.Structure):
    """vrt_hip_camera (include/vrt_hip.h)."""
    _fields_ = [("position", C.c_float * 3), ("front", C.c_float * 3), ("up", C.c_float * 3),
                ("world_up", C.c_float * 3), ("right", C.c_float * 3), ("view", C.c_float * 16),
                ("focal_length", C.c_float), ("w", C.c_uint64), ("h", C.c_uint64)]


def _L():
    from . import lib
    return lib()


class Camera:
    """yaw/pitch camera of vrt/camera.cpp: a view over the library's vrt_hip_camera (csrc/vrt_host_camera.cpp), which
    evaluates lookAt -> translate -> inverse -> mat4*vec4 in glm's order of operations, unfused -- the view matrix
    and the plane points are the reference's to the last bit (numpy's own cos/sin and sums are not: a last-bit
    difference in a ray shows up as up to 5e-4 of radiance for small sigma)."""

    def __init__(self, position, w, h, yaw=-90.0, pitch=0.0, focal=1.0, up=(0, 1, 0), front=(0, 0, 1)):
        self._c = _CCamera()
        self.w, self.h = int(w), int(h)
        p, u, f = (np.ascontiguousarray(v, f32) for v in (position, up, front))
        _L().vrt_hip_camera_init(C.byref(self._c), p.ctypes.data_as(_f32p), u.ctypes.data_as(_f32p),
                                 f.ctypes.data_as(_f32p), float(yaw), float(pitch), self.w, self.h, float(focal))

    position = property(lambda self: np.array(self._c.position[:], f32))
    front = property(lambda self: np.array(self._c.front[:], f32))
    right = property(lambda self: np.array(self._c.right[:], f32))
    up = property(lambda self: np.array(self._c.up[:], f32))
    focal = property(lambda self: f32(self._c.focal_length))

    def turn(self, yaw, pitch=0.0):
        _L().vrt_hip_camera_turn(C.byref(self._c), float(yaw), float(pitch), 1)

    @property
    def view(self):
        """translate(lookAtRH(pos, pos+front, up), focal*front), column-major float[16] (glm layout)."""
        return np.array(self._c.view[:], f32)

    def plane(self):
        """The three w*h projection-plane arrays, inverse(view) * (x, y, 0, 1) per pixel (camera.cpp:60-69)."""
        xs, ys, zs = (np.empty(self.w * self.h, f32) for _ in range(3))
        _L().vrt_hip_camera_plane(C.byref(self._c), xs.ctypes.data_as(_f32p), ys.ctypes.data_as(_f32p),
                                  zs.ctypes.data_as(_f32p))
        return xs, ys, zs

    def orbit(self, deg):
        """main.cpp:252, 330: position = rotate(I, radians(deg), +Y) * position (the caller then turns)."""
        _L().vrt_hip_camera_orbit(C.byref(self._c), float(deg))


def cli_camera(w, h, camera_offset=-4.0, focal=1.0, initial_rot=0.0):
    """Camera of volumetric-ray-tracer/main.cpp:247-255.  Returns (camera, angle)."""
    cam = Camera((0.0, 0.0, camera_offset), w, h, -90.0, 0.0, focal)
    cam.orbit(initial_rot)
    angle = f32(-90.0) - f32(initial_rot)
    cam.turn(angle, 0.0)
    return cam, angle


def orbit_step(cam, angle, deg):
    """One step of the frame loop's orbit (main.cpp:329-334).  Returns the new angle."""
    cam.orbit(deg)
    angle = f32(f32(angle) - f32(deg))
    cam.turn(angle, 0.0)
    return angle
