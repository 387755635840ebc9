"""Host-side producers either side of the hot path (product code, numpy float32):
scene synthesis, OBJ vertices -> Gaussians, the yaw/pitch camera and its view matrix.

Mirrors (reference paths relative to /root/reference/src):
  grid_scene      volumetric-ray-tracer/main.cpp:194-205
  read_obj        vrt/gaussians-from-file.cpp:7-44
  Camera          vrt/camera.cpp:7-71 (turn / update), orbit step main.cpp:330-334
"""
import numpy as np

GAUSSIAN = np.dtype([("albedo", np.float32, 4), ("mu", np.float32, 4), ("sigma", np.float32),
                     ("magnitude", np.float32)])
f32 = np.float32


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


def _normalize(v):
    v = v.astype(f32)
    return (v * (f32(1) / np.sqrt(np.dot(v, v).astype(f32)))).astype(f32)


class Camera:
    """yaw/pitch camera of vrt/camera.cpp.  Rays leave `position` through the plane
    position + x*right + y*up - focal*front  (the closed form of camera.cpp:52,60-69)."""

    def __init__(self, position, w, h, yaw=-90.0, pitch=0.0, focal=1.0, up=(0, 1, 0)):
        self.position = np.asarray(position, f32)
        self.world_up = np.asarray(up, f32)
        self.focal = f32(focal)
        self.w, self.h = int(w), int(h)
        self.turn(yaw, pitch)

    def turn(self, yaw, pitch=0.0):
        p = min(max(float(pitch), -89.0), 89.0)
        ry, rp = np.radians(f32(yaw), dtype=f32), np.radians(f32(p), dtype=f32)
        front = np.array([np.cos(ry) * np.cos(rp), np.sin(rp), np.sin(ry) * np.cos(rp)], f32)
        self.front = _normalize(front)
        self.right = _normalize(np.cross(self.front, self.world_up).astype(f32))
        self.up = _normalize(np.cross(self.right, self.front).astype(f32))

    @property
    def view(self):
        """translate(lookAtRH(pos, pos+front, up), focal*front), column-major float[16] (glm layout)."""
        s, u, f, e = self.right, self.up, self.front, self.position
        m = np.zeros((4, 4), f32)  # m[col][row]
        m[0][0], m[1][0], m[2][0] = s
        m[0][1], m[1][1], m[2][1] = u
        m[0][2], m[1][2], m[2][2] = -f
        m[3][0] = -np.dot(s, e)
        m[3][1] = -np.dot(u, e)
        m[3][2] = np.dot(f, e)
        m[3][3] = 1
        t = (self.focal * f).astype(f32)
        m[3] = m[0] * t[0] + m[1] * t[1] + m[2] * t[2] + m[3]
        return m.ravel().copy()

    def plane(self):
        """The three w*h projection-plane arrays (camera.cpp:60-69) via the closed form."""
        x = (f32(-1) + np.arange(self.w, dtype=f32) / f32(self.w / 2.0)).astype(f32)
        y = (f32(-1) + np.arange(self.h, dtype=f32) / f32(self.h / 2.0)).astype(f32)
        base = (self.position - self.focal * self.front).astype(f32)
        pts = base[None, None, :] + x[None, :, None] * self.right[None, None, :] + y[:, None, None] * self.up[None, None, :]
        pts = pts.astype(f32)
        return (np.ascontiguousarray(pts[..., 0]).ravel(), np.ascontiguousarray(pts[..., 1]).ravel(),
                np.ascontiguousarray(pts[..., 2]).ravel())

    def orbit(self, deg):
        """main.cpp:330-334: position = rotate(I, radians(deg), +Y) * position."""
        a = np.radians(f32(deg), dtype=f32)
        c, s = np.cos(a), np.sin(a)
        x, y, z = self.position
        self.position = np.array([c * x + s * z, y, -s * x + c * z], f32)


def cli_camera(w, h, camera_offset=-4.0, focal=1.0, initial_rot=0.0):
    """Camera of volumetric-ray-tracer/main.cpp:247-255.  Returns (camera, angle)."""
    cam = Camera((0.0, 0.0, camera_offset), w, h, -90.0, 0.0, focal)
    cam.orbit(initial_rot)
    angle = f32(-90.0) - f32(initial_rot)
    cam.turn(angle, 0.0)
    return cam, angle
