"""ctypes front-end of the CPU oracle (TEST INFRASTRUCTURE -- never imported by the product).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
See oracle/vrt_oracle.h for what each function restates (reference file:line) and for the
pinning status of the oracle.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

EXP_LIBM, EXP_VCL, EXP_FAST, EXP_SPLINE = 0, 1, 2, 3
ERF_LIBM, ERF_AS, ERF_SPLINE, ERF_SPLINE_MIRROR, ERF_TAYLOR = 0, 1, 2, 3, 4
PACK_TRUNC, PACK_ROUND, ALPHA_OPAQUE, ALPHA_COMPUTED = 0, 1, 0, 2

# numpy dtype of gaussian_t (vrt/types.h:195-200): albedo(4) mu(4) sigma magnitude = 40 bytes
GAUSSIAN = np.dtype([("albedo", np.float32, 4), ("mu", np.float32, 4), ("sigma", np.float32),
                     ("magnitude", np.float32)])
assert GAUSSIAN.itemsize == 40


class Camera(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("front", C.c_float * 3), ("up", C.c_float * 3),
                ("world_up", C.c_float * 3), ("right", C.c_float * 3), ("view", C.c_float * 16),
                ("focal_length", C.c_float), ("w", C.c_uint64), ("h", C.c_uint64)]


def build(force=False):
    """Compile liboracle.so (and oracle/_ref when /root/reference is present)."""
    so = os.path.join(_HERE, "liboracle.so")
    if force or not os.path.exists(so):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/src/vrt") and (force or not os.path.exists(os.path.join(_HERE, "_ref", "libref_approx.so"))):
        subprocess.call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
    return so


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _up(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32)) if a is not None else None


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        f32p, u32p, vp = C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.c_void_p
        for name in ("oracle_as_erf", "oracle_spline_erf", "oracle_spline_erf_mirror", "oracle_taylor_erf",
                     "oracle_vcl_exp", "oracle_fast_exp", "oracle_spline_exp"):
            getattr(L, name).restype = C.c_float
            getattr(L, name).argtypes = [C.c_float]
        L.oracle_exp.restype = C.c_float
        L.oracle_exp.argtypes = [C.c_int, C.c_float]
        L.oracle_erf.restype = C.c_float
        L.oracle_erf.argtypes = [C.c_int, C.c_float]
        L.oracle_transmittance.restype = C.c_float
        L.oracle_transmittance.argtypes = [f32p, f32p, C.c_float, vp, C.c_size_t, C.c_int, C.c_int]
        L.oracle_transmittance_step.restype = C.c_float
        L.oracle_transmittance_step.argtypes = [f32p, f32p, C.c_float, C.c_float, vp, C.c_size_t]
        L.oracle_density.restype = C.c_float
        L.oracle_density.argtypes = [f32p, vp, C.c_size_t]
        L.oracle_radiance.restype = None
        L.oracle_radiance.argtypes = [f32p, f32p, vp, C.c_size_t, C.c_int, C.c_int, f32p]
        L.oracle_grid_scene.restype = C.c_size_t
        L.oracle_grid_scene.argtypes = [C.c_uint, vp]
        L.oracle_read_obj.restype = C.c_long
        L.oracle_read_obj.argtypes = [C.c_char_p, vp, C.c_size_t]
        L.oracle_camera_init.restype = None
        L.oracle_camera_init.argtypes = [C.POINTER(Camera), f32p, f32p, f32p, C.c_float, C.c_float, C.c_uint64,
                                         C.c_uint64, C.c_float]
        L.oracle_camera_turn.restype = None
        L.oracle_camera_turn.argtypes = [C.POINTER(Camera), C.c_float, C.c_float]
        L.oracle_camera_plane.restype = None
        L.oracle_camera_plane.argtypes = [C.POINTER(Camera), f32p, f32p, f32p]
        L.oracle_orbit_step.restype = None
        L.oracle_orbit_step.argtypes = [C.POINTER(Camera), f32p, C.c_float]
        L.oracle_tile_gaussians.restype = C.c_size_t
        L.oracle_tile_gaussians.argtypes = [C.c_float, C.c_float, vp, C.c_size_t, f32p, C.POINTER(C.c_uint64),
                                            C.POINTER(C.c_uint64), C.POINTER(u32p), C.POINTER(u32p)]
        L.oracle_free.restype = None
        L.oracle_free.argtypes = [vp]
        L.oracle_render_image.restype = None
        L.oracle_render_image.argtypes = [C.c_uint32, C.c_uint32, u32p, f32p, f32p, f32p, f32p, f32p, vp,
                                          C.c_size_t, C.c_int, C.c_int, C.c_int, u32p, C.c_size_t, C.c_int]
        L.oracle_render_image_tiled.restype = None
        L.oracle_render_image_tiled.argtypes = [C.c_uint32, C.c_uint32, u32p, f32p, f32p, f32p, f32p, f32p, vp,
                                                C.c_size_t, C.c_float, C.c_float, C.c_uint64, C.c_uint64, u32p,
                                                u32p, C.c_int, C.c_int, C.c_int, u32p, C.c_size_t, C.c_int]
        L.oracle_pack_pixel.restype = C.c_uint32
        L.oracle_pack_pixel.argtypes = [f32p, C.c_int]
        L.oracle_simd_render_tiled.restype = C.c_uint64
        L.oracle_simd_render_tiled.argtypes = [C.c_uint32, C.c_uint32, u32p, f32p, f32p, f32p, f32p, vp, C.c_size_t,
                                               C.c_float, C.c_float, C.c_uint64, C.c_uint64, u32p, u32p, u32p,
                                               C.c_size_t, C.c_int, C.c_uint64, C.POINTER(C.c_int)]
        _LIB = L
    return _LIB


def ref_lib():
    """The REAL reference approx.cpp built in place (oracle/_ref); None when absent."""
    global _REF
    if _REF is None:
        so = os.path.join(_HERE, "_ref", "libref_approx.so")
        if not os.path.exists(so):
            return None
        _REF = C.CDLL(so)
    return _REF


def ref_map(name, x):
    """Apply reference function `name` (see ref_approx_driver.cpp) elementwise."""
    R = ref_lib()
    if R is None:
        raise RuntimeError("oracle/_ref not built (reference tree absent)")
    x = np.ascontiguousarray(x, dtype=np.float32)
    W = R.ref_simd_floats()
    n = ((x.size + W - 1) // W) * W
    xin = np.zeros(n, np.float32)
    xin[:x.size] = x.ravel()
    out = np.empty(n, np.float32)
    fn = getattr(R, name)
    fn.restype = None
    fn.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_size_t]
    fn(_fp(xin), _fp(out), n)
    return out[:x.size].reshape(x.shape)


# ---------------------------------------------------------------------------------------------
def vec4(v):
    a = np.zeros(4, np.float32)
    a[:len(v)] = v
    return a


def map_scalar(fn_name, x):
    f = getattr(lib(), fn_name)
    x = np.asarray(x, np.float32)
    return np.array([f(float(v)) for v in x.ravel()], np.float32).reshape(x.shape)


def gaussians(albedo, mu, sigma, magnitude):
    n = len(sigma)
    g = np.zeros(n, GAUSSIAN)
    a = np.asarray(albedo, np.float32)
    g["albedo"][:, :a.shape[1]] = a
    m = np.asarray(mu, np.float32)
    g["mu"][:, :m.shape[1]] = m
    g["sigma"] = sigma
    g["magnitude"] = magnitude
    return g


def grid_scene(dim):
    d = dim & 0xFF
    g = np.zeros(d * d, GAUSSIAN)
    n = lib().oracle_grid_scene(dim, g.ctypes.data)
    assert n == d * d
    return g


def read_obj(path):
    n = lib().oracle_read_obj(path.encode(), None, 0)
    if n < 0:
        raise IOError(path)
    g = np.zeros(n, GAUSSIAN)
    k = lib().oracle_read_obj(path.encode(), g.ctypes.data, n)
    assert k == n
    return g


def transmittance(o, n, s, g, exp_kind=EXP_LIBM, erf_kind=ERF_LIBM):
    o, n = vec4(o), vec4(n)
    s = np.atleast_1d(np.asarray(s, np.float32))
    return np.array([lib().oracle_transmittance(_fp(o), _fp(n), float(v), g.ctypes.data, len(g), exp_kind, erf_kind)
                     for v in s], np.float32)


def transmittance_step(o, n, s, delta, g):
    o, n = vec4(o), vec4(n)
    return lib().oracle_transmittance_step(_fp(o), _fp(n), float(s), float(delta), g.ctypes.data, len(g))


def density(pt, g):
    pt = vec4(pt)
    return lib().oracle_density(_fp(pt), g.ctypes.data, len(g))


def radiance(o, n, g, exp_kind=EXP_LIBM, erf_kind=ERF_AS):
    o, n = vec4(o), vec4(n)
    out = np.zeros(4, np.float32)
    lib().oracle_radiance(_fp(o), _fp(n), g.ctypes.data, len(g), exp_kind, erf_kind, _fp(out))
    return out


def camera(position, w, h, yaw=-90.0, pitch=0.0, focal=1.0, up=(0, 1, 0), front=(0, 0, 1)):
    c = Camera()
    p, u, f = (np.asarray(v, np.float32) for v in (position, up, front))
    lib().oracle_camera_init(C.byref(c), _fp(p), _fp(u), _fp(f), yaw, pitch, w, h, focal)
    return c


def cli_camera(w, h, camera_offset=-4.0, focal=1.0, initial_rot=0.0):
    """Camera set-up of volumetric-ray-tracer/main.cpp:247-255. Returns (camera, angle)."""
    c = camera((0.0, 0.0, camera_offset), w, h, -90.0, 0.0, focal)
    angle = np.array([-90.0], np.float32)
    lib().oracle_orbit_step(C.byref(c), _fp(angle), float(initial_rot))
    return c, angle


def orbit_step(c, angle, deg):
    lib().oracle_orbit_step(C.byref(c), _fp(angle), float(deg))


def camera_plane(c):
    n = c.w * c.h
    xs, ys, zs = (np.empty(n, np.float32) for _ in range(3))
    lib().oracle_camera_plane(C.byref(c), _fp(xs), _fp(ys), _fp(zs))
    return xs, ys, zs


def camera_view(c):
    return np.array(c.view, np.float32)


def tile_gaussians(tw, th, g, view):
    view = np.ascontiguousarray(view, np.float32)
    tw_, th_ = C.c_uint64(), C.c_uint64()
    off, idx = C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint32)()
    nloop = lib().oracle_tile_gaussians(tw, th, g.ctypes.data, len(g), _fp(view), C.byref(tw_), C.byref(th_),
                                        C.byref(off), C.byref(idx))
    nt = max(nloop, tw_.value * th_.value)
    offsets = np.ctypeslib.as_array(off, shape=(nt + 1,)).copy()
    indices = np.ctypeslib.as_array(idx, shape=(max(int(offsets[-1]), 1),)).copy()[:int(offsets[-1])]
    lib().oracle_free(off)
    lib().oracle_free(idx)
    return dict(tw=np.float32(tw), th=np.float32(th), w=tw_.value, h=th_.value, offsets=offsets, indices=indices,
                nloop=nloop)


def render(w, h, plane, origin, g, tiles=None, exp_kind=EXP_VCL, erf_kind=ERF_AS, pack=PACK_ROUND | ALPHA_COMPUTED,
           pixels=None, threads=None, want_image=True):
    """Reference-semantics render. Returns (image u32 [h*w] or None, radiance float32 [npix,4])."""
    xs, ys, zs = plane
    origin = vec4(origin)
    threads = threads or os.cpu_count()
    img = np.zeros(w * h, np.uint32) if want_image else None
    if pixels is not None:
        pixels = np.ascontiguousarray(pixels, np.uint32)
        npix = pixels.size
    else:
        npix = w * h
    rad = np.zeros((npix, 4), np.float32)
    if tiles is None:
        lib().oracle_render_image(w, h, _up(img), _fp(rad), _fp(xs), _fp(ys), _fp(zs), _fp(origin), g.ctypes.data,
                                  len(g), exp_kind, erf_kind, pack, _up(pixels), npix, threads)
    else:
        lib().oracle_render_image_tiled(w, h, _up(img), _fp(rad), _fp(xs), _fp(ys), _fp(zs), _fp(origin),
                                        g.ctypes.data, len(g), float(tiles["tw"]), float(tiles["th"]), tiles["w"],
                                        tiles["h"], _up(tiles["offsets"]), _up(tiles["indices"]), exp_kind, erf_kind,
                                        pack, _up(pixels), npix, threads)
    return img, rad


def simd_render_tiled(w, h, plane, origin, g, tiles, tile_subset=None, threads=None, max_rows=0):
    """CPU baseline port of mode 8. Returns (image, inner_terms, simd_width)."""
    xs, ys, zs = plane
    origin = vec4(origin)
    threads = threads or os.cpu_count()
    img = np.zeros(w * h, np.uint32)
    sw = C.c_int(0)
    sub = np.ascontiguousarray(tile_subset, np.uint32) if tile_subset is not None else None
    terms = lib().oracle_simd_render_tiled(w, h, _up(img), _fp(xs), _fp(ys), _fp(zs), _fp(origin), g.ctypes.data,
                                           len(g), float(tiles["tw"]), float(tiles["th"]), tiles["w"], tiles["h"],
                                           _up(tiles["offsets"]), _up(tiles["indices"]), _up(sub),
                                           0 if sub is None else sub.size, threads, max_rows, C.byref(sw))
    return img, terms, sw.value
