"""MI355X-native volumetric Gaussian ray tracer: Python (ctypes) face of libvrt_hip.so.

The product is the C-ABI shared library built from csrc/ (include/vrt_hip.h); this module is
only the thin binding the tests, bench.py and the smoke entry use.  It mirrors the reference's
operator set for the hot path -- render_image / simd_render_image, radiance, transmittance,
tile_gaussians (reference src/vrt/rt.h, rt.cpp) -- and fails loudly when the HIP library or a
GPU is missing: there is no CPU fallback.

The directory name contains '-', so import it through `load()` in tests/conftest.py /
__graft_entry__.py (importlib by path) under the module name `sgrt_amd`.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VRT_HIP_LIB") or os.path.join(_HERE, "lib", "libvrt_hip.so")   # override: A/B builds
INCLUDE_DIR = os.path.join(os.path.dirname(_HERE), "include")

EXP_LIBM, EXP_VCL, EXP_FAST, EXP_SPLINE = 0, 1, 2, 3
ERF_LIBM, ERF_AS, ERF_SPLINE, ERF_SPLINE_MIRROR, ERF_TAYLOR = 0, 1, 2, 3, 4
TABLE_STEP_DEFAULT = 0.05   # vrt_hip_set_table_step: the library's default (0 = exact kernels only)
PACK_TRUNC, PACK_ROUND, ALPHA_OPAQUE, ALPHA_COMPUTED = 0, 1, 0, 2

_lib = None


class VrtHipError(RuntimeError):
    pass


class Stats(C.Structure):
    _fields_ = [("kernel_ms", C.c_double), ("tiling_ms", C.c_double), ("rays", C.c_uint64), ("blocks", C.c_uint64),
                ("list_entries", C.c_uint64), ("tile_entries", C.c_uint64), ("overflow_blocks", C.c_uint64),
                ("lane_entries", C.c_uint64), ("lane_max_entries", C.c_uint64), ("shaded_blocks", C.c_uint64), ("dense_blocks", C.c_uint64),
                ("dense_busy_frac", C.c_double), ("table_blocks", C.c_uint64),
                ("lane_pairs", C.c_uint64), ("dense_visits_full", C.c_uint64), ("dense_visits_zero", C.c_uint64),
                ("dense_visits_common", C.c_uint64), ("table_nodes", C.c_uint64), ("table_retries", C.c_uint64),
                ("table_skips", C.c_uint64), ("table_declined", C.c_uint64), ("table_coarser", C.c_uint64), ("table_empty", C.c_uint64), ("table_phase_ticks", C.c_uint64 * 8), ("dense_launch_skips", C.c_uint64)]


def build(verbose=False):
    """Compile csrc/ for gfx950 with hipcc (cross-compiles without a GPU)."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-j4", "-C", os.path.join(_HERE, "csrc")], stdout=out)   # the three kernel translation units side by side
    return LIB_PATH


# every symbol include/vrt_hip.h declares: (restype, argtypes)
_f32p, _u32p, _vp = C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.c_void_p
SYMBOLS = {
    "vrt_hip_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "vrt_hip_destroy": (None, [_vp]),
    "vrt_hip_last_error": (C.c_char_p, [_vp]),
    "vrt_hip_version": (C.c_char_p, []),
    "vrt_hip_set_gaussians": (C.c_int, [_vp, C.c_size_t] + [_f32p] * 9),
    "vrt_hip_set_gaussians_aos": (C.c_int, [_vp, C.c_size_t, _vp]),
    "vrt_hip_tile_gaussians": (C.c_int, [_vp, C.c_float, C.c_float, _f32p]),
    "vrt_hip_tile_gaussians_device": (C.c_int, [_vp, C.c_float, C.c_float, _f32p, _vp]),
    "vrt_hip_set_tiles": (C.c_int, [_vp, C.c_float, C.c_float, C.c_uint64, C.c_uint64, _u32p, _u32p]),
    "vrt_hip_clear_tiles": (C.c_int, [_vp]),
    "vrt_hip_get_tile_counts": (C.c_int, [_vp, _u32p, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "vrt_hip_get_tile_indices": (C.c_int, [_vp, C.c_uint64, _u32p, C.c_size_t, _u32p]),
    "vrt_hip_set_plane": (C.c_int, [_vp, C.c_uint32, C.c_uint32, _f32p, _f32p, _f32p]),
    "vrt_hip_set_camera": (C.c_int, [_vp, C.c_uint32, C.c_uint32, _f32p, _f32p, _f32p, _f32p, C.c_float]),
    "vrt_hip_set_options": (C.c_int, [_vp, C.c_int, C.c_int, C.c_float]),
    "vrt_hip_render": (C.c_int, [_vp, _f32p, C.c_int, _u32p, _f32p]),
    "vrt_hip_render_device": (C.c_int, [_vp, _f32p, C.c_int, _vp, _vp, _vp]),
    "vrt_hip_frame_device": (C.c_int, [_vp, C.c_float, C.c_float, _f32p, _f32p, C.c_int, _vp, C.c_int, _vp]),
    "vrt_hip_frame_retained_device": (C.c_int, [_vp, C.c_float, C.c_float, _f32p, _f32p, C.c_int, _vp, _vp]),
    "vrt_hip_set_shard": (C.c_int, [_vp, C.c_int, C.c_int]),
    "vrt_hip_shard_pixels": (C.c_size_t, [_vp]),
    "vrt_hip_render_shard_device": (C.c_int, [_vp, _f32p, C.c_int, _vp, _vp]),
    "vrt_hip_set_table_step": (C.c_int, [_vp, C.c_float]),
    "vrt_hip_set_table_budget": (C.c_int, [_vp, C.c_float]),
    "vrt_hip_set_cull_prune": (C.c_int, [_vp, C.c_float]),
    "vrt_hip_set_camera_view": (C.c_int, [_vp, C.c_uint32, C.c_uint32, _f32p]),
    "vrt_hip_frame": (C.c_int, [_vp, C.c_float, C.c_float, _f32p, _f32p, C.c_int, _vp, C.c_int]),
    "vrt_hip_sync": (C.c_int, [_vp]),
    "vrt_hip_assemble_shards_device": (C.c_int, [_vp, _vp, _vp, _vp]),
    "vrt_hip_assemble_shards_strided_device": (C.c_int, [_vp, _vp, C.c_size_t, _vp, _vp]),
    "vrt_hip_image_pixels": (C.c_size_t, [_vp]),
    "vrt_hip_sparse_shard_words": (C.c_size_t, [_vp]),
    "vrt_hip_frame_sparse_device": (C.c_int, [_vp, C.c_float, C.c_float, _f32p, _f32p, C.c_int, _vp, _vp]),
    "vrt_hip_frame_batch_device": (C.c_int, [_vp, C.c_int, C.c_float, C.c_float, _f32p, _f32p, C.c_int, _vp, C.c_int, _vp]),
    "vrt_hip_scatter_sparse_device": (C.c_int, [_vp, C.POINTER(_vp), C.c_int, C.c_int, _vp, _vp]),
    "vrt_hip_scatter_sparse_retained_device": (C.c_int, [_vp, C.POINTER(_vp), C.c_int, C.c_int, _vp, _vp]),
    "vrt_hip_scatter_sparse_batch_device": (C.c_int, [_vp, _vp, C.c_int, C.c_size_t, C.c_int, C.c_int, _vp, C.c_int, _vp]),
    "vrt_hip_group_create": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(_vp)]),
    "vrt_hip_group_destroy": (None, [_vp]),
    "vrt_hip_group_size": (C.c_int, [_vp]),
    "vrt_hip_group_ctx": (_vp, [_vp, C.c_int]),
    "vrt_hip_group_last_error": (C.c_char_p, [_vp]),
    "vrt_hip_group_frame": (C.c_int, [_vp, C.c_float, C.c_float, _f32p, _f32p, C.c_int, _u32p, C.c_int]),
    "vrt_hip_group_image_device": (_vp, [_vp]),
    "vrt_hip_group_sync": (C.c_int, [_vp]),
    "vrt_hip_group_frame_batch": (C.c_int, [_vp, C.c_int, C.c_float, C.c_float, _f32p, _f32p, C.c_int, C.POINTER(C.c_void_p), C.c_int]),
    "vrt_hip_group_batch_image_device": (_vp, [_vp, C.c_int]),
    "vrt_hip_copy_state": (C.c_int, [_vp, _vp]),
    "vrt_hip_state_generation": (C.c_uint64, [_vp]),
    "vrt_hip_get_image_size": (C.c_int, [_vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "vrt_hip_transmittance": (C.c_int, [_vp, _f32p, _f32p, _f32p, C.c_size_t, _f32p]),
    "vrt_hip_transmittance_rays": (C.c_int, [_vp, C.c_size_t, _f32p, _f32p, _f32p, _f32p]),
    "vrt_hip_radiance": (C.c_int, [_vp, C.c_size_t, _f32p, _f32p, _f32p]),
    "vrt_hip_transmittance_step": (C.c_int, [_vp, _f32p, _f32p, _f32p, C.c_size_t, C.c_float, _f32p]),
    "vrt_hip_density": (C.c_int, [_vp, C.c_size_t, _f32p, _f32p]),
    "vrt_hip_eval_erf": (C.c_int, [_vp, C.c_int, _f32p, C.c_size_t, _f32p]),
    "vrt_hip_eval_exp": (C.c_int, [_vp, C.c_int, _f32p, C.c_size_t, _f32p]),
    "vrt_hip_camera_init": (None, [_vp, _f32p, _f32p, _f32p, C.c_float, C.c_float, C.c_uint64, C.c_uint64, C.c_float]),
    "vrt_hip_camera_turn": (None, [_vp, C.c_float, C.c_float, C.c_int]),
    "vrt_hip_camera_refresh": (None, [_vp]),
    "vrt_hip_camera_plane": (None, [_vp, _f32p, _f32p, _f32p]),
    "vrt_hip_camera_orbit": (None, [_vp, C.c_float]),
    "vrt_hip_mat4_inverse": (None, [_f32p, _f32p]),
    "vrt_hip_get_stats": (C.c_int, [_vp, C.POINTER(Stats)]),
    "vrt_hip_enable_stats": (C.c_int, [_vp, C.c_int]),
    "vrt_hip_enable_kernel_timing": (C.c_int, [_vp, C.c_int]),
    "vrt_hip_get_kernel_timing": (C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                           C.POINTER(C.c_uint64)]),
}


def lib():
    """dlopen libvrt_hip.so; raises if it has not been built (never falls back to anything)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VrtHipError(f"{LIB_PATH} is missing: run __graft_entry__.build() (hipcc, gfx950)")
        try:
            # PyTorch ships its own libamdhip64; load it first so the process has ONE HIP runtime
            # (torch is the plumbing for device buffers / streams / RCCL in tests and bench.py)
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(_f32p)


def _f3(v):
    return np.ascontiguousarray(np.asarray(v, np.float32)[:3])


class Renderer:
    """One context = one GPU.  Mirrors the reference call set (main.cpp:257-296)."""

    def __init__(self, device=0, _handle=None):
        self._L = lib()
        self._owned = _handle is None
        if _handle is None:
            h = _vp()
            rc = self._L.vrt_hip_create(device, C.byref(h))
            if rc != 0:
                raise VrtHipError(f"vrt_hip_create({device}) failed ({rc}): {self._L.vrt_hip_last_error(None).decode()}")
            _handle = h
        self._h = _handle
        self.n = 0
        self.w = self.h = 0
        self.table_step = float(os.environ.get("VRT_HIP_TABLE_STEP", TABLE_STEP_DEFAULT))

    def close(self):
        if getattr(self, "_h", None):
            if self._owned:
                self._L.vrt_hip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != 0:
            raise VrtHipError(f"{what} failed ({rc}): {self._L.vrt_hip_last_error(self._h).decode()}")

    # ---- scene ----
    def set_gaussians(self, g):
        """g: structured array with fields albedo[4], mu[4], sigma, magnitude (gaussian_t, types.h:195-200)."""
        g = np.ascontiguousarray(g)
        assert g.dtype.itemsize == 40
        self.n = len(g)
        self._chk(self._L.vrt_hip_set_gaussians_aos(self._h, len(g), g.ctypes.data), "set_gaussians_aos")

    def set_gaussians_soa(self, mu, albedo, sigma, magnitude, alpha=None):
        mu = np.asarray(mu, np.float32)
        albedo = np.asarray(albedo, np.float32)
        cols = [np.ascontiguousarray(mu[:, i]) for i in range(3)] + [np.ascontiguousarray(albedo[:, i]) for i in range(3)]
        a = None if alpha is None else np.ascontiguousarray(alpha, np.float32)
        s = np.ascontiguousarray(sigma, np.float32)
        m = np.ascontiguousarray(magnitude, np.float32)
        self.n = len(s)
        self._chk(self._L.vrt_hip_set_gaussians(self._h, len(s), *[_fp(c) for c in cols], None if a is None else _fp(a),
                                                _fp(s), _fp(m)), "set_gaussians")

    # ---- tiles ----
    def tile_gaussians(self, tw, th, view):
        v = np.ascontiguousarray(view, np.float32).ravel()
        assert v.size == 16
        self._chk(self._L.vrt_hip_tile_gaussians(self._h, tw, th, _fp(v)), "tile_gaussians")

    def tile_gaussians_device(self, tw, th, view, stream=0):
        v = np.ascontiguousarray(view, np.float32).ravel()
        self._chk(self._L.vrt_hip_tile_gaussians_device(self._h, tw, th, _fp(v), stream or None), "tile_gaussians_device")

    def set_tiles(self, tiles):
        off = np.ascontiguousarray(tiles["offsets"], np.uint32)
        idx = np.ascontiguousarray(tiles["indices"], np.uint32)
        self._chk(self._L.vrt_hip_set_tiles(self._h, float(tiles["tw"]), float(tiles["th"]), int(tiles["w"]),
                                            int(tiles["h"]), off.ctypes.data_as(_u32p),
                                            idx.ctypes.data_as(_u32p) if idx.size else None), "set_tiles")

    def clear_tiles(self):
        self._chk(self._L.vrt_hip_clear_tiles(self._h), "clear_tiles")

    def tile_counts(self):
        tw, th = C.c_uint64(), C.c_uint64()
        self._chk(self._L.vrt_hip_get_tile_counts(self._h, None, 0, C.byref(tw), C.byref(th)), "get_tile_counts")
        counts = np.zeros(tw.value * th.value, np.uint32)
        self._chk(self._L.vrt_hip_get_tile_counts(self._h, counts.ctypes.data_as(_u32p), counts.size, None, None),
                  "get_tile_counts")
        return counts.reshape(th.value, tw.value)

    def tile_indices(self, t):
        cnt = C.c_uint32()
        self._chk(self._L.vrt_hip_get_tile_indices(self._h, t, None, 0, C.byref(cnt)), "get_tile_indices")
        out = np.zeros(max(cnt.value, 1), np.uint32)
        self._chk(self._L.vrt_hip_get_tile_indices(self._h, t, out.ctypes.data_as(_u32p), out.size, C.byref(cnt)),
                  "get_tile_indices")
        return out[:cnt.value]

    # ---- rays ----
    def set_plane(self, w, h, xs, ys, zs):
        xs, ys, zs = (np.ascontiguousarray(a, np.float32) for a in (xs, ys, zs))
        assert xs.size == ys.size == zs.size == w * h
        self.w, self.h = w, h
        self._chk(self._L.vrt_hip_set_plane(self._h, w, h, _fp(xs), _fp(ys), _fp(zs)), "set_plane")

    def set_camera(self, w, h, pos, right, up, front, focal=1.0):
        self.w, self.h = w, h
        self._chk(self._L.vrt_hip_set_camera(self._h, w, h, _fp(_f3(pos)), _fp(_f3(right)), _fp(_f3(up)),
                                             _fp(_f3(front)), focal), "set_camera")

    def set_camera_view(self, w, h, view):
        """In-kernel rays that are bit for bit the reference's: plane point = inverse(view) * (x, y, 0, 1)."""
        self.w, self.h = w, h
        v = np.ascontiguousarray(np.asarray(view, np.float32).reshape(16))
        self._chk(self._L.vrt_hip_set_camera_view(self._h, w, h, _fp(v)), "set_camera_view")

    def set_options(self, exp_kind=EXP_VCL, erf_kind=ERF_AS, cull_eps=1e-9):
        self._chk(self._L.vrt_hip_set_options(self._h, exp_kind, erf_kind, cull_eps), "set_options")

    # ---- render ----
    def render(self, origin, pack=PACK_ROUND | ALPHA_COMPUTED, want_radiance=True):
        """Returns (image u32 [h,w], radiance f32 [h,w,4] or None)."""
        img = np.zeros(self.w * self.h, np.uint32)
        rad = np.zeros((self.w * self.h, 4), np.float32) if want_radiance else None
        self._chk(self._L.vrt_hip_render(self._h, _fp(_f3(origin)), pack, img.ctypes.data_as(_u32p),
                                         _fp(rad) if rad is not None else None), "render")
        return img.reshape(self.h, self.w), (rad.reshape(self.h, self.w, 4) if rad is not None else None)

    def render_device(self, origin, pack, d_image, d_radiance=0, stream=0):
        self._chk(self._L.vrt_hip_render_device(self._h, _fp(_f3(origin)), pack, d_image, d_radiance or None,
                                                stream or None), "render_device")

    def frame_call(self, tw, th, view, origin, pack, shard=False, retained=False):
        """Pre-marshalled per-frame call (tile_gaussians + render in ONE C call): returns f(d_out, stream).
        The argument arrays are converted once, so an animation loop pays ~1 us of Python per frame.
        retained: vrt_hip_frame_retained_device -- the caller promises that d_out still holds this context's previous frame."""
        v = np.ascontiguousarray(view, np.float32).ravel().copy()
        o = _f3(origin).copy()
        fn, h, vp, op = self._L.vrt_hip_frame_device, self._h, _fp(v), _fp(o)
        tw, th, pack, shard = float(tw), float(th), int(pack), int(bool(shard))
        if retained:
            fnr = self._L.vrt_hip_frame_retained_device

            def call_retained(d_out, stream=0, _keep=(v, o)):
                rc = fnr(h, tw, th, vp, op, pack, d_out, stream or None)
                if rc != 0:
                    self._chk(rc, "frame_retained_device")
            return call_retained

        def call(d_out, stream=0, _keep=(v, o)):
            rc = fn(h, tw, th, vp, op, pack, d_out, shard, stream or None)
            if rc != 0:
                self._chk(rc, "frame_device")
        return call

    def frame_view_call(self, tw, th, view, origin, pack):
        """Like frame_call for a frame loop whose camera MOVES: sets the in-kernel rays of `view` (vrt_hip_set_camera_view)
        and renders the frame, two C calls behind one pre-marshalled Python call: returns f(d_out, stream)."""
        v = np.ascontiguousarray(view, np.float32).ravel().copy()
        o = _f3(origin).copy()
        setv, fn, h, vp, op = self._L.vrt_hip_set_camera_view, self._L.vrt_hip_frame_device, self._h, _fp(v), _fp(o)
        tw, th, pack, w_, h_ = float(tw), float(th), int(pack), int(self.w), int(self.h)

        def call(d_out, stream=0, _keep=(v, o)):
            rc = setv(h, w_, h_, vp)
            if rc == 0:
                rc = fn(h, tw, th, vp, op, pack, d_out, 0, stream or None)
            if rc != 0:
                self._chk(rc, "frame_device")
        return call

    def frame(self, tw, th, view, origin, pack, want_image=True, wait=True):
        """vrt_hip_frame: tile_gaussians + render on the context's own stream and buffer (what the CLI calls)."""
        v = np.ascontiguousarray(view, np.float32).ravel()
        img = np.zeros(self.w * self.h, np.uint32) if want_image else None
        self._chk(self._L.vrt_hip_frame(self._h, float(tw), float(th), _fp(v), _fp(_f3(origin)), int(pack),
                                        img.ctypes.data if want_image else None, int(wait)), "frame")
        return img.reshape(self.h, self.w) if want_image else None

    def sync(self):
        self._chk(self._L.vrt_hip_sync(self._h), "sync")

    def set_shard(self, rank, world):
        self._chk(self._L.vrt_hip_set_shard(self._h, rank, world), "set_shard")

    def shard_pixels(self):
        return self._L.vrt_hip_shard_pixels(self._h)

    def render_shard_device(self, origin, pack, d_shard, stream=0):
        self._chk(self._L.vrt_hip_render_shard_device(self._h, _fp(_f3(origin)), pack, d_shard, stream or None),
                  "render_shard_device")

    # ---- sparse shards (only the non-empty 32x32 cells travel) ----
    def sparse_shard_words(self):
        return self._L.vrt_hip_sparse_shard_words(self._h)

    def frame_sparse_call(self, tw, th, view, origin, pack):
        """Pre-marshalled vrt_hip_frame_sparse_device: returns f(d_sparse, stream)."""
        v = np.ascontiguousarray(view, np.float32).ravel().copy()
        o = _f3(origin).copy()
        fn, h, vp, op = self._L.vrt_hip_frame_sparse_device, self._h, _fp(v), _fp(o)
        tw, th, pack = float(tw), float(th), int(pack)

        def call(d_sparse, stream=0, _keep=(v, o)):
            rc = fn(h, tw, th, vp, op, pack, d_sparse, stream or None)
            if rc != 0:
                self._chk(rc, "frame_sparse_device")
        return call

    def frame_batch_call(self, others, tw, th, views, origins, pack, out_kind=0):
        """Pre-marshalled vrt_hip_frame_batch_device: frame i of a batch is rendered by (self, *others)[i] with views[i],
        origins[i]; returns f(out_ptrs, stream, n=None) -- the first n frames into the device pointers out_ptrs[i]
        (out_kind 0: raster frames, 1: compact shards, 2: sparse shards), every kernel launched once for all of them."""
        ctxs = [self, *others]
        v = np.ascontiguousarray(views, np.float32).reshape(len(ctxs), 16).copy()
        o = np.ascontiguousarray(origins, np.float32).reshape(len(ctxs), 3).copy()
        harr = (_vp * len(ctxs))(*[c_._h.value for c_ in ctxs])
        fn, vp, op = self._L.vrt_hip_frame_batch_device, _fp(v), _fp(o)
        tw, th, pack, out_kind = float(tw), float(th), int(pack), int(out_kind)

        def call(out_ptrs, stream=0, n=None, _keep=(v, o, ctxs)):
            n = len(ctxs) if n is None else int(n)
            outs = (_vp * n)(*[int(p_) for p_ in out_ptrs[:n]])
            rc = fn(harr, n, tw, th, vp, op, pack, outs, out_kind, stream or None)
            if rc != 0:
                self._chk(rc, "frame_batch_device")
        return call

    def scatter_sparse_device(self, shard_ptrs, pack, d_image, stream=0, retained=False):
        """Background + every stored cell of every shard (device pointers readable from this context's GPU).  retained:
        d_image still holds this context's previous retained assembly -- only cells that went dark are reset."""
        arr = (_vp * len(shard_ptrs))(*[int(p) for p in shard_ptrs])
        fn = self._L.vrt_hip_scatter_sparse_retained_device if retained else self._L.vrt_hip_scatter_sparse_device
        self._chk(fn(self._h, arr, len(shard_ptrs), int(pack), d_image, stream or None), "scatter_sparse_device")

    def scatter_sparse_batch_device(self, shard_ptrs, frame_stride_words, nframes, pack, image_ptrs, stream=0, retained=False):
        """vrt_hip_scatter_sparse_batch_device: frame f from shard s at shard_ptrs[s] + 4 * f * frame_stride_words bytes into
        image_ptrs[f], all frames in one launch."""
        arr = (_vp * len(shard_ptrs))(*[int(p) for p in shard_ptrs])
        imgs = (_vp * nframes)(*[int(p) for p in image_ptrs[:nframes]])
        self._chk(self._L.vrt_hip_scatter_sparse_batch_device(self._h, arr, len(shard_ptrs), int(frame_stride_words), int(nframes), int(pack),
                                                              imgs, int(bool(retained)), stream or None), "scatter_sparse_batch_device")

    def assemble_shards_device(self, d_gathered, d_image, stream=0, rank_stride_px=None):
        """Scatter a rank-major gather result into raster order; rank_stride_px > shard_pixels() when the gather
        carried several frames per rank (d_gathered then points at the frame's slice of rank 0)."""
        if rank_stride_px is None:
            rc = self._L.vrt_hip_assemble_shards_device(self._h, d_gathered, d_image, stream or None)
        else:
            rc = self._L.vrt_hip_assemble_shards_strided_device(self._h, d_gathered, rank_stride_px, d_image, stream or None)
        self._chk(rc, "assemble_shards_device")

    # ---- point queries ----
    def transmittance(self, o, n, s):
        s = np.ascontiguousarray(np.atleast_1d(s), np.float32)
        out = np.zeros_like(s)
        self._chk(self._L.vrt_hip_transmittance(self._h, _fp(_f3(o)), _fp(_f3(n)), _fp(s), s.size, _fp(out)),
                  "transmittance")
        return out

    def transmittance_rays(self, origins, dirs, s):
        """broadcast_transmittance (rt.h:102-127): one sample point per ray."""
        origins = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        dirs = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        s = np.ascontiguousarray(s, np.float32).ravel()
        assert len(origins) == len(dirs) == s.size
        out = np.zeros_like(s)
        self._chk(self._L.vrt_hip_transmittance_rays(self._h, s.size, _fp(origins), _fp(dirs), _fp(s), _fp(out)),
                  "transmittance_rays")
        return out

    def transmittance_step(self, o, n, s, delta):
        s = np.ascontiguousarray(np.atleast_1d(s), np.float32)
        out = np.zeros_like(s)
        self._chk(self._L.vrt_hip_transmittance_step(self._h, _fp(_f3(o)), _fp(_f3(n)), _fp(s), s.size, delta,
                                                     _fp(out)), "transmittance_step")
        return out

    def density(self, pts):
        pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 3)
        out = np.zeros(len(pts), np.float32)
        self._chk(self._L.vrt_hip_density(self._h, len(pts), _fp(pts), _fp(out)), "density")
        return out

    def radiance(self, origins, dirs):
        origins = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        dirs = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        out = np.zeros((len(dirs), 4), np.float32)
        self._chk(self._L.vrt_hip_radiance(self._h, len(dirs), _fp(origins), _fp(dirs), _fp(out)), "radiance")
        return out

    def eval_erf(self, kind, x):
        x = np.ascontiguousarray(x, np.float32)
        y = np.zeros_like(x)
        self._chk(self._L.vrt_hip_eval_erf(self._h, kind, _fp(x), x.size, _fp(y)), "eval_erf")
        return y

    def eval_exp(self, kind, x):
        x = np.ascontiguousarray(x, np.float32)
        y = np.zeros_like(x)
        self._chk(self._L.vrt_hip_eval_exp(self._h, kind, _fp(x), x.size, _fp(y)), "eval_exp")
        return y

    def enable_stats(self, on=True):
        self._chk(self._L.vrt_hip_enable_stats(self._h, int(on)), "enable_stats")

    def set_table_step(self, step):
        """Table mode for dense blocks (default 0.05; 0 = exact kernels only): see vrt_hip_set_table_step in include/vrt_hip.h."""
        self._chk(self._L.vrt_hip_set_table_step(self._h, float(step)), "set_table_step")
        self.table_step = float(step)

    def set_cull_prune(self, kappa):
        """Budget factor of the block kernel's ray-level prune (default 6; 0 = off): see vrt_hip_set_cull_prune in include/vrt_hip.h."""
        self._chk(self._L.vrt_hip_set_cull_prune(self._h, float(kappa)), "set_cull_prune")

    def set_table_budget(self, budget):
        """Largest worst-case change of a ray's radiance the table kernel may cause (default 2.5e-5)."""
        self._chk(self._L.vrt_hip_set_table_budget(self._h, float(budget)), "set_table_budget")

    def enable_kernel_timing(self, on=True):
        self._chk(self._L.vrt_hip_enable_kernel_timing(self._h, int(on)), "enable_kernel_timing")

    def kernel_timing(self):
        """Mean HIP-event durations (ms) of the render kernel, the dense kernel and the list kernels."""
        r, d, l, n = C.c_double(), C.c_double(), C.c_double(), C.c_uint64()
        self._chk(self._L.vrt_hip_get_kernel_timing(self._h, C.byref(r), C.byref(d), C.byref(l), C.byref(n)),
                  "get_kernel_timing")
        return dict(render_ms=r.value, dense_ms=d.value, lists_ms=l.value, launches=n.value)

    def stats(self):
        s = Stats()
        self._chk(self._L.vrt_hip_get_stats(self._h, C.byref(s)), "get_stats")
        return {k: (list(getattr(s, k)) if k == "table_phase_ticks" else getattr(s, k)) for k, _ in Stats._fields_}


class Group:
    """Several GPUs from one process (vrt_hip_group_*): one member context per entry of `devices`; a device listed twice
    gives two members on it (protocol tests on a one-GPU box)."""

    def __init__(self, devices):
        self._L = lib()
        d = (C.c_int * len(devices))(*devices)
        h = _vp()
        rc = self._L.vrt_hip_group_create(d, len(devices), C.byref(h))
        if rc != 0:
            raise VrtHipError(f"vrt_hip_group_create failed ({rc}): {self._L.vrt_hip_group_last_error(None).decode()}")
        self._g = h
        self.members = [Renderer(_handle=_vp(self._L.vrt_hip_group_ctx(h, i))) for i in range(len(devices))]

    def frame(self, tw, th, view, origin, pack, want_image=True, wait=True):
        w, h = self.members[0].w, self.members[0].h
        img = np.zeros(w * h, np.uint32) if want_image else None
        v = np.ascontiguousarray(view, np.float32).ravel()
        rc = self._L.vrt_hip_group_frame(self._g, float(tw), float(th), _fp(v), _fp(_f3(origin)), int(pack),
                                         img.ctypes.data_as(_u32p) if want_image else None, int(wait))
        if rc != 0:
            raise VrtHipError(f"group_frame failed ({rc}): {self._L.vrt_hip_group_last_error(self._g).decode()}")
        return img.reshape(h, w) if want_image else None

    def frame_batch(self, tw, th, views, origins, pack, want_images=True):
        """vrt_hip_group_frame_batch: len(views) frames with one launch of each kernel per member and one assembly launch;
        returns the images [n, h, w] (or None)."""
        n = len(views)
        w, h = self.members[0].w, self.members[0].h
        v = np.ascontiguousarray(np.asarray(views, np.float32).reshape(n, 16))
        o = np.ascontiguousarray(np.asarray(origins, np.float32).reshape(n, 3))
        imgs = np.zeros((n, h, w), np.uint32) if want_images else None
        ptrs = (C.c_void_p * n)(*[imgs[i].ctypes.data for i in range(n)]) if want_images else None
        rc = self._L.vrt_hip_group_frame_batch(self._g, n, float(tw), float(th), _fp(v), _fp(o), int(pack), ptrs, 1)
        if rc != 0:
            raise VrtHipError(f"group_frame_batch failed ({rc}): {self._L.vrt_hip_group_last_error(self._g).decode()}")
        return imgs

    def sync(self):
        rc = self._L.vrt_hip_group_sync(self._g)
        if rc != 0:
            raise VrtHipError(f"group_sync failed ({rc}): {self._L.vrt_hip_group_last_error(self._g).decode()}")

    def close(self):
        if getattr(self, "_g", None):
            for m in self.members:
                m.close()
            self._L.vrt_hip_group_destroy(self._g)
            self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
