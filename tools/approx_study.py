#!/usr/bin/env python3
"""GPU analogue of the reference's approximation study (tests/accuracy.cpp, tests/approx_cycles.cpp, tests/img-error.cpp):
for every (Exp, Erf) kernel variant the library builds -- accuracy of the elementwise function on the reference's grids
(accuracy.cpp:19,35-39), frame time at `-g 16 -w 1024` and `-g 64 -w 2048`, and the mean squared u8-channel error of the
rendered image against the expf/erff variant (img-error.cpp:44-57).  Prints a markdown table.

    python tools/approx_study.py > profiles/rNN_approx_study.md        (on an MI355X)
"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import torch
from scipy.special import erf as erf64
from conftest import load_pkg

pkg = load_pkg()
from sgrt_amd import scene

r = pkg.Renderer(0)
EXP = {"expf": pkg.EXP_LIBM, "vcl_exp": pkg.EXP_VCL, "fast_exp": pkg.EXP_FAST, "spline_exp": pkg.EXP_SPLINE}
ERF = {"erff": pkg.ERF_LIBM, "A&S": pkg.ERF_AS, "spline": pkg.ERF_SPLINE, "spline_mirror": pkg.ERF_SPLINE_MIRROR, "taylor": pkg.ERF_TAYLOR}
VARIANTS = [("expf", "erff"), ("expf", "A&S"), ("vcl_exp", "erff"), ("vcl_exp", "A&S"), ("fast_exp", "A&S"), ("spline_exp", "A&S"),
            ("vcl_exp", "spline"), ("vcl_exp", "spline_mirror"), ("vcl_exp", "taylor")]

xe = np.arange(-6.0, 6.0001, 0.1, dtype=np.float32)
xx = np.arange(-16.0, 0.0001, 0.1, dtype=np.float32)
print("## Elementwise accuracy on the reference's grids (max abs error vs float64)\n")
print("| function | max abs err |\n|---|---|")
for name, k in ERF.items():
    print(f"| erf: {name} on [-6, 6] | {np.abs(r.eval_erf(k, xe).astype(np.float64) - erf64(xe.astype(np.float64))).max():.3e} |")
for name, k in EXP.items():
    lo = -9.0 if name == "spline_exp" else -16.0    # spline_exp is defined on [-9, 0] (approx.cpp:141-188)
    x = xx[xx >= lo]
    print(f"| exp: {name} on [{lo:g}, 0] | {np.abs(r.eval_exp(k, x).astype(np.float64) - np.exp(x.astype(np.float64))).max():.3e} |")


def frames(g, w, ek, rk, reps):
    cam, _ = scene.cli_camera(w, w)
    r.set_gaussians(scene.grid_scene(g)); r.set_camera_view(w, w, cam.view)
    r.set_options(ek, rk, 1e-9)
    r.tile_gaussians(2 / 16, 2 / 16, cam.view)
    pack = pkg.PACK_ROUND | pkg.ALPHA_COMPUTED
    img, _ = r.render(cam.position, pack=pack, want_radiance=False)
    out = torch.zeros(w * w, dtype=torch.int32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    f = r.frame_call(2 / 16, 2 / 16, cam.view, cam.position, pack)
    for _ in range(10):
        f(out.data_ptr(), s)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        f(out.data_ptr(), s)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, img


def channels(img):
    return np.stack([(img >> s) & 0xFF for s in (16, 8, 0)], -1).astype(np.float64) / 255.0


print("\n## Kernel variants: frame time and image error against expf/erff\n")
print("| Exp | Erf | `-g 16 -w 1024` ms | `-g 64 -w 2048` ms | MSE vs expf/erff (`-g 16`) | max abs u8 diff |\n|---|---|---|---|---|---|")
ref = None
for en, rn in VARIANTS:
    ms2, img2 = frames(16, 1024, EXP[en], ERF[rn], 200)
    ms4, _ = frames(64, 2048, EXP[en], ERF[rn], 200)
    if ref is None:
        ref = img2
    a, b = channels(img2), channels(ref)
    mse = ((a - b) ** 2).sum(-1).mean()
    print(f"| {en} | {rn} | {ms2:.4f} | {ms4:.4f} | {mse:.3e} | {int(np.abs(a - b).max() * 255 + 0.5)} |")
