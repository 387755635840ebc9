#!/usr/bin/env python3
"""Is `notes/teapot.png` of the reference tree (fixture: tests/golden/notes/teapot.png) an output of the renderer's
scalar / opaque-alpha modes?

512 x 512 RGBA, alpha 255 everywhere: what `render_image` (modes 1-3, 5-7: truncating pack, rt.h:239-243, 279-283) and the
untiled `simd_render_image` (mode 4: rounding pack, rt.h:329-333) write.  The slides embed it as a "preliminary result"
(notes/slides.md:147); its command line is not recorded.  This tool searches, with the GPU renderer, the CLI's parameter
space -- camera offset (-c), focal length, initial rotation (-i), tiled / untiled, truncating / rounding pack, the
conventions (vertical flip, channel order) -- and, because an earlier revision may have used another sigma rule, sigma.
Everything is scored on the RGB bytes (alpha carries no information here).

    python tools/notes_png_fit.py > profiles/r03_notes_png_fit.md      (on the MI355X box)
"""
import itertools
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
from conftest import GOLDEN, load_pkg  # noqa: E402

pkg = load_pkg()
from sgrt_amd import scene  # noqa: E402
from PIL import Image  # noqa: E402

png = np.array(Image.open(os.path.join(GOLDEN, "notes", "teapot.png"))).astype(np.int16)     # [h, w, 4] file order
H, W = png.shape[:2]
g0 = scene.read_obj(os.path.join(GOLDEN, "test-objects", "teapot.obj"))
r = pkg.Renderer(0)
nrender = 0
cur_sigma = [None]


def frame(offset=-4.0, focal=1.0, rot=0.0, tiles=16, sigma=0.05, rounding=False, exact=False):
    """The CLI's frame with opaque alpha as the bytes stbi_write_png stores (u32 A|R|G|B little-endian); tiles = 0: untiled."""
    global nrender
    nrender += 1
    if cur_sigma[0] != sigma:
        g = g0.copy(); g["sigma"] = sigma
        r.set_gaussians(g); cur_sigma[0] = sigma
    r.set_table_step(0.0 if exact else pkg.TABLE_STEP_DEFAULT)
    cam, _ = scene.cli_camera(W, H, camera_offset=offset, focal=focal, initial_rot=rot)
    r.set_camera_view(W, H, cam.view)
    if tiles:
        r.tile_gaussians(2.0 / tiles, 2.0 / tiles, cam.view)
    else:
        r.clear_tiles()
    pack = (pkg.PACK_ROUND if rounding else pkg.PACK_TRUNC) | pkg.ALPHA_OPAQUE
    img, _ = r.render(cam.position, pack, want_radiance=False)
    return img.view(np.uint8).reshape(H, W, 4).astype(np.int16)


def score(img, ref=png):
    d = np.abs(img[..., :3] - ref[..., :3])
    return float(d.mean()), int(d.max()), float((d <= 1).mean())


CONVS = [(flip, order) for flip in (False, True) for order in ((0, 1, 2), (2, 1, 0))]


def conv_apply(img, conv):
    flip, order = conv
    v = img[::-1] if flip else img
    return np.concatenate([v[..., list(order)], v[..., 3:]], axis=2)


def label(conv):
    return f"flip={int(conv[0])} order={''.join('BGR'[i] for i in conv[1])}A"


lit = png[..., :3].sum(2) > 0
ys, xs = np.nonzero(lit)
print("# notes/teapot.png vs the renderer: parameter search (`tools/notes_png_fit.py`)\n")
print(f"{len(g0)} Gaussians; PNG {W}x{H}, alpha in [{png[..., 3].min()}, {png[..., 3].max()}], {lit.mean() * 100:.1f} % of the pixels lit, "
      f"lit bounding box x {xs.min()}..{xs.max()}, y {ys.min()}..{ys.max()}; channel maxima (file order) {[int(png[..., c].max()) for c in range(3)]}\n")

# ---- 1. coarse grid: camera distance x focal length x sigma, best convention per cell ------------------------------
print("## 1. coarse grid, tiled (--tiles 16), truncating pack, rotation 0: mean abs diff of the RGB bytes, best convention per cell\n")
best = (1e9, None)
for sigma in (0.03, 0.05, 0.075, 0.1, 0.15):
    offsets = [-2.0, -2.5, -3.0, -3.5, -4.0, -5.0, -6.0, -8.0]
    focals = [0.5, 0.75, 1.0, 1.25, 1.5, 2.0, 2.5]
    print(f"sigma {sigma:g}\n\n| -c \\\\ focal | " + " | ".join(f"{f:g}" for f in focals) + " |\n|---|" + "---|" * len(focals))
    for off in offsets:
        row = []
        for fo in focals:
            img = frame(off, fo, sigma=sigma)
            m = min((score(conv_apply(img, c))[0], c) for c in CONVS)
            row.append(f"{m[0]:.2f}")
            if m[0] < best[0]:
                best = (m[0], dict(offset=off, focal=fo, rot=0.0, sigma=sigma), m[1])
        print(f"| {off:g} | " + " | ".join(row) + " |")
    print()
print(f"best cell: {best[1]}, {label(best[2])}: {best[0]:.3f}\n")

# ---- 2. rotation scan at the best cell's sigma (an orbit frame or -i) ------------------------------------------------
p, conv = dict(best[1]), best[2]
print("## 2. rotation scan (best distance / focal length of the grid per angle, its sigma)\n")
print("| rotation | -c | focal | mean abs diff | max | within 1 LSB | convention |\n|---|---|---|---|---|---|---|")
scan = []
for rot in np.arange(0.0, 360.0, 15.0):
    cell = (1e9,)
    for off in (-3.0, -4.0, -5.0):
        for fo in (0.75, 1.0, 1.5, 2.0):
            img = frame(off, fo, rot=float(rot), sigma=p["sigma"])
            for c in CONVS:
                sc = score(conv_apply(img, c))
                if sc[0] < cell[0]:
                    cell = (sc[0], sc[1], sc[2], off, fo, c)
    scan.append((cell[0], float(rot), cell[3], cell[4], cell[5]))
    print(f"| {rot:.0f} | {cell[3]:g} | {cell[4]:g} | {cell[0]:.3f} | {cell[1]} | {cell[2] * 100:.2f} % | {label(cell[5])} |")
scan.sort()
if scan[0][0] < best[0]:
    p.update(rot=scan[0][1], offset=scan[0][2], focal=scan[0][3]); conv = scan[0][4]
print(f"\nstart of the refinement: {p}, {label(conv)}\n")


def f(q):
    return score(conv_apply(frame(q["offset"], q["focal"], q["rot"], 16, q["sigma"]), conv))


# ---- 3. coordinate descent over distance, focal length, rotation and sigma ------------------------------------------
print("## 3. coordinate descent (tiled, truncating pack)\n")
print("| step | -c | focal | rot | sigma | mean abs diff | max | within 1 LSB |\n|---|---|---|---|---|---|---|---|")
cur = f(p)
step = dict(rot=4.0, focal=0.1, offset=0.25, sigma=0.01)
k = 0
for it in range(10):
    for key in ("offset", "focal", "rot", "sigma"):
        improved = True
        while improved:
            improved = False
            for sgn in (+1, -1):
                q = dict(p); q[key] = p[key] + sgn * step[key]
                if q["sigma"] < 0.005 or q["focal"] < 0.1:
                    continue
                sc = f(q)
                if sc[0] < cur[0] - 1e-4:
                    p, cur, improved = q, sc, True
                    k += 1
                    print(f"| {k} | {p['offset']:.4f} | {p['focal']:.4f} | {p['rot']:.3f} | {p['sigma']:.4f} | {cur[0]:.3f} | {cur[1]} | {cur[2] * 100:.2f} % |")
                    break
    step = {a: b / 2 for a, b in step.items()}
print(f"\nbest: {p}: mean abs diff {cur[0]:.3f}, max {cur[1]}, within 1 LSB {cur[2] * 100:.2f} %\n")

# ---- 4. the discrete choices at the best point ----------------------------------------------------------------------
print("## 4. modes at the best point\n")
print("| variant | mean abs diff | max | within 1 LSB |\n|---|---|---|---|")
for lab, kw in (("tiled (modes 5-7), truncating pack", dict(tiles=16, rounding=False)), ("tiled, rounding pack", dict(tiles=16, rounding=True)),
                ("untiled (modes 1-3), truncating pack", dict(tiles=0, rounding=False)), ("untiled (mode 4), rounding pack", dict(tiles=0, rounding=True)),
                ("tiled --tiles 8, truncating", dict(tiles=8, rounding=False)), ("tiled --tiles 32, truncating", dict(tiles=32, rounding=False)),
                ("tiled, truncating, exact kernels (table step 0)", dict(tiles=16, rounding=False, exact=True))):
    sc = score(conv_apply(frame(p["offset"], p["focal"], p["rot"], sigma=p["sigma"], **kw), conv))
    print(f"| {lab} | {sc[0]:.3f} | {sc[1]} | {sc[2] * 100:.2f} % |")
img = conv_apply(frame(p["offset"], p["focal"], p["rot"], 16, p["sigma"]), conv)
d = np.abs(img[..., :3] - png[..., :3])
ylit, xlit = np.nonzero(img[..., :3].sum(2) > 0)
print(f"\nrendered lit bounding box x {xlit.min()}..{xlit.max()}, y {ylit.min()}..{ylit.max()}; channel maxima {[int(img[..., c].max()) for c in range(3)]}")
hist = np.bincount(d.ravel(), minlength=12)
print("\nhistogram of |difference| over the RGB values: " + ", ".join(f"{k_}: {v}" for k_, v in enumerate(hist) if v))
# ---- 5. one channel at a time: which of this build's radiance channels does each PNG channel follow? ------------------
print("\n## 5. channel by channel at the best point: correlation with this build's float radiance (R, G, B and w = the plain sum of the emission terms)\n")
r.set_table_step(pkg.TABLE_STEP_DEFAULT)
cam, _ = scene.cli_camera(W, H, camera_offset=p["offset"], focal=p["focal"], initial_rot=p["rot"])
r.set_camera_view(W, H, cam.view); r.tile_gaussians(2.0 / 16, 2.0 / 16, cam.view)
_, rad = r.render(cam.position, pkg.PACK_TRUNC | pkg.ALPHA_OPAQUE)
rad = rad.reshape(-1, 4).astype(np.float64)
P = png.reshape(-1, 4)[:, :3].astype(np.float64)
sel = (P.sum(1) > 0) | (rad[:, 3] > 1e-3)
print("| PNG channel (file order) | corr with R | G | B | w |\n|---|---|---|---|---|")
for c in range(3):
    print(f"| {c} | " + " | ".join(f"{np.corrcoef(P[sel][:, c], rad[sel][:, k_])[0, 1]:.3f}" for k_ in range(4)) + " |")
print("\nPNG channel 0 against floor(255 min(1, w)) -- the opaque truncating pack of a white scene -- refined over distance, focal length, rotation, sigma:\n")
print("| -c | focal | rot | sigma | Exp/Erf | mean abs diff | max | within 1 step | equal |\n|---|---|---|---|---|---|---|---|---|")


def wscore(q, kinds=(pkg.EXP_VCL, pkg.ERF_AS)):
    global nrender
    nrender += 1
    if cur_sigma[0] != q["sigma"]:
        g = g0.copy(); g["sigma"] = q["sigma"]; r.set_gaussians(g); cur_sigma[0] = q["sigma"]
    r.set_options(kinds[0], kinds[1], 1e-9)
    c_, _ = scene.cli_camera(W, H, camera_offset=q["offset"], focal=q["focal"], initial_rot=q["rot"])
    r.set_camera_view(W, H, c_.view); r.tile_gaussians(2.0 / 16, 2.0 / 16, c_.view)
    _, rd = r.render(c_.position, pkg.PACK_TRUNC | pkg.ALPHA_OPAQUE)
    d_ = np.abs(np.floor(np.minimum(rd[..., 3].astype(np.float64), 1.0) * 255) - png[..., 0])
    return float(d_.mean()), int(d_.max()), float((d_ <= 1).mean()), float((d_ == 0).mean())


q = dict(p); wc = wscore(q)
st = dict(rot=1.0, focal=0.02, offset=0.05, sigma=0.002)
for it in range(9):
    for key in ("offset", "focal", "rot", "sigma"):
        improved = True
        while improved:
            improved = False
            for sgn in (+1, -1):
                q2 = dict(q); q2[key] = q[key] + sgn * st[key]
                sc = wscore(q2)
                if sc[0] < wc[0] - 1e-5:
                    q, wc, improved = q2, sc, True
                    break
    st = {a: b / 2 for a, b in st.items()}
for lab, kinds in (("vcl_exp / A&S erf", (pkg.EXP_VCL, pkg.ERF_AS)), ("expf / erff (modes 1, 5)", (pkg.EXP_LIBM, pkg.ERF_LIBM)), ("expf / A&S erf", (pkg.EXP_LIBM, pkg.ERF_AS))):
    sc = wscore(q, kinds)
    print(f"| {q['offset']:.4f} | {q['focal']:.4f} | {q['rot']:.3f} | {q['sigma']:.5f} | {lab} | {sc[0]:.3f} | {sc[1]} | {sc[2] * 100:.2f} % | {sc[3] * 100:.2f} % |")
r.set_options(pkg.EXP_VCL, pkg.ERF_AS, 1e-9)
lit_share = float(((png[..., :3].sum(2) > 0)).mean())
print(f"\n({lit_share * 100:.1f} % of the pixels are lit: of THOSE, {(wc[2] - (1 - lit_share)) / lit_share * 100:.0f} % are within one step.)\n")
print("Reading: the silhouette and one channel follow this revision's geometry and emission sum closely but not to the step, the other two "
      "channels follow another albedo rule: the image comes from an earlier revision of the renderer and is no golden vector for the scalar / "
      "opaque-alpha modes of this one.\n")
verdict = ("REPRODUCED within 1 LSB on >= 99 % of the values" if cur[2] >= 0.99 and cur[1] <= 2 else
           ("CLOSE (see the histogram)" if cur[0] < 0.5 else "NOT reproducible from CLI parameters of this revision"))
print(f"\n**{verdict}** ({nrender} frames rendered)")
r.close()
