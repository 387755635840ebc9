"""The one artefact in the reference tree that the reference renderer itself produced, used as a golden vector.

`thesis/images/teapot.png` (fixture: tests/golden/thesis/teapot.png) is the `-o` output of the reference's tiled
SIMD-over-pixels mode: 1024 x 1024 RGBA with the computed alpha of rt.h:373.  Its command line is not recorded; the search
in tools/thesis_png_fit.py (table: profiles/r02_thesis_png_fit_teapot.md) identifies it as

    volumetric-ray-tracer -f test-objects/teapot.obj -w 1024 --focal-length 1.7        (-c -4, --tiles 16, mode 8, rotation 0)

at which this repository's frame equals the PNG in 4,185,372 of 4,194,304 channel values and is off by ONE u8 step in
the other 8,932 -- the reference's own rcp-estimate noise (hard part 1).  That pins, against a
reference-produced output: the OBJ loader, the camera (focal length 1.7: view matrix and plane points), tile_gaussians,
broadcast_radiance / broadcast_transmittance with A&S erf and VCL exp, the rounding pack with computed alpha and the
B,G,R,A byte order of the PNG (main.cpp:306).

CPU tests: the ORACLE (oracle/vrt_oracle.c) and the CPU baseline port (oracle/vrt_cpu_simd.*) against the PNG on seeded
pixel / tile samples.  GPU tests: the HIP path, full frame, through the Python binding and through the CLI binary.

The second image, `thesis/images/cube.png`, is `-f test-objects/cube.obj -w 1024 --focal-length 1.5 -i 30` (386 Gaussians of
sigma 0.15, the camera turned by 30 degrees: profiles/r02_thesis_png_fit_cube.md) -- the same chain at another focal
length and with the orbit step / camera_t::turn of main.cpp:252-255 in it.
"""
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

W = H = 1024
FOCAL = 1.7
OBJ = os.path.join(GOLDEN, "test-objects", "teapot.obj")
CUBE_OBJ = os.path.join(GOLDEN, "test-objects", "cube.obj")
CUBE_FOCAL, CUBE_ROT = 1.5, 30.0
BIN = os.path.join(ROOT, "simd-gaussian-ray-tracing_amd", "bin")


def thesis_png(name="teapot"):
    from PIL import Image
    png = np.array(Image.open(os.path.join(GOLDEN, "thesis", f"{name}.png")))
    assert png.shape == (H, W, 4) and png.dtype == np.uint8
    return png.astype(np.int16)


def file_bytes(img_u32):
    """u32 A<<24|R<<16|G<<8|B, little-endian, written as if RGBA (main.cpp:306): file bytes are B, G, R, A."""
    return np.asarray(img_u32, np.uint32).view(np.uint8).reshape(-1, 4).astype(np.int16)


def sample_pixels(png, n, seed):
    """Seeded sample of covered pixels: a third from the brightest, a third from the silhouette (alpha 1..40, where tile
    sets and culling matter most), a third uniform over the covered area; plus a few background pixels."""
    rng = np.random.default_rng(seed)
    a = png[..., 3].ravel()
    lum = png[..., :3].sum(-1).ravel()
    covered = np.flatnonzero(a > 0)
    bright = covered[np.argsort(lum[covered])[-20000:]]
    edge = np.flatnonzero((a > 0) & (a <= 40))
    k = n // 3
    pix = np.concatenate([rng.choice(bright, k, replace=False), rng.choice(edge, k, replace=False),
                          rng.choice(covered, n - 2 * k, replace=False), rng.choice(np.flatnonzero(a == 0), 8, replace=False)])
    return np.unique(pix).astype(np.uint32)


def test_oracle_reproduces_the_reference_image(oracle):
    """The scalar restatement (exact divides, A&S erf, VCL exp) against the reference's own output: <= 1 u8 step on every
    sampled channel value, and equal on almost all of them."""
    png = thesis_png().reshape(-1, 4)
    g = oracle.read_obj(OBJ)
    assert len(g) == 3644 and float(g["sigma"][0]) == np.float32(0.05)
    cam, _ = oracle.cli_camera(W, H, focal=FOCAL)
    tiles = oracle.tile_gaussians(2 / 16, 2 / 16, g, oracle.camera_view(cam))
    pix = sample_pixels(thesis_png(), 420, seed=2024)
    img, rad = oracle.render(W, H, oracle.camera_plane(cam), cam.position[:], g, tiles, pixels=pix)
    got = file_bytes(img[pix])
    d = np.abs(got - png[pix])
    assert d.max() <= 1, (int(d.max()), pix[np.argmax(d.max(1))])
    assert (d == 0).mean() >= 0.98
    assert png[pix][:, 3].max() > 200 and (png[pix][:, 3] == 0).sum() >= 8      # the sample spans opaque .. background


def test_cpu_baseline_port_reproduces_the_reference_image(oracle):
    """oracle/vrt_cpu_simd.* -- the SIMD-over-pixels port timed as `cpu_baseline` (kind "port") -- renders whole pixel rows of
    four tiles of the same frame: <= 1 u8 step from the reference's output, i.e. the port does the reference's work."""
    png = thesis_png()
    g = oracle.read_obj(OBJ)
    cam, _ = oracle.cli_camera(W, H, focal=FOCAL)
    tiles = oracle.tile_gaussians(2 / 16, 2 / 16, g, oracle.camera_view(cam))
    subset = [6 * 16 + 7, 7 * 16 + 3, 9 * 16 + 12, 10 * 16 + 8]                  # lid, handle, spout base, body
    rows = 6
    img, terms, width = oracle.simd_render_tiled(W, H, oracle.camera_plane(cam), cam.position[:], g, tiles, tile_subset=subset, max_rows=rows)
    assert width in (8, 16) and terms > 0
    img = img.reshape(H, W)
    worst, n = 0, 0
    for t in subset:
        ty, tx = divmod(t, 16)
        got = file_bytes(img[ty * 64:ty * 64 + rows, tx * 64:(tx + 1) * 64]).reshape(rows, 64, 4)
        ref = png[ty * 64:ty * 64 + rows, tx * 64:(tx + 1) * 64]
        worst = max(worst, int(np.abs(got - ref).max()))
        n += int((ref[..., 3] > 0).sum())
    assert worst <= 1, worst
    assert n > 600                                                               # the sampled rows are on the teapot


def test_oracle_reproduces_the_rotated_cube_image(oracle):
    """thesis/images/cube.png: another object, sigma, focal length -- and a camera that was turned (-i 30): the orbit step
    (glm::rotate about +Y), camera_t::turn and the view matrix of a rotated pose are in this one."""
    png = thesis_png("cube").reshape(-1, 4)
    g = oracle.read_obj(CUBE_OBJ)
    assert len(g) == 386 and float(g["sigma"][0]) == np.float32(0.15)
    cam, angle = oracle.cli_camera(W, H, focal=CUBE_FOCAL, initial_rot=CUBE_ROT)
    assert angle[0] == np.float32(-120.0)
    tiles = oracle.tile_gaussians(2 / 16, 2 / 16, g, oracle.camera_view(cam))
    pix = sample_pixels(thesis_png("cube"), 900, seed=386)
    img, _ = oracle.render(W, H, oracle.camera_plane(cam), cam.position[:], g, tiles, pixels=pix)
    d = np.abs(file_bytes(img[pix]) - png[pix])
    assert d.max() <= 1, (int(d.max()), pix[np.argmax(d.max(1))])
    assert (d == 0).mean() >= 0.97


@pytest.mark.gpu
def test_gpu_frame_is_the_rotated_cube_image(pkg, renderer):
    from sgrt_amd import scene
    png = thesis_png("cube")
    g = scene.read_obj(CUBE_OBJ)
    cam, _ = scene.cli_camera(W, H, focal=CUBE_FOCAL, initial_rot=CUBE_ROT)
    renderer.set_gaussians(g)
    renderer.set_options(pkg.EXP_VCL, pkg.ERF_AS, 1e-9)
    renderer.set_camera_view(W, H, cam.view)
    renderer.tile_gaussians(2 / 16, 2 / 16, cam.view)
    img, _ = renderer.render(cam.position, pkg.PACK_ROUND | pkg.ALPHA_COMPUTED, want_radiance=False)
    d = np.abs(file_bytes(img).reshape(H, W, 4) - png)
    assert d.max() <= 1
    assert (d == 0).mean() >= 0.99


@pytest.mark.gpu
def test_gpu_frame_is_the_reference_image(pkg, renderer):
    """The HIP path, full frame, product camera: every channel value within one u8 step of the reference's PNG and
    >= 99.5 % of them equal (measured: 8,932 of 4,194,304 differ)."""
    from sgrt_amd import scene
    png = thesis_png()
    g = scene.read_obj(OBJ)
    cam, _ = scene.cli_camera(W, H, focal=FOCAL)
    renderer.set_gaussians(g)
    renderer.set_options(pkg.EXP_VCL, pkg.ERF_AS, 1e-9)
    renderer.set_camera_view(W, H, cam.view)
    renderer.tile_gaussians(2 / 16, 2 / 16, cam.view)
    img, _ = renderer.render(cam.position, pkg.PACK_ROUND | pkg.ALPHA_COMPUTED, want_radiance=False)
    d = np.abs(file_bytes(img).reshape(H, W, 4) - png)
    assert d.max() <= 1
    assert (d == 0).mean() >= 0.995
    # the same frame from reference-style plane arrays (camera_t::projection_plane)
    renderer.set_plane(W, H, *cam.plane())
    img2, _ = renderer.render(cam.position, pkg.PACK_ROUND | pkg.ALPHA_COMPUTED, want_radiance=False)
    np.testing.assert_array_equal(img, img2)


@pytest.mark.gpu
def test_cli_writes_the_reference_image(tmp_path):
    """The shipped binary with the command line the search found: its PNG against the reference's PNG, byte for byte
    within one u8 step."""
    from PIL import Image
    out = tmp_path / "teapot.png"
    p = subprocess.run([os.path.join(BIN, "volumetric-ray-tracer"), "-q", "-f", OBJ, "-w", "1024", "--focal-length", "1.7", "-o", str(out)],
                       cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and p.stdout.startswith("TIME: "), p.stderr
    got = np.array(Image.open(out)).astype(np.int16)
    d = np.abs(got - thesis_png())
    assert d.max() <= 1 and (d == 0).mean() >= 0.995
    out2 = tmp_path / "cube.png"
    p = subprocess.run([os.path.join(BIN, "volumetric-ray-tracer"), "-q", "-f", CUBE_OBJ, "-w", "1024", "--focal-length", "1.5", "-i", "30",
                        "-o", str(out2)], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    d = np.abs(np.array(Image.open(out2)).astype(np.int16) - thesis_png("cube"))
    assert d.max() <= 1 and (d == 0).mean() >= 0.99


@pytest.mark.gpu
def test_oracle_pin_light_subset_on_the_gpu_box(oracle):
    """A light subset of the two oracle-vs-PNG pins above, marked `gpu` so that the driver's GPU record shows the pin of the
    checker itself green next to the HIP path's (round-2 verdict): 120 + 150 sampled pixels, a few seconds of CPU."""
    for name, obj, kw, n, seed in (("teapot", OBJ, dict(focal=FOCAL), 120, 7), ("cube", CUBE_OBJ, dict(focal=CUBE_FOCAL, initial_rot=CUBE_ROT), 150, 8)):
        png = thesis_png(name)
        g = oracle.read_obj(obj)
        cam, _ = oracle.cli_camera(W, H, **kw)
        tiles = oracle.tile_gaussians(2 / 16, 2 / 16, g, oracle.camera_view(cam))
        pix = sample_pixels(png, n, seed=seed)
        img, _ = oracle.render(W, H, oracle.camera_plane(cam), cam.position[:], g, tiles, pixels=pix)
        d = np.abs(file_bytes(img[pix]) - png.reshape(-1, 4)[pix])
        assert d.max() <= 1, (name, int(d.max()))
        assert (d == 0).mean() >= 0.97, name


@pytest.mark.gpu
def test_notes_teapot_png_soft_check(pkg, renderer):
    """`notes/teapot.png` (fixture tests/golden/notes/teapot.png): 512 x 512, alpha 255 everywhere -- an opaque-alpha render
    of the teapot that the slides call a preliminary result.  tools/notes_png_fit.py (profiles/r03_notes_png_fit.md) finds
    no CLI setting of THIS revision that reproduces it: its colours follow another albedo rule, and even the channel that
    carries the plain sum of the emission terms (file channel 0 tracks this build's w radiance with correlation 0.999) is up
    to 17 steps off on a quarter of the lit pixels -- an earlier revision of the code, so no golden vector for the scalar
    modes.  What it does agree with is kept as a soft check of camera, OBJ loader and the opaque truncating pack: the lit
    silhouette to a pixel and that channel within one step on >= 95 % of the image at -c -3.44."""
    from PIL import Image
    from sgrt_amd import scene
    png = np.array(Image.open(os.path.join(GOLDEN, "notes", "teapot.png"))).astype(np.int64)
    assert png.shape == (512, 512, 4) and (png[..., 3] == 255).all()
    g = scene.read_obj(OBJ)
    cam, _ = scene.cli_camera(512, 512, camera_offset=-3.44)
    renderer.set_gaussians(g)
    renderer.set_options(pkg.EXP_LIBM, pkg.ERF_LIBM, 1e-9)          # modes 1 / 5: expf, erff
    try:
        renderer.set_camera_view(512, 512, cam.view)
        renderer.tile_gaussians(2 / 16, 2 / 16, cam.view)
        img, rad = renderer.render(cam.position, pkg.PACK_TRUNC | pkg.ALPHA_OPAQUE)
    finally:
        renderer.set_options(pkg.EXP_VCL, pkg.ERF_AS, 1e-9)
    assert ((img >> 24) == 255).all()
    w8 = np.floor(np.minimum(rad[..., 3].astype(np.float64), 1.0) * 255)
    d = np.abs(w8 - png[..., 0])
    assert (d <= 1).mean() >= 0.95 and d.mean() <= 0.3
    ys, xs = np.nonzero(png[..., :3].sum(2) > 0)
    yr, xr = np.nonzero(w8 > 0)
    assert abs(int(xs.min()) - int(xr.min())) <= 2 and abs(int(xs.max()) - int(xr.max())) <= 2
    assert abs(int(ys.min()) - int(yr.min())) <= 2 and abs(int(ys.max()) - int(yr.max())) <= 2
