"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same inputs.

Tolerances (stated per BASELINE.json north_star): float radiance within 1e-4 max-abs of the
reference semantics; packed u8 channels within 1 LSB.  With culling disabled (cull_eps = 0) the HIP
path evaluates the reference's full sum and must agree to 2e-5 (fp32 re-association only).
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

TOL = 1e-4          # north_star: "images within 1e-4 max-abs of reference"
TOL_NOCULL = 2e-5   # same sum, different association / 1-ulp rcp,exp


def channels(img):
    img = np.asarray(img).reshape(-1)
    return ((img[:, None] >> np.array([0, 8, 16, 24], np.uint32)) & 255).astype(np.int32)


def setup_scene(pkg, oracle, renderer, g, w, h, tiles_n=16, plane_from="oracle", camera_offset=-4.0, rot=0.0):
    cam, _ = oracle.cli_camera(w, h, camera_offset=camera_offset, initial_rot=rot)
    plane = oracle.camera_plane(cam)
    view = oracle.camera_view(cam)
    origin = np.array(cam.position[:], np.float32)
    renderer.set_gaussians(g)
    renderer.set_plane(w, h, *plane)
    if tiles_n:
        renderer.tile_gaussians(2.0 / tiles_n, 2.0 / tiles_n, view)
        tiles = oracle.tile_gaussians(2.0 / tiles_n, 2.0 / tiles_n, g.view(oracle.GAUSSIAN), view)
    else:
        renderer.clear_tiles()
        tiles = None
    return cam, plane, origin, tiles


@pytest.mark.parametrize("eps,tol", [(1e-9, TOL), (0.0, TOL_NOCULL)])
def test_cfg1_full_frame(pkg, oracle, renderer, eps, tol):
    """-g 4 -w 256 (BASELINE configs[0]): every pixel, tiled mode-8 semantics."""
    w = h = 256
    g = oracle.grid_scene(4)
    cam, plane, origin, tiles = setup_scene(pkg, oracle, renderer, g, w, h)
    renderer.set_options(pkg.EXP_VCL, pkg.ERF_AS, eps)
    img, rad = renderer.render(origin)
    oimg, orad = oracle.render(w, h, plane, origin, g, tiles)
    err = np.abs(rad.reshape(-1, 4) - orad).max()
    assert err <= tol, err
    assert np.abs(channels(img) - channels(oimg)).max() <= 1
    assert rad.max() > 0.15  # the image is not empty


def test_cfg1_untiled_and_scalar_packing(pkg, oracle, renderer):
    """Untiled overloads (rt.h:227-247, 315-337): opaque alpha, truncating vs rounding pack."""
    w = h = 64
    g = oracle.grid_scene(4)
    cam, plane, origin, _ = setup_scene(pkg, oracle, renderer, g, w, h, tiles_n=0)
    renderer.set_options(pkg.EXP_VCL, pkg.ERF_AS, 0.0)
    for pack in (pkg.PACK_TRUNC | pkg.ALPHA_OPAQUE, pkg.PACK_ROUND | pkg.ALPHA_OPAQUE):
        img, rad = renderer.render(origin, pack=pack)
        oimg, orad = oracle.render(w, h, plane, origin, g, None, pack=pack)
        assert np.abs(rad.reshape(-1, 4) - orad).max() <= TOL_NOCULL
        assert (img.reshape(-1) >> 24 == 0xFF).all()
        assert np.abs(channels(img) - channels(oimg)).max() <= 1


@pytest.mark.parametrize("dim,w,npix", [(16, 1024, 256), (64, 2048, 160)])
def test_grid_sparse_pixels(pkg, oracle, renderer, dim, w, npix):
    """-g 16 -w 1024 and -g 64 -w 2048 (BASELINE configs[1], [3]): full GPU frame, oracle on a seeded pixel subset."""
    h = w
    g = oracle.grid_scene(dim)
    cam, plane, origin, tiles = setup_scene(pkg, oracle, renderer, g, w, h)
    renderer.set_options(pkg.EXP_VCL, pkg.ERF_AS, 1e-9)
    img, rad = renderer.render(origin)
    rng = np.random.default_rng(1234 + dim)
    # half uniformly random, half near the brightest pixels
    bright = np.argsort(rad.reshape(-1, 4)[:, 0] + rad.reshape(-1, 4)[:, 2])[-4 * npix:]
    pix = np.unique(np.concatenate([rng.integers(0, w * h, npix // 2), rng.choice(bright, npix // 2)])).astype(np.uint32)
    _, orad = oracle.render(w, h, plane, origin, g, tiles, pixels=pix, want_image=False)
    err = np.abs(rad.reshape(-1, 4)[pix] - orad).max()
    assert err <= TOL, err
    assert orad.max() > 1e-3


def test_tile_binning_matches_reference_algorithm(pkg, oracle, renderer):
    """Device tile_gaussians == rt.cpp:29-69 restatement, exact index lists (order preserved)."""
    for g, rot in [(oracle.grid_scene(4), 0.0), (oracle.grid_scene(16), 0.0), (oracle.grid_scene(64), 0.0),
                   (oracle.read_obj(os.path.join(GOLDEN, "test-objects", "monkey.obj")), 0.0),
                   (oracle.read_obj(os.path.join(GOLDEN, "test-objects", "teapot.obj")), 33.0)]:
        cam, _ = oracle.cli_camera(256, 256, initial_rot=rot)
        view = oracle.camera_view(cam)
        renderer.set_gaussians(g)
        renderer.tile_gaussians(2 / 16, 2 / 16, view)
        tiles = oracle.tile_gaussians(2 / 16, 2 / 16, g, view)
        counts = renderer.tile_counts()
        assert counts.shape == (16, 16)
        np.testing.assert_array_equal(counts.ravel(), np.diff(tiles["offsets"]))
        for t in (0, 17, 100, 135, 255):
            np.testing.assert_array_equal(renderer.tile_indices(t),
                                          tiles["indices"][tiles["offsets"][t]:tiles["offsets"][t + 1]])


def test_device_binning_matches_hand_derived_tile_counts(pkg, oracle, renderer):
    """Device tile_gaussians against tests/golden/tile_counts.json: the per-axis counts SURVEY.md 8(c) derived by hand
    from rt.cpp:58-59 for the three grid configs -- a known answer that is not this repository's output."""
    import json
    gold = json.load(open(os.path.join(GOLDEN, "tile_counts.json")))
    cam, _ = oracle.cli_camera(64, 64)
    for name, sc in gold["scenes"].items():
        renderer.set_gaussians(oracle.grid_scene(sc["grid"]))
        renderer.tile_gaussians(gold["tw"], gold["tw"], oracle.camera_view(cam))
        pa = np.array(sc["per_axis"])
        np.testing.assert_array_equal(renderer.tile_counts(), np.outer(pa, pa), err_msg=name)


def test_host_tiles_equal_device_tiles(pkg, oracle, renderer):
    """vrt_hip_set_tiles (caller-made tiles_t) and vrt_hip_tile_gaussians give the same image."""
    w = h = 128
    g = oracle.grid_scene(8)
    cam, plane, origin, tiles = setup_scene(pkg, oracle, renderer, g, w, h)
    img_a, rad_a = renderer.render(origin)
    renderer.set_tiles(tiles)
    img_b, rad_b = renderer.render(origin)
    np.testing.assert_array_equal(img_a, img_b)
    np.testing.assert_array_equal(rad_a, rad_b)


def test_camera_basis_mode_matches_plane_mode(pkg, oracle, renderer):
    """In-kernel ray generation (closed form of camera.cpp:52,60-69) vs the plane arrays."""
    from sgrt_amd import scene
    w = h = 128
    g = oracle.grid_scene(8)
    for rot in (0.0, 47.0):
        cam, plane, origin, tiles = setup_scene(pkg, oracle, renderer, g, w, h, rot=rot)
        _, rad_p = renderer.render(origin)
        pc, _ = scene.cli_camera(w, h, initial_rot=rot)
        renderer.set_camera(w, h, pc.position, pc.right, pc.up, pc.front, 1.0)
        _, rad_c = renderer.render(pc.position)
        assert np.abs(rad_p - rad_c).max() <= 2e-5


def test_view_mode_rays_are_the_reference_rays(pkg, oracle, renderer):
    """vrt_hip_set_camera_view: in-kernel rays from inverse(view), evaluated like camera.cpp:60-69 with glm's arithmetic.
    Given the same view matrix they are bit for bit the rays of the plane arrays, so the images are IDENTICAL -- also
    where the closed-form basis mode drifts (small sigma far from the camera: |oc|^2 - mubar^2 amplifies the last bit of
    a ray direction)."""
    rng = np.random.default_rng(3)
    n = 400
    g = oracle.gaussians(rng.uniform(0, 1, size=(n, 4)), rng.normal(size=(n, 3)) * 0.6, rng.uniform(0.03, 0.1, n), rng.uniform(0.2, 2.0, n))
    for (w, h, rot, tiles_n) in [(96, 64, 33.0, 4), (128, 128, 210.0, 16), (100, 60, 0.0, 0)]:
        cam, plane, origin, tiles = setup_scene(pkg, oracle, renderer, g, w, h, tiles_n=tiles_n, rot=rot)
        renderer.set_options(pkg.EXP_VCL, pkg.ERF_AS, 1e-9)
        img_p, rad_p = renderer.render(origin)
        renderer.set_camera_view(w, h, oracle.camera_view(cam))
        img_v, rad_v = renderer.render(origin)
        assert rad_p.max() > 0.05
        np.testing.assert_array_equal(rad_v, rad_p)
        np.testing.assert_array_equal(img_v, img_p)


def test_transmittance_three_gaussians(pkg, oracle, renderer):
    """Scene and sweep of the reference's tests/transmittance.cpp:9-31."""
    g = oracle.gaussians([[0, 1, 0, .1], [0, 0, 1, .7], [1, 0, 0, 1]], [[.3, .3, .5], [-.3, -.3, 0], [0, 0, 2]],
                         [.1, .4, .75], [2., .7, 1.])
    o, n = (0, 0, -5), (0, 0, 1)
    k = np.arange(-6, 6.0001, 0.1, dtype=np.float32)
    s = (np.float32(7.0) + k * np.float32(0.75)).astype(np.float32)
    renderer.set_gaussians(g)
    for ek, rk in [(pkg.EXP_LIBM, pkg.ERF_LIBM), (pkg.EXP_VCL, pkg.ERF_AS), (pkg.EXP_LIBM, pkg.ERF_AS)]:
        renderer.set_options(ek, rk, 1e-9)
        T = renderer.transmittance(o, n, s)
        To = oracle.transmittance(o, n, s, g, ek, rk)
        assert np.abs(T - To).max() <= 2e-6, (ek, rk)
    # the reference test's own property: analytic T ~ numeric integral (fast_exp error ~ 3e-2 relative)
    renderer.set_options(pkg.EXP_LIBM, pkg.ERF_LIBM, 1e-9)
    T = renderer.transmittance(o, n, s)
    Ts = renderer.transmittance_step(o, n, s, 0.01)
    assert np.abs(T - Ts).max() < 0.05
    D = renderer.density(np.stack([np.zeros_like(s), np.zeros_like(s), -5 + s], 1))
    Do = np.array([oracle.density((0, 0, -5 + float(v)), g) for v in s], np.float32)
    assert np.abs(D - Do).max() <= 1e-6
    renderer.set_options(pkg.EXP_VCL, pkg.ERF_AS, 1e-9)


def test_broadcast_transmittance_rays(pkg, oracle, renderer):
    """vrt_hip_transmittance_rays == broadcast_transmittance (rt.h:102-127): every ray its own origin, direction and
    sample point -- against the oracle's transmittance per ray, and equal to the one-ray entry point."""
    g = oracle.gaussians([[0, 1, 0, .1], [0, 0, 1, .7], [1, 0, 0, 1]], [[.3, .3, .5], [-.3, -.3, 0], [0, 0, 2]],
                         [.1, .4, .75], [2., .7, 1.])
    rng = np.random.default_rng(11)
    n = 130                                     # not a multiple of the wave size
    origins = (rng.normal(size=(n, 3)) * 0.3 + np.array([0, 0, -5])).astype(np.float32)
    d = rng.normal(size=(n, 3)) * 0.08 + np.array([0, 0, 1])
    dirs = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    s = rng.uniform(-1.0, 9.0, n).astype(np.float32)        # incl. sample points behind the origin (T > 1, rt.h:121-126)
    renderer.set_gaussians(g)
    for ek, rk in [(pkg.EXP_VCL, pkg.ERF_AS), (pkg.EXP_LIBM, pkg.ERF_LIBM)]:
        renderer.set_options(ek, rk, 1e-9)
        T = renderer.transmittance_rays(origins, dirs, s)
        To = np.array([oracle.transmittance(o, v, [sv], g, ek, rk)[0] for o, v, sv in zip(origins, dirs, s)], np.float32)
        assert np.abs(T - To).max() <= 2e-6, (ek, rk)
        for i in (0, 64, 129):
            assert T[i] == renderer.transmittance(origins[i], dirs[i], s[i:i + 1])[0]
    assert (To < 0.9).any() and (To > 0.99).any()          # the sample spans attenuated and free rays
    assert renderer.transmittance_rays(origins[:0], dirs[:0], s[:0]).size == 0
    renderer.set_options(pkg.EXP_VCL, pkg.ERF_AS, 1e-9)


def test_radiance_arbitrary_rays(pkg, oracle, renderer):
    """vrt_hip_radiance == radiance / broadcast_radiance (rt.h:146-223) incl. the w (alpha) component."""
    g = oracle.gaussians([[0, 1, 0, .1], [0, 0, 1, .7], [1, 0, 0, 1]], [[.3, .3, .5], [-.3, -.3, 0], [0, 0, 2]],
                         [.1, .4, .75], [2., .7, 1.])
    rng = np.random.default_rng(7)
    origins = (rng.normal(size=(100, 3)) * 0.3 + np.array([0, 0, -5])).astype(np.float32)
    d = rng.normal(size=(100, 3)) * 0.08 + np.array([0, 0, 1])
    dirs = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    renderer.set_gaussians(g)
    renderer.set_options(pkg.EXP_VCL, pkg.ERF_AS, 0.0)
    out = renderer.radiance(origins, dirs)
    ref = np.stack([oracle.radiance(o, n, g, oracle.EXP_VCL, oracle.ERF_AS) for o, n in zip(origins, dirs)])
    assert np.abs(out - ref).max() <= TOL_NOCULL
    assert ref.max() > 0.05
    renderer.set_options(pkg.EXP_VCL, pkg.ERF_AS, 1e-9)


def test_device_erf_exp_against_reference_tables(pkg, oracle, renderer):
    """Device approximations vs golden tables produced by the REAL reference approx.cpp (tests/golden)."""
    gold = np.load(os.path.join(GOLDEN, "approx_ref.npz"))
    x, xe = gold["erf_x"], gold["exp_x"]
    # A&S: device uses a 1-ulp reciprocal; the reference scalar a divide, its SIMD twin rcp14 (2^-14)
    assert np.abs(renderer.eval_erf(pkg.ERF_AS, x) - gold["as_erf"]).max() <= 5e-7
    assert np.abs(renderer.eval_erf(pkg.ERF_AS, x) - gold["simd_as_erf"]).max() <= 4e-5
    assert np.abs(renderer.eval_erf(pkg.ERF_SPLINE, x) - gold["spline_erf"]).max() <= 5e-7
    assert np.abs(renderer.eval_erf(pkg.ERF_SPLINE_MIRROR, x) - gold["spline_erf_mirror"]).max() <= 5e-7
    assert np.abs(renderer.eval_erf(pkg.ERF_TAYLOR, x) - gold["taylor_erf"]).max() <= 2e-6
    assert np.abs(renderer.eval_erf(pkg.ERF_LIBM, x) - gold["libm_erf"]).max() <= 5e-7
    v = renderer.eval_exp(pkg.EXP_VCL, xe)
    assert (np.abs(v - gold["vcl_exp"]) <= 2.5e-7 * np.abs(gold["vcl_exp"])).all()   # ~2 ulp
    assert (renderer.eval_exp(pkg.EXP_VCL, np.array([-87.4, -100.0], np.float32)) == 0).all()
    # fast_exp builds the float's bit pattern from a*x+b, a cancelling sum of two ~1e9 terms: the reference binary
    # (-ffast-math => fused multiply-add) and the unfused source semantics differ by the product's rounding
    # (<= 64 integer steps = 8e-6 relative); the device follows the unfused source, bit-exact with the oracle
    fe = renderer.eval_exp(pkg.EXP_FAST, xe)
    assert (np.abs(fe - gold["fast_exp"]) <= 1e-5 * np.abs(gold["fast_exp"])).all()
    np.testing.assert_array_equal(fe, oracle.map_scalar("oracle_fast_exp", xe))
    assert np.abs(renderer.eval_exp(pkg.EXP_SPLINE, xe) - gold["spline_exp"]).max() <= 1e-7


def test_device_erf_exp_against_the_reference_held_tables(pkg, oracle, renderer):
    """vrt_hip_eval_erf / _exp against the tables the reference itself holds (thesis/plots/cmp_{erf,exp}_approx.tex = the output of
    tests/accuracy.cpp:9-58; fixture tests/golden/thesis_plots.npz): every series on the reference's own float-accumulated grids."""
    plots = np.load(os.path.join(GOLDEN, "thesis_plots.npz"))
    # device erf: the A&S rational with a 1-ulp reciprocal instead of the scalar reference's divide; the others as in the source
    for kind, series, bound in ((pkg.ERF_AS, "abramowitz_stegun", 3e-7), (pkg.ERF_SPLINE, "spline", 3e-7),
                                (pkg.ERF_SPLINE_MIRROR, "spline_mirror", 3e-7), (pkg.ERF_TAYLOR, "taylor", 2e-6),
                                (pkg.ERF_LIBM, "std_erf", 3e-7)):
        x, y = plots[f"cmp_erf_approx/{series}/x"], plots[f"cmp_erf_approx/{series}/y"]
        assert len(x) == 121
        err = np.abs(renderer.eval_erf(kind, x) - y).max()
        assert err <= bound, (series, err)
    xe = plots["cmp_exp_approx/vcl/x"]
    v, y = renderer.eval_exp(pkg.EXP_VCL, xe), plots["cmp_exp_approx/vcl/y"]
    assert (np.abs(v - y) <= 2.5e-7 * np.abs(y)).all()                       # ~2 ulp: v_exp_f32-free restatement of exp_f
    assert (v.view(np.uint32) == y.view(np.uint32)).mean() >= 0.5
    fe, fy = renderer.eval_exp(pkg.EXP_FAST, xe), plots["cmp_exp_approx/fast/y"]
    assert (np.abs(fe - fy) <= 1e-5 * np.abs(fy)).all()
    se, sy = renderer.eval_exp(pkg.EXP_SPLINE, xe), plots["cmp_exp_approx/spline/y"]
    assert np.abs(se - sy).max() <= 1e-7
    le, ly = renderer.eval_exp(pkg.EXP_LIBM, xe), plots["cmp_exp_approx/std_exp/y"]
    assert (np.abs(le - ly) <= 2.5e-7 * np.abs(ly)).all()


def test_obj_scene_sparse(pkg, oracle, renderer):
    """-f monkey.obj at 512^2, rotated view: irregular per-ray Gaussian counts; oracle on sparse pixels."""
    w = h = 512
    g = oracle.read_obj(os.path.join(GOLDEN, "test-objects", "monkey.obj"))
    cam, plane, origin, tiles = setup_scene(pkg, oracle, renderer, g, w, h, rot=20.0)
    renderer.set_options(pkg.EXP_VCL, pkg.ERF_AS, 1e-9)
    img, rad = renderer.render(origin)
    rng = np.random.default_rng(99)
    lum = rad.reshape(-1, 4)[:, :3].sum(1)
    cand = np.nonzero(lum > 0.05)[0]
    pix = np.unique(np.concatenate([rng.choice(cand, 40), rng.integers(0, w * h, 8)])).astype(np.uint32)
    _, orad = oracle.render(w, h, plane, origin, g, tiles, pixels=pix, want_image=False)
    err = np.abs(rad.reshape(-1, 4)[pix] - orad).max()
    assert err <= TOL, err


def test_empty_and_edge_inputs(pkg, oracle, renderer):
    """Empty scene, a single Gaussian, a Gaussian behind the camera, non-multiple-of-8 tile sizes."""
    w = h = 40  # tiles of 10x10 pixels with --tiles 4: blocks are ragged
    cam, _ = oracle.cli_camera(w, h)
    plane = oracle.camera_plane(cam)
    origin = np.array(cam.position[:], np.float32)
    renderer.set_plane(w, h, *plane)
    renderer.set_options(pkg.EXP_VCL, pkg.ERF_AS, 0.0)
    # empty
    g0 = oracle.grid_scene(4)[:0]
    renderer.set_gaussians(g0)
    renderer.clear_tiles()
    img, rad = renderer.render(origin)
    assert (rad == 0).all() and (img == 0).all()
    # one Gaussian in front, one behind the camera (T > 1 behind: rt.h has no clamp)
    g = oracle.gaussians([[1, .5, .25, 1], [.2, .4, .6, 1]], [[0, 0, 1], [0.1, 0, -6]], [.3, .5], [1., 2.])
    renderer.set_gaussians(g)
    renderer.tile_gaussians(2 / 4, 2 / 4, oracle.camera_view(cam))
    tiles = oracle.tile_gaussians(2 / 4, 2 / 4, g, oracle.camera_view(cam))
    img, rad = renderer.render(origin)
    oimg, orad = oracle.render(w, h, plane, origin, g, tiles)
    assert np.abs(rad.reshape(-1, 4) - orad).max() <= TOL_NOCULL
    assert np.abs(channels(img) - channels(oimg)).max() <= 1
    renderer.clear_tiles()
    _, rad = renderer.render(origin)
    _, orad = oracle.render(w, h, plane, origin, g, None)
    assert np.abs(rad.reshape(-1, 4) - orad).max() <= TOL_NOCULL
    renderer.set_options(pkg.EXP_VCL, pkg.ERF_AS, 1e-9)


def test_alternate_approximations_render(pkg, oracle, renderer):
    """Exp/Erf template arguments of the reference as kernel variants."""
    w = h = 64
    g = oracle.grid_scene(4)
    cam, plane, origin, tiles = setup_scene(pkg, oracle, renderer, g, w, h)
    for ek, rk, tol in [(pkg.EXP_LIBM, pkg.ERF_LIBM, 2e-5), (pkg.EXP_FAST, pkg.ERF_AS, 2e-5),
                        (pkg.EXP_SPLINE, pkg.ERF_AS, 2e-5), (pkg.EXP_VCL, pkg.ERF_SPLINE, 2e-5),
                        # spline_erf_mirror jumps by 0.107 at x = 0 (approx.cpp:45-55: +-0.0537 either side) and
                        # taylor_erf by 4.7e-3 at |x| = 2 (approx.cpp:75-76): a sample point within float noise of
                        # such a jump may land on either side, so these two variants get the jump x weight as
                        # tolerance instead of the re-association tolerance
                        (pkg.EXP_VCL, pkg.ERF_SPLINE_MIRROR, 2e-2),
                        (pkg.EXP_VCL, pkg.ERF_TAYLOR, 5e-3)]:
        renderer.set_options(ek, rk, 0.0)
        _, rad = renderer.render(origin)
        _, orad = oracle.render(w, h, plane, origin, g, tiles, exp_kind=ek, erf_kind=rk, want_image=False)
        assert np.abs(rad.reshape(-1, 4) - orad).max() <= tol, (ek, rk)
    renderer.set_options(pkg.EXP_VCL, pkg.ERF_AS, 1e-9)


def test_sharded_render_assembles_to_full_frame(pkg, oracle, renderer):
    """Tile shards of 2 and 3 'ranks' rendered on this GPU and assembled == single-GPU frame."""
    import torch
    w = h = 128
    g = oracle.grid_scene(8)
    cam, plane, origin, tiles = setup_scene(pkg, oracle, renderer, g, w, h)
    renderer.set_shard(0, 1)
    full, _ = renderer.render(origin, want_radiance=False)
    pack = pkg.PACK_ROUND | pkg.ALPHA_COMPUTED
    st = torch.cuda.current_stream().cuda_stream
    for world in (2, 3):
        shards = []
        for rank in range(world):
            renderer.set_shard(rank, world)
            n = renderer.shard_pixels()
            buf = torch.zeros(n, dtype=torch.int32, device="cuda")
            renderer.render_shard_device(origin, pack, buf.data_ptr(), st)
            torch.cuda.synchronize()
            shards.append(buf)
        gathered = torch.cat(shards)
        out = torch.zeros(w * h, dtype=torch.int32, device="cuda")
        renderer.assemble_shards_device(gathered.data_ptr(), out.data_ptr(), st)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(out.cpu().numpy().view(np.uint32).reshape(h, w), full)
        # several frames per gather: [rank][frame][shard], frame 1 of 3 holds the image, the others garbage
        n, frames = shards[0].numel(), 3
        batched = torch.full((world, frames, n), 0x7F7F7F7F, dtype=torch.int32, device="cuda")
        for rank in range(world):
            batched[rank, 1] = shards[rank]
        out.zero_()
        renderer.assemble_shards_device(batched.data_ptr() + 4 * n, out.data_ptr(), st, rank_stride_px=frames * n)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(out.cpu().numpy().view(np.uint32).reshape(h, w), full)
        with pytest.raises(RuntimeError):
            renderer.assemble_shards_device(batched.data_ptr(), out.data_ptr(), st, rank_stride_px=n - 1)
    renderer.set_shard(0, 1)


def _blob_scene(oracle, n, seed, spread=0.15, sigma=(0.25, 0.45), mag=(0.01, 0.05)):
    """n wide, faint Gaussians around the view axis: every ray sees all of them (lists as long as the scene)."""
    rng = np.random.default_rng(seed)
    mu = rng.normal(size=(n, 3)) * spread + np.array([0, 0, 1.0])
    return oracle.gaussians(rng.uniform(0.1, 1, size=(n, 4)), mu, rng.uniform(*sigma, n), rng.uniform(*mag, n))


@pytest.mark.parametrize("n,w,what", [(300, 16, "per-ray lists > 48: blocks handed to the 16-wave kernel"),
                                      (1100, 8, "block candidates > 1024: LDS overflow, streamed list"),
                                      (4200, 4, "cell list > 4096: cell slot overflow, tile list used")])
def test_capacity_overflow_paths(pkg, oracle, renderer, n, w, what):
    """Every fixed-capacity structure of the kernels has a correct (slower) way out; heterogeneous sigma/magnitude."""
    g = _blob_scene(oracle, n, 5 + n)
    cam, plane, origin, tiles = setup_scene(pkg, oracle, renderer, g, w, w, tiles_n=1)
    renderer.set_options(pkg.EXP_VCL, pkg.ERF_AS, 1e-9)
    _, orad = oracle.render(w, w, plane, origin, g, tiles, want_image=False)
    scale = max(1.0, float(orad.max()))
    assert orad.max() > 0.05
    renderer.enable_stats(True)
    try:
        renderer.set_table_step(0.0)                       # the exact kernels' own ways out
        img, rad = renderer.render(origin)
        st = renderer.stats()
        assert np.abs(rad.reshape(-1, 4) - orad).max() <= TOL * scale, what
        if n == 300:
            assert st["dense_blocks"] > 0
        else:
            assert st["overflow_blocks"] > 0
        renderer.set_table_step(pkg.TABLE_STEP_DEFAULT)    # table mode: up to 2048 survivors per block, beyond that declined
        img, rad = renderer.render(origin)
        st = renderer.stats()
        assert np.abs(rad.reshape(-1, 4) - orad).max() <= TOL * scale, what + " (table mode)"
        assert st["table_blocks"] + st["table_declined"] == st["dense_blocks"] > 0
        assert (st["table_declined"] > 0) == (n > 2048)
    finally:
        renderer.enable_stats(False)


def test_erf_saturation_thresholds(pkg, renderer):
    """The dense kernel skips terms whose Erf is EXACTLY +-1: the thresholds it uses must really saturate."""
    sat = {pkg.ERF_AS: 5.5, pkg.ERF_LIBM: 4.2, pkg.ERF_SPLINE: 3.1, pkg.ERF_SPLINE_MIRROR: 2.9, pkg.ERF_TAYLOR: 2.0}
    for kind, s in sat.items():
        x = np.concatenate([np.linspace(s, s + 3, 4001), np.geomspace(s, 1e6, 2000)]).astype(np.float32)
        x = x[x >= np.float32(s)]
        assert (renderer.eval_erf(kind, x) == 1.0).all(), kind
        assert (renderer.eval_erf(kind, -x) == -1.0).all(), kind
    # and the A&S threshold is tight: just below it the value is not yet 1
    assert renderer.eval_erf(pkg.ERF_AS, np.array([5.3], np.float32))[0] < 1.0


@pytest.mark.parametrize("w,h,tiles_n", [(100, 100, 5), (96, 64, 3), (100, 100, 16), (72, 40, 2), (33, 100, 5), (64, 256, 5)])
def test_ragged_geometry(pkg, oracle, renderer, w, h, tiles_n):
    """Tile sizes that are not multiples of 8 or 32, non-square images, and the reference's truncated tile size
    (rt.h:348-349) with its row stride tile_w*tiles_w != width (rt.h:364-365): same pixels, same values.  In the last
    two the stride differs so much that a tile's rows wrap around the image edge several times."""
    g = oracle.grid_scene(6)
    cam, plane, origin, tiles = setup_scene(pkg, oracle, renderer, g, w, h, tiles_n=tiles_n)
    renderer.set_options(pkg.EXP_VCL, pkg.ERF_AS, 1e-9)
    img, rad = renderer.render(origin)
    tile_w, tile_h = int(np.float32(w) * tiles["tw"] / np.float32(2)), int(np.float32(h) * tiles["th"] / np.float32(2))
    stride = tile_w * tiles["w"]
    n_written = stride * tile_h * tiles["h"]
    pix = np.arange(min(n_written, w * h), dtype=np.uint32)
    oimg, orad = oracle.render(w, h, plane, origin, g, tiles, pixels=pix)
    assert np.abs(rad.reshape(-1, 4)[pix] - orad).max() <= TOL
    assert np.abs(channels(img.reshape(-1)[pix]) - channels(oimg[pix])).max() <= 1
    assert orad.max() > 0.01
    # pixels the reference never writes stay cleared here
    assert (img.reshape(-1)[len(pix):] == 0).all()


def test_error_reporting(pkg, oracle, renderer):
    """Status codes + messages instead of the reference's _exit(1) (definitions.h:23-30)."""
    import ctypes as C
    L = pkg.lib()
    g = oracle.grid_scene(2)
    renderer.set_gaussians(g)
    h = renderer._h
    off = (C.c_uint32 * 2)(0, 1)
    bad = (C.c_uint32 * 1)(99)
    assert L.vrt_hip_set_tiles(h, 2.0, 2.0, 1, 1, off, bad) == -1
    assert b"out of range" in L.vrt_hip_last_error(h)
    assert L.vrt_hip_set_tiles(h, -1.0, 2.0, 1, 1, off, bad) == -1
    assert L.vrt_hip_set_options(h, 9, 1, C.c_float(0.0)) == -1
    assert L.vrt_hip_set_options(h, pkg.EXP_FAST, pkg.ERF_TAYLOR, C.c_float(0.0)) == -1   # pair not instantiated
    assert L.vrt_hip_set_shard(h, 3, 2) == -1
    fresh = pkg.Renderer(0)
    try:
        with pytest.raises(pkg.VrtHipError, match="set_plane/set_camera"):
            fresh.render((0, 0, -4))
        fresh.set_gaussians(g[:0])
        cam, _ = oracle.cli_camera(16, 16)
        fresh.set_plane(16, 16, *oracle.camera_plane(cam))
        img, rad = fresh.render((0, 0, -4))            # empty scene renders an empty image
        assert (img == 0).all()
    finally:
        fresh.close()
    renderer.clear_tiles()


def test_animation_loop_matches_fresh_contexts(pkg, oracle, renderer):
    """The CLI's orbit loop (main.cpp:257-334): re-tiling and re-rendering from a moving camera through ONE context
    (cached tables, double-buffered queues, launch feedback) must equal an independent render of every pose,
    including when the dense kernel gets dropped and picked up again."""
    import torch
    from sgrt_amd import scene
    w = h = 128
    g = oracle.read_obj(os.path.join(GOLDEN, "test-objects", "cube.obj"))
    renderer.set_gaussians(g)
    renderer.set_options(pkg.EXP_VCL, pkg.ERF_AS, 1e-9)
    cam, angle = scene.cli_camera(w, h)
    out = torch.zeros(w * h, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    pack = pkg.PACK_ROUND | pkg.ALPHA_COMPUTED
    frames = []
    for k in range(14):
        renderer.set_camera(w, h, cam.position, cam.right, cam.up, cam.front, 1.0)
        renderer.frame_call(2 / 16, 2 / 16, cam.view, cam.position, pack)(out.data_ptr(), st)
        torch.cuda.synchronize()
        frames.append((out.cpu().numpy().view(np.uint32).copy(), cam.position.copy(), cam.right.copy(), cam.up.copy(),
                       cam.front.copy(), cam.view.copy()))
        cam.orbit(25.0); angle = np.float32(angle - np.float32(25.0)); cam.turn(angle, 0.0)
    fresh = pkg.Renderer(0)
    try:
        fresh.set_gaussians(g)
        for k in (0, 5, 9, 13):
            img, pos, right, up, front, view = frames[k]
            fresh.set_camera(w, h, pos, right, up, front, 1.0)
            fresh.tile_gaussians(2 / 16, 2 / 16, view)
            ref, _ = fresh.render(pos, want_radiance=False)
            np.testing.assert_array_equal(img, ref.reshape(-1))
            assert (img >> 24).max() > 100
    finally:
        fresh.close()


def test_arbitrary_plane_arrays(pkg, oracle, renderer):
    """The plane arrays are caller data: a non-pinhole ray pattern (barrel-distorted plane) must still render
    exactly -- the tile/cell cone culls need pinhole rays and switch themselves off, the per-block cull works
    from the actual lane rays."""
    w = h = 96
    g = oracle.grid_scene(8)
    cam, _ = oracle.cli_camera(w, h)
    xs, ys, zs = (a.copy() for a in oracle.camera_plane(cam))
    r2 = xs * xs + ys * ys
    xs = (xs * (1 + 0.25 * r2)).astype(np.float32)        # not an affine function of (row, column) any more
    ys = (ys * (1 + 0.25 * r2)).astype(np.float32)
    zs = (zs + 0.1 * r2).astype(np.float32)
    origin = np.array(cam.position[:], np.float32)
    view = oracle.camera_view(cam)
    renderer.set_gaussians(g)
    renderer.set_plane(w, h, xs, ys, zs)
    renderer.set_options(pkg.EXP_VCL, pkg.ERF_AS, 1e-9)
    for tiles_n in (4, 0):
        if tiles_n:
            renderer.tile_gaussians(2 / tiles_n, 2 / tiles_n, view)
            tiles = oracle.tile_gaussians(2 / tiles_n, 2 / tiles_n, g, view)
        else:
            renderer.clear_tiles()
            tiles = None
        img, rad = renderer.render(origin)
        oimg, orad = oracle.render(w, h, (xs, ys, zs), origin, g, tiles,
                                   pack=oracle.PACK_ROUND | (oracle.ALPHA_COMPUTED if tiles_n else oracle.ALPHA_OPAQUE))
        assert np.abs(rad.reshape(-1, 4) - orad).max() <= TOL
        assert orad.max() > 0.05
    renderer.clear_tiles()


def test_img_error_experiment(pkg, oracle, renderer):
    """The reference's tests/img-error.cpp as a parity test: 16x16 grid, sigma .25, magnitude 3, origin 0, identity view,
    tile_gaussians(1/8, 1/8); mean squared u8-channel error of the SIMD variants against the scalar expf/erff image
    (img-error.cpp:18-60).  Run at 64^2 instead of 256^2 so that the CPU oracle finishes in seconds; the three MSEs
    measured on the GPU images must equal the ones measured on the oracle's images."""
    w = h = 64
    g = oracle.grid_scene(16)
    g["sigma"] = 0.25
    g["magnitude"] = 3.0
    cam = oracle.camera((0.0, 0.0, 0.0), w, h)                      # camera_create_info_t{} defaults
    plane = oracle.camera_plane(cam)
    origin = np.zeros(3, np.float32)
    view = np.eye(4, dtype=np.float32).ravel()                     # glm::mat4(1.f)
    tiles = oracle.tile_gaussians(1 / 8, 1 / 8, g, view)
    renderer.set_gaussians(g)
    renderer.set_plane(w, h, *plane)
    renderer.tile_gaussians(1 / 8, 1 / 8, view)

    def both(ek, rk, pack):
        renderer.set_options(ek, rk, 1e-9)
        gi, _ = renderer.render(origin, pack=pack, want_radiance=False)
        oi, _ = oracle.render(w, h, plane, origin, g, tiles, exp_kind=ek, erf_kind=rk, pack=pack)
        return channels(gi)[:, :3] / 255.0, channels(oi)[:, :3] / 255.0

    scalar = pkg.PACK_TRUNC | pkg.ALPHA_OPAQUE
    simd = pkg.PACK_ROUND | pkg.ALPHA_COMPUTED
    ref_g, ref_o = both(pkg.EXP_LIBM, pkg.ERF_LIBM, scalar)
    assert np.abs(ref_g - ref_o).max() <= 1 / 255 + 1e-9 and ref_o.max() > 0.5
    for ek, rk in [(pkg.EXP_VCL, pkg.ERF_AS), (pkg.EXP_FAST, pkg.ERF_AS)]:
        var_g, var_o = both(ek, rk, simd)
        mse_g = ((ref_g - var_g) ** 2).sum(1).mean()
        mse_o = ((ref_o - var_o) ** 2).sum(1).mean()
        assert abs(mse_g - mse_o) <= 0.05 * mse_o + 2e-6, (ek, rk, mse_g, mse_o)
        assert mse_o < 1e-3      # the approximations are image-level accurate (the thesis's conclusion)
    renderer.set_options(pkg.EXP_VCL, pkg.ERF_AS, 1e-9)
    renderer.clear_tiles()


def test_work_queues_and_scheduling_do_not_change_the_image(pkg, oracle, renderer, monkeypatch):
    """The one-wave kernel hands blocks beyond its grid size out through work counters; which wave shades a block must
    not show.  One persistent wave per CU (256 waves for 1024 shaded blocks: three quarters of them come from the queues)
    against the default grid, repeated so that both counter sets are used; bit-identical images and radiances."""
    w = h = 1024
    g = oracle.grid_scene(16)
    cam, plane, origin, tiles = setup_scene(pkg, oracle, renderer, g, w, h)
    ref_img, ref_rad = renderer.render(origin)
    assert (ref_img >> 24).max() > 0
    monkeypatch.setenv("VRT_HIP_RENDER_WAVES", "1")
    r1 = pkg.Renderer(0)
    try:
        setup_scene(pkg, oracle, r1, g, w, h)
        for _ in range(3):
            img, rad = r1.render(origin)
            np.testing.assert_array_equal(img, ref_img)
            np.testing.assert_array_equal(rad, ref_rad)
    finally:
        r1.close()
    # Round-3 advisor finding: the mask of queues a wave has seen run out was 32 bits wide for 64 queues.  One, two and three waves for
    # the whole frame (VRT_HIP_RENDER_GRID): a wave works its way through all 64 queues, the lowest relative index first, so at the end
    # of its launch only queues with relative index >= 32 still hold entries -- and with two or three waves they race for the last ones
    # (the bits of queues 32..63 are set).  Every block must still be shaded exactly once.
    for grid in ("1", "2", "3"):
        monkeypatch.setenv("VRT_HIP_RENDER_GRID", grid)
        r2 = pkg.Renderer(0)
        try:
            setup_scene(pkg, oracle, r2, g, w, h)
            for _ in range(2):
                img, rad = r2.render(origin)
                np.testing.assert_array_equal(img, ref_img, err_msg=f"grid {grid}")
                np.testing.assert_array_equal(rad, ref_rad, err_msg=f"grid {grid}")
        finally:
            r2.close()


def test_cull_threshold_scales_with_the_scene_size(pkg, oracle, renderer):
    """50,000 Gaussians (a random cloud of narrow ones): the tile-level threshold is cull_eps * 4096 / N, so that what a ray can
    lose at that level -- up to N dropped Gaussians -- stays where it is for N = 4096 (round-2 verdict: the bound 3 eps (N + 4096)
    passed the 1e-4 tolerance from N ~ 29k at a fixed threshold).  Sparse pixels against the oracle, and the whole frame against
    the full sum (cull_eps = 0) of the same context."""
    rng = np.random.default_rng(50000)
    n, w = 50000, 256
    mu = rng.uniform(-1.0, 1.0, size=(n, 3)) * np.array([1.0, 1.0, 0.6]) + np.array([0, 0, 0.2])
    g = oracle.gaussians(rng.uniform(0.2, 1, size=(n, 4)), mu, rng.uniform(0.008, 0.02, n), rng.uniform(0.3, 1.5, n))
    cam, plane, origin, tiles = setup_scene(pkg, oracle, renderer, g, w, w)
    renderer.set_options(pkg.EXP_VCL, pkg.ERF_AS, 1e-9)
    try:
        img, rad = renderer.render(origin)
        # (oracle pixels from the four central tiles: the reference's tile test has a slack of |x| (rt.cpp:58-59), so an outer
        # tile of this scene holds tens of thousands of Gaussians and one of its pixels costs the oracle 5 N_t^2 ~ 1e9 terms)
        lum = rad.reshape(w, w, 4)[:, :, :3].sum(2)
        c0, c1 = w // 2 - w // 16, w // 2 + w // 16
        ys, xs = np.nonzero(lum[c0:c1, c0:c1] > 0.2 * lum[c0:c1, c0:c1].max())
        sel = rng.choice(len(ys), 12, replace=False)
        pix = np.unique(np.concatenate([(ys[sel] + c0) * w + xs[sel] + c0, rng.integers(c0, c1, 4) * w + rng.integers(c0, c1, 4)])).astype(np.uint32)
        _, orad = oracle.render(w, w, plane, origin, g, tiles, pixels=pix, want_image=False)
        assert orad.max() > 0.05
        assert np.abs(rad.reshape(-1, 4)[pix] - orad).max() <= TOL
        renderer.set_options(pkg.EXP_VCL, pkg.ERF_AS, 0.0)
        renderer.set_table_step(0.0)
        _, full = renderer.render(origin)
        renderer.set_options(pkg.EXP_VCL, pkg.ERF_AS, 1e-9)
        _, exact = renderer.render(origin)
        assert np.abs(exact.astype(np.float64) - full).max() <= 2.5e-5          # the cull's whole budget, at twelve times 4096 Gaussians
    finally:
        renderer.set_options(pkg.EXP_VCL, pkg.ERF_AS, 1e-9)
        renderer.set_gaussians(oracle.grid_scene(4))


@pytest.mark.parametrize("name,w,rot", [("g64", 1024, 47.0), ("teapot", 512, 0.0), ("monkey", 512, 20.0), ("cube", 256, 30.0)])
def test_per_tile_cull_slack_keeps_the_error_bound(pkg, oracle, renderer, monkeypatch, name, w, rot):
    """Level-wise cull thresholds (TileLists::cull_ref_n): a level that n candidates enter drops below cull_eps * 1365 / n,
    so each of the three levels under the tile level loses less than ~3 * 1365 * cull_eps along a ray and the frame less than
    3 * cull_eps * (N + 4096) = 2.5e-5 for N <= 4096, whatever the scene.  Checked against the full sum
    (cull_eps = 0) of the same context; one fixed threshold (VRT_HIP_CULL_REF_N=0) shades more entries for less error, and
    both stay far inside the parity tolerance of 1e-4."""
    from sgrt_amd import scene
    g = scene.grid_scene(64) if name == "g64" else scene.read_obj(os.path.join(GOLDEN, "test-objects", name + ".obj"))
    cam, _ = scene.cli_camera(w, w, initial_rot=rot)

    def frame(r, eps, kappa=0.0):
        r.set_gaussians(g)
        r.set_camera_view(w, w, cam.view)
        r.set_options(pkg.EXP_VCL, pkg.ERF_AS, eps)
        r.set_cull_prune(kappa)
        r.set_table_step(0.0)   # the exact kernels: this test is about what the CULL loses (the table kernel has its own budget)
        r.tile_gaussians(2 / 16, 2 / 16, cam.view)
        r.enable_stats(True)
        _, rad = r.render(cam.position)
        st = r.stats()
        r.enable_stats(False)
        return rad.astype(np.float64), st["lane_entries"] + st["dense_visits_full"]

    full, _ = frame(renderer, 0.0)
    rad, work = frame(renderer, 1e-9)
    # the budgeted ray-level prune on top (round 3, default kappa 6): a block-kernel ray may lose 3 * 6 * 1365 * cull_eps = 2.46e-5 more --
    # for it this replaces the table budget, so a frame's worst case stays 5e-5 -- and shades fewer entries for it
    radp, workp = frame(renderer, 1e-9, 6.0)
    monkeypatch.setenv("VRT_HIP_CULL_REF_N", "0")
    r0 = pkg.Renderer(0)
    try:
        rad0, work0 = frame(r0, 1e-9)
    finally:
        r0.close()
        renderer.set_options(pkg.EXP_VCL, pkg.ERF_AS, 1e-9)
        renderer.set_table_step(pkg.TABLE_STEP_DEFAULT)
        renderer.set_cull_prune(6.0)
    assert np.abs(rad - full).max() <= 2.5e-5
    assert np.abs(rad0 - full).max() <= np.abs(rad - full).max() + 1e-6
    assert work < work0
    assert np.abs(radp - full).max() <= 5e-5 and np.abs(radp - rad).max() <= 2.46e-5
    assert workp <= work and (name != "g64" or workp < work)   # (at 1024^2 most of this scene's blocks are dense: the prune is the block kernel's)


def test_kernel_timing_modes(pkg, oracle, renderer):
    """vrt_hip_enable_kernel_timing: 1 = four events per frame, 2 = the one-wave kernel only, 3 = that on every 8th frame."""
    import torch
    w = h = 256
    g = oracle.grid_scene(8)
    cam, plane, origin, tiles = setup_scene(pkg, oracle, renderer, g, w, h)
    out = torch.zeros(w * h, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    frame = renderer.frame_call(2 / 16, 2 / 16, oracle.camera_view(cam), origin, pkg.PACK_ROUND | pkg.ALPHA_COMPUTED)
    for mode, launches, full in [(1, 16, True), (2, 16, False), (3, 2, False)]:
        renderer.enable_kernel_timing(mode)
        for _ in range(16):
            frame(out.data_ptr(), st)
        torch.cuda.synchronize()
        kt = renderer.kernel_timing()
        assert kt["launches"] == launches, (mode, kt)
        assert 0 < kt["render_ms"] < 5.0
        assert (kt["lists_ms"] > 0) == full and (kt["dense_ms"] > 0) == full, (mode, kt)
    renderer.enable_kernel_timing(False)


def test_frames_enqueued_back_to_back_with_a_moving_camera(pkg, oracle, renderer):
    """An orbit whose frames are enqueued without waiting (the CLI's --frames loop): the host runs many frames ahead of
    the GPU, and the library sizes its dense launch from reports of EARLIER frames.  Which kernel shades a block never
    depends on those reports (only how many workgroups the dense launch gets), so every frame equals the one a waiting
    caller gets, bit for bit -- including the views that do need the dense kernel."""
    import torch
    from sgrt_amd import scene
    w = h = 1024
    g = oracle.grid_scene(64)
    renderer.set_gaussians(g)
    pack = pkg.PACK_ROUND | pkg.ALPHA_COMPUTED
    st = torch.cuda.current_stream().cuda_stream
    cams = [scene.cli_camera(w, h, initial_rot=float(a))[0] for a in (0, 0, 0, 0, 20, 45, 60, 80, 100, 100, 100, 140)]
    bufs = [torch.zeros(w * h, dtype=torch.int32, device="cuda") for _ in cams]
    for cam, buf in zip(cams, bufs):                                   # no synchronisation inside this loop
        renderer.set_camera(w, h, cam.position, cam.right, cam.up, cam.front, float(cam.focal))
        renderer.frame_call(2 / 16, 2 / 16, cam.view, cam.position, pack)(buf.data_ptr(), st)
    torch.cuda.synchronize()
    dense_seen = False
    for cam, buf in zip(cams, bufs):
        renderer.set_camera(w, h, cam.position, cam.right, cam.up, cam.front, float(cam.focal))
        renderer.tile_gaussians(2 / 16, 2 / 16, cam.view)
        renderer.enable_stats(True)
        img, _ = renderer.render(cam.position, pack, want_radiance=False)
        dense_seen |= renderer.stats()["dense_blocks"] > 0
        renderer.enable_stats(False)
        np.testing.assert_array_equal(buf.cpu().numpy().view(np.uint32).reshape(h, w), img)
    assert dense_seen      # the orbit does pass through views that need the dense kernel


def test_dense_launch_is_left_out_only_while_nothing_changed(pkg, oracle, renderer):
    """Once a frame has REPORTED that it had no dense cell and no handed-over block, later frames of the same scene, options,
    rays and camera launch no dense kernel at all (the report is exact for them).  The first frame after any change must
    bring it back: here the context is driven into that state (frames waited for, so the reports are in), then the camera
    turns to a view with dense cells / the scene becomes a dense one / the cull threshold changes -- every frame equals
    the frame of a fresh context."""
    import torch
    from sgrt_amd import scene
    w = h = 2048                       # `-g 64` has no dense cell at this size seen straight on, and has some at 47 degrees
    grid = scene.grid_scene(64)
    monkey = scene.read_obj(os.path.join(GOLDEN, "test-objects", "monkey.obj"))
    pack = pkg.PACK_ROUND | pkg.ALPHA_COMPUTED
    st = torch.cuda.current_stream().cuda_stream
    cam0 = scene.cli_camera(w, h)[0]
    cam47 = scene.cli_camera(w, h, initial_rot=47.0)[0]

    def fresh(g, cam, eps=1e-9):
        renderer.set_gaussians(g)
        renderer.set_options(pkg.EXP_VCL, pkg.ERF_AS, eps)
        renderer.set_camera_view(w, h, cam.view)
        renderer.tile_gaussians(2 / 16, 2 / 16, cam.view)
        renderer.enable_stats(True)
        img, _ = renderer.render(cam.position, pack, want_radiance=False)
        dense = renderer.stats()["dense_blocks"]
        renderer.enable_stats(False)
        return img.reshape(-1), dense

    def fresh_tiles(g, cam, tiles_n):
        renderer.set_gaussians(g)
        renderer.set_options(pkg.EXP_VCL, pkg.ERF_AS, 1e-9)
        renderer.set_camera_view(w, h, cam.view)
        renderer.tile_gaussians(2 / tiles_n, 2 / tiles_n, cam.view)
        renderer.enable_stats(True)
        img = renderer.render(cam.position, pack, want_radiance=False)[0].reshape(-1)
        dense = renderer.stats()["dense_blocks"]
        renderer.enable_stats(False)
        return img, dense

    r = pkg.Renderer(0)
    buf = torch.zeros(w * h, dtype=torch.int32, device="cuda")
    try:
        def settle(g, cam, eps=1e-9, n=6):
            r.set_gaussians(g)
            r.set_options(pkg.EXP_VCL, pkg.ERF_AS, eps)
            r.set_camera_view(w, h, cam.view)
            for _ in range(n):
                r.frame_call(2 / 16, 2 / 16, cam.view, cam.position, pack)(buf.data_ptr(), st)
                torch.cuda.synchronize()

        def frame(cam):
            r.set_camera_view(w, h, cam.view)
            r.frame_call(2 / 16, 2 / 16, cam.view, cam.position, pack)(buf.data_ptr(), st)
            torch.cuda.synchronize()
            return buf.cpu().numpy().view(np.uint32)

        want0, d0 = fresh(grid, cam0)
        want47, d47 = fresh(grid, cam47)
        assert d0 == 0 and d47 > 0
        settle(grid, cam0)
        np.testing.assert_array_equal(frame(cam0), want0)
        np.testing.assert_array_equal(frame(cam47), want47)          # the camera turned: dense cells
        np.testing.assert_array_equal(frame(cam47), want47)
        np.testing.assert_array_equal(frame(cam0), want0)
        settle(grid, cam0)
        wantm, dm = fresh(monkey, cam0)
        assert dm > 0
        r.set_gaussians(monkey)                                       # another scene behind the same camera
        np.testing.assert_array_equal(frame(cam0), wantm)
        # another tile grid behind the same camera, scene and options (round-2 advisor: prepare_tile_grid stamped nothing, so
        # the first frame on the new grid ran without the dense launch): the launch is back for exactly one frame of the new
        # grid, then -- that frame having reported -- left out again
        settle(grid, cam0)
        skips = r.stats()["dense_launch_skips"]
        assert skips > 0
        for tiles_n in (4, 16, 8):
            r.frame_call(2 / tiles_n, 2 / tiles_n, cam0.view, cam0.position, pack)(buf.data_ptr(), st)
            torch.cuda.synchronize()
            assert r.stats()["dense_launch_skips"] == skips, tiles_n          # first frame of this grid: not skipped
            want_t, dense_t = fresh_tiles(grid, cam0, tiles_n)
            np.testing.assert_array_equal(buf.cpu().numpy().view(np.uint32), want_t)
            for _ in range(3):
                r.frame_call(2 / tiles_n, 2 / tiles_n, cam0.view, cam0.position, pack)(buf.data_ptr(), st)
                torch.cuda.synchronize()
            np.testing.assert_array_equal(buf.cpu().numpy().view(np.uint32), want_t)
            # ... settled again -- unless this grid HAS dense cells (8 tiles: longer tile lists, more careful thresholds below them,
            # some cell lists beyond the dense threshold: the case the missing stamp got wrong)
            assert (r.stats()["dense_launch_skips"] > skips) == (dense_t == 0), (tiles_n, dense_t)
            skips = r.stats()["dense_launch_skips"]
        # untiled (clear_tiles) -> device tiles
        r.clear_tiles()
        r.render_device(cam0.position, pack, buf.data_ptr(), 0, st)
        torch.cuda.synchronize()
        skips = r.stats()["dense_launch_skips"]
        r.frame_call(2 / 16, 2 / 16, cam0.view, cam0.position, pack)(buf.data_ptr(), st)
        torch.cuda.synchronize()
        assert r.stats()["dense_launch_skips"] == skips
        np.testing.assert_array_equal(buf.cpu().numpy().view(np.uint32), want0)
        settle(grid, cam0)
        want_e, _ = fresh(grid, cam0, eps=0.0)
        r.set_options(pkg.EXP_VCL, pkg.ERF_AS, 0.0)                   # other options
        np.testing.assert_array_equal(frame(cam0), want_e)
        renderer.set_options(pkg.EXP_VCL, pkg.ERF_AS, 1e-9)
    finally:
        r.close()


def test_frames_on_the_context_stream_then_on_a_caller_stream(pkg, oracle, renderer):
    """vrt_hip_frame without waiting (the CLI's animation loop: frames in flight on the context's own stream) followed at
    once by a frame of another camera on a caller's stream: the library orders the two itself (lists and queue counters of
    the first frame must not be rewritten under it)."""
    import torch
    from sgrt_amd import scene
    w = h = 1024
    g = oracle.grid_scene(64)
    renderer.set_gaussians(g)
    renderer.set_options(pkg.EXP_VCL, pkg.ERF_AS, 1e-9)
    pack = pkg.PACK_ROUND | pkg.ALPHA_COMPUTED
    cam_a, cam_b = scene.cli_camera(w, h, initial_rot=50.0)[0], scene.cli_camera(w, h, initial_rot=0.0)[0]
    side = torch.cuda.Stream()
    out = torch.zeros(w * h, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for _ in range(3):
        renderer.set_camera_view(w, h, cam_a.view)
        for _k in range(3):
            renderer.frame(2 / 16, 2 / 16, cam_a.view, cam_a.position, pack, want_image=False, wait=False)     # 50 degrees: heavy
        renderer.set_camera_view(w, h, cam_b.view)
        renderer.frame_call(2 / 16, 2 / 16, cam_b.view, cam_b.position, pack)(out.data_ptr(), side.cuda_stream)
        renderer.set_camera_view(w, h, cam_a.view)
        img_a = renderer.frame(2 / 16, 2 / 16, cam_a.view, cam_a.position, pack)                                   # waits
        torch.cuda.synchronize()
        got_b = out.cpu().numpy().view(np.uint32).reshape(h, w)
        fresh = pkg.Renderer(0)
        try:
            fresh.set_gaussians(g)
            for cam, got in ((cam_a, img_a), (cam_b, got_b)):
                fresh.set_camera_view(w, h, cam.view)
                fresh.tile_gaussians(2 / 16, 2 / 16, cam.view)
                ref, _ = fresh.render(cam.position, pack, want_radiance=False)
                np.testing.assert_array_equal(got, ref)
        finally:
            fresh.close()


@pytest.mark.parametrize("name,step,rot", [("monkey", None, 20.0), ("monkey", None, 150.0), ("teapot", None, 0.0), ("teapot", 0.1, 0.0),
                                           ("cube", None, 30.0)])
def test_table_mode_stays_inside_the_tolerance(pkg, oracle, renderer, name, step, rot):
    """Table mode (the default; vrt_hip_set_table_step): dense blocks interpolate the transmittance exponent from nodes along
    each ray.  Not the reference's summation order, but every ray's worst-case deviation is bounded by the kernel itself
    (budget 2.5e-5): the frame is within that of the exact kernels, within the 1e-4 tolerance of the ORACLE on the bright
    pixels of the test objects, and blocks the table cannot cover are shaded exactly."""
    w = h = 512
    g = oracle.read_obj(os.path.join(GOLDEN, "test-objects", name + ".obj"))
    cam, plane, origin, tiles = setup_scene(pkg, oracle, renderer, g, w, h, rot=rot)
    renderer.set_options(pkg.EXP_VCL, pkg.ERF_AS, 1e-9)
    renderer.set_table_step(0.0)
    renderer.enable_stats(True)
    _, exact = renderer.render(origin)
    st0 = renderer.stats()
    assert st0["table_blocks"] == 0 and st0["dense_blocks"] > 0
    renderer.set_table_step(pkg.TABLE_STEP_DEFAULT if step is None else step)
    try:
        img, rad = renderer.render(origin)
        st = renderer.stats()
        assert st["table_blocks"] > st["table_empty"] and st["table_blocks"] + st["table_declined"] == st0["dense_blocks"]
        assert st["table_declined"] <= 0.02 * st0["dense_blocks"]            # the bound holds for (nearly) every block
        assert np.abs(rad - exact).max() <= 2.5e-5 + 5e-6                    # the budget + fp32 noise of two summation orders
        rng = np.random.default_rng(7)
        lum = rad.reshape(-1, 4)[:, :3].sum(1)
        pix = np.unique(rng.choice(np.nonzero(lum > 0.05)[0], 40)).astype(np.uint32)
        _, orad = oracle.render(w, h, plane, origin, g, tiles, pixels=pix, want_image=False)
        assert np.abs(rad.reshape(-1, 4)[pix] - orad).max() <= TOL
        # a budget only black rays can meet: (nearly) every block declined after its second attempt, the exact kernel shades them
        renderer.set_table_budget(1e-12)
        _, rad2 = renderer.render(origin)
        st2 = renderer.stats()
        assert st2["table_blocks"] + st2["table_declined"] == st0["dense_blocks"]
        assert st2["table_declined"] >= 0.7 * (st0["dense_blocks"] - st["table_empty"])
        assert np.abs(rad2 - exact).max() <= 1e-6
        renderer.set_table_budget(2.5e-5)
        # a step nothing can meet: the sample range needs more nodes than eight segments hold
        renderer.set_table_step(1e-5)
        _, rad3 = renderer.render(origin)
        st3 = renderer.stats()
        assert st3["table_blocks"] == st["table_empty"] and st3["table_declined"] == st0["dense_blocks"] - st["table_empty"]
        # (the fallback is the exact kernel's arithmetic over the table kernel's 8 waves per block; the exact launch sums a ray's partial
        # radiances over 16: VRT_HIP_DENSE_WAVES=8 makes the two bit-equal)
        assert np.abs(rad3 - exact).max() <= 1e-6
    finally:
        renderer.enable_stats(False)
        renderer.set_table_budget(2.5e-5)
        renderer.set_table_step(pkg.TABLE_STEP_DEFAULT)


def test_more_ranks_than_tiles(pkg, oracle, renderer):
    """4 tiles on 5 and 8 'ranks' (one context playing every rank in turn): ranks without a tile render nothing, launch no
    list kernel, and must leave the queue counters as the next render expects them (tests/fuzz_parity.py found that they
    did not); the assembled frame and the full-frame render that follows are the single-rank image."""
    import torch
    w, h = 96, 64
    g = oracle.grid_scene(6)
    cam, plane, origin, tiles = setup_scene(pkg, oracle, renderer, g, w, h, tiles_n=2)
    renderer.set_shard(0, 1)
    full, full_rad = renderer.render(origin)
    assert (full >> 24).max() > 0
    pack = pkg.PACK_ROUND | pkg.ALPHA_COMPUTED
    st = torch.cuda.current_stream().cuda_stream
    for world in (5, 8):
        shards = []
        for rank in range(world):
            renderer.set_shard(rank, world)
            buf = torch.zeros(renderer.shard_pixels(), dtype=torch.int32, device="cuda")
            renderer.render_shard_device(origin, pack, buf.data_ptr(), st)
            torch.cuda.synchronize()
            shards.append(buf)
        assert len({b.numel() for b in shards}) == 1          # one fixed-size gather moves the frame
        out = torch.zeros(w * h, dtype=torch.int32, device="cuda")
        renderer.assemble_shards_device(torch.cat(shards).data_ptr(), out.data_ptr(), st)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(out.cpu().numpy().view(np.uint32).reshape(h, w), full)
        renderer.set_shard(0, 1)
        again, again_rad = renderer.render(origin)
        np.testing.assert_array_equal(again, full)
        np.testing.assert_array_equal(again_rad, full_rad)


def test_retained_frame_buffer_equals_fresh_frames(pkg, renderer):
    """vrt_hip_frame renders into the library's own buffer, which still holds the context's previous frame: the list kernel
    then clears only the cells that were lit in that frame and are empty now (per-cell stamps), not all of them.  Every
    frame of a long-lived context must equal the frame of a fresh one: a camera that orbits (cells go dark and light up),
    the same view twice, another image size, another background (opaque / computed alpha), another tile grid, a
    vrt_hip_render in between (it writes the same buffer), another scene, tiles that do not cover the image."""
    from sgrt_amd import scene
    scenes = {"g64": scene.grid_scene(64), "g16": scene.grid_scene(16), "monkey": scene.read_obj(os.path.join(GOLDEN, "test-objects", "monkey.obj"))}
    mode8, opaque = pkg.PACK_ROUND | pkg.ALPHA_COMPUTED, pkg.PACK_TRUNC | pkg.ALPHA_OPAQUE
    steps = [("g64", 768, 16, mode8, 0.0), ("g64", 768, 16, mode8, 10.0), ("g64", 768, 16, mode8, 10.0), ("g64", 768, 16, mode8, 55.0),
             ("g64", 768, 16, mode8, 0.0), ("g64", 512, 16, mode8, 0.0), ("g64", 512, 16, opaque, 0.0), ("g64", 512, 16, opaque, 30.0),
             ("g64", 512, 8, opaque, 30.0), ("g64", 512, 8, mode8, 31.0), ("render", 512, 8, mode8, 31.0), ("g64", 512, 8, mode8, 32.0),
             ("g16", 512, 8, mode8, 32.0), ("monkey", 512, 8, mode8, 32.0), ("monkey", 512, 8, mode8, 200.0), ("g16", 500, 16, mode8, 5.0),
             ("g16", 500, 16, mode8, 50.0), ("g64", 768, 16, mode8, 0.0)]
    r, r2 = pkg.Renderer(0), pkg.Renderer(0)
    bufs, cur2 = {}, None
    try:
        cur = None
        for k, (name, w, tiles_n, pack, rot) in enumerate(steps):
            cam = scene.cli_camera(w, w, initial_rot=rot)[0]
            if name == "render":               # another image into the library's buffer, by another entry point
                r.set_camera_view(w, w, cam.view)
                r.tile_gaussians(2 / tiles_n, 2 / tiles_n, cam.view)
                r.render(cam.position, pack, want_radiance=False)
                continue
            if name != cur:
                r.set_gaussians(scenes[name]); cur = name
            r.set_camera_view(w, w, cam.view)
            got = r.frame(2 / tiles_n, 2 / tiles_n, cam.view, cam.position, pack)
            want = single_frame_ref(pkg, renderer, scenes[name], cam, w, tiles_n, pack)
            np.testing.assert_array_equal(got, want, err_msg=f"step {k}: {name} {w} px, {tiles_n} tiles, rot {rot}")
            # the same promise for a CALLER's buffer (vrt_hip_frame_retained_device): a second context with a device buffer of its
            # own per image size, frames not waited for one by one
            import torch
            if name != cur2:
                r2.set_gaussians(scenes[name]); cur2 = name
            r2.set_camera_view(w, w, cam.view)
            buf = bufs.setdefault(w, torch.full((w * w,), 0x7F7F7F7F, dtype=torch.int32, device="cuda"))   # garbage: the first frame must clear it all
            r2.frame_call(2 / tiles_n, 2 / tiles_n, cam.view, cam.position, pack, retained=True)(buf.data_ptr(), torch.cuda.current_stream().cuda_stream)
            wrote = (tiles_n * int(np.float32(w) * np.float32(2 / tiles_n) / np.float32(2))) ** 2 if w % tiles_n else w * w
            torch.cuda.synchronize()
            got2 = buf.cpu().numpy().view(np.uint32)
            if w % tiles_n == 0:
                np.testing.assert_array_equal(got2.reshape(w, w), want, err_msg=f"caller's buffer, step {k}")
    finally:
        r.close()
        r2.close()


def single_frame_ref(pkg, renderer, g, cam, w, tiles_n, pack):
    renderer.set_gaussians(g)
    renderer.set_options(pkg.EXP_VCL, pkg.ERF_AS, 1e-9)
    renderer.set_camera_view(w, w, cam.view)
    renderer.tile_gaussians(2.0 / tiles_n, 2.0 / tiles_n, cam.view)
    return renderer.render(cam.position, pack, want_radiance=False)[0]


def test_chunked_tile_level_gives_the_same_lists(pkg, oracle, monkeypatch):
    """Round 3: for scenes beyond 8192 Gaussians the tile level tests the bounding spheres of 64 consecutive Gaussians first
    (launch_build_chunks) and only the members of the chunks that are left.  The chunk test is conservative and keeps index order,
    so the lists -- and with them every bit of the frame -- are the ones of the per-Gaussian pass (VRT_HIP_CHUNKS=0); forced on
    (VRT_HIP_CHUNKS=2) the same holds for the small scenes."""
    from sgrt_amd import scene
    rng = np.random.default_rng(11)
    n = 20000
    cloud = np.zeros(n, scene.GAUSSIAN)
    cloud["mu"][:, :3] = rng.normal(size=(n, 3)) * np.array([0.8, 0.8, 0.3]) + np.array([0, 0, 1])
    cloud["mu"][:, 3] = 0
    cloud["sigma"] = rng.uniform(0.004, 0.02, n); cloud["magnitude"] = rng.uniform(0.2, 2, n); cloud["albedo"] = rng.uniform(0, 1, (n, 4))
    order = np.lexsort((cloud["mu"][:, 0], np.floor(cloud["mu"][:, 1] * 8)))   # rows of a coarse raster: chunks with some locality
    cases = [("g64", scene.grid_scene(64), 1024, 47.0), ("teapot", scene.read_obj(os.path.join(GOLDEN, "test-objects", "teapot.obj")), 512, 20.0),
             ("cloud", cloud[order], 768, 10.0), ("cloud unsorted", cloud, 512, 0.0)]
    for name, g, w, rot in cases:
        cam, _ = scene.cli_camera(w, w, initial_rot=rot)
        out = {}
        for mode in ("0", "2", "1"):
            monkeypatch.setenv("VRT_HIP_CHUNKS", mode)
            r = pkg.Renderer(0)
            try:
                r.set_gaussians(g); r.set_camera_view(w, w, cam.view); r.tile_gaussians(2 / 16, 2 / 16, cam.view)
                r.enable_stats(True)
                img, rad = r.render(cam.position)
                st = r.stats()
                out[mode] = (img, rad, st["tile_entries"], st["list_entries"])
            finally:
                r.close()
        for mode in ("2", "1"):
            assert out[mode][2:] == out["0"][2:], (name, mode, out[mode][2:], out["0"][2:])
            np.testing.assert_array_equal(out[mode][0], out["0"][0], err_msg=f"{name} chunks {mode}")
            np.testing.assert_array_equal(out[mode][1], out["0"][1], err_msg=f"{name} chunks {mode}")
        assert out["0"][1][..., :3].max() > 0


def test_frames_with_more_blocks_than_waves(pkg, oracle, monkeypatch):
    """Frames whose blocks outnumber the block kernel's resident waves are worked off through the work queues; since round 3 a wave
    claims its next entry before it shades the current block (VRT_HIP_CLAIM_EARLY, default from 1/8 of the grid size in entries;
    a kernel variant of its own, chosen from what the previous frame reported).
    Which wave shades a block must not matter: the frame equals the one of the old order (claim after the block) bit for bit, with
    the default grid and with a small one (VRT_HIP_RENDER_WAVES=3: every wave walks through a dozen blocks), and the oracle's pixels."""
    from sgrt_amd import scene
    g = scene.grid_scene(16)
    w = 3072
    cam, _ = scene.cli_camera(w, w, initial_rot=12.0)
    frames = {}
    for claim, waves in (("0", "13"), ("8", "13"), ("1000000", "3"), ("8", "5")):
        monkeypatch.setenv("VRT_HIP_CLAIM_EARLY", claim)
        monkeypatch.setenv("VRT_HIP_RENDER_WAVES", waves)
        r = pkg.Renderer(0)
        try:
            r.set_gaussians(g); r.set_camera_view(w, w, cam.view); r.tile_gaussians(2 / 16, 2 / 16, cam.view)
            first = r.render(cam.position)      # reports its block count; the kernel variant of the NEXT frame follows the report
            img, rad = r.render(cam.position)
            np.testing.assert_array_equal(first[0], img); np.testing.assert_array_equal(first[1], rad)
            r.enable_stats(True)
            r.render(cam.position, want_radiance=False)
            assert r.stats()["shaded_blocks"] > 2 * 13 * 256
            frames[(claim, waves)] = (img, rad)
        finally:
            r.close()
    ref = frames[("0", "13")]
    for k, (img, rad) in frames.items():
        np.testing.assert_array_equal(img, ref[0], err_msg=str(k))
        np.testing.assert_array_equal(rad, ref[1], err_msg=str(k))
    lum = ref[1][..., :3].sum(-1).reshape(-1)
    pix = np.random.default_rng(5).choice(np.flatnonzero(lum > 0.25 * lum.max()), 96, replace=False).astype(np.uint32)
    ocam, _ = oracle.cli_camera(w, w, initial_rot=12.0)
    view = oracle.camera_view(ocam)
    tiles = oracle.tile_gaussians(2 / 16, 2 / 16, g.view(oracle.GAUSSIAN), view)
    orad = oracle.render(w, w, oracle.camera_plane(ocam), ocam.position[:], g.view(oracle.GAUSSIAN), tiles, pixels=pix, want_image=False)[1]
    assert np.abs(ref[1].reshape(-1, 4)[pix] - orad).max() <= TOL
