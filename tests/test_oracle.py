"""CPU tests of the oracle (the checker must itself be pinned before it judges the HIP path).

  - erf/exp restatements against the tables the reference ITSELF holds (thesis/plots/cmp_{erf,exp}_*.tex = the output of
    tests/accuracy.cpp; tests/golden/thesis_plots.npz), against golden tables generated from the REAL reference approx.cpp
    (tests/golden/approx_ref.npz, made by tests/golden/gen_approx_golden.py) and, where /root/reference is present,
    against that reference compiled in place (oracle/_ref);
  - the reference test's own property (tests/transmittance.cpp:24-31): analytic transmittance ==
    numeric line integral of the density;
  - known answers for the scene producers, camera and tile binning;
  - the SIMD baseline port against the scalar restatement.
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN

OBJ = os.path.join(GOLDEN, "test-objects")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "approx_ref.npz"))


def test_erf_exp_restatements_match_reference_tables(oracle, gold):
    x, xe = gold["erf_x"], gold["exp_x"]
    # scalar variants: the reference is built with -ffast-math (FMA contraction), the oracle without
    assert np.abs(oracle.map_scalar("oracle_as_erf", x) - gold["as_erf"]).max() <= 3e-7
    assert np.abs(oracle.map_scalar("oracle_spline_erf", x) - gold["spline_erf"]).max() <= 2e-7
    assert np.abs(oracle.map_scalar("oracle_spline_erf_mirror", x) - gold["spline_erf_mirror"]).max() <= 2e-7
    assert np.abs(oracle.map_scalar("oracle_taylor_erf", x) - gold["taylor_erf"]).max() <= 6e-7
    # SIMD A&S uses rcp14 (2^-14): the survey's "hard part 1" gap
    assert np.abs(oracle.map_scalar("oracle_as_erf", x) - gold["simd_as_erf"]).max() <= 4e-5
    # VCL exp: every operation is pinned (explicit FMAs) -> bit exact
    np.testing.assert_array_equal(oracle.map_scalar("oracle_vcl_exp", xe), gold["vcl_exp"])
    assert (np.abs(oracle.map_scalar("oracle_fast_exp", xe) - gold["fast_exp"]) <= 1e-5 * np.abs(gold["fast_exp"])).all()
    assert np.abs(oracle.map_scalar("oracle_spline_exp", xe) - gold["spline_exp"]).max() <= 1.2e-7
    # A&S error bound quoted by the thesis (|err| <= 5e-4) against the true erf
    assert np.abs(oracle.map_scalar("oracle_as_erf", x) - gold["libm_erf"]).max() <= 5e-4


@pytest.fixture(scope="module")
def plots():
    """What the REFERENCE ITSELF holds for erf / exp: the output of its accuracy experiment (tests/accuracy.cpp:9-58), kept in the
    reference tree as thesis plot data (tests/golden/gen_thesis_plot_golden.py parses the numbers into thesis_plots.npz)."""
    return np.load(os.path.join(GOLDEN, "thesis_plots.npz"))


# restatement / device kernel name -> (plot series, stated bound on |restatement - the reference's printed value|)
THESIS_ERF = {"oracle_as_erf": ("abramowitz_stegun", 1e-7), "oracle_spline_erf": ("spline", 1e-7),
              "oracle_spline_erf_mirror": ("spline_mirror", 1e-7), "oracle_taylor_erf": ("taylor", 6e-7)}   # the reference binary is a -ffast-math build (FMA contraction): last-bit differences


def test_restatements_match_the_reference_held_tables(oracle, plots):
    """Row a12 pinned by reference-HELD vectors (round-3 verdict, missing item 1): every series of cmp_erf_approx.tex (121 points on
    the float-accumulated grid -6, -5.9, ... , -5.7000003, ...) and cmp_exp_approx.tex (160 points) against the oracle restatements."""
    for name, (series, bound) in THESIS_ERF.items():
        x, y = plots[f"cmp_erf_approx/{series}/x"], plots[f"cmp_erf_approx/{series}/y"]
        assert len(x) == 121 and x[0] == -6.0 and x[3] == np.float32(-5.7000003)
        got = oracle.map_scalar(name, x)
        assert np.abs(got - y).max() <= bound, (name, np.abs(got - y).max())
    # most points agree to the bit (119 / 118 / 117 of 121 for A&S / spline / mirror)
    x = plots["cmp_erf_approx/abramowitz_stegun/x"]
    assert (oracle.map_scalar("oracle_as_erf", x).view(np.uint32) == plots["cmp_erf_approx/abramowitz_stegun/y"].view(np.uint32)).sum() >= 115
    # exp: VCL bit for bit (every operation pinned); fast_exp within the product's rounding of the FMA build; spline to 1.2e-7 relative
    xe = plots["cmp_exp_approx/vcl/x"]
    assert len(xe) == 160 and xe[0] == -16.0
    np.testing.assert_array_equal(oracle.map_scalar("oracle_vcl_exp", xe), plots["cmp_exp_approx/vcl/y"])
    fe, fy = oracle.map_scalar("oracle_fast_exp", xe), plots["cmp_exp_approx/fast/y"]
    assert (np.abs(fe - fy) <= 1e-5 * np.abs(fy)).all()
    se, sy = oracle.map_scalar("oracle_spline_exp", xe), plots["cmp_exp_approx/spline/y"]
    assert (np.abs(se - sy) <= 2e-7 * np.abs(sy) + 1e-12).all()
    # the columns the approximations are judged against: std::erf / std::exp of the reference's libm, and SVML within its 4 ulp
    import math
    xs, ys = plots["cmp_erf_approx/std_erf/x"], plots["cmp_erf_approx/std_erf/y"]
    assert np.abs(np.array([math.erf(float(v)) for v in xs]) - ys).max() <= 6e-8
    assert np.abs(plots["cmp_erf_approx/svml/y"] - ys).max() <= 2.4e-7
    xs, ys = plots["cmp_exp_approx/std_exp/x"], plots["cmp_exp_approx/std_exp/y"]
    assert (np.abs(np.exp(xs.astype(np.float64)) - ys) <= 6e-8 * ys).all()
    # the thesis's statement about A&S (|err| <= 5e-4) holds on the reference's own table
    assert np.abs(plots["cmp_erf_approx/abramowitz_stegun/y"] - plots["cmp_erf_approx/std_erf/y"]).max() <= 5e-4


def test_error_plots_follow_from_the_tables(oracle, plots):
    """cmp_erf_err.tex is |std::erf - approximation| / |x| of the same run (julia/cmp_erf.jl:19-23; spline and mirror clamped to 0.1):
    derived data, reproduced from the approx table's own columns and -- away from x = 0, where the division amplifies the last
    bit -- from the oracle's restatements.  cmp_exp_err.tex comes from another run of the experiment (its x column is accumulated
    differently: -0.09999881 against -0.0999975): only its statement is checked, VCL exp within 2 ulp of std::exp."""
    std = plots["cmp_erf_approx/std_erf/y"].astype(np.float64)
    for fn, series in (("oracle_as_erf", "abramowitz_stegun"), ("oracle_taylor_erf", "taylor")):
        x, e = plots[f"cmp_erf_err/{series}/x"], plots[f"cmp_erf_err/{series}/y"]
        np.testing.assert_array_equal(x, plots[f"cmp_erf_approx/{series}/x"])
        far = np.abs(x) > 0.05
        table = np.abs((std - plots[f"cmp_erf_approx/{series}/y"]) / x)
        assert np.abs(table - e)[far].max() <= 1e-7, series
        mine = np.abs((std - oracle.map_scalar(fn, x)) / x)
        assert np.abs(mine - e)[far].max() <= 1e-5, series
    for series in ("spline", "mirror"):   # clamped to 0.1 by the plotting script
        assert plots[f"cmp_erf_err/{series}/y"].max() <= 0.1 + 1e-7
    x, e = plots["cmp_exp_err/vcl/x"], plots["cmp_exp_err/vcl/y"]
    assert (e * np.abs(x) <= 2.4e-7 * np.exp(x.astype(np.float64))).all()
    # taylor_erf.tex is Julia's own evaluation (julia/approx_erf.jl), no output of the C++ path: only its erf column is checked
    import math
    xt, yt = plots["taylor_erf/erf/x"], plots["taylor_erf/erf/y"]
    assert np.abs(np.array([math.erf(float(v)) for v in xt]) - yt).max() <= 1e-7


def test_against_reference_built_in_place(oracle):
    """Only where /root/reference exists (this container): live comparison with oracle/_ref."""
    if oracle.ref_lib() is None:
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    rng = np.random.default_rng(42)
    x = rng.uniform(-6, 6, 4096).astype(np.float32)       # approx_cycles.cpp:67-73 ranges
    xe = rng.uniform(-10, 0, 4096).astype(np.float32)
    assert np.abs(oracle.map_scalar("oracle_as_erf", x) - oracle.ref_map("ref_as_erf", x)).max() <= 5e-7
    np.testing.assert_array_equal(oracle.map_scalar("oracle_vcl_exp", xe), oracle.ref_map("ref_simd_vcl_exp", xe))
    wide = np.linspace(-100, 5, 3001).astype(np.float32)
    np.testing.assert_array_equal(oracle.map_scalar("oracle_vcl_exp", wide), oracle.ref_map("ref_simd_vcl_exp", wide))


def three_gaussians(oracle):
    # tests/transmittance.cpp:9
    return oracle.gaussians([[0, 1, 0, .1], [0, 0, 1, .7], [1, 0, 0, 1]], [[.3, .3, .5], [-.3, -.3, 0], [0, 0, 2]],
                            [.1, .4, .75], [2., .7, 1.])


def test_transmittance_equals_numeric_integral(oracle):
    """The reference's transmittance experiment as an assertion: T(s) = exp(-int_0^s density)."""
    g = three_gaussians(oracle)
    o, n = (0, 0, -5), (0, 0, 1)
    for k in np.arange(-6, 6.01, 1.0):
        s = 7.0 + k * 0.75
        T = oracle.transmittance(o, n, s, g)[0]
        # float64 trapezoid of the density along the ray
        t = np.linspace(0, s, 20001)
        dens = np.zeros_like(t)
        for q in g:
            d2 = (0 - q["mu"][0]) ** 2 + (0 - q["mu"][1]) ** 2 + (-5 + t - q["mu"][2]) ** 2
            dens += q["magnitude"] * np.exp(-d2 / (2 * float(q["sigma"]) ** 2))
        ref = np.exp(-np.trapz(dens, t))
        assert abs(T - ref) <= 2e-6, (s, T, ref)
        # the reference's own (coarser) cross-check: Riemann sum + fast_exp, a few % accurate
        assert abs(oracle.transmittance_step(o, n, s, 0.01, g) - ref) <= 0.05
    # A&S instead of erff moves T by < 1e-3 (|erf error| <= 5e-4, weights sum ~1)
    s = np.float32(7.0)
    assert abs(oracle.transmittance(o, n, s, g, oracle.EXP_VCL, oracle.ERF_AS)[0] - oracle.transmittance(o, n, s, g)[0]) < 1e-3


def test_radiance_is_order_independent_and_has_alpha(oracle):
    """'the order of the Gaussians is irrelevant to the method' (thesis/main.tex:254-255)."""
    g = three_gaussians(oracle)
    o, n = (0.1, -0.05, -5), (0, 0, 1)
    a = oracle.radiance(o, n, g)
    b = oracle.radiance(o, n, g[::-1].copy())
    assert np.abs(a - b).max() <= 1e-6
    assert a[3] > 0  # w = sum albedo.w * inner (rt.h:220, 373)


def test_grid_scene_known_answers(oracle):
    g = oracle.grid_scene(4)  # main.cpp:194-205
    assert len(g) == 16
    np.testing.assert_allclose(g["mu"][0], [-0.75, -0.75, 1, 0])
    np.testing.assert_allclose(g["mu"][5], [-0.25, -0.25, 1, 0])      # i = 1, j = 1 (i-major)
    np.testing.assert_allclose(g["mu"][1], [-0.75, -0.25, 1, 0])      # j runs along y
    np.testing.assert_allclose(g["albedo"][0], [1, 0, 0, 1])
    np.testing.assert_allclose(g["albedo"][15], [1 - 15 / 16, 0, 15 / 16, 1])
    assert (g["sigma"] == 0.125).all() and (g["magnitude"] == 1).all()
    assert len(oracle.grid_scene(64)) == 4096
    assert len(oracle.grid_scene(260)) == 16     # grid_dim truncated to u8 (main.cpp:196): 260 -> 4


def test_obj_loader(oracle):
    counts = {"cube.obj": (386, 0.15), "monkey.obj": (507, 0.15), "simple_cube.obj": (8, 0.3), "sphere.obj": (42, 0.3),
              "teapot.obj": (3644, 0.05)}     # gaussians-from-file.cpp:26-30
    for name, (n, sig) in counts.items():
        g = oracle.read_obj(os.path.join(OBJ, name))
        assert len(g) == n
        assert np.allclose(g["sigma"], sig) and (g["magnitude"] == 1).all() and (g["albedo"][:, 3] == 1).all()
        v = g["mu"][:, :3]
        nv = v / np.linalg.norm(v, axis=1, keepdims=True)
        assert np.abs(g["albedo"][:, :3] - (nv * 0.5 + 0.5)).max() <= 1e-6
    t = oracle.read_obj(os.path.join(OBJ, "teapot.obj"))
    np.testing.assert_allclose(t["mu"][0, :3], [-1.766694, -0.302723, 0.011649], rtol=1e-7)
    with pytest.raises(IOError):
        oracle.read_obj(os.path.join(OBJ, "missing.obj"))


def test_camera_closed_form(oracle):
    """plane = pos + x right + y up - focal front; default CLI camera: plane (x, y, -3), rays toward +z."""
    c, angle = oracle.cli_camera(64, 32)
    xs, ys, zs = oracle.camera_plane(c)
    xs, ys, zs = (a.reshape(32, 64) for a in (xs, ys, zs))
    assert np.abs(zs + 3).max() <= 1e-6
    np.testing.assert_allclose(xs[0], -1 + np.arange(64) / 32.0, atol=2e-7)
    np.testing.assert_allclose(ys[:, 0], -1 + np.arange(32) / 16.0, atol=2e-7)   # row 0 is y = -1: no flip
    # rotated: compare with the closed form
    for rot in (33.0, 190.0):
        c, _ = oracle.cli_camera(16, 16, initial_rot=rot)
        xs, ys, zs = oracle.camera_plane(c)
        pos, right, up, front = (np.array(v[:], np.float64) for v in (c.position, c.right, c.up, c.front))
        x = -1 + np.arange(16) / 8.0
        P = pos[None, None] + x[None, :, None] * right + x[:, None, None] * up - front
        got = np.stack([xs, ys, zs], -1).reshape(16, 16, 3)
        assert np.abs(got - P).max() <= 5e-6
        assert abs(np.linalg.norm(pos) - 4) <= 1e-5       # orbit keeps the radius (main.cpp:330-334)
    # view-space depth used by tiling: grid plane z = 1 seen from z = -4 with focal 1 -> 4
    c, _ = oracle.cli_camera(8, 8)
    V = oracle.camera_view(c).reshape(4, 4).T
    assert abs((V @ np.array([0.3, -0.2, 1, 1]))[2] - 4) <= 1e-5


def test_tile_binning_known_answers(oracle):
    c, _ = oracle.cli_camera(256, 256)
    view = oracle.camera_view(c)
    # golden: the per-axis counts SURVEY.md 8(c) derived by hand from rt.cpp:58-59 (tests/golden/tile_counts.json);
    # the inclusion test is separable, so tile (ty, tx) holds per_axis[ty] * per_axis[tx] Gaussians
    import json
    gold = json.load(open(os.path.join(GOLDEN, "tile_counts.json")))
    for name, sc in gold["scenes"].items():
        t = oracle.tile_gaussians(gold["tw"], gold["tw"], oracle.grid_scene(sc["grid"]), view)
        assert (t["w"], t["h"], t["nloop"]) == (16, 16, 256)
        pa = np.array(sc["per_axis"])
        np.testing.assert_array_equal(np.diff(t["offsets"]).reshape(16, 16), np.outer(pa, pa), err_msg=name)
    t4 = oracle.tile_gaussians(2 / 16, 2 / 16, oracle.grid_scene(64), view)
    d = np.diff(t4["offsets"])
    assert (d.min(), d.max()) == (1156, 1681) and abs(d.mean() - 1610.015625) < 1e-9   # cfg4 (mean 1610)
    # order preserved (rt.cpp:52-62)
    for k in (0, 100, 255):
        idx = t4["indices"][t4["offsets"][k]:t4["offsets"][k + 1]]
        assert (np.diff(idx.astype(np.int64)) > 0).all()
    # near-plane discard proj.z < 1 (rt.cpp:38): a Gaussian just behind the projection plane is in no tile
    g = oracle.gaussians([[1, 1, 1, 1]] * 2, [[0, 0, -2.5], [0, 0, 1]], [0.1, 0.1], [1, 1])
    t = oracle.tile_gaussians(1.0, 1.0, g, view)
    assert (t["w"], t["h"]) == (2, 2) and set(t["indices"].tolist()) == {1}
    # non power-of-two tile count: float loops (rt.cpp:47-49) vs ceil(2/tw) (types.h:280)
    t = oracle.tile_gaussians(2 / 10, 2 / 10, oracle.grid_scene(4), view)
    assert t["w"] == 10 and t["h"] == 10 and t["nloop"] in (100, 110, 121)


def test_pixel_packing(oracle):
    import ctypes as C
    L = oracle.lib()
    def pack(c, flags):
        a = np.array(c, np.float32)
        return L.oracle_pack_pixel(a.ctypes.data_as(C.POINTER(C.c_float)), flags)
    c = [0.5, 0.25, 2.0, 0.999]
    # A<<24 | R<<16 | G<<8 | B (rt.h:243); trunc: (u32)(0.5*255) = 127; round: 127.5 -> 128 (nearest even)
    assert pack(c, oracle.PACK_TRUNC | oracle.ALPHA_OPAQUE) == (0xFF << 24) | (127 << 16) | (63 << 8) | 255
    assert pack(c, oracle.PACK_ROUND | oracle.ALPHA_OPAQUE) == (0xFF << 24) | (128 << 16) | (64 << 8) | 255
    assert pack(c, oracle.PACK_ROUND | oracle.ALPHA_COMPUTED) == (255 << 24) | (128 << 16) | (64 << 8) | 255
    assert pack([0, 0, 0, 0.3], oracle.PACK_ROUND | oracle.ALPHA_COMPUTED) == (76 << 24)   # 76.5 -> 76 (even)


def test_render_tiled_equals_untiled_when_tiles_hold_everything(oracle):
    """With one tile holding every Gaussian the tiled and untiled drivers agree (rt.h:227 vs 251)."""
    w = h = 16
    g = oracle.grid_scene(2)
    c, _ = oracle.cli_camera(w, h)
    plane = oracle.camera_plane(c)
    tiles = dict(tw=np.float32(2), th=np.float32(2), w=1, h=1, offsets=np.array([0, 4], np.uint32),
                 indices=np.arange(4, dtype=np.uint32))
    img_t, rad_t = oracle.render(w, h, plane, c.position[:], g, tiles, pack=oracle.PACK_ROUND | oracle.ALPHA_OPAQUE)
    img_u, rad_u = oracle.render(w, h, plane, c.position[:], g, None, pack=oracle.PACK_ROUND | oracle.ALPHA_OPAQUE)
    # (atol: worker threads may or may not flush denormals depending on what else the process imported)
    np.testing.assert_allclose(rad_t, rad_u, rtol=0, atol=1e-37)
    np.testing.assert_array_equal(img_t, img_u)
    # sparse pixel evaluation returns the same numbers as the dense one
    pix = np.array([0, 17, 100, 255], np.uint32)
    _, rad_s = oracle.render(w, h, plane, c.position[:], g, None, pixels=pix, want_image=False)
    np.testing.assert_allclose(rad_s, rad_u[pix], rtol=0, atol=1e-37)


def test_simd_baseline_port_matches_scalar_oracle(oracle):
    """The CPU baseline (SIMD over pixels, rcp estimates) against the exact-divide scalar restatement:
    <= 1 LSB per u8 channel on cfg1 (-g 4 -w 256) -- the survey's SIMD-vs-scalar gap is ~5e-6 here."""
    w = h = 256
    g = oracle.grid_scene(4)
    c, _ = oracle.cli_camera(w, h)
    plane = oracle.camera_plane(c)
    tiles = oracle.tile_gaussians(2 / 16, 2 / 16, g, oracle.camera_view(c))
    img_s, _ = oracle.render(w, h, plane, c.position[:], g, tiles)
    img_v, terms, width = oracle.simd_render_tiled(w, h, plane, c.position[:], g, tiles, threads=4)
    assert width in (8, 16) and terms == 256 * 256 * 5 * 81            # 2.65e7 inner terms
    sh = np.array([0, 8, 16, 24], np.uint32)
    d = np.abs(((img_s[:, None] >> sh) & 255).astype(int) - ((img_v[:, None] >> sh) & 255).astype(int))
    assert d.max() <= 1
    # bounded sample: two tiles, first 3 rows
    img_p, terms_p, _ = oracle.simd_render_tiled(w, h, plane, c.position[:], g, tiles, tile_subset=[17, 34], threads=2, max_rows=3)
    assert terms_p == 2 * 3 * 16 * 5 * 81
    rows = img_p.reshape(h, w)[16:19, 16:32]
    np.testing.assert_array_equal(rows, img_v.reshape(h, w)[16:19, 16:32])
