"""CPU tests of the boundary and the host-side mirror (no compute calls: there is no GPU here).

  - libvrt_hip.so loads and exports every symbol include/vrt_hip.h declares;
  - without a GPU the library refuses to create a context (no CPU fallback), with an error string;
  - the product's host producers (scene.py: grid scene, OBJ loader, camera, view matrix) agree with
    the oracle's restatement of the reference;
  - the host mirror of the shard map (sharding.py) covers every tile exactly once.
"""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

OBJ = os.path.join(GOLDEN, "test-objects")


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "vrt_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vrt_hip_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(pkg):
    syms = declared_symbols()
    assert len(syms) >= 25
    lib = C.CDLL(pkg.LIB_PATH)
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    # and the Python binding knows each of them (argtypes set => a typo cannot silently pass)
    assert sorted(pkg.SYMBOLS) == syms
    assert pkg.lib().vrt_hip_version().startswith(b"vrt_hip")


def test_no_gpu_means_no_context(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.VrtHipError) as e:
        pkg.Renderer(0)
    assert "no HIP device" in str(e.value) or "failed" in str(e.value)
    L = pkg.lib()
    h = C.c_void_p()
    assert L.vrt_hip_create(0, C.byref(h)) == -3 and not h.value        # VRT_HIP_ERR_NO_DEVICE
    assert L.vrt_hip_create(0, None) == -1                                # VRT_HIP_ERR_INVALID
    assert L.vrt_hip_set_options(None, 1, 1, 0.0) == -1
    # the multi-GPU group fails the same way, with a message, and the size queries of a missing context read 0
    g = C.c_void_p()
    dev = (C.c_int * 2)(0, 0)
    assert L.vrt_hip_group_create(dev, 2, C.byref(g)) == -3 and not g.value
    assert b"member 0" in L.vrt_hip_group_last_error(None)
    assert L.vrt_hip_group_create(dev, 0, C.byref(g)) == -1
    assert L.vrt_hip_group_size(None) == 0 and L.vrt_hip_group_ctx(None, 0) is None
    assert L.vrt_hip_sparse_shard_words(None) == 0 and L.vrt_hip_image_pixels(None) == 0


def test_host_scene_matches_oracle(pkg, oracle):
    from sgrt_amd import scene
    for d in (4, 16, 64, 260):
        a, b = scene.grid_scene(d), oracle.grid_scene(d)
        assert a.tobytes() == b.tobytes(), d
    for name in ("cube.obj", "monkey.obj", "simple_cube.obj", "sphere.obj", "teapot.obj"):
        a, b = scene.read_obj(os.path.join(OBJ, name)), oracle.read_obj(os.path.join(OBJ, name))
        assert len(a) == len(b)
        np.testing.assert_array_equal(a["mu"], b["mu"])
        np.testing.assert_array_equal(a["sigma"], b["sigma"])
        assert np.abs(a["albedo"] - b["albedo"]).max() <= 1.2e-7


def _same_camera(pc, oc, oracle, what):
    for f in ("position", "front", "right", "up"):
        np.testing.assert_array_equal(getattr(pc, f), np.array(getattr(oc, f)[:], np.float32), err_msg=f"{what} {f}")
    np.testing.assert_array_equal(pc.view, oracle.camera_view(oc), err_msg=f"{what} view")
    for got, ref, ax in zip(pc.plane(), oracle.camera_plane(oc), "xyz"):
        np.testing.assert_array_equal(got, ref, err_msg=f"{what} plane {ax}")


def test_host_camera_is_the_oracle_camera_bit_for_bit(pkg, oracle):
    """camera.cpp:7-71 + main.cpp:247-255, 330-334: the product's camera (library arithmetic in glm's order) against
    the oracle's restatement -- BIT equality of basis, view matrix and every plane point, at rotated poses, other focal
    lengths / offsets / aspect ratios, and along an orbit.  (A last-bit difference in a ray is up to 5e-4 of radiance
    for small sigma; a tolerance here would hide exactly that.)"""
    from sgrt_amd import scene
    for rot in (0.0, 20.0, 33.0, 47.0, 123.0, 359.0):
        pc, pa = scene.cli_camera(32, 24, initial_rot=rot)
        oc, oa = oracle.cli_camera(32, 24, initial_rot=rot)
        assert np.float32(pa) == oa[0]
        _same_camera(pc, oc, oracle, f"rot {rot}")
    for off, focal in ((-4.0, 1.0), (-2.5, 0.7), (-9.0, 2.25)):
        pc, _ = scene.cli_camera(17, 40, camera_offset=off, focal=focal, initial_rot=61.0)
        oc, _ = oracle.cli_camera(17, 40, camera_offset=off, focal=focal, initial_rot=61.0)
        _same_camera(pc, oc, oracle, f"offset {off} focal {focal}")
    # a pitched camera (the CLI never pitches; camera_t does)
    pc = scene.Camera((0.3, -0.2, -4.0), 16, 16, yaw=-70.0, pitch=25.0, focal=1.5)
    oc = oracle.camera((0.3, -0.2, -4.0), 16, 16, yaw=-70.0, pitch=25.0, focal=1.5)
    _same_camera(pc, oc, oracle, "pitched")
    # the orbit loop, 36 steps of 10 degrees and 360 steps of 1 degree (cfg5): every pose identical, not just the last
    for steps, deg in ((36, 10.0), (360, 1.0)):
        pc, ang = scene.cli_camera(8, 8)
        oc, oa = oracle.cli_camera(8, 8)
        for k in range(steps):
            ang = scene.orbit_step(pc, ang, deg)
            oracle.orbit_step(oc, oa, deg)
            assert np.float32(ang) == oa[0]
            _same_camera(pc, oc, oracle, f"orbit {deg} step {k}")
        assert np.abs(pc.position - np.array([0, 0, -4])).max() <= 1e-4   # incremental float rotation stays on the circle


def test_cxx_mirror_camera_is_the_oracle_camera_bit_for_bit(pkg, oracle, tmp_path):
    """vrt::camera_t of include/vrt/vrt.hpp, compiled HERE with -O3 -ffast-math -march=native (the reference's own
    flags, CMakeLists.txt:8 -- the worst case for a header): view matrix and plane arrays still equal the oracle's
    bit for bit, because the arithmetic is the library's."""
    import subprocess
    src = os.path.join(ROOT, "tests", "host", "camera_dump.cpp")
    exe = str(tmp_path / "camera_dump")
    libdir = os.path.dirname(pkg.LIB_PATH)
    subprocess.check_call(["g++", "-O3", "-ffast-math", "-march=native", "-std=c++17", "-I", os.path.join(ROOT, "include"), src,
                           "-o", exe, "-L", libdir, "-lvrt_hip", f"-Wl,-rpath,{libdir}"])
    for rot, steps in ((0.0, 0), (33.0, 0), (47.0, 3), (123.0, 0), (359.0, 2)):
        out = subprocess.check_output([exe, "12", "10", str(rot), str(steps), "7.5"])
        got = np.frombuffer(out, np.float32)
        oc, oa = oracle.cli_camera(12, 10, initial_rot=rot)
        for _ in range(steps):
            oracle.orbit_step(oc, oa, 7.5)
        ref = np.concatenate([np.array(oc.position[:], np.float32), oracle.camera_view(oc), *oracle.camera_plane(oc)])
        np.testing.assert_array_equal(got, ref, err_msg=f"rot {rot} steps {steps}")


def test_shard_table_partitions_the_frame(pkg):
    from sgrt_amd import sharding
    for tiles_w, tiles_h, world in [(16, 16, 1), (16, 16, 2), (16, 16, 8), (16, 16, 3), (10, 7, 4), (1, 1, 2)]:
        tab = sharding.shard_table(tiles_w, tiles_h, world)
        owned = tab[tab >= 0]
        assert sorted(owned.tolist()) == list(range(tiles_w * tiles_h))
        if tiles_w % world == 0:
            assert (tab >= 0).all() and tab.shape[1] == tiles_w * tiles_h // world
        img = np.arange(tiles_h * 4 * tiles_w * 8, dtype=np.uint32).reshape(tiles_h * 4, tiles_w * 8)
        shards = np.stack([sharding.extract_shard(img, tab, r, tiles_w, 8, 4) for r in range(world)])
        np.testing.assert_array_equal(sharding.assemble(shards, tab, tiles_w, 8, 4, *img.shape), img)
    # 8 GPUs, 16x16 tiles: ranks in 4 x 2 bricks, brick rows shifted by two tiles -- every 4 x 2 window of tiles (wherever it
    # lies) holds every rank once, so
    # an object that covers a few tiles in the middle of the frame is spread over all ranks
    tab = sharding.shard_table(16, 16, 8)
    owner = np.empty(256, np.int64)
    for r in range(8):
        owner[tab[r]] = r
    owner = owner.reshape(16, 16)
    for y0 in range(15):
        for x0 in range(13):
            assert sorted(owner[y0:y0 + 2, x0:x0 + 4].ravel().tolist()) == list(range(8))
    # the lit tiles of `-g 64 -w 2048` (4 x 4 tiles in the middle, 3/4/4/3 cells wide): the busiest rank carries 25 of the 196 cells
    cells = np.outer([3, 4, 4, 3], [3, 4, 4, 3])
    load = np.bincount(owner[6:10, 6:10].ravel(), weights=cells.ravel(), minlength=8)
    assert load.max() / load.sum() <= 25 / 196 + 1e-9        # 12.8 % (ideal 12.5 %)
    for world in (2, 4):
        tab = sharding.shard_table(16, 16, world)
        o = np.empty(256, np.int64)
        for r in range(world):
            o[tab[r]] = r
        load = np.bincount(o.reshape(16, 16)[6:10, 6:10].ravel(), weights=cells.ravel(), minlength=world)
        assert load.max() == load.min()


def test_bench_gather_batches_of_a_short_run():
    """bench.py, N > 1: the driver times 20 steps; a gather batch must then be a fraction of the run (round-2 verdict: one
    batch of 20 was a single un-overlapped render -> gather -> assemble chain)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for steps in (4, 12, 20, 40, 100, 127):
        F = bench.gather_batch_frames(32, steps)
        assert 1 <= F <= 32 and (steps + F - 1) // F >= 4, (steps, F)
    assert bench.gather_batch_frames(32, 400) == 32 and bench.gather_batch_frames(32, 20) == 5
    assert bench.gather_batch_frames(32, 1) == 1 and bench.gather_batch_frames(4, 400) == 4
