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


def test_host_camera_matches_oracle(pkg, oracle):
    from sgrt_amd import scene
    for rot in (0.0, 20.0, 123.0, 359.0):
        pc, pa = scene.cli_camera(32, 32, initial_rot=rot)
        oc, oa = oracle.cli_camera(32, 32, initial_rot=rot)
        assert abs(float(pa) - float(oa[0])) <= 1e-4
        for f in ("position", "front", "right", "up"):
            assert np.abs(getattr(pc, f) - np.array(getattr(oc, f)[:])).max() <= 1e-6, (rot, f)
        assert np.abs(pc.view - oracle.camera_view(oc)).max() <= 2e-6
        got = np.stack(pc.plane(), -1)
        ref = np.stack(oracle.camera_plane(oc), -1)
        assert np.abs(got - ref).max() <= 3e-6
    # orbit loop of main.cpp:330-334, 36 frames of 10 degrees: incremental float rotation stays on the circle
    pc, ang = scene.cli_camera(8, 8)
    oc, oa = oracle.cli_camera(8, 8)
    for _ in range(36):
        pc.orbit(10.0); ang = np.float32(ang - np.float32(10.0)); pc.turn(ang, 0.0)
        oracle.orbit_step(oc, oa, 10.0)
    assert np.abs(pc.position - np.array(oc.position[:])).max() <= 2e-5
    assert np.abs(pc.position - np.array([0, 0, -4])).max() <= 1e-4


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
    # 8 GPUs, 16x16 tiles: each rank gets 2 tiles of every tile row and every tile column (diagonal deal)
    tab = sharding.shard_table(16, 16, 8)
    for r in range(8):
        ty, tx = np.divmod(tab[r], 16)
        assert (np.bincount(ty, minlength=16) == 2).all() and (np.bincount(tx, minlength=16) == 2).all()
