"""GPU test of the host side above the C ABI: the drop-in CLI and the vrt:: C++ mirror."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu
BIN = os.path.join(ROOT, "simd-gaussian-ray-tracing_amd", "bin")


def run(args, cwd):
    return subprocess.run(args, cwd=cwd, capture_output=True, text=True, timeout=300)


def test_cli_single_frame_matches_oracle(tmp_path, oracle):
    """volumetric-ray-tracer -g 4 -w 256 -q -o out.png: TIME line + PNG bytes == the reference semantics
    (mode 8: rounding pack, computed alpha, u32 A|R|G|B stored little-endian as 'RGBA' -> B,G,R,A)."""
    from PIL import Image
    out = tmp_path / "out.png"
    p = run([os.path.join(BIN, "volumetric-ray-tracer"), "-g", "4", "-w", "256", "-q", "--plane-arrays", "-o", str(out)], tmp_path)
    assert p.returncode == 0, p.stderr
    assert re.fullmatch(r"TIME: [0-9.e+-]+ ms\n", p.stdout), p.stdout
    png = np.array(Image.open(out))
    assert png.shape == (256, 256, 4)
    g = oracle.grid_scene(4)
    cam, _ = oracle.cli_camera(256, 256)
    tiles = oracle.tile_gaussians(2 / 16, 2 / 16, g, oracle.camera_view(cam))
    img, _ = oracle.render(256, 256, oracle.camera_plane(cam), cam.position[:], g, tiles)
    ref = np.stack([(img >> s) & 255 for s in (0, 8, 16, 24)], -1).reshape(256, 256, 4).astype(np.int32)  # bytes B,G,R,A
    assert np.abs(png.astype(np.int32) - ref).max() <= 1
    assert png[..., 3].max() > 40 and png[..., 0].max() > 40   # alpha and the byte-0 (blue) channel carry signal


def test_cli_frames_and_modes(tmp_path):
    """--frames N writes <stem>_<k>.png and prints AVG. TIME (main.cpp:299-315); -f loads an OBJ; -m selects a mode."""
    obj = os.path.join(GOLDEN, "test-objects", "sphere.obj")
    p = run([os.path.join(BIN, "volumetric-ray-tracer"), "-f", obj, "-w", "64", "-q", "--frames", "3", "-r", "90", "--tiles", "4",
             "-o", "orbit.png"], tmp_path)
    assert p.returncode == 0, p.stderr
    assert re.fullmatch(r"AVG\. TIME: [0-9.e+-]+ ms \(3 frames\)\n", p.stdout), p.stdout
    assert sorted(os.listdir(tmp_path)) == ["orbit_1.png", "orbit_2.png", "orbit_3.png"]
    from PIL import Image
    a, b = (np.array(Image.open(tmp_path / f"orbit_{k}.png")) for k in (1, 3))
    assert (a != b).any() and a[..., 3].max() > 0
    for mode in ("1", "4", "5"):
        q = run([os.path.join(BIN, "volumetric-ray-tracer"), "-g", "2", "-w", "32", "-q", "-m", mode, "-o", f"m{mode}.png"], tmp_path)
        assert q.returncode == 0 and q.stdout.startswith("TIME: "), q.stderr
        img = np.array(Image.open(tmp_path / f"m{mode}.png"))
        assert (img[..., 3] == 255).all()      # scalar / untiled modes write opaque alpha (rt.h:239, 279, 329)


def test_cpp_mirror_example_runs(tmp_path):
    """The reference's transmittance experiment and render calls, compiled against include/vrt/vrt.hpp."""
    p = run([os.path.join(BIN, "api_example")], tmp_path)
    assert p.returncode == 0, p.stderr
    lines = p.stdout.strip().splitlines()
    assert lines[0] == "s, T, T_s, err, D" and len(lines) == 21
    first = [float(v) for v in lines[1].split(",")]
    assert abs(first[1] - 1.0) < 0.2            # T near 1 for the sample point nearest the origin
    last = [float(v) for v in lines[13].split(",")]
    assert 0 < last[1] < first[1]               # transmittance decreases along the ray
    assert lines[14].startswith("radiance ") and lines[15].startswith("aborted: 0")
    # broadcast_transmittance / broadcast_radiance (rt.h:102-127, 205-223): lane l of the W-wide call == the one-ray call
    lanes = [ln for ln in lines if ln.startswith("lane ")]
    assert len(lanes) == 4
    for ln in lanes:
        v = [float(x) for x in re.findall(r"[-+]?\d[-+0-9.e]*", ln.split(":", 1)[1])]
        assert len(v) == 10 and v[0] == v[1] and v[2:6] == v[6:10], ln
        assert 0 < v[0] <= 1.0001
    # a tiles_t survives an untiled render and another scene's tiling (stale tiles_t rendered silently wrong)
    m = re.fullmatch(r"tiles_t reuse: (\d+) pixels differ from the first tiled image \(untiled: (\d+), other scene: (\d+)\)", lines[-1])
    assert m and int(m.group(1)) == 0 and int(m.group(2)) > 0 and int(m.group(3)) > 0, lines[-1]
