"""VRT_HIP_TIMELINE of one `-g 64 -w 2048` frame (the library prints the per-phase stamps of the list kernel and the block kernel on stderr).
    VRT_HIP_TIMELINE=1 python3 tools/timeline.py [grid] [width]  2> profiles/rNN_timeline.txt"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
os.environ.setdefault("VRT_HIP_TIMELINE", "1")
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 64
w = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
from conftest import load_pkg
pkg = load_pkg()
from sgrt_amd import scene
cam, _ = scene.cli_camera(w, w)
r = pkg.Renderer(0)
r.set_gaussians(scene.grid_scene(grid)); r.set_camera_view(w, w, cam.view); r.tile_gaussians(2 / 16, 2 / 16, cam.view)
for _ in range(3):  # image only, like the frame path (16-byte clears); the lists are rebuilt per frame, as in a frame loop
    r.tile_gaussians(2 / 16, 2 / 16, cam.view); r.render(cam.position, want_radiance=False)
sys.stderr.write(f"# -g {grid} -w {w}: the frame below (third of three)\n"); sys.stderr.flush()
r.tile_gaussians(2 / 16, 2 / 16, cam.view)   # (same camera: the cone table's rows are found)
r.render(cam.position, want_radiance=False)
r.close()
