# Workload for the PMC passes: 8 frames of the bench configuration (-g 64 -w 2048) followed by 4 frames of a scene
# whose only Gaussian is behind the camera (every cell inactive: the list kernel clears exactly w*h*4 bytes with the
# same 4-byte-per-lane stores) -- the calibration point for WRITE_SIZE in this access pattern.
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
from conftest import load_pkg
pkg = load_pkg()
from sgrt_amd import scene
w = 2048
r = pkg.Renderer(0)
cam, _ = scene.cli_camera(w, w)
r.set_camera_view(w, w, cam.view)
img = torch.zeros(w * w, dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
pack = pkg.PACK_ROUND | pkg.ALPHA_COMPUTED
r.set_gaussians(scene.grid_scene(64))
f = r.frame_call(2 / 16, 2 / 16, cam.view, cam.position, pack)
for _ in range(8):
    f(img.data_ptr(), s)
torch.cuda.synchronize()
g = scene.grid_scene(1)
g["mu"][:, 2] = -10.0      # behind the camera: in no tile (rt.cpp:38)
r.set_gaussians(g)
f = r.frame_call(2 / 16, 2 / 16, cam.view, cam.position, pack)
for _ in range(4):
    f(img.data_ptr(), s)
torch.cuda.synchronize()
