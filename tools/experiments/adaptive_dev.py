"""Deviation of the frame from the full sum (cull_eps = 0) with the per-tile slack on / off, and the frame times."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np
from conftest import load_pkg, GOLDEN
pkg = load_pkg()
from sgrt_amd import scene
import torch
OBJ = os.path.join(GOLDEN, "test-objects")
scenes = [("g64", scene.grid_scene(64), 1024, 0.0), ("g64r", scene.grid_scene(64), 1024, 47.0), ("g16", scene.grid_scene(16), 512, 0.0), ("g64w", scene.grid_scene(64), 2048, 0.0),
          ("teapot", scene.read_obj(os.path.join(OBJ, "teapot.obj")), 512, 0.0), ("monkey", scene.read_obj(os.path.join(OBJ, "monkey.obj")), 512, 20.0),
          ("cube", scene.read_obj(os.path.join(OBJ, "cube.obj")), 512, 30.0)]
def frame(r, g, w, rot, eps):
    cam, _ = scene.cli_camera(w, w, initial_rot=rot)
    r.set_gaussians(g); r.set_camera_view(w, w, cam.view)
    r.set_options(pkg.EXP_VCL, pkg.ERF_AS, eps)
    r.tile_gaussians(2/16, 2/16, cam.view)
    img, rad = r.render(cam.position, want_radiance=True)
    img_t = torch.zeros(w*w, dtype=torch.int32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        r.tile_gaussians_device(2/16, 2/16, cam.view, s)
        r.render_device(cam.position, pkg.PACK_ROUND | pkg.ALPHA_COMPUTED, img_t.data_ptr(), 0, s)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    return img, rad, dt
res = {}
for mode in ("1365.33", "4096", "0"):
    os.environ["VRT_HIP_CULL_REF_N"] = mode
    r = pkg.Renderer(0)
    for name, g, w, rot in scenes:
        img, rad, dt = frame(r, g, w, rot, 1e-9)
        if mode == "1365.33":
            res[name] = frame(r, g, w, rot, 0.0)[:2]
        full_img, full_rad = res[name]
        d = np.abs(rad.astype(np.float64) - full_rad.astype(np.float64))
        du8 = np.abs(img.view(np.uint8).astype(np.int16) - full_img.view(np.uint8).astype(np.int16))
        print(f"adaptive={mode} {name:7s} w={w} frame {dt*1e3:8.3f} ms  max |rad - full| {d.max():.3e}  mean {d.mean():.3e}  u8 differing {int((du8>0).sum())} max {int(du8.max())}", flush=True)
    r.close()
