import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
from conftest import load_pkg, GOLDEN
pkg = load_pkg()
from sgrt_amd import scene
import torch
r = pkg.Renderer(0)
def run(name, g, w, eps=1e-9, reps=5):
    cam, _ = scene.cli_camera(w, w)
    r.set_gaussians(g); r.set_camera_view(w, w, cam.view)
    r.set_options(pkg.EXP_VCL, pkg.ERF_AS, eps)
    r.tile_gaussians(2/16, 2/16, cam.view)
    r.enable_stats(True)
    img, _ = r.render(cam.position, want_radiance=False)
    st = r.stats(); r.enable_stats(False)
    sb = max(st["shaded_blocks"], 1)
    img_t = torch.zeros(w*w, dtype=torch.int32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        r.tile_gaussians_device(2/16, 2/16, cam.view, s)
        r.render_device(cam.position, pkg.PACK_ROUND | pkg.ALPHA_COMPUTED, img_t.data_ptr(), 0, s)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    rays_l = st["lane_entries"]
    print(f"{name:10s} w={w} eps={eps:g} frame {dt*1e3:8.3f} ms  {w*w/dt/1e6:9.1f} Mray/s | shaded blocks {sb} of {st['blocks']}, block list {st['list_entries']/sb:.1f}, cell list {st['tile_entries']/sb:.1f}, ray list {rays_l/(sb*64):.1f}, longest {st['lane_max_entries']/sb:.1f}, dense blocks {st['dense_blocks']} (>1024 survivors: {st['overflow_blocks']}) busy {st['dense_busy_frac']:.2f}, dense visits full/zero/-2A {st['dense_visits_full']:.3e}/{st['dense_visits_zero']:.3e}/{st['dense_visits_common']:.3e}")
teapot = scene.read_obj(os.path.join(GOLDEN, "test-objects", "teapot.obj"))
monkey = scene.read_obj(os.path.join(GOLDEN, "test-objects", "monkey.obj"))
cube = scene.read_obj(os.path.join(GOLDEN, "test-objects", "cube.obj"))
run("g64", scene.grid_scene(64), 2048, reps=20)
run("g16", scene.grid_scene(16), 1024, reps=20)
run("teapot", teapot, 2048)
run("teapot", teapot, 2048, eps=1e-7)
run("monkey", monkey, 4096, reps=3)
run("cube", cube, 256, reps=20)
