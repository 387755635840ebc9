"""Table mode against the exact kernels on the OBJ scenes: frame time, deviation of the float radiance, what the table
kernel did (nodes per block, retries at the reduced spacing, declined blocks).  python tools/table_check.py [step ...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
from conftest import load_pkg, GOLDEN
pkg = load_pkg()
from sgrt_amd import scene
import torch
steps = [float(a) for a in sys.argv[1:]] or [0.05]
r = pkg.Renderer(0)

def frame(g, w, cam, step, budget=None, reps=3):
    r.set_gaussians(g); r.set_camera_view(w, w, cam.view)
    r.set_options(pkg.EXP_VCL, pkg.ERF_AS, 1e-9)
    r.set_table_step(step)
    if budget: r.set_table_budget(budget)
    r.tile_gaussians(2 / 16, 2 / 16, cam.view)
    r.enable_stats(True)
    img, rad = r.render(cam.position, want_radiance=True)
    st = r.stats(); r.enable_stats(False)
    img_t = torch.zeros(w * w, dtype=torch.int32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        r.tile_gaussians_device(2 / 16, 2 / 16, cam.view, s)
        r.render_device(cam.position, pkg.PACK_ROUND | pkg.ALPHA_COMPUTED, img_t.data_ptr(), 0, s)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        r.tile_gaussians_device(2 / 16, 2 / 16, cam.view, s)
        r.render_device(cam.position, pkg.PACK_ROUND | pkg.ALPHA_COMPUTED, img_t.data_ptr(), 0, s)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    return rad, img, st, dt

for name, w, rot in (("teapot", 2048, 0.0), ("monkey", 4096, 0.0), ("monkey", 4096, 180.0), ("cube", 1024, 30.0)):
    g = scene.read_obj(os.path.join(GOLDEN, "test-objects", name + ".obj"))
    cam, _ = scene.cli_camera(w, w, initial_rot=rot)
    rad0, img0, st0, dt0 = frame(g, w, cam, 0.0)
    print(f"{name} {w}^2 rot {rot}: exact {dt0*1e3:8.3f} ms, dense blocks {st0['dense_blocks']}", flush=True)
    for step in steps:
        rad, img, st, dt = frame(g, w, cam, step)
        d = np.abs(rad - rad0)
        ch = ((img.reshape(-1)[:, None] >> np.array([0, 8, 16, 24])) & 255).astype(int) - ((img0.reshape(-1)[:, None] >> np.array([0, 8, 16, 24])) & 255).astype(int)
        tb = max(st["table_blocks"], 1)
        print(f"   step {step:5.3f}: {dt*1e3:8.3f} ms ({dt0/dt:4.1f}x)  max |d radiance| {d.max():.2e}  u8 steps {np.abs(ch).max()} ({(ch != 0).sum()} values)"
              f" | table blocks {st['table_blocks']} of {st['dense_blocks']} dense, declined {st['table_declined']}, second attempts {st['table_retries']},"
              f" phases us/block {[round(t*0.01/tb,1) for t in st['table_phase_ticks']]} coarser than requested {st['table_coarser']}, nodes per block {st['table_nodes']/tb:.0f}, saturated visits {st['table_skips']:.3e}", flush=True)
