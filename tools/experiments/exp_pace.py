import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
from conftest import load_pkg
pkg = load_pkg()
import torch
from sgrt_amd import scene
W = 2048
g = scene.grid_scene(64); cam, _ = scene.cli_camera(W, W); pack = pkg.PACK_ROUND | pkg.ALPHA_COMPUTED; tw = th = 2/16
def mk():
    r = pkg.Renderer(0); r.set_gaussians(g); r.set_options(pkg.EXP_VCL, pkg.ERF_AS, 1e-9); r.set_camera_view(W, W, cam.view)
    r.tile_gaussians_device(tw, th, cam.view, 0); return r
def wait(streams):
    evs = []
    for s in streams:
        e = torch.cuda.Event(); e.record(s); evs.append(e)
    for e in evs:
        while not e.query(): pass
    torch.cuda.synchronize()
for nctx in (3, 4):
    ctxs = [mk() for _ in range(nctx)]
    imgs = [torch.zeros(W * W, dtype=torch.int32, device="cuda") for _ in range(nctx)]
    calls = [c.frame_call(tw, th, cam.view, cam.position, pack, shard=False) for c in ctxs]
    ss = [torch.cuda.Stream() for _ in range(nctx)]
    def run(k, pace_cycles=0):
        for i in range(k):
            j = i % nctx
            if pace_cycles and i < nctx and j:
                with torch.cuda.stream(ss[j]):
                    torch.cuda._sleep(int(pace_cycles * j))
            calls[j](imgs[j].data_ptr(), ss[j].cuda_stream)
    run(2000); wait(ss)
    for pace_us in (0, 3, 6, 9, 12):
        res = []
        for rep in range(5):
            run(5); wait(ss)
            t0 = time.perf_counter(); run(20, pace_us * 2347); wait(ss)    # 1e6 _sleep cycles = 426 us
            res.append((time.perf_counter() - t0) / 20 * 1e6)
        print(f"contexts {nctx} pace {pace_us}: K=20 us per frame {sorted(res)}", flush=True)
    # calibrate _sleep
    torch.cuda.synchronize(); t0 = time.perf_counter(); torch.cuda._sleep(1000000); torch.cuda.synchronize(); print("1e6 sleep cycles =", (time.perf_counter() - t0) * 1e6, "us")
    for c in ctxs: c.close()
