"""Frame batches: us per frame of -g 64 -w 2048 for n frames per launch (whole frames; and the shard of one of 8 ranks)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np
from conftest import load_pkg
pkg = load_pkg()
from sgrt_amd import scene
import torch
w = 2048
g = scene.grid_scene(64)
cam, _ = scene.cli_camera(w, w)
pack = pkg.PACK_ROUND | pkg.ALPHA_COMPUTED
tw = th = 2 / 16
def ctxs_for(n, shard):
    out = []
    for _ in range(n):
        r = pkg.Renderer(0)
        r.set_gaussians(g); r.set_options(pkg.EXP_VCL, pkg.ERF_AS, 1e-9); r.set_camera_view(w, w, cam.view); r.set_shard(*shard)
        out.append(r)
    return out
def wait(streams):
    evs = []
    for s in streams:
        e = torch.cuda.Event(); e.record(s); evs.append(e)
    for e in evs:
        while not e.query(): pass
    torch.cuda.synchronize()
for shard, kind in (((0, 1), 0),):
    for n, nstreams in ((1, 1), (3, 3), (4, 4), (8, 4), (16, 4), (8, 2), (16, 2), (12, 3), (24, 3), (32, 4)):
        per = max(1, n // nstreams)
        groups = [ctxs_for(per, shard) for _ in range(nstreams)]
        streams = [torch.cuda.Stream() for _ in range(nstreams)]
        st0 = streams[0].cuda_stream
        for gr in groups:
            for r in gr: r.tile_gaussians_device(tw, th, cam.view, st0)
        words = groups[0][0].sparse_shard_words() if kind == 2 else w * w
        bufs = [[torch.zeros(words, dtype=torch.int32, device="cuda") for _ in gr] for gr in groups]
        calls = [gr[0].frame_batch_call(gr[1:], tw, th, [cam.view] * per, [cam.position] * per, pack, out_kind=kind) for gr in groups]
        ptrs = [[b.data_ptr() for b in bb] for bb in bufs]
        def run(k):
            for i in range(k):
                j = i % nstreams
                calls[j](ptrs[j], streams[j].cuda_stream)
        run(40 * nstreams); wait(streams)
        reps = max(4 * nstreams, 1600 // (per))
        t0 = time.perf_counter(); run(reps); wait(streams); dt = time.perf_counter() - t0
        print(f"shard {shard} kind {kind}: {per} frames per launch x {nstreams} streams: {dt / (reps * per) * 1e6:7.2f} us per frame", flush=True)
        for gr in groups:
            for r in gr: r.close()
