"""One GPU plays, one after the other, every rank of an N-GPU tile-sharded job of `-g 64 -w 2048` (no collective: the shards
a gather would deliver are rendered once and kept) and reports the time per frame each rank needs for ITS part:

  frame by frame   vrt_hip_frame_sparse_device per frame, one context (what bench.py did for N > 1 until round 2)
  batches          vrt_hip_frame_batch_device: F frames per launch of each kernel, three groups of F contexts on three streams
  + assembly       rank 0 only: the F frames of a batch assembled from the N shards, plain (background fill + cells, frame
                   by frame) and batched + retained (one launch per batch, only cells that went dark are reset)

The slowest rank sets the frame rate of the job; `bound` = frame time on one GPU / that.  What this cannot show is the
RCCL gather itself (N - 1 prefixes of ~0.1 MB per frame, once per batch, on its own stream).

    python tools/emulate_ranks.py [F=16] > profiles/rNN_multigpu_emulation.md
"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from conftest import load_pkg  # noqa: E402

pkg = load_pkg()
import torch  # noqa: E402
from sgrt_amd import scene  # noqa: E402

W = 2048
F = int(sys.argv[1]) if len(sys.argv) > 1 else 16
g = scene.grid_scene(64)
cam, _ = scene.cli_camera(W, W)
pack = pkg.PACK_ROUND | pkg.ALPHA_COMPUTED
tw = th = 2 / 16
st = torch.cuda.current_stream().cuda_stream


def mk(rank, world):
    r = pkg.Renderer(0)
    r.set_gaussians(g)
    r.set_options(pkg.EXP_VCL, pkg.ERF_AS, 1e-9)
    r.set_camera_view(W, W, cam.view)
    r.set_shard(rank, world)
    r.tile_gaussians_device(tw, th, cam.view, st)
    return r


def wait(streams):
    evs = []
    for s in streams:
        e = torch.cuda.Event()
        e.record(s)
        evs.append(e)
    for e in evs:
        while not e.query():
            pass
    torch.cuda.synchronize()


def timed(fn, frames_per_call, settle, reps, streams):
    t_settle = time.perf_counter()      # at least `settle` calls and 40 ms: the dense-launch feedback settles, the clocks ramp up
    i = 0
    while i < settle or time.perf_counter() - t_settle < 0.04:
        fn()
        i += 1
        if i % 8 == 0:
            wait(streams)
    wait(streams)
    best = None
    for _ in range(2):                  # best of two: a box shared with nobody still shows an outlier now and then
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        wait(streams)
        dt = (time.perf_counter() - t0) / (reps * frames_per_call) * 1e6
        best = dt if best is None else min(best, dt)
    return best


# the frame on one GPU: three contexts in flight (bench.py's N = 1)
ctxs = [mk(0, 1) for _ in range(3)]
ss = [torch.cuda.Stream() for _ in range(3)]
imgs = [torch.zeros(W * W, dtype=torch.int32, device="cuda") for _ in range(3)]
calls = [c.frame_call(tw, th, cam.view, cam.position, pack, shard=False) for c in ctxs]
k = [0]


def one_gpu():
    i = k[0] % 3
    k[0] += 1
    calls[i](imgs[i].data_ptr(), ss[i].cuda_stream)


t1 = timed(one_gpu, 1, 60, 1500, ss)
for c in ctxs:
    c.close()
print(f"# `-g 64 -w 2048` tile-sharded over N ranks, every rank played by one MI355X (tools/emulate_ranks.py, F = {F})\n")
print(f"One GPU, whole frames, three in flight: **{t1:.2f} us per frame**.\n")
print("| N | rank | lit cells | frame by frame | batches of F | + plain assembly | + batched retained assembly |")
print("|---|---|---|---|---|---|---|")
for world in (2, 4, 8):
    shards = []
    for rk in range(world):
        r = mk(rk, world)
        words = r.sparse_shard_words()
        b = torch.zeros(words, dtype=torch.int32, device="cuda")
        r.frame_sparse_call(tw, th, cam.view, cam.position, pack)(b.data_ptr(), st)
        torch.cuda.synchronize()
        shards.append(b)
        r.close()
    cells = [int(b[0]) for b in shards]
    prefix = ((4 + int(shards[0][1]) + 3) // 4 * 4) + 1024 * max(cells)
    recv = [b[:prefix].repeat(F) for b in shards]           # [rank][frame][prefix]: every frame the same shard
    worst = 0.0
    for rank in range(world):
        single = mk(rank, world)
        sbuf = torch.zeros(words, dtype=torch.int32, device="cuda")
        sc = single.frame_sparse_call(tw, th, cam.view, cam.position, pack)
        t_single = timed(lambda: sc(sbuf.data_ptr(), st), 1, 30, 600, [torch.cuda.current_stream()])
        single.close()
        groups = [[mk(rank, world) for _ in range(F)] for _ in range(3)]
        rs = [torch.cuda.Stream() for _ in range(3)]
        asm = torch.cuda.Stream()
        bufs = [[torch.zeros(words, dtype=torch.int32, device="cuda") for _ in range(F)] for _ in range(3)]
        bcalls = [gr[0].frame_batch_call(gr[1:], tw, th, [cam.view] * F, [cam.position] * F, pack, out_kind=2) for gr in groups]
        bp = [[b.data_ptr() for b in bb] for bb in bufs]
        img = torch.zeros(W * W, dtype=torch.int32, device="cuda")
        fimgs = [torch.zeros(W * W, dtype=torch.int32, device="cuda") for _ in range(F)]
        n = [0]

        def batch(mode):
            b = n[0] % 3
            n[0] += 1
            bcalls[b](bp[b], rs[b].cuda_stream)
            if mode == "plain":
                for f in range(F):
                    groups[0][0].scatter_sparse_device([t.data_ptr() + 4 * f * prefix for t in recv], pack, img.data_ptr(), asm.cuda_stream)
            elif mode == "batched":
                groups[0][0].scatter_sparse_batch_device([t.data_ptr() for t in recv], prefix, F, pack, [im.data_ptr() for im in fimgs],
                                                         asm.cuda_stream, retained=True)

        t_batch = timed(lambda: batch(None), F, 15, 960 // F, rs + [asm])
        # once more on three other streams: which hardware queues a stream triple lands on depends on what the process created
        # before ("Which streams"); bench.py probes for that, here the better of two triples counts
        rs[:] = [torch.cuda.Stream() for _ in range(3)]
        t_batch = min(t_batch, timed(lambda: batch(None), F, 15, 960 // F, rs + [asm]))
        t_plain = t_ret = None
        if rank == 0:
            t_plain = timed(lambda: batch("plain"), F, 6, 480 // F, rs + [asm])
            t_ret = timed(lambda: batch("batched"), F, 6, 960 // F, rs + [asm])
        worst = max(worst, t_ret if rank == 0 else t_batch)
        print(f"| {world} | {rank} | {cells[rank]} | {t_single:.2f} | {t_batch:.2f} | {'' if t_plain is None else f'{t_plain:.2f}'} | "
              f"{'' if t_ret is None else f'{t_ret:.2f}'} |", flush=True)
        for gr in groups:
            for r in gr:
                r.close()
    print(f"| {world} | **slowest** | | | | | **{worst:.2f} us -> bound {t1 / worst:.1f}x** |", flush=True)
