"""GPU tests of frame batches (vrt_hip_frame_batch_device): n frames, n contexts, every kernel launched once with the
frame as grid.y -- the result of each frame must be what vrt_hip_frame_device / vrt_hip_frame_sparse_device gives for it
alone, bit for bit: moving cameras, sparse and dense scenes, raster frames, compact and sparse shards, batches back to
back (the argument ring, the counter generations), a shorter batch after a longer one; and the refusals."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
OBJ = os.path.join(GOLDEN, "test-objects")


def make_contexts(pkg, n, g, w, h, cam0, shard=(0, 1), table=None):
    ctxs = []
    for _ in range(n):
        r = pkg.Renderer(0)
        r.set_gaussians(g)
        r.set_options(pkg.EXP_VCL, pkg.ERF_AS, 1e-9)
        if table is not None:
            r.set_table_step(table)
        r.set_camera_view(w, h, cam0.view)
        r.set_shard(*shard)
        ctxs.append(r)
    return ctxs


@pytest.mark.parametrize("name,w,h,tiles_n,n,table", [("g64", 768, 768, 16, 5, None), ("g16", 512, 384, 8, 16, None), ("monkey", 256, 256, 8, 3, None),
                                                       ("monkey", 256, 256, 8, 3, 0.0), ("cube", 200, 136, 5, 4, None), ("cube", 200, 136, 5, 4, 0.0)])
def test_batch_equals_single_frames(pkg, renderer, name, w, h, tiles_n, n, table):
    """table = None: the library's default (dense blocks through the table kernel and, behind it, the exact kernel for what it
    declines -- both batched); 0.0: the exact kernels only."""
    import torch
    from sgrt_amd import scene
    g = {"g64": lambda: scene.grid_scene(64), "g16": lambda: scene.grid_scene(16),
         "monkey": lambda: scene.read_obj(os.path.join(OBJ, "monkey.obj")),
         "cube": lambda: scene.read_obj(os.path.join(OBJ, "cube.obj"))}[name]()
    pack = pkg.PACK_ROUND | pkg.ALPHA_COMPUTED
    cams = [scene.cli_camera(w, h, initial_rot=7.0 * i)[0] for i in range(2 * n)]
    tw = th = 2.0 / tiles_n
    # reference: every frame on its own, one context
    renderer.set_gaussians(g)
    renderer.set_options(pkg.EXP_VCL, pkg.ERF_AS, 1e-9)
    if table is not None:
        renderer.set_table_step(table)
    renderer.set_shard(0, 1)
    want = []
    for c in cams:
        renderer.set_camera_view(w, h, c.view)
        renderer.tile_gaussians(tw, th, c.view)
        want.append(renderer.render(c.position, pack, want_radiance=False)[0].reshape(-1))
    st = torch.cuda.current_stream().cuda_stream
    ctxs = make_contexts(pkg, n, g, w, h, cams[0], table=table)
    try:
        # (pixels no tile covers -- 136 rows in 5 tiles of 27 -- are never written: 0 as in vrt_hip_frame's own buffer)
        fill = 0 if name == "cube" else 0x55
        outs = [torch.full((w * h,), fill, dtype=torch.int32, device="cuda") for _ in range(n)]
        ptrs = [o.data_ptr() for o in outs]
        for half in range(2):                       # two batches back to back, other cameras in the second
            cs = cams[half * n:(half + 1) * n]
            for r, c in zip(ctxs, cs):
                r.set_camera_view(w, h, c.view)
            f = ctxs[0].frame_batch_call(ctxs[1:], tw, th, [c.view for c in cs], [c.position for c in cs], pack)
            f(ptrs, st)
            if half == 0:
                f(ptrs, st)                         # the same batch again right behind (argument ring, counter generations)
            torch.cuda.synchronize()
            for i in range(n):
                np.testing.assert_array_equal(outs[i].cpu().numpy().view(np.uint32), want[half * n + i], err_msg=f"batch {half} frame {i}")
        # a shorter batch through the same contexts
        m = max(1, n // 2)
        for o in outs:
            o.fill_(0 if name == "cube" else 0x33)
        f = ctxs[0].frame_batch_call(ctxs[1:m], tw, th, [c.view for c in cams[n:n + m]], [c.position for c in cams[n:n + m]], pack)
        f(ptrs, st)
        torch.cuda.synchronize()
        for i in range(m):
            np.testing.assert_array_equal(outs[i].cpu().numpy().view(np.uint32), want[n + i])
    finally:
        for r in ctxs:
            r.close()


@pytest.mark.parametrize("name,w,tiles_n,world,rank", [("g64", 1024, 16, 8, 3), ("monkey", 256, 8, 3, 1), ("g4", 96, 2, 8, 7)])
def test_batch_of_sparse_shards(pkg, renderer, name, w, tiles_n, world, rank):
    """out_kind 2: the shard of rank `rank` for n frames in one batch == vrt_hip_frame_sparse_device frame by frame
    (header, keys and the pixels of every stored cell)."""
    import torch
    from sgrt_amd import scene
    g = {"g64": lambda: scene.grid_scene(64), "g4": lambda: scene.grid_scene(4),
         "monkey": lambda: scene.read_obj(os.path.join(OBJ, "monkey.obj"))}[name]()
    pack = pkg.PACK_ROUND | pkg.ALPHA_COMPUTED
    n = 4
    cams = [scene.cli_camera(w, w, initial_rot=11.0 * i)[0] for i in range(n)]
    tw = th = 2.0 / tiles_n
    st = torch.cuda.current_stream().cuda_stream
    ctxs = make_contexts(pkg, n + 1, g, w, w, cams[0], shard=(rank, world))
    try:
        single = ctxs[n]
        single.tile_gaussians_device(tw, th, cams[0].view, st)
        words = single.sparse_shard_words()
        want = []
        for c in cams:
            single.set_camera_view(w, w, c.view)
            buf = torch.zeros(words, dtype=torch.int32, device="cuda")
            single.frame_sparse_call(tw, th, c.view, c.position, pack)(buf.data_ptr(), st)
            torch.cuda.synchronize()
            want.append(buf.cpu().numpy().view(np.uint32))
        for r, c in zip(ctxs[:n], cams):
            r.set_camera_view(w, w, c.view)
            r.tile_gaussians_device(tw, th, c.view, st)
        bufs = [torch.zeros(words, dtype=torch.int32, device="cuda") for _ in range(n)]
        f = ctxs[0].frame_batch_call(ctxs[1:n], tw, th, [c.view for c in cams], [c.position for c in cams], pack, out_kind=2)
        f([b.data_ptr() for b in bufs], st)
        torch.cuda.synchronize()
        stored = 0
        for i in range(n):
            got = bufs[i].cpu().numpy().view(np.uint32)
            cells, cap = int(want[i][0]), int(want[i][1])
            np.testing.assert_array_equal(got[:4], want[i][:4])
            off = (4 + cap + 3) // 4 * 4
            # which slot a cell gets is decided by the order in which tiles file their cells (atomics): compare by key
            def by_key(buf):
                order = np.argsort(buf[4:4 + cells], kind="stable")
                return buf[4:4 + cells][order], buf[off:off + 1024 * cells].reshape(cells, 1024)[order]
            gk, gp = by_key(got)
            wk, wp = by_key(want[i])
            np.testing.assert_array_equal(gk, wk)
            np.testing.assert_array_equal(gp, wp)
            stored += cells
        assert stored > 0 or name == "g4"                  # g4: rank 7 of 8 owns none of the four tiles
    finally:
        for r in ctxs:
            r.close()


def test_batch_of_compact_shards(pkg, renderer):
    """out_kind 1: rank 1 of 3's compact shard [slot][tile_h][tile_w] for n frames in one batch == vrt_hip_frame_device(..,
    shard = 1) frame by frame."""
    import torch
    from sgrt_amd import scene
    g = scene.grid_scene(16)
    w, tiles_n, n = 512, 8, 3
    pack = pkg.PACK_ROUND | pkg.ALPHA_COMPUTED
    cams = [scene.cli_camera(w, w, initial_rot=17.0 * i)[0] for i in range(n)]
    tw = th = 2.0 / tiles_n
    st = torch.cuda.current_stream().cuda_stream
    ctxs = make_contexts(pkg, n + 1, g, w, w, cams[0], shard=(1, 3))
    try:
        single = ctxs[n]
        single.tile_gaussians_device(tw, th, cams[0].view, st)
        px = single.shard_pixels()
        assert px > 0
        want = []
        for c in cams:
            single.set_camera_view(w, w, c.view)
            buf = torch.full((px,), 0x44, dtype=torch.int32, device="cuda")
            single.frame_call(tw, th, c.view, c.position, pack, shard=True)(buf.data_ptr(), st)
            torch.cuda.synchronize()
            want.append(buf.cpu().numpy())
        for r, c in zip(ctxs[:n], cams):
            r.set_camera_view(w, w, c.view)
        bufs = [torch.full((px,), 0x44, dtype=torch.int32, device="cuda") for _ in range(n)]
        ctxs[0].frame_batch_call(ctxs[1:n], tw, th, [c.view for c in cams], [c.position for c in cams], pack, out_kind=1)([b.data_ptr() for b in bufs], st)
        torch.cuda.synchronize()
        for i in range(n):
            np.testing.assert_array_equal(bufs[i].cpu().numpy(), want[i], err_msg=f"frame {i}")
        assert any(int((w_ != 0x44).sum()) > 0 and int((w_ & 0xFFFFFF).max()) > 0 for w_ in want)
    finally:
        for r in ctxs:
            r.close()


def test_batch_refusals(pkg, renderer):
    import torch
    from sgrt_amd import scene
    g = scene.grid_scene(8)
    w = 128
    cam, _ = scene.cli_camera(w, w)
    ctxs = make_contexts(pkg, 2, g, w, w, cam)
    out = [torch.zeros(w * w, dtype=torch.int32, device="cuda") for _ in range(2)]
    ptrs = [o.data_ptr() for o in out]
    try:
        # frames before the refusals: the contexts' list and queue generations are in use (a refused batch must leave them
        # in a state from which the next frame is right)
        for r_, o_ in zip(ctxs, out):
            for _ in range(3):
                r_.frame_call(2 / 4, 2 / 4, cam.view, cam.position, 0)(o_.data_ptr(), 0)
        torch.cuda.synchronize()
        before = out[0].clone()
        assert int((before != 0).sum()) > 0 and bool((out[0] == out[1]).all())
        with pytest.raises(pkg.VrtHipError, match="own"):     # one context cannot hold two frames
            ctxs[0].frame_batch_call([ctxs[0]], 2 / 4, 2 / 4, [cam.view] * 2, [cam.position] * 2, 0)(ptrs, 0)
        ctxs[1].set_options(pkg.EXP_LIBM, pkg.ERF_LIBM, 1e-9)
        with pytest.raises(pkg.VrtHipError, match="differ"):
            ctxs[0].frame_batch_call([ctxs[1]], 2 / 4, 2 / 4, [cam.view] * 2, [cam.position] * 2, 0)(ptrs, 0)
        ctxs[1].set_options(pkg.EXP_VCL, pkg.ERF_AS, 1e-9)
        ctxs[1].set_camera_view(64, 64, cam.view)
        with pytest.raises(pkg.VrtHipError, match="differ"):
            ctxs[0].frame_batch_call([ctxs[1]], 2 / 4, 2 / 4, [cam.view] * 2, [cam.position] * 2, 0)(ptrs, 0)
        ctxs[1].set_camera_view(w, w, cam.view)
        ctxs[1].set_table_step(0.1)                            # the frames of a batch share ONE table setting
        with pytest.raises(pkg.VrtHipError, match="differ"):
            ctxs[0].frame_batch_call([ctxs[1]], 2 / 4, 2 / 4, [cam.view] * 2, [cam.position] * 2, 0)(ptrs, 0)
        ctxs[1].set_table_step(ctxs[0].table_step)
        # single frames right after the refused batch (its contexts had advanced their generations for kernels that never ran)
        for r_, o_ in zip(ctxs, out):
            o_.zero_()
            r_.frame_call(2 / 4, 2 / 4, cam.view, cam.position, 0)(o_.data_ptr(), 0)
        torch.cuda.synchronize()
        assert bool((out[0] == before).all()) and bool((out[1] == before).all())
        # and it still works afterwards
        ctxs[0].frame_batch_call([ctxs[1]], 2 / 4, 2 / 4, [cam.view] * 2, [cam.position] * 2, 0)(ptrs, 0)
        torch.cuda.synchronize()
        assert bool((out[0] == before).all()) and bool((out[1] == before).all())
    finally:
        for r in ctxs:
            r.close()
