"""Multi-process (world_size 2, 3 and 8, gloo, CPU) test of the N > 1 path: tile shards -> gather -> assemble.

The shard map and buffer layout are the ones csrc/vrt_hip_api.cpp uses on the GPUs (mirrored on the
host in sharding.py; their equality with the device code is checked on the GPU by
test_gpu_parity.py::test_sharded_render_assembles_to_full_frame).  Here every rank produces its shard
of a small frame with the CPU oracle standing in for the HIP render, the shards travel through the same
torch.distributed gather bench.py issues over RCCL, and rank 0 must end up with the single-process frame.
The second test drives sharding.FrameGatherer -- the batched, double-buffered frame loop bench.py runs for N > 1 --
with frames that differ from one another, so that a frame assembled from the wrong batch slot would show.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from conftest import load_pkg
    load_pkg()
    from sgrt_amd import scene, sharding
    import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        w = h = 64
        tiles_n = 4
        g = scene.grid_scene(4).view(O.GAUSSIAN)
        cam, _ = scene.cli_camera(w, h)
        plane = cam.plane()
        tiles = O.tile_gaussians(2 / tiles_n, 2 / tiles_n, g, cam.view)
        tab = sharding.shard_table(tiles_n, tiles_n, world)
        tile_w = tile_h = w // tiles_n
        # this rank's pixels only (the oracle renders the listed raster indices)
        mine = []
        for t in tab[rank][tab[rank] >= 0]:
            ty, tx = divmod(int(t), tiles_n)
            rows, cols = np.meshgrid(np.arange(ty * tile_h, (ty + 1) * tile_h), np.arange(tx * tile_w, (tx + 1) * tile_w), indexing="ij")
            mine.append((rows * w + cols).ravel())
        pix = np.concatenate(mine).astype(np.uint32) if mine else np.zeros(0, np.uint32)
        img, _ = O.render(w, h, plane, cam.position, g, tiles, pixels=pix, threads=1)
        shard = torch.from_numpy(sharding.extract_shard(img.reshape(h, w), tab, rank, tiles_n, tile_w, tile_h).view(np.int32))
        gathered = sharding.gather_frame(dist, shard, rank, world, dst=0)
        if rank == 0:
            frame = sharding.assemble(gathered.numpy().view(np.uint32), tab, tiles_n, tile_w, tile_h, h, w)
            np.save(out_path, frame)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_tile_sharded_gather_reproduces_the_frame(world, tmp_path, oracle, pkg):
    from sgrt_amd import scene
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    frame = np.load(out)
    w = h = 64
    g = scene.grid_scene(4).view(oracle.GAUSSIAN)
    cam, _ = scene.cli_camera(w, h)
    tiles = oracle.tile_gaussians(2 / 4, 2 / 4, g, cam.view)
    full, _ = oracle.render(w, h, cam.plane(), cam.position, g, tiles, threads=2)
    np.testing.assert_array_equal(frame, full.reshape(h, w))
    assert (frame >> 24).max() > 0


def _loop_worker(rank, world, port, out_path, nsteps, frames_per_gather):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_pkg
    load_pkg()
    from sgrt_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tiles_n, tile_w, tile_h = 4, 8, 4
        h, w = tiles_n * tile_h, tiles_n * tile_w
        tab = sharding.shard_table(tiles_n, tiles_n, world)
        base = (np.arange(h * w, dtype=np.uint32) * 2654435761 & 0x7FFFFFFF).reshape(h, w)   # frame k = base + k
        npx = tab.shape[1] * tile_w * tile_h
        fg = sharding.FrameGatherer(dist, rank, world, npx, frames_per_gather, "cpu")
        which, counter, frames = {}, [0], []

        def render(b, f):
            k = counter[0]; counter[0] += 1
            which[(b, f)] = k
            shard = sharding.extract_shard(base + np.uint32(k), tab, rank, tiles_n, tile_w, tile_h)
            fg.shard_frame(b, f).copy_(torch.from_numpy(shard.view(np.int32).ravel()))

        def assemble(b, f):
            view, stride = fg.gathered_frame(b, f)
            parts = np.stack([view[q * stride: q * stride + npx].numpy() for q in range(world)]).view(np.uint32)
            frames.append((which[(b, f)], sharding.assemble(parts, tab, tiles_n, tile_w, tile_h, h, w)))

        fg.run(nsteps, render, assemble)
        if rank == 0:
            assert [k for k, _ in frames] == list(range(nsteps))           # every frame, in order
            for k, fr in frames:
                np.testing.assert_array_equal(fr, base + np.uint32(k))
            np.save(out_path, np.array([len(frames)]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,nsteps,frames_per_gather", [(2, 7, 3), (3, 4, 4), (8, 5, 2), (2, 3, 1)])
def test_batched_frame_loop(world, nsteps, frames_per_gather, tmp_path, pkg):
    out = str(tmp_path / "n.npy")
    mp.spawn(_loop_worker, args=(world, _free_port(), out, nsteps, frames_per_gather), nprocs=world, join=True)
    assert int(np.load(out)[0]) == nsteps


def _sparse_loop_worker(rank, world, port, out_path, nsteps, frames_per_gather, grow=False, exchange="all_gather"):
    """The sparse protocol (sharding.SparseFrameGatherer, what bench.py runs over RCCL for N > 1) on CPU tensors: frames
    whose lit cells move from frame to frame, tiles that are not a multiple of the 32-px cell, more ranks than lit tiles."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_pkg
    load_pkg()
    from sgrt_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tiles_n, tile_w, tile_h = 4, 80, 40           # 3 x 2 cells per tile, the last column / row of cells partial
        h, w = tiles_n * tile_h, tiles_n * tile_w
        tab = sharding.shard_table(tiles_n, tiles_n, world)
        cap = sharding.sparse_capacity(tab, tile_w, tile_h)
        words = sharding.sparse_words(cap)

        def frame_image(k):                            # a blob that moves with k over a zero background
            yy, xx = np.mgrid[0:h, 0:w]
            radius = (6 + 22 * (k // frames_per_gather)) if grow else (18 + 3 * (k % 4))   # grow: every batch fuller than the prefix
            blob = ((yy - (20 + 9 * k) % h) ** 2 + (xx - (30 + 31 * k) % w) ** 2) < radius ** 2
            return np.where(blob, (np.uint32(0x01000000) + (yy * w + xx + k).astype(np.uint32)), np.uint32(0)).astype(np.uint32)

        fg = sharding.SparseFrameGatherer(dist, rank, world, words, cap, frames_per_gather, "cpu", nbuf=2 if world == 8 else 3, exchange=exchange)
        which, counter, frames = {}, [0], []

        def render(b, f):
            k = counter[0]; counter[0] += 1
            which[(b, f)] = k
            sh = sharding.extract_sparse(frame_image(k), tab, rank, tiles_n, tile_w, tile_h)
            fg.shard_frame(b, f).copy_(torch.from_numpy(sh.view(np.int32)))

        def assemble(b, f):
            parts = [t.numpy() for t in fg.gathered_shards(b, f)]
            frames.append((which[(b, f)], sharding.scatter_sparse(parts, tiles_n, tile_w, tile_h, h, w)))

        fg.run(nsteps, render, assemble)
        if rank == 0:
            assert [k for k, _ in frames] == list(range(nsteps))
            for k, fr in frames:
                np.testing.assert_array_equal(fr, frame_image(k))
            dense_bytes = (world - 1) * nsteps * tab.shape[1] * tile_w * tile_h * 4
            np.save(out_path, np.array([len(frames), fg.bytes_moved, dense_bytes, fg.regathered]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,nsteps,frames_per_gather,grow,exchange",
                         [(2, 7, 3, False, "all_gather"), (3, 5, 2, False, "all_gather"), (8, 4, 4, False, "all_gather"), (2, 8, 2, True, "all_gather"),
                          (3, 8, 2, True, "all_gather"), (2, 7, 3, False, "gather"), (2, 8, 2, True, "gather")])
def test_sparse_shard_frame_loop(world, nsteps, frames_per_gather, grow, exchange, tmp_path, pkg):
    """grow: the lit area grows from batch to batch faster than the margin on the travelling prefix -- the check that runs
    one batch late must notice and gather those batches again in full before they are assembled.  exchange: one all_gather per
    batch (the cell counts are read from the received headers; round 4, the default) or gather + all-reduce (rounds 2-3)."""
    out = str(tmp_path / "n.npy")
    mp.spawn(_sparse_loop_worker, args=(world, _free_port(), out, nsteps, frames_per_gather, grow, exchange), nprocs=world, join=True)
    n, moved, dense, regathered = np.load(out)
    assert int(n) == nsteps
    assert moved < dense or grow    # fewer bytes travel than the compact shards would take
    assert (regathered > 0) == grow


def test_sparse_shard_layout_round_trip(pkg):
    """extract_sparse / scatter_sparse (host mirror of RenderTarget::sparse and scatter_sparse_kernel): any image comes
    back, for tile sizes that are and are not multiples of the cell, for every world size."""
    from sgrt_amd import sharding
    rng = np.random.default_rng(5)
    for tiles_n, tile_w, tile_h, world in [(4, 64, 64, 1), (4, 64, 64, 3), (2, 100, 33, 2), (3, 32, 96, 8), (1, 70, 70, 2)]:
        h, w = tiles_n * tile_h, tiles_n * tile_w
        img = np.zeros((h, w), np.uint32)
        for _ in range(6):
            y, x = int(rng.integers(0, h)), int(rng.integers(0, w))
            img[y:y + int(rng.integers(1, 40)), x:x + int(rng.integers(1, 40))] = rng.integers(1, 2 ** 32, dtype=np.uint64).astype(np.uint32)
        tab = sharding.shard_table(tiles_n, tiles_n, world)
        shards = [sharding.extract_sparse(img, tab, r, tiles_n, tile_w, tile_h) for r in range(world)]
        assert all(int(s[1]) == sharding.sparse_capacity(tab, tile_w, tile_h) for s in shards)
        np.testing.assert_array_equal(sharding.scatter_sparse(shards, tiles_n, tile_w, tile_h, h, w), img)
        # a prefix that just covers the stored cells is enough (what the gather ships)
        cut = [s[:sharding.sparse_pixel_offset(int(s[1])) + int(s[0]) * 1024] for s in shards]
        np.testing.assert_array_equal(sharding.scatter_sparse(cut, tiles_n, tile_w, tile_h, h, w), img)
