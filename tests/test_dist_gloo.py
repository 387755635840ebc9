"""Multi-process (world_size 2 and 3, gloo, CPU) test of the N > 1 path: tile shards -> gather -> assemble.

The shard map and buffer layout are the ones csrc/vrt_hip_api.cpp uses on the GPUs (mirrored on the
host in sharding.py; their equality with the device code is checked on the GPU by
test_gpu_parity.py::test_sharded_render_assembles_to_full_frame).  Here every rank produces its shard
of a small frame with the CPU oracle standing in for the HIP render, the shards travel through the same
torch.distributed gather bench.py issues over RCCL, and rank 0 must end up with the single-process frame.
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
