"""Host-side mirror of the tile -> GPU shard map of csrc/vrt_hip_api.cpp (shard_owner / rebuild_shard)
and of the shard-buffer layout, plus the gather protocol bench.py runs over RCCL.

A frame's reference tiles (tiles_w x tiles_h, row-major tile id t = ty*tiles_w + tx, rt.cpp:47-51) are
dealt to ranks along diagonals: owner(t) = (t + t // tiles_w) % world.  Every rank renders its tiles
into a compact buffer [slot][tile_h][tile_w] (slots padded to the same count on every rank so that one
fixed-size gather moves the frame); rank 0 scatters the rank-major concatenation into raster order --
the copy loop of the reference's simd_render_image (rt.h:388-399).
"""
import numpy as np


def shard_owner(t, tiles_w, world):
    return (t + t // tiles_w) % world


def shard_table(tiles_w, tiles_h, world):
    """tile id of every (rank, slot); -1 pads ranks that own fewer tiles.  Shape [world, slots]."""
    owned = [[] for _ in range(world)]
    for t in range(tiles_w * tiles_h):
        owned[shard_owner(t, tiles_w, world)].append(t)
    slots = max(len(o) for o in owned)
    tab = np.full((world, slots), -1, np.int64)
    for r, o in enumerate(owned):
        tab[r, :len(o)] = o
    return tab


def extract_shard(image, table, rank, tiles_w, tile_w, tile_h):
    """The compact shard buffer rank `rank` would produce from a raster image [H, W]."""
    out = np.zeros((table.shape[1], tile_h, tile_w), image.dtype)
    for s, t in enumerate(table[rank]):
        if t < 0:
            continue
        ty, tx = divmod(int(t), tiles_w)
        out[s] = image[ty * tile_h:(ty + 1) * tile_h, tx * tile_w:(tx + 1) * tile_w]
    return out


def assemble(gathered, table, tiles_w, tile_w, tile_h, height, width):
    """Scatter the rank-major concatenation [world, slots, tile_h, tile_w] into a raster image."""
    img = np.zeros((height, width), gathered.dtype)
    g = gathered.reshape(table.shape[0], table.shape[1], tile_h, tile_w)
    for r in range(table.shape[0]):
        for s, t in enumerate(table[r]):
            if t < 0:
                continue
            ty, tx = divmod(int(t), tiles_w)
            img[ty * tile_h:(ty + 1) * tile_h, tx * tile_w:(tx + 1) * tile_w] = g[r, s]
    return img


def gather_frame(dist, shard, rank, world, dst=0):
    """One framebuffer gather: every rank contributes its (equal-sized) shard tensor; dst receives
    the rank-major concatenation.  Works on any torch.distributed backend (RCCL on GPUs, gloo in tests)."""
    import torch
    if rank == dst:
        out = torch.empty((world,) + tuple(shard.shape), dtype=shard.dtype, device=shard.device)
        dist.gather(shard, list(out.unbind(0)), dst=dst)
        return out
    dist.gather(shard, None, dst=dst)
    return None
