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


class FrameGatherer:
    """The frame loop of an N-rank job: every rank renders its shard of `frames` consecutive frames into one
    buffer, and the shards travel to rank 0 in ONE gather per batch ([rank][frame][shard] on rank 0) -- fewer,
    larger collectives: a 2048^2 frame is ~0.06 ms of GPU work, less than one RCCL call costs.  Two buffers:
    the gather of batch i overlaps the rendering of batch i + 1.  Every frame is rendered, gathered and handed
    to `assemble`.

    dist     torch.distributed (initialised); device: where the shard buffers live ("cuda" for RCCL, "cpu" for gloo)
    npx      u32 per shard (vrt_hip_shard_pixels(): the same on every rank)
    stage    True: the tensors are on a GPU but the backend cannot move GPU memory (gloo rehearsal on a one-GPU
             box): the gather is staged through host memory, synchronously
    """

    def __init__(self, dist, rank, world, npx, frames, device, stage=False):
        import torch
        self.dist, self.rank, self.world, self.npx, self.F, self.stage = dist, rank, world, int(npx), max(1, int(frames)), stage
        self.shard = [torch.zeros(self.F * self.npx, dtype=torch.int32, device=device) for _ in range(2)]
        self.gathered = [torch.zeros(world * self.F * self.npx, dtype=torch.int32, device=device) if rank == 0 else None
                         for _ in range(2)]
        self._views = {}

    def shard_frame(self, b, f):
        """Where frame f of buffer b is rendered."""
        return self.shard[b][f * self.npx:(f + 1) * self.npx]

    def gathered_frame(self, b, f):
        """(tensor view starting at rank 0's shard of frame f of buffer b, rank stride in u32) -- rank 0 only."""
        return self.gathered[b][f * self.npx:], self.F * self.npx

    def _io(self, b, nf):
        if (b, nf) not in self._views:
            n, F = self.npx, self.F
            dst = [self.gathered[b][q * F * n: q * F * n + nf * n] for q in range(self.world)] if self.rank == 0 else None
            self._views[(b, nf)] = (self.shard[b][: nf * n], dst)
        return self._views[(b, nf)]

    def start(self, b, nf):
        """Begin the gather of the first nf frames of buffer b; returns a handle with wait()."""
        import torch
        src, dst = self._io(b, nf)
        if not self.stage:
            return self.dist.gather(src, dst, dst=0, async_op=True)
        torch.cuda.synchronize()
        host = src.cpu()
        out = [torch.empty_like(host) for _ in range(self.world)] if self.rank == 0 else None
        self.dist.gather(host, out, dst=0)
        if self.rank == 0:
            for q in range(self.world):
                dst[q].copy_(out[q])
        return None

    def run(self, nsteps, render, assemble):
        """nsteps frames: render(b, f) fills shard_frame(b, f) (stream-ordered with the collective on GPUs);
        assemble(b, f) is called on rank 0 for every gathered frame."""
        pending, nfs = [None, None], [0, 0]
        busy = [False, False]

        def finish(b):
            if pending[b] is not None:
                pending[b].wait()
            if self.rank == 0:
                for f in range(nfs[b]):
                    assemble(b, f)
            pending[b], busy[b] = None, False

        F = self.F
        for i in range((nsteps + F - 1) // F):
            b, nf = i & 1, min(F, nsteps - i * F)
            if busy[b]:
                finish(b)
            for f in range(nf):
                render(b, f)
            nfs[b], pending[b], busy[b] = nf, self.start(b, nf), True
            if busy[1 - b]:
                finish(1 - b)
        for b in (0, 1):
            if busy[b]:
                finish(b)
