"""Host-side mirror of the tile -> GPU shard map of csrc/vrt_hip_api.cpp (shard_owner / rebuild_shard)
and of the shard-buffer layout, plus the gather protocol bench.py runs over RCCL.

A frame's reference tiles (tiles_w x tiles_h, row-major tile id t = ty*tiles_w + tx, rt.cpp:47-51) are
dealt to ranks in a x b bricks (a * b = world, as square as the divisors allow), every row of bricks shifted by half a
brick: owner(tx, ty) = (tx + (a // 2) * (ty // b)) % a + a * (ty % b), so that every a x b window of tiles holds every
rank once and a rank's tiles do not line up in columns.  Every rank renders its tiles
into a compact buffer [slot][tile_h][tile_w] (slots padded to the same count on every rank so that one
fixed-size gather moves the frame); rank 0 scatters the rank-major concatenation into raster order --
the copy loop of the reference's simd_render_image (rt.h:388-399).
"""
import numpy as np

CELL = 32                 # csrc/vrt_kernels.h: second-level cull region = the unit of a sparse shard
SPARSE_HDR_WORDS = 4      # [0] cells stored, [1] capacity, [2] cells per tile, [3] 0


def shard_owner(t, tiles_w, world):
    b = max(d for d in range(1, int(world ** 0.5) + 1) if world % d == 0)
    a = world // b
    ty, tx = divmod(t, tiles_w)
    return (tx + (a // 2) * (ty // b)) % a + a * (ty % b)


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


# ---- sparse shards (include/vrt_hip.h, "Sparse shards"): only the 32x32 cells that hold something travel ----------------
def sparse_capacity(table, tile_w, tile_h):
    """Cells a shard buffer can hold: slots per rank x cells per tile (the same on every rank)."""
    return table.shape[1] * (-(-tile_w // CELL)) * (-(-tile_h // CELL))


def sparse_pixel_offset(cap):
    return (SPARSE_HDR_WORDS + cap + 3) // 4 * 4


def sparse_words(cap):
    return sparse_pixel_offset(cap) + cap * CELL * CELL


def extract_sparse(image, table, rank, tiles_w, tile_w, tile_h, background=0):
    """The sparse shard rank `rank` would produce from a raster image [H, W] (u32): every cell of its tiles that is not
    all background, in tile / cell order (the device files them in whatever order its workgroups finish; any order is a
    valid shard)."""
    cx, cy = -(-tile_w // CELL), -(-tile_h // CELL)
    cap = sparse_capacity(table, tile_w, tile_h)
    out = np.zeros(sparse_words(cap), np.uint32)
    out[1], out[2] = cap, cx * cy
    P, n = sparse_pixel_offset(cap), 0
    for t in table[rank]:
        if t < 0:
            continue
        ty, tx = divmod(int(t), tiles_w)
        tile = image[ty * tile_h:(ty + 1) * tile_h, tx * tile_w:(tx + 1) * tile_w]
        for ci in range(cx * cy):
            y0, x0 = (ci // cx) * CELL, (ci % cx) * CELL
            cell = np.full((CELL, CELL), background, np.uint32)
            part = tile[y0:y0 + CELL, x0:x0 + CELL]
            cell[:part.shape[0], :part.shape[1]] = part
            if (part != background).any():
                out[SPARSE_HDR_WORDS + n] = int(t) * cx * cy + ci
                out[P + n * CELL * CELL: P + (n + 1) * CELL * CELL] = cell.ravel()
                n += 1
    out[0] = n
    return out


def scatter_sparse(shards, tiles_w, tile_w, tile_h, height, width, background=0):
    """Raster image from sparse shards (each a u32 array, full or a prefix): background, then every stored cell."""
    img = np.full((height, width), background, np.uint32)
    cx, cy = -(-tile_w // CELL), -(-tile_h // CELL)
    for sh in shards:
        sh = np.asarray(sh).view(np.uint32)
        n, cap = int(sh[0]), int(sh[1])
        P = sparse_pixel_offset(cap)
        for s_ in range(n):
            key = int(sh[SPARSE_HDR_WORDS + s_])
            t, ci = divmod(key, cx * cy)
            ty, tx = divmod(t, tiles_w)
            y0, x0 = ty * tile_h + (ci // cx) * CELL, tx * tile_w + (ci % cx) * CELL
            hh, ww = min(CELL, (ty + 1) * tile_h - y0), min(CELL, (tx + 1) * tile_w - x0)
            cell = sh[P + s_ * CELL * CELL: P + (s_ + 1) * CELL * CELL].reshape(CELL, CELL)
            img[y0:y0 + hh, x0:x0 + ww] = cell[:hh, :ww]
    return img


class SparseFrameGatherer:
    """FrameGatherer for sparse shards.  A shard's size is data (header word 0), a gather wants equal sizes: the ranks
    gather the prefix [0, pixel offset + 1024 * cells) of every frame's shard, `cells` being the fullest shard any rank
    has produced so far plus a quarter -- 0.8 MB instead of 16 MB per `-g 64 -w 2048` frame.  The first batch learns that
    number with one tiny all-reduce(MAX) and a host read; later batches only CHECK it, one batch late and off the
    critical path (the all-reduced maximum of a batch is copied to the host asynchronously and looked at when the batch
    is finished: if a shard was fuller than the prefix that travelled, the batch is gathered again in full before it is
    assembled) -- the frame loop never waits for the GPU.

    exchange  "all_gather" (default, round 4): ONE collective per batch.  Every rank receives every shard prefix, so every rank sees
              every shard's header -- the cell counts travel with the data and the per-batch all-reduce is gone (only the very first
              batch still learns its prefix size that way).  A `-g 64 -w 2048` frame is 0.84 MB at N = 8: the links do not notice that
              eight ranks receive it instead of one, but a batch is then one RCCL launch instead of two, and a short run (the driver's
              20 steps are four batches of 17 us of rendering each at N = 8) is bound by exactly those launches.
              "gather": gather to rank 0 + a 1-element all-reduce(MAX) per batch (rounds 2-3).

    words    u32 per shard buffer (vrt_hip_sparse_shard_words(), the same on every rank); cap: its capacity in cells
    """

    def __init__(self, dist, rank, world, words, cap, frames, device, stage=False, nbuf=2, exchange="all_gather"):
        import torch
        assert exchange in ("all_gather", "gather")
        self.exchange, self.side = exchange, None   # side: the stream the received headers are looked at on (all_gather on a GPU)
        self.dist, self.rank, self.world, self.words, self.cap = dist, rank, world, int(words), int(cap)
        self.F, self.stage, self.device = max(1, int(frames)), stage, device
        self.P = sparse_pixel_offset(self.cap)
        self.NB = NB = max(2, int(nbuf))   # batches in flight: one being rendered, the others travelling / being assembled
        self.shard = [torch.zeros(self.F * self.words, dtype=torch.int32, device=device) for _ in range(NB)]
        self.recv = [None] * NB      # rank 0: [world, nf, prefix] of the batch in flight
        self.prefix = [0] * NB
        self.sent = [None] * NB      # CUDA event: the send copy of buffer b's last batch has been taken
        self.cells_sent = [0] * NB   # cells of the prefix that travelled for the batch in buffer b
        self.most = [None] * NB      # pinned host copy of the batch's all-reduced fullest shard (None: known, fits)
        self.most_ready = [None] * NB
        self.cells_hint = None       # fullest shard seen so far (all ranks agree: it is all-reduced)
        self.regathered = 0
        self.bytes_moved = 0
        self.frames_moved = 0
        # per-phase timers (off in a timed region: a pair of events per phase and batch).  With `timing` on, run() notes for every
        # batch: "render" (unless the caller's render_batch notes it on its own stream), "gather" = from the moment the collective is
        # enqueued until the frame stream may use its result (an UPPER bound of the collective: the frame stream also carries the
        # older batches' assembly), "gather_nccl" = the collective's own duration where the backend reports it
        # (TORCH_NCCL_ENABLE_TIMING=1), "assemble", and the host time spent inside run().  phase_ms() sums them.
        self.timing = False
        self._spans = []
        self._g0 = [None] * NB
        self.host_ms = 0.0
        self.timed_frames = 0

    def mark(self, stream=None):
        """A time stamp for the phase timers: a CUDA event on `stream` (default: the current one), or the host clock."""
        if not self.timing:
            return None
        if str(self.device) != "cpu":
            import torch
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(stream) if stream is not None else ev.record()
            return ev
        import time
        return time.perf_counter()

    def note(self, kind, a, b):
        if a is not None and b is not None:
            self._spans.append((kind, a, b))

    def phase_ms(self):
        """{phase: total ms} of everything noted so far (waits for the events), then forgets it."""
        out = {}
        for kind, a, b in self._spans:
            if isinstance(a, float):
                ms = (b - a) * 1e3 if kind != "gather_nccl" else float(b)
            else:
                b.synchronize()
                ms = a.elapsed_time(b)
            out[kind] = out.get(kind, 0.0) + ms
        out["host_in_run"] = self.host_ms
        out["frames"] = self.timed_frames
        out["batches_gathered_twice"] = self.regathered
        self._spans, self.host_ms, self.timed_frames = [], 0.0, 0
        return out

    def shard_frame(self, b, f):
        return self.shard[b][f * self.words:(f + 1) * self.words]

    def gathered_shards(self, b, f):
        """Rank 0: the `world` shard prefixes of frame f of buffer b (tensor views, one per rank)."""
        return [self.recv[b][q, f] for q in range(self.world)]

    def _gather(self, b, nf, cells):
        import torch
        frames = self.shard[b].view(self.F, self.words)[:nf]
        prefix = self.P + int(cells) * CELL * CELL
        send = frames[:, :prefix].contiguous()
        if send.is_cuda:                                               # the shard buffer may be rendered into again after this
            self.sent[b] = torch.cuda.Event()
            self.sent[b].record()
        self.prefix[b], self.cells_sent[b] = prefix, int(cells)
        everyone = self.exchange == "all_gather"
        if self.rank == 0 or everyone:
            self.recv[b] = torch.empty((self.world, nf, prefix), dtype=torch.int32, device=self.device)
        if self.rank == 0:
            self.bytes_moved += (self.world - 1) * nf * prefix * 4
            self.frames_moved += nf
        if everyone:
            if not self.stage:
                if send.is_cuda:    # recv[b] is the stack of the ranks' (nf, prefix) blocks along its first dimension
                    return self.dist.all_gather_into_tensor(self.recv[b], send, async_op=True)
                return self.dist.all_gather(list(self.recv[b].unbind(0)), send, async_op=True)
            host = send.cpu()
            out = [torch.empty_like(host) for _ in range(self.world)]
            self.dist.all_gather(out, host)
            for q in range(self.world):
                self.recv[b][q].copy_(out[q])
            return None
        dst = list(self.recv[b].unbind(0)) if self.rank == 0 else None
        if not self.stage:
            return self.dist.gather(send, dst, dst=0, async_op=True)
        host = send.cpu()
        out = [torch.empty_like(host) for _ in range(self.world)] if self.rank == 0 else None
        self.dist.gather(host, out, dst=0)
        if self.rank == 0:
            for q in range(self.world):
                dst[q].copy_(out[q])
        return None

    def start(self, b, nf):
        import torch
        frames = self.shard[b].view(self.F, self.words)[:nf]
        self._g0[b] = self.mark()
        if self.exchange == "all_gather" and self.cells_hint is not None and not self.stage:
            # no all-reduce: every rank will see every shard's header (word 0 = cells stored) in what it receives; the fullest one
            # is looked at when the batch is finished -- on a side stream that waits for the collective, copied to pinned memory
            self.most[b] = self.most_ready[b] = None
            work = self._gather(b, nf, min(self.cap, self.cells_hint + self.cells_hint // 4 + 8))
            if self.recv[b].is_cuda:
                if self.side is None:
                    self.side = torch.cuda.Stream()
                with torch.cuda.stream(self.side):
                    work.wait()                                           # (the side stream waits, nobody else)
                    seen = self.recv[b][:, :, 0].max().reshape(1).to(torch.int64)
                    host = torch.empty(1, dtype=torch.int64, pin_memory=True)
                    host.copy_(seen, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record()
                self.recv[b].record_stream(self.side)
                self.most[b], self.most_ready[b] = host, ev
            else:
                self.most[b] = "headers"                                  # CPU tensors: read them once the collective has been waited for
            return work
        most = frames[:, 0].max().reshape(1).to(torch.int64)          # cells of the fullest shard of this batch, this rank
        if self.stage:
            most = most.cpu()
        self.dist.all_reduce(most, op=self.dist.ReduceOp.MAX)
        self.most[b] = self.most_ready[b] = None
        if self.cells_hint is None or self.stage:
            self.cells_hint = max(self.cells_hint or 0, int(most.item()))   # first batch (and the CPU-staged path): host read
            cells = self.cells_hint
        else:
            cells = min(self.cap, self.cells_hint + self.cells_hint // 4 + 8)
            if most.is_cuda:
                host = torch.empty(1, dtype=torch.int64, pin_memory=True)
                host.copy_(most, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
            else:                                                       # CPU tensors (the gloo tests): nothing to wait for
                host, ev = most.clone(), None
            self.most[b], self.most_ready[b] = host, ev
        return self._gather(b, nf, cells)

    def run(self, nsteps, render, assemble, render_batch=None, assemble_batch=None):
        """nsteps frames: render(b, f) fills shard_frame(b, f) -- or render_batch(b, nf) the first nf frames of buffer b at
        once; assemble(b, f) is called on rank 0 for every gathered frame -- or assemble_batch(b, nf) once per batch (frame
        f's shard of rank q is recv[b][q, f]: base recv[b][q], frame stride prefix[b] words)."""
        NB = self.NB
        pending, nfs, busy, order = [None] * NB, [0] * NB, [False] * NB, []

        import time
        t_run = time.perf_counter()

        def finish(b):
            if pending[b] is not None:
                pending[b].wait()
                if self.timing:
                    try:                                              # the collective's own duration, where the backend keeps it
                        self.note("gather_nccl", 0.0, float(pending[b]._get_duration()))
                    except Exception:
                        pass
            if self.most[b] is not None:                              # the deferred check of this batch's prefix
                if self.most_ready[b] is not None:
                    self.most_ready[b].synchronize()
                most = int(self.recv[b][:, :, 0].max().item()) if isinstance(self.most[b], str) else int(self.most[b].item())
                self.most[b] = None
                self.cells_hint = max(self.cells_hint, most)
                if most > self.cells_sent[b]:                         # a shard was fuller than what travelled: once more, in full
                    self.regathered += 1
                    again = self._gather(b, nfs[b], most)
                    if again is not None:
                        again.wait()
            self.note("gather", self._g0[b], self.mark())
            if self.rank == 0:
                a0 = self.mark()
                if assemble_batch is not None:
                    assemble_batch(b, nfs[b])
                else:
                    for f in range(nfs[b]):
                        assemble(b, f)
                self.note("assemble", a0, self.mark())
            pending[b], busy[b] = None, False

        F = self.F
        for i in range((nsteps + F - 1) // F):
            b, nf = i % NB, min(F, nsteps - i * F)
            if busy[b]:
                order.remove(b)
                finish(b)
            if render_batch is not None:
                render_batch(b, nf)                # (notes its own "render" span: it runs on a stream of its own)
            else:
                r0 = self.mark()
                for f in range(nf):
                    render(b, f)
                self.note("render", r0, self.mark())
            nfs[b], pending[b], busy[b] = nf, self.start(b, nf), True
            order.append(b)
            while len(order) > NB - 1:            # oldest first: frames are assembled in the order they were rendered
                finish(order.pop(0))
        while order:
            finish(order.pop(0))
        if self.timing:
            self.host_ms += (time.perf_counter() - t_run) * 1e3
            self.timed_frames += nsteps
