#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path on MI355X.

Metric (BASELINE.json): Mrays/sec (whole node) + ms/frame, 2048^2 image, 64x64 Gaussian grid
(`volumetric-ray-tracer -g 64 -w 2048`, default --tiles 16, mode-8 packing).

A step = one frame = what the reference times as `TIME:` (main.cpp:260-296): tile binning of all
Gaussians + render (+ for N > 1: RCCL gather of the tile shards to rank 0 + assembly into raster
order).  Inputs (scene tables, camera) are resident in HBM before the timed region starts.
N = 1 keeps four frames in flight (four library contexts on four HIP streams = the HIP runtime's four hardware queues,
--frames-in-flight);
the strictly serial figures are reported next to the headline under "serial".

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  Extra keys: "roofline" (render kernel: algorithmic HBM bytes per launch
over its HIP-event duration, next to the VALU-op rate that actually bounds it) and "cpu_baseline"
(the oracle's SIMD port of the reference's mode 8, timed on this host on a bounded sample).
"""
import argparse
import importlib.util
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))


def load_pkg():
    d = os.path.join(ROOT, "simd-gaussian-ray-tracing_amd")
    spec = importlib.util.spec_from_file_location("sgrt_amd", os.path.join(d, "__init__.py"),
                                                  submodule_search_locations=[d])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["sgrt_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# fp32 VALU issue peak (MI355X_MICROARCH.md: a wave64 VALU instruction issues over 2 cycles on its SIMD; 256 CUs x 4
# SIMDs; 2.4 GHz max clock) = 1.2288e12 wave-instructions/s = the 157.3 TFLOP/s vector figure / (64 lanes x 2 flop).
VALU_PEAK_WAVE_INSTR = 256 * 4 * 2.4e9 / 2
# What a pure v_fma_f32 loop sustains on this part, with the in-kernel clock that explains it (tools/ubench/valu.hip,
# profiles/r02_valu_ubench.txt): the same 2 cycles per instruction at the clock the chip holds under a VALU-saturating
# load, which is below 2.4 GHz (guide, "DVFS give-back").  Reported next to the spec figure, never instead of it.
VALU_MEASURED_FILE = os.path.join(ROOT, "profiles", "r02_valu_ubench.json")


def kernel_source_sha():
    """Identity of the kernel sources a profile was taken with (tools/pmc_traffic.py stores it in its summary)."""
    import hashlib
    h = hashlib.sha256()
    for name in ("vrt_kernels_common.hpp", "vrt_block_kernel.hip", "vrt_table_kernel.hip", "vrt_kernels.hip", "vrt_kernels.h",
                 "vrt_device_math.h"):
        with open(os.path.join(ROOT, "simd-gaussian-ray-tracing_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def latest_pmc_summary():
    """(path, dict) of the newest profiles/r*_pmc_traffic.json, or (None, None)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    for path in reversed(files):
        try:
            return os.path.relpath(path, ROOT), json.load(open(path))
        except (OSError, ValueError):
            continue
    return None, None


def host_threads():
    """Hardware threads this process may really use: the affinity mask, capped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(scene_mod, g, w, h, tiles_n, grid_dim, budget_note):
    """The reference CPU path stand-in (oracle/vrt_cpu_simd.*: own SIMD port of mode 8, one task per image tile on a
    thread pool like rt.h:355-386) on a bounded sample: the first rows of EVERY tile -- 256 tasks, so every hardware
    thread has work -- scaled to the whole frame by inner-term count.  `cores` = threads started; `busy_threads` =
    process CPU time / wall time of the sample, i.e. how many of them really ran."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import oracle as O
    O.build()
    cam, _ = O.cli_camera(w, h)
    plane = O.camera_plane(cam)
    og = g.view(O.GAUSSIAN)
    tiles = O.tile_gaussians(2.0 / tiles_n, 2.0 / tiles_n, og, O.camera_view(cam))
    nt = np.diff(tiles["offsets"]).astype(np.float64)
    tile_px = (w // tiles_n) * (h // tiles_n)
    frame_terms = float((nt ** 2).sum() * 5 * tile_px)
    ntiles = tiles_n * tiles_n
    threads = max(1, min(host_threads(), ntiles))
    subset = np.arange(ntiles, dtype=np.uint32)
    tile_w = w // tiles_n

    def sample(rows):
        c0, t0 = time.process_time(), time.perf_counter()
        _, terms, simd = O.simd_render_tiled(w, h, plane, cam.position[:], og, tiles, subset, threads, max_rows=rows)
        return terms, simd, time.perf_counter() - t0, time.process_time() - c0

    # calibrate on one pixel row of each tile, then size the sample for ~15 s of wall time
    terms, simd, dt, _ = sample(1)
    rows = int(max(1, min(h // tiles_n, round(15.0 / max(dt, 1e-3)))))
    terms, simd, dt, cpu_s = sample(rows)
    sample_rays = ntiles * rows * tile_w
    terms_per_s = terms / dt
    frame_s = frame_terms / terms_per_s
    # BASELINE configs[0] (`-g 4 -w 256 -q`, CPU SIMD path, single thread): the plumbing case, whole frame
    g1 = scene_mod.grid_scene(4).view(O.GAUSSIAN)
    cam1, _ = O.cli_camera(256, 256)
    tiles1 = O.tile_gaussians(2.0 / 16, 2.0 / 16, g1, O.camera_view(cam1))
    plane1 = O.camera_plane(cam1)
    t1 = time.perf_counter()
    _, terms1, _ = O.simd_render_tiled(256, 256, plane1, cam1.position[:], g1, tiles1, None, 1)
    dt1 = time.perf_counter() - t1
    # how the port's speed relates to the real reference's on ONE host (the build container: profiles/rNN_cpu_port_vs_reference.md,
    # made by tools/cpu_port_vs_reference.py against the survey's probe of the T-SIMD build; the reference cannot be rebuilt here)
    import glob
    import re
    vs_ref, vs_ref_file = None, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_cpu_port_vs_reference.md")), reverse=True):
        m = re.search(r"port_vs_reference_same_host:\s*([0-9.]+)\s*\.\.\s*([0-9.]+)", open(path).read())
        if m:
            vs_ref, vs_ref_file = [float(m.group(1)), float(m.group(2))], os.path.relpath(path, ROOT)
            break
    return {
        "value": (w * h) / frame_s / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
        "port_vs_reference_same_host": vs_ref,
        "note": ("own AVX-512 / AVX2 restatement of the reference's mode 8 doing its full per-term work, with the reference's call "
                 "structure (from_gaussian_t and the A&S erf are real calls, four-component dot products, VCL exp's range check); "
                 f"measured in the build container at {vs_ref[0]:.2f}-{vs_ref[1]:.2f} x the speed of the real T-SIMD build on the same "
                 f"host and workloads ({vs_ref_file}): divide by that for the reference's own speed" if vs_ref else
                 "own AVX-512 / AVX2 restatement of the reference's mode 8 doing its full per-term work"),
        "busy_threads": cpu_s / dt, "terms_per_s_per_busy_thread": terms / max(cpu_s, 1e-9),
        "sample": (f"all {ntiles} tiles x first {rows} pixel row(s) = {sample_rays} rays on {threads} threads "
                   f"({cpu_s / dt:.1f} busy on average), {terms:.3e} (ray,i,k,j) inner terms in {dt:.2f} s "
                   f"({terms_per_s:.3e} terms/s, SIMD width {simd}); frame = {frame_terms:.3e} terms -> "
                   f"{frame_s:.1f} s/frame extrapolated by term count{budget_note}"),
        "ms_per_frame_extrapolated": frame_s * 1e3,
        "cfg1_single_thread": {"workload": "-g 4 -w 256, 1 thread, whole frame", "ms_per_frame": dt1 * 1e3,
                               "Mrays_per_s": 65536 / dt1 / 1e6, "terms_per_s": terms1 / dt1},
    }


def parity_check(scene_mod, g, w, h, tiles_n, rad, npix, seed, cull_eps, cull_prune, table_budget, dense_blocks):
    """max |GPU - oracle| of the float radiance on `npix` seeded LIT pixels of the frame the timed loop renders (the oracle as the
    checker, after the timed region), with the a-priori bound of what the timed settings may cost beside it:
    every dropped Gaussian changes a ray by less than 3 x its sigma*mag*exp(-x); tile level 3 eps N (N <= 4096: eps shrinks with N
    beyond), the three lower levels 3 x 1365 eps each, the budgeted prune 3 kappa 1365 eps -- or, for a ray of a dense block, the table
    kernel's budget instead of ray level + prune.  The bound is on the deviation from the reference's full sum; the oracle differs
    from that by fp32 re-association only (a few 1e-6)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import oracle as O
    O.build()
    cam, _ = O.cli_camera(w, h)
    og = g.view(O.GAUSSIAN)
    tiles = O.tile_gaussians(2.0 / tiles_n, 2.0 / tiles_n, og, O.camera_view(cam))
    rad = rad.reshape(-1, 4)
    lum = rad[:, :3].sum(1)
    lit = np.flatnonzero(lum > 0.05 * lum.max())
    rng = np.random.default_rng(seed)
    pix = np.sort(rng.choice(lit, min(npix, lit.size), replace=False)).astype(np.uint32)
    t0 = time.perf_counter()
    orad = O.render(w, h, O.camera_plane(cam), cam.position[:], og, tiles, pixels=pix, want_image=False, threads=host_threads())[1]
    dt = time.perf_counter() - t0
    ref_n = 4096.0 / 3.0
    n = len(g)
    levels = 3.0 * cull_eps * min(n, 4096) + 3 * 3.0 * ref_n * cull_eps
    prune = 3.0 * cull_prune * ref_n * cull_eps
    bound_block = levels + prune
    bound_dense = 3.0 * cull_eps * min(n, 4096) + 2 * 3.0 * ref_n * cull_eps + table_budget
    return {"pixels": int(pix.size), "pixel_choice": f"seeded ({seed}) among the {lit.size} pixels above 5 % of the peak luminance",
            "max_abs": float(np.abs(rad[pix] - orad).max()), "bound": max(bound_block, bound_dense if dense_blocks else 0.0),
            "bound_terms": {"tile_cell_block_ray_levels": levels, "budgeted_prune": prune,
                            "dense_blocks_table_budget_instead_of_ray_level_and_prune": bound_dense if dense_blocks else None},
            "tolerance": 1e-4, "oracle_peak_radiance": float(orad[:, :3].max()), "oracle_seconds": dt,
            "settings": "the timed loop's (cull_eps, ray-level prune, table kernel as configured)"}


def gather_batch_frames(gather_frames, steps):
    """Frames per RCCL gather for a run of `steps` timed steps: --gather-frames, but never more than a quarter of the run."""
    return max(1, min(int(gather_frames), max(int(steps), 1) // 4 if steps >= 4 else 1))


def pick_streams(n, frame_a, frame_b, fixed=()):
    """n HIP streams that overlap pairwise (and with the streams in `fixed`).  The HIP runtime runs all streams of a process
    on a handful of hardware queues (four by default; a stream gets one when it is first used), and two streams on one
    queue execute strictly one after the other: measured with this workload (`profiles/r02_stream_pairs.txt`), frames
    alternating between torch's pool streams i and i + 4 take the serial 48.8 us per frame, any other pair 32.7.  Which
    streams collide depends on everything the process created before, so the choice is made by measurement: candidates
    are probed against the streams already chosen with the workload itself (frame_a(stream), frame_b(stream): one frame
    each on two different contexts; 48 frames per probe)."""
    import torch

    def span(sa, sb, k=48):
        for i in range(8):
            (frame_a if i % 2 == 0 else frame_b)((sa if i % 2 == 0 else sb).cuda_stream)
        sa.synchronize(); sb.synchronize()
        t0 = time.perf_counter()
        for i in range(k):
            (frame_a if i % 2 == 0 else frame_b)((sa if i % 2 == 0 else sb).cuda_stream)
        sa.synchronize(); sb.synchronize()
        return (time.perf_counter() - t0) / k

    cands = [torch.cuda.Stream() for _ in range(12)]
    serial = min(span(cands[0], cands[0]) for _ in range(2))
    chosen, log = [], []
    for c in cands:
        ts = [span(c, o) for o in list(fixed) + chosen]
        ok = all(t < 0.85 * serial for t in ts)
        log.append((round(serial * 1e6, 1), [round(t * 1e6, 1) for t in ts], ok))
        if ok:
            chosen.append(c)
        if len(chosen) == n:
            break
    for c in cands:                       # fewer hardware queues than streams asked for: fill up
        if len(chosen) < n and c not in chosen:
            chosen.append(c)
    return chosen, log


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--grid", type=int, default=64)
    ap.add_argument("--width", type=int, default=2048)
    ap.add_argument("--tiles", type=int, default=16)
    ap.add_argument("--cull-eps", type=float, default=1e-9)
    ap.add_argument("--table-step", type=float, default=0.05, help="node spacing of the table kernel (library default 0.05; 0 = exact kernels only)")
    ap.add_argument("--parity-pixels", type=int, default=256, help="lit pixels the oracle checks after the timed region (N = 1; with --no-cpu-baseline: none)")
    ap.add_argument("--cull-prune", type=float, default=6.0, help="budget factor of the block kernel's ray-level prune (library default 6; 0 = off)")
    ap.add_argument("--gather-frames", type=int, default=32, help="N > 1: frames per RCCL gather (one collective per batch)")
    ap.add_argument("--exchange", choices=["all_gather", "gather"], default="all_gather",
                    help="N > 1: the collective of a batch -- all_gather (one RCCL launch per batch: the cell counts travel with the data) "
                         "or gather to rank 0 + a 1-element all-reduce per batch (rounds 2-3)")
    ap.add_argument("--no-stream-probe", action="store_true", help="take torch's next pool streams as they come (see pick_streams)")
    ap.add_argument("--setup-ms", type=float, default=50.0, help="untimed set-up frames before the warm-up steps, in milliseconds of wall time")
    ap.add_argument("--no-batch", action="store_true", help="N > 1: launch every frame of a gather batch on its own (round-2 baseline)")
    ap.add_argument("--frames-in-flight", type=int, default=2,
                    help="N = 1: library contexts (each on its own HIP stream) the timed frames alternate between; 1 = strictly serial "
                         "frames; default 2 = double buffering, whatever --steps is (round 4: every context costs a timed region ~30 us "
                         "of start-up and drain -- 20 steps read 24.5 us per step with two contexts, 26.5 with four -- and the steady "
                         "state loses 2 %: 21.2 against 20.7 us; 1..4 are measured after the timed region: in_flight_sweep)")
    ap.add_argument("--parallel", choices=["tiles", "frames"], default="tiles",
                    help="N > 1: 'tiles' = one frame's tiles sharded over the ranks + gather (the headline, SURVEY 8e); 'frames' = "
                         "every rank renders whole frames of its own, no collective (the replicas-only alternative: weak scaling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--plane-arrays", action="store_true", help="feed reference-style plane arrays (12 B/ray reads)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the HIP path has no CPU fallback", file=sys.stderr)
        sys.exit(3)
    # VRT_BENCH_BACKEND=gloo is a rehearsal mode for a one-GPU box: all ranks share GPU 0 and the shards travel
    # through host memory; the driver's multi-GPU runs use the default (nccl = RCCL over xGMI, one GPU per rank).
    backend = os.environ.get("VRT_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    pkg = load_pkg()
    from sgrt_amd import scene

    w = h = args.width
    g = scene.grid_scene(args.grid)
    cam, _ = scene.cli_camera(w, h)
    origin = cam.position
    view = cam.view
    tw = th = 2.0 / args.tiles
    pack = pkg.PACK_ROUND | pkg.ALPHA_COMPUTED

    # N = 1: the frames alternate between `--frames-in-flight` library contexts, each on its own HIP stream (default 4: one per
    # hardware queue of the HIP runtime; a fifth stream shares a queue with the first and serialises behind it):
    # frame k+1's list kernel and the head of its render kernel run while frame k's render kernel drains -- the
    # double-buffered frame loop of any renderer.  Every frame does all of its work; the strictly serial figures are
    # measured after the timed region and reported next to the headline ("serial").  N > 1: one context per rank
    # (the frame rate is set by the gather there).
    solo = world == 1 or args.parallel == "frames"   # this rank renders whole frames on its own
    if args.frames_in_flight <= 0:
        args.frames_in_flight = 2
    nfl = max(1, args.frames_in_flight) if solo else 1          # frames in flight of the timed region
    # contexts: N = 1 also measures 1..4 frames in flight after the timed region (--frames-in-flight 1 = a strictly serial process:
    # one context, nothing ever overlaps -- what the serial rocprofv3 summaries under profiles/ are taken with)
    nctx = (max(nfl, 4) if (world == 1 and nfl > 1) else nfl) if solo else 1

    def make_renderer():
        r_ = pkg.Renderer(local_rank)
        r_.set_gaussians(g)
        if args.plane_arrays:
            r_.set_plane(w, h, *cam.plane())
        else:
            r_.set_camera_view(w, h, view)   # in-kernel rays = the reference's plane points, bit for bit (camera.cpp:60-69)
        r_.set_options(pkg.EXP_VCL, pkg.ERF_AS, args.cull_eps)
        r_.set_cull_prune(args.cull_prune)
        r_.set_table_step(args.table_step)
        r_.set_shard(*((0, 1) if solo else (rank, world)))
        return r_

    ctxs = [make_renderer() for _ in range(nctx)]
    r = ctxs[0]
    # every context on a stream of its own; with one context per rank (N > 1, tile sharding) it is torch's current stream,
    # which the collective and the assembly are ordered with (the legacy default stream as one of SEVERAL frame streams
    # costs throughput: 42.5 instead of ~36 us per frame with four contexts)
    for r_ in ctxs:
        r_.tile_gaussians_device(tw, th, view, 0)     # also sizes the tile grid (one-time host sync)
    torch.cuda.synchronize()
    images = [torch.zeros(w * h, dtype=torch.int32, device="cuda") for _ in range(nctx)]
    image = images[0]
    frames = [r_.frame_call(tw, th, view, origin, pack, shard=not solo) for r_ in ctxs]   # tile_gaussians + render, one C call
    frame = frames[0]
    stream_probe = None
    if solo and nctx > 1 and not os.environ.get("VRT_BENCH_DEFAULT_STREAM"):
        if args.no_stream_probe:
            streams = [torch.cuda.Stream() for _ in range(nctx)]
        else:
            streams, stream_probe = pick_streams(nctx, lambda st_: frames[0](images[0].data_ptr(), st_),
                                                 lambda st_: frames[1](images[1].data_ptr(), st_))
    else:
        streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(nctx - 1)]
    sps = [s_.cuda_stream for s_ in streams]
    sp = sps[0]
    img_ptrs = [im.data_ptr() for im in images]
    img_ptr = img_ptrs[0]
    # N > 1: sharding.SparseFrameGatherer -- every rank renders its tile shard of F consecutive frames as SPARSE shards
    # (only the 32x32-px cells a Gaussian reaches are stored: ~0.8 MB of this 16 MB frame), ONE gather per F frames moves
    # the used prefixes to rank 0 (double-buffered against the next batch's rendering), rank 0 assembles every gathered
    # frame: background + stored cells (vrt_hip_scatter_sparse_device).  The same class runs under gloo in
    # tests/test_dist_gloo.py.
    # (a short run -- the driver times 20 steps -- takes smaller batches, so that at least four of them exist and rendering,
    # gather and assembly of different batches overlap as in the steady state: one batch of 20 would be a single
    # un-overlapped render -> gather -> assemble chain)
    F = gather_batch_frames(args.gather_frames, args.steps)
    if not solo:
        from sgrt_amd.sharding import SparseFrameGatherer, sparse_pixel_offset
        words = r.sparse_shard_words()
        cap = None
        for c_ in range(1, words):          # capacity from the buffer size: words = offset(cap) + 1024 cap
            if sparse_pixel_offset(c_) + 1024 * c_ == words:
                cap = c_
                break
        # three batches in flight (one being rendered, two travelling / being assembled), each with its own stream:
        # measured on one of 8 shards 4.9-5.1 us per frame, against 6.1 with two and 7.9 with one
        NB = 2 if args.no_batch else 3
        fg = SparseFrameGatherer(dist, rank, world, words, cap, F, "cuda", stage=backend != "nccl", nbuf=NB, exchange=args.exchange)
        sparse_frame = r.frame_sparse_call(tw, th, view, origin, pack)
        shard_ptr = [t_.data_ptr() for t_ in fg.shard]

        def render_shard(b, f):
            sparse_frame(shard_ptr[b] + 4 * f * words, sp)

        # The F frames of a gather batch go out as ONE launch of each kernel (vrt_hip_frame_batch_device: grid.y = frame;
        # F contexts per buffer, every frame needs its own lists and queues), on a stream of their own per buffer: a rank
        # that owns 1/N of the tiles has a few microseconds of work per frame -- launched frame by frame it is bound by
        # launch latency (measured on one of 8 shards: 19 us per frame, against 2-4 us in batches of 16).
        render_shard_batch = assemble_batch = None
        if not args.no_batch:
            # rank 0 assembles the F frames of a batch with one launch, each into a frame buffer of its own
            images = images + [torch.zeros(w * h, dtype=torch.int32, device="cuda") for _ in range(F - len(images))]
            img_ptrs = [im.data_ptr() for im in images]

            def assemble_batch(b, nf):
                r.scatter_sparse_batch_device([fg.recv[b][q].data_ptr() for q in range(world)], fg.prefix[b], nf, pack, img_ptrs, sp,
                                              retained=True)

            groups = [[r] + [make_renderer() for _ in range(F - 1)]] + [[make_renderer() for _ in range(F)] for _ in range(NB - 1)]
            if args.no_stream_probe:
                rstreams = [torch.cuda.Stream() for _ in range(NB)]
            else:
                pa = groups[0][0].frame_sparse_call(tw, th, view, origin, pack)
                pb = groups[1][0].frame_sparse_call(tw, th, view, origin, pack)
                rstreams, stream_probe = pick_streams(NB, lambda st_: pa(shard_ptr[0], st_), lambda st_: pb(shard_ptr[1], st_),
                                                      fixed=[torch.cuda.current_stream()])
            for gr in groups:
                for r_ in gr[1:] if gr[0] is r else gr:
                    r_.tile_gaussians_device(tw, th, view, sp)
            torch.cuda.synchronize()
            batch_calls = [gr[0].frame_batch_call(gr[1:], tw, th, [view] * F, [origin] * F, pack, out_kind=2) for gr in groups]
            batch_ptrs = [[shard_ptr[b] + 4 * f * words for f in range(F)] for b in range(NB)]
            streams = streams + rstreams

            def render_shard_batch(b, nf):
                if fg.sent[b] is not None:
                    rstreams[b].wait_event(fg.sent[b])     # the previous batch in this buffer has been copied out
                r0 = fg.mark(rstreams[b])
                batch_calls[b](batch_ptrs[b], rstreams[b].cuda_stream, nf)
                fg.note("render", r0, fg.mark(rstreams[b]))
                done = torch.cuda.Event()
                done.record(rstreams[b])
                torch.cuda.current_stream().wait_event(done)

        def assemble(b, f):
            # `image` is written by nothing but this call: retained assembly (only cells that went dark are reset, not 16.8 MB)
            r.scatter_sparse_device([t_.data_ptr() for t_ in fg.gathered_shards(b, f)], pack, img_ptr, sp, retained=True)

    def run(nsteps, in_flight=nfl, serial=False):
        if solo:
            for k in range(nsteps):
                i = k % in_flight
                frames[i](img_ptrs[i], sps[i])
        else:
            fg.run(nsteps, render_shard, assemble, *((render_shard_batch, assemble_batch) if not serial else ()))

    def barrier_local():
        evs = []
        for s_ in streams:
            ev = torch.cuda.Event()
            ev.record(s_)
            evs.append(ev)
        for ev in evs:
            ev.synchronize()

    def barrier():
        # poll the streams' completion first: hipDeviceSynchronize alone wakes up ~0.1-0.2 ms after the GPU is done
        # (interrupt-driven wait), which would be charged to the K timed steps
        evs = []
        for s_ in streams:
            ev = torch.cuda.Event()
            ev.record(s_)
            evs.append(ev)
        for ev in evs:
            while not ev.query():
                pass
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # set-up, not warm-up: the dense-launch feedback of the library needs a few frames of one scene PER CONTEXT to settle
    # (frame batches: every context renders one frame per batch, two groups of F contexts)
    run(8 if (solo or args.no_batch) else 12 * F)
    # ... and the GPU a few tens of milliseconds of this work to reach its clocks: the first frames after an idle phase
    # (context creation is ~1 s of host work) run up to 2x slower (tools/emulate_ranks.py: 11.7 vs 5.2 us per frame on the
    # same shard).  Still set-up: the W warm-up steps follow, then exactly K timed steps.
    t_setup = time.perf_counter()
    while (time.perf_counter() - t_setup) * 1e3 < args.setup_ms:
        run(16 if (solo or args.no_batch) else NB * F)
        if solo:
            barrier_local()
    run(args.warmup)
    barrier()
    # HIP events around the dominant kernel, on the stream it runs on, live in the timed region -- on every 8th frame:
    # an event costs the stream 2-3 us and a frame is ~55 us, so bracketing every launch would take a tenth of the
    # throughput being measured.  The per-kernel breakdown of the launch sequence (four events per frame, every frame)
    # is measured in a second loop after the timed region.
    r.enable_kernel_timing(3)
    t0 = time.perf_counter()
    run(args.steps)
    t_enqueued = time.perf_counter() - t0      # the host's share: all K steps are enqueued (the GPU may be far behind, or waiting)
    barrier()
    elapsed = time.perf_counter() - t0
    if os.environ.get("VRT_BENCH_DEBUG") and not solo:
        for b in range(fg.NB):
            print(f"[debug rank {rank}] shard buffer {b} headers:", fg.shard[b].view(fg.F, fg.words)[:, :4].cpu().tolist(), "prefix", fg.prefix[b], file=sys.stderr)
            if rank == 0 and fg.recv[b] is not None:
                print(f"[debug rank {rank}] recv {b} headers:", fg.recv[b][:, :, :4].cpu().tolist(), file=sys.stderr)
        if rank == 0:
            print("[debug] lit pixels per image:", [int((im != 0).sum()) for im in images], file=sys.stderr)
    kt = r.kernel_timing()
    r.enable_kernel_timing(False)
    # after the timed region: strictly serial frames on one context (no events), then the per-kernel breakdown
    n_serial = max(200, min(args.steps, 400))   # its own sample size: a 20-step timed region says little about one frame's latency
    barrier()
    t1 = time.perf_counter()
    run(n_serial, 1, serial=True)
    barrier()
    serial_ms = (time.perf_counter() - t1) / max(n_serial, 1) * 1e3
    # What a frame loop with a MOVING camera costs (N = 1): the same workload, the view turned by 1e-4 degrees per step, so
    # nothing a static view lets the library keep (per-origin tables, tile cones, the dense-launch report) applies: every
    # frame sets its rays, rebuilds the cone table and runs all launches.  Same frames in flight, then serial.
    moving = None
    sweep = None
    if solo and world == 1 and not args.plane_arrays:
        poses = [scene.cli_camera(w, h, initial_rot=1e-4 * k)[0] for k in range(64)]
        mcalls = [[r_.frame_view_call(tw, th, p_.view, p_.position, pack) for p_ in poses] for r_ in ctxs]

        def run_moving(nsteps, in_flight):
            for k in range(nsteps):
                i = k % in_flight
                mcalls[i][k % 64](img_ptrs[i], sps[i])

        run_moving(64, nfl)
        barrier()
        t2 = time.perf_counter()
        run_moving(n_serial, nfl)
        barrier()
        mov_ms = (time.perf_counter() - t2) / n_serial * 1e3
        # the last frame of every context against a fresh render of its pose
        last = {(k % nfl): k % 64 for k in range(n_serial)}
        mov_ok = True
        for i_, k_ in last.items():
            r.set_camera_view(w, h, poses[k_].view)
            r.tile_gaussians(tw, th, poses[k_].view)
            ref_, _ = r.render(poses[k_].position, pack, want_radiance=False)
            mov_ok = mov_ok and bool((images[i_].cpu().numpy().view(np.uint32) == ref_.reshape(-1)).all())
        # (three samples, the median reported: one in three runs of this section read 130-170 us per frame on an otherwise
        # unchanged build -- a stall of ~20 ms somewhere in 200 frames -- while a stand-alone loop of 2000 such frames is steady at
        # 38.8 us, profiles/r04_experiments.md)
        mov_serial_samples = []
        for _ in range(3):
            t2 = time.perf_counter()
            run_moving(n_serial, 1)
            barrier()
            mov_serial_samples.append((time.perf_counter() - t2) / n_serial * 1e3)
        mov_serial_ms = sorted(mov_serial_samples)[1]
        moving = {"what": "the view turned by 1e-4 degrees per step: rays, cone table, lists and every launch per frame",
                  "frames_in_flight": nfl, "ms_per_step": mov_ms, "value": w * h / (mov_ms * 1e-3) / 1e6,
                  "serial_ms_per_frame": mov_serial_ms, "serial_ms_per_frame_samples": mov_serial_samples, "serial_value": w * h / (mov_serial_ms * 1e-3) / 1e6, "steps": n_serial,
                  "frames_equal_reference": mov_ok}
        # back to the static view of the timed region (the statistics pass and the frame check below use it)
        for r_ in ctxs:
            r_.set_camera_view(w, h, view)
        run(2 * nctx, nctx)
        barrier()
        sweep = {}
        for nf in range(1, nctx + 1):
            run(4 * nf, nf)
            barrier()
            t2 = time.perf_counter()
            run(n_serial, nf)
            barrier()
            sweep[str(nf)] = (time.perf_counter() - t2) / n_serial * 1e3
        run(2 * nctx, nctx)   # every frame buffer holds the static frame again
        barrier()
        # the same loop with the frame buffers declared "retained" (vrt_hip_frame_retained_device: each buffer is written by its
        # context only and still holds its previous frame, like the reference's `image`): the list kernel then resets only the
        # cells that went dark instead of writing 16 MB of background over background.  Reported next to the headline, which
        # keeps the full clear of a caller's buffer.
        rframes = [r_.frame_call(tw, th, view, origin, pack, retained=True) for r_ in ctxs]
        for k in range(4 * nfl):
            rframes[k % nfl](img_ptrs[k % nfl], sps[k % nfl])
        barrier()
        t2 = time.perf_counter()
        for k in range(n_serial):
            rframes[k % nfl](img_ptrs[k % nfl], sps[k % nfl])
        barrier()
        retained_ms = (time.perf_counter() - t2) / n_serial * 1e3
        t2 = time.perf_counter()
        for k in range(n_serial):
            rframes[0](img_ptrs[0], sps[0])
        barrier()
        retained_serial_ms = (time.perf_counter() - t2) / n_serial * 1e3
        sweep["retained_frame_buffers"] = {"frames_in_flight": nfl, "ms_per_step": retained_ms, "serial_ms_per_frame": retained_serial_ms}
    # What the speed costs in accuracy (round-3 verdict): the same loop at EXACT settings -- no budgeted prune, table kernel off
    # (cull_eps stays: its thresholds' share is <= 2.5e-6) -- in flight and serial, after the timed region.
    exact = None
    if solo and world == 1 and not args.plane_arrays:
        for r_ in ctxs:
            r_.set_cull_prune(0.0)
            r_.set_table_step(0.0)
        run(8 * nctx, nctx)
        barrier()
        t2 = time.perf_counter()
        run(n_serial)
        barrier()
        ex_ms = (time.perf_counter() - t2) / n_serial * 1e3
        t2 = time.perf_counter()
        run(n_serial, 1, serial=True)
        barrier()
        ex_serial_ms = (time.perf_counter() - t2) / n_serial * 1e3
        _, ex_rad = r.render(origin, pack)
        exact = {"what": "--cull-prune 0 and the table kernel off (vrt_hip_set_table_step(0)); level-wise cull thresholds as timed",
                 "frames_in_flight": nfl, "ms_per_step": ex_ms, "value": w * h / (ex_ms * 1e-3) / 1e6,
                 "ms_per_frame": ex_serial_ms, "serial_value": w * h / (ex_serial_ms * 1e-3) / 1e6, "steps": n_serial}
        for r_ in ctxs:
            r_.set_cull_prune(args.cull_prune)
            r_.set_table_step(args.table_step)
        run(4 * nctx, nctx)
        barrier()
    # N > 1: where a rank's time goes -- the timed loop once more with the gatherer's phase timers on (sharding.py: "render" on the
    # batch's render stream, "gather" from enqueue until the frame stream may use it, "assemble" on rank 0), all ranks to rank 0
    per_rank = None
    if not solo:
        fg.timing = True
        fg.phase_ms()
        run(args.steps)
        barrier()
        ph = fg.phase_ms()
        fg.timing = False
        nfr = max(1, ph.get("frames", 0))
        mine = torch.tensor([rank, ph.get("render", 0.0) / nfr, ph.get("gather", 0.0) / nfr, ph.get("gather_nccl", -nfr) / nfr,
                             ph.get("assemble", 0.0) / nfr, ph.get("host_in_run", 0.0) / nfr, float(ph.get("batches_gathered_twice", 0))],
                            dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [{"rank": int(v[0]), "render_ms_per_frame": float(v[1]), "gather_ms_per_frame_upper_bound": float(v[2]),
                     "gather_collective_ms_per_frame": (float(v[3]) if float(v[3]) >= 0 else None), "assemble_ms_per_frame": float(v[4]),
                     "host_ms_per_frame_in_loop": float(v[5]), "batches_gathered_twice": int(v[6])} for v in (t_.cpu().tolist() for t_ in allr)]
    # N > 1 control (after the timed region): the same number of WHOLE frames on every rank, no exchange at all -- what `--parallel frames`
    # times, N x one GPU by construction.  Beside `value` it separates what the tile-sharded path loses to its collective from what a rank
    # loses to anything else (a busy host, a slow device).  The collective part is unconditional: a rank whose control fails reports NaN.
    control = None
    if not solo:
        c_ms = float("nan")
        try:
            c_ctx = []
            for _ in range(2):
                rc = pkg.Renderer(local_rank)
                rc.set_gaussians(g)
                rc.set_camera_view(w, h, view) if not args.plane_arrays else rc.set_plane(w, h, *cam.plane())
                rc.set_options(pkg.EXP_VCL, pkg.ERF_AS, args.cull_eps)
                rc.set_cull_prune(args.cull_prune)
                rc.set_table_step(args.table_step)
                rc.set_shard(0, 1)
                rc.tile_gaussians_device(tw, th, view, 0)                 # sizes the tile grid (one-time host sync)
                c_ctx.append(rc)
            c_img = [torch.zeros(w * h, dtype=torch.int32, device="cuda") for _ in c_ctx]
            c_st = [torch.cuda.Stream() for _ in c_ctx]
            c_fr = [rc.frame_call(tw, th, view, origin, pack) for rc in c_ctx]
            for i_ in range(8):                                           # set-up: the library's per-context feedback settles
                c_fr[i_ % 2](c_img[i_ % 2].data_ptr(), c_st[i_ % 2].cuda_stream)
            torch.cuda.synchronize()
        except Exception as e_:                                           # (reported, never raised: the ranks must reach the barrier together)
            c_fr = None
            control = {"error": repr(e_)[:200]}
        dist.barrier()
        if c_fr is not None:
            t2 = time.perf_counter()
            for i_ in range(args.steps):
                c_fr[i_ % 2](c_img[i_ % 2].data_ptr(), c_st[i_ % 2].cuda_stream)
            for s_ in c_st:
                s_.synchronize()
            c_ms = (time.perf_counter() - t2) * 1e3
        c_t = torch.tensor([c_ms], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        allc = [torch.zeros_like(c_t) for _ in range(world)]
        dist.all_gather(allc, c_t)
        c_all = [float(t_.item()) for t_ in allc]
        if control is None and all(v == v for v in c_all):
            control = {"what": "every rank renders --steps whole frames on its own (two in flight), no collective: `--parallel frames`",
                       "ms_per_rank": c_all, "value": w * h * args.steps * world / (max(c_all) * 1e-3) / 1e6, "scaling": "weak"}
        elif control is None:
            control = {"error": "a rank's control loop failed", "ms_per_rank": c_all}
        if c_fr is not None:
            for rc in c_ctx:
                rc.close()
    r.enable_kernel_timing(1)
    run(max(50, min(args.steps, 100)), 1, serial=True)
    barrier()
    seq = r.kernel_timing()
    r.enable_kernel_timing(False)
    kt["lists_ms"], kt["dense_ms"], kt["render_serial_ms"] = seq["lists_ms"], seq["dense_ms"], seq["render_ms"]
    if not kt["launches"] or not kt["render_ms"] > 0:   # frame batches carry no per-launch events: the serial launch's duration
        kt["render_ms"] = seq["render_ms"]

    red_dev = "cuda" if backend == "nccl" else "cpu"
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    kern_ms = torch.tensor([kt["render_ms"]], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(kern_ms, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    kernel_ms = float(kern_ms.item())

    if rank == 0:
        rays = w * h * args.steps * (world if (solo and world > 1) else 1)   # frames mode: every rank renders K whole frames
        ms_per_step = elapsed / args.steps * 1e3
        # ---- statistics pass (outside the timed region): list lengths, shaded blocks ----
        counts = r.tile_counts()
        n_entries = int(counts.sum())
        r.enable_stats(True)
        r.set_shard(0, 1)
        _img, _rad = r.render(origin, pack)
        # the gathered + assembled frame of the timed loop must be the single-GPU frame, bit for bit
        bad = [(i_, int((im.cpu().numpy().view(np.uint32) != _img.reshape(-1)).sum())) for i_, im in enumerate(images)]
        bad = [b_ for b_ in bad if b_[1]]
        frame_ok = not bad
        if bad:
            print(f"bench.py: frame buffers that differ from the single-GPU frame (buffer, pixels): {bad}", file=sys.stderr)
        st = r.stats()
        r.enable_stats(False)
        r.set_shard(*((0, 1) if solo else (rank, world)))
        sb = max(st["shaded_blocks"], 1)
        # ---- roofline.  Algorithmic HBM bytes: 4 B per ray written (+12 B per ray
        # read with --plane-arrays), the Gaussian rows once (64 B each), the candidate lists once (4 B per entry).
        # Dominant kernel = render_kernel: it writes the rays it shades and reads their cells' lists; the other rays
        # of the frame are cleared by the list kernel, so the FRAME figure is given next to it.
        per_ray = 4 + (12 if args.plane_arrays else 0)
        share = 1 if solo else world   # tile sharding: a rank's kernels see 1/world of the frame
        render_bytes = (sb * 64 * per_ray + 4 * st["tile_entries"] + 64 * len(g)) / share
        frame_bytes = (w * h * per_ray + 64 * len(g) + 4 * n_entries) / share
        # the dominant kernel's duration is that of the SERIAL launch (frames in flight stretch a launch: it shares the GPU with
        # the other contexts' kernels, and is then no component of the step); the overlapped figure is kept beside it
        kernel_ms_overlapped = kernel_ms
        kernel_ms = kt["render_serial_ms"] if (solo and kt["render_serial_ms"] > 0) else kernel_ms
        render_gbs = render_bytes / (kernel_ms * 1e-3) / 1e9
        # the whole frame: wall time per step of the timed region (one rank: nothing but the launch sequence is in it)
        frame_ms = ms_per_step if solo else kt["lists_ms"] + kt["render_ms"] + kt["dense_ms"]
        frame_gbs = frame_bytes / (frame_ms * 1e-3) / 1e9
        # ---- VALU (what actually bounds the path).  ALGORITHMIC lane-ops: per (ray, candidate of its block)
        # one cull / precompute = 14 lane-ops, per surviving (emitter i, sample k, absorber j) of a ray 16 lane-ops
        # (1 fma + A&S erf with 1 rcp + 1 fma); both counted by the statistics pass above, nothing skipped is credited.
        # 64 lane-ops = one wave-instruction equivalent.
        culls = 64 * st["list_entries"]
        terms = 5 * st["lane_pairs"]
        alg_wave_instr = (14 * culls + 16 * terms) / 64.0
        valu = {"algorithmic": {"ray_candidate_culls": culls, "surviving_ikj_terms": terms,
                                "lane_ops": 14 * culls + 16 * terms, "wave_instruction_equivalents": alg_wave_instr,
                                "achieved_per_s": alg_wave_instr / (kernel_ms * 1e-3),
                                "frac": alg_wave_instr / (kernel_ms * 1e-3) / VALU_PEAK_WAVE_INSTR,
                                "frac_serial_launch": alg_wave_instr / (kt["render_serial_ms"] * 1e-3) / VALU_PEAK_WAVE_INSTR,
                                # per FRAME TIME of the timed region: what the GPU as a whole sustains with frames in flight
                                # (a launch that shares the GPU with other frames' kernels lasts longer than its share of it)
                                "frac_of_frame_time": (alg_wave_instr / (ms_per_step * 1e-3) / VALU_PEAK_WAVE_INSTR) if solo else None},
                "peak_per_s": VALU_PEAK_WAVE_INSTR,
                "peak_note": "spec: 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 VALU instruction (MI355X_MICROARCH.md)"}
        try:
            valu["measured_fma_issue"] = json.load(open(VALU_MEASURED_FILE))
        except (OSError, ValueError):
            pass
        # ---- constants from separate PMC passes (rocprofv3 --pmc cannot run inside this process): HBM traffic and
        # EXECUTED instruction counts per launch.  Used only when they were taken with these very kernel sources.
        traffic, traffic_frame = None, None
        pmc_path, pmc = latest_pmc_summary()
        sha = kernel_source_sha()
        if pmc is not None and world == 1 and args.grid == 64 and w == 2048 and not args.plane_arrays:
            if pmc.get("kernel_source_sha") == sha:
                try:
                    traffic = pmc["render_kernel_traffic_bytes_per_launch"]["lower"]
                    traffic_frame = pmc["frame_hbm_bytes"]["write"] + pmc["frame_hbm_bytes"]["fetch_raw"]
                    sq = pmc.get("sq_counters_bench_scene_per_launch", {})
                    nv = sq.get("render_kernel", {}).get("SQ_INSTS_VALU")
                    nl = sq.get("build_tile_lists_kernel", {}).get("SQ_INSTS_VALU", 0.0)
                    if nv:
                        rate = nv / (kernel_ms * 1e-3)
                        valu["executed"] = {"wave_instructions_per_launch": nv, "achieved_per_s": rate,
                                            "frac": rate / VALU_PEAK_WAVE_INSTR,
                                            "frac_serial_launch": nv / (kt["render_serial_ms"] * 1e-3) / VALU_PEAK_WAVE_INSTR,
                                            "algorithmic_share": alg_wave_instr / nv,
                                            "frame_wave_instructions": nv + nl,
                                            "frame_frac_of_frame_time": ((nv + nl) / (ms_per_step * 1e-3) / VALU_PEAK_WAVE_INSTR) if solo else None}
                    valu["from"] = {"file": pmc_path, "kernel_source_sha": sha, "kind": "constants of separate --pmc passes"}
                except KeyError:
                    traffic, traffic_frame = None, None
            else:
                valu["from"] = {"file": pmc_path, "dropped": "taken with other kernel sources "
                                f"({pmc.get('kernel_source_sha')} != {sha}): traffic and executed counts omitted"}
        result = {
            "metric": "Mrays/sec (whole node), 2048^2 image, 64x64 Gaussian grid", "value": rays / elapsed / 1e6,
            "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            # ms/frame of the metric = what ONE caller of the vrt:: API waits for a frame (strictly serial frames on one
            # context, measured after the timed region); ms_per_step is the timed loop's wall time per step, which with
            # several frames in flight is a throughput figure
            "ms_per_frame": serial_ms if solo else ms_per_step,
            # how long the host took to ENQUEUE the K timed steps (rank 0): where this approaches ms_per_step the loop is host-bound
            "host_enqueue_ms_per_step": t_enqueued / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak" if (solo and world > 1) else "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"-g {args.grid} -w {w} (tiles {args.tiles}, mode-8 packing, cull_eps {args.cull_eps:g}, ray-level prune {args.cull_prune:g}, "
                                   f"{'plane arrays' if args.plane_arrays else 'in-kernel rays'})",
                       "gaussians": int(len(g)), "rays_per_frame": w * h, "tile_list_entries": n_entries,
                       "parallelism": (f"whole frames on each of {world} ranks, no collective" if (solo and world > 1) else
                                       f"tile-shard x{world}" + (f" + RCCL {args.exchange} of sparse shards every {F} frames, assembled on rank 0" if world > 1 else "")
                                       + (f", {F} frames per kernel launch" if (world > 1 and not args.no_batch) else "")),
                       "shard_transport": (None if solo else {"format": "sparse: 32x32-px cells that hold something", "bytes_per_frame":
                                           fg.bytes_moved / max(1, fg.frames_moved), "compact_shards_would_be": (world - 1) * w * h * 4 // world,
                                           "batches_gathered_twice": fg.regathered}),
                       "frames_per_gather": (None if solo else F),
                       "gather_batches_in_timed_region": (None if solo else (args.steps + F - 1) // F),
                       "frames_in_flight": nfl,
                       "stream_probe_us_per_frame": stream_probe,
                       "frame_equals_single_gpu_frame": frame_ok},
            "roofline": {"bound": "hbm", "achieved": render_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": render_gbs / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_from": (valu.get("from") if traffic is not None else None),
                         "kernel": "render_kernel<VCL,AS,4>", "kernel_ms": kernel_ms, "kernel_ms_with_frames_in_flight": kernel_ms_overlapped,
                         "algorithmic_bytes": render_bytes,
                         # (an event pair with nothing between reads ~5 us on this stack -- measured in round 1 on an empty
                         # launch slot; rocprofv3's kernel durations under profiles/ are the event figures minus that)
                         # strictly serial frames, after the timed region: four events per frame
                         "launch_sequence_ms": {"lists": kt["lists_ms"], "render_kernel": kt["render_serial_ms"],
                                                "render_dense_kernel": kt["dense_ms"]},
                         "frame": {"algorithmic_bytes": frame_bytes, "ms": frame_ms, "achieved": frame_gbs,
                                   "frac": frame_gbs / HBM_PEAK_GBS, "traffic": traffic_frame},
                         "note": "the path is VALU/transcendental-bound, not HBM-bound (SURVEY 7 hard part 4): see valu; "
                                 "traffic = WRITE_SIZE + raw FETCH_SIZE of separate PMC passes (profiles/); with several frames "
                                 "in flight a launch shares the GPU with the other contexts' kernels and lasts longer "
                                 "(kernel_ms_with_frames_in_flight); kernel_ms is the serial launch, a component of ms_per_frame"},
            # the same loop with one context: frame k+1 starts when frame k is done
            # the same loop at exact settings, and what the timed settings deviate by (checked by the oracle below: "parity")
            "exact_settings": exact,
            "collective": ({"backend": dist.get_backend(), "world_size": dist.get_world_size(), "frame_parallel_control": control, "per_rank": per_rank,
                            "per_rank_note": "a repeat of the timed loop with phase timers on (after the timed region); gather = from enqueue "
                                             "until the frame stream may use the result (upper bound of the collective; its own duration "
                                             "where TORCH_NCCL_ENABLE_TIMING=1 makes the backend keep it)"} if world > 1 else None),
            "serial": {"frames_in_flight": 1, "ms_per_step": serial_ms, "value": w * h / (serial_ms * 1e-3) / 1e6, "steps": n_serial},
            "moving_camera": moving,
            "in_flight_sweep_ms_per_step": sweep,   # frames in flight -> ms per step; + the retained-buffer variant of the loop
            "valu": {**valu, "blocks": st["blocks"], "shaded_blocks": st["shaded_blocks"], "dense_blocks": st["dense_blocks"],
                     "mean_cell_list": st["tile_entries"] / sb, "mean_block_list": st["list_entries"] / sb,
                     "mean_ray_list": st["lane_entries"] / (sb * 64), "mean_block_longest_ray_list": st["lane_max_entries"] / sb,
                     "dense_overflow_blocks": st["overflow_blocks"]},
        }
        if not args.no_cpu_baseline and world == 1:
            # the oracle as the CHECKER of the frame the timed loop renders (after the timed region), then as the CPU baseline
            if args.parity_pixels > 0 and not args.plane_arrays:
                result["parity"] = parity_check(scene, g, w, h, args.tiles, _rad, args.parity_pixels, 2048 + args.grid, args.cull_eps,
                                                args.cull_prune, 2.5e-5, st["dense_blocks"] > 0)
                if exact is not None:
                    result["parity"]["timed_vs_exact_settings_max_abs"] = float(np.abs(_rad - ex_rad).max())
            result["cpu_baseline"] = cpu_baseline(scene, g, w, h, args.tiles, args.grid, "")
            result["speedup_vs_cpu_baseline"] = result["value"] / result["cpu_baseline"]["value"]
        print(json.dumps(result))
    barrier()
    for r_ in ctxs:
        r_.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
