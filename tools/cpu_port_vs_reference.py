"""How fast is the CPU baseline port (oracle/vrt_cpu_simd.*) against the REAL reference on the same host?

The reference's render path (rt.h) needs glm, which the image lacks, so it cannot be rebuilt here without stand-in headers; what
exists is the survey's probe of the real reference in THIS container (Intel Xeon @ 2.1 GHz, one thread, AVX-512,
g++ 11.4 -O3 -march=native -ffast-math, mode 4 = untiled simd_render_image):

    N = 256 Gaussians (-g 16), 64 x 64 rays   2.55 s   (5.26e8 (ray, i, k, j) inner terms / s)
    N = 16  Gaussians (-g 4), 256 x 256 rays  0.159 s  (5.27e8)

This script times the port on the same two workloads, one thread, on the machine it runs on (run it in the build container for the
same-host comparison) and prints a table:

    python tools/cpu_port_vs_reference.py > profiles/rNN_cpu_port_vs_reference.md
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O  # noqa: E402

SURVEY = {(16, 64): 2.55, (4, 256): 0.159}   # (grid, width) -> seconds of the real reference, SURVEY.md section 6


def one(grid, w, repeats=3):
    g = O.grid_scene(grid)
    cam, _ = O.cli_camera(w, w)
    plane = O.camera_plane(cam)
    n = len(g)
    # untiled (mode 4): one tile that holds every Gaussian, covering the whole image
    tiles = dict(tw=np.float32(2.0), th=np.float32(2.0), w=1, h=1, offsets=np.array([0, n], np.uint32), indices=np.arange(n, dtype=np.uint32))
    best, terms, simd = 1e30, 0, 0
    for _ in range(repeats):
        t0 = time.perf_counter()
        _, terms, simd = O.simd_render_tiled(w, w, plane, cam.position[:], g, tiles, None, 1)
        best = min(best, time.perf_counter() - t0)
    return n, terms, simd, best


def main():
    O.build()
    cpu = ""
    try:
        cpu = [ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")][0]
    except (OSError, IndexError):
        pass
    print(f"# CPU baseline port against the reference's own speed, one thread, untiled mode-4 workloads of SURVEY.md section 6\n")
    print(f"host: {cpu}; port = oracle/vrt_cpu_simd.inc; reference seconds = the survey's probe of the real T-SIMD build in the build container\n")
    print("| workload | inner terms | port, s (best of 3) | port, terms/s | reference, s (survey) | port speed / reference speed |")
    print("|---|---|---|---|---|---|")
    ratios = []
    for (grid, w), ref_s in SURVEY.items():
        n, terms, simd, dt = one(grid, w)
        ratios.append(ref_s / dt)
        print(f"| N = {n} (`-g {grid}`), {w} x {w} rays, SIMD width {simd} | {terms:.3e} | {dt:.3f} | {terms / dt:.3e} | {ref_s} | {ref_s / dt:.2f} |")
    print(f"\nport_vs_reference_same_host: {min(ratios):.2f} .. {max(ratios):.2f}")


if __name__ == "__main__":
    main()
