#!/usr/bin/env python3
"""Summarise rocprofv3 PMC passes of tools/pmc_workload.py into per-launch HBM traffic per kernel.

  cd /tmp && export TMPDIR=/tmp
  for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $c --kernel-trace --output-format csv -d OUT/pmc_$c -o pmc -- \
      python3 tools/pmc_workload.py; done
  python tools/pmc_traffic.py OUT > profiles/rNN_pmc_traffic.json

Units and gfx950 corrections follow MI355X_MICROARCH.md (HBM section): counters are in KiB; WRITE_SIZE is exact for
this kernel family (checked here: the calibration frames of an empty scene must write w*h*4 bytes + ~50 KB of list
bookkeeping); FETCH_SIZE under-reports WIDE (16 B/lane) streaming reads by 2x -- these kernels read 4..16 B per lane
through gathers, so the raw value is a lower bound and 2x it an upper bound; both are reported.
"""
import collections
import csv
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_sha():
    """Same identity bench.py computes: the summary is only used with the kernel sources it was taken with."""
    h = hashlib.sha256()
    for name in ("vrt_kernels_common.hpp", "vrt_block_kernel.hip", "vrt_table_kernel.hip", "vrt_kernels.hip", "vrt_kernels.h",
                 "vrt_device_math.h"):
        with open(os.path.join(ROOT, "simd-gaussian-ray-tracing_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


out = sys.argv[1]
res = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = list(csv.DictReader(open(f"{out}/pmc_{counter}/pmc_counter_collection.csv")))
    per = collections.defaultdict(list)
    for r in rows:
        name = r["Kernel_Name"]
        key = ("render_dense_kernel" if "render_dense" in name else "render_kernel" if "render_kernel" in name
               else "build_tile_lists_kernel" if "build_tile" in name else None)
        if key:
            per[key].append(float(r["Counter_Value"]) * 1024.0)
    for k, v in per.items():
        # the first 8 launches are the bench scene, the last 4 the empty calibration scene
        res.setdefault(k, {})[counter] = {"bench_scene_bytes_per_launch": sum(v[:8]) / 8, "empty_scene_bytes_per_launch": sum(v[8:]) / max(len(v[8:]), 1)}
w = 2048
cal = res["build_tile_lists_kernel"]["WRITE_SIZE"]["empty_scene_bytes_per_launch"]
summary = {
    "workload": "-g 64 -w 2048, tiles 16 (tools/pmc_workload.py)",
    "kernel_source_sha": kernel_source_sha(),
    "write_size_calibration": {"expected_bytes_clear_only": w * w * 4, "measured_bytes_list_kernel_empty_scene": cal,
                               "ratio": cal / (w * w * 4)},
    "per_kernel": res,
    "frame_hbm_bytes": {
        "write": sum(res[k]["WRITE_SIZE"]["bench_scene_bytes_per_launch"] for k in ("build_tile_lists_kernel", "render_kernel")),
        "fetch_raw": sum(res[k]["FETCH_SIZE"]["bench_scene_bytes_per_launch"] for k in ("build_tile_lists_kernel", "render_kernel")),
    },
    "render_kernel_traffic_bytes_per_launch": {
        "lower": res["render_kernel"]["WRITE_SIZE"]["bench_scene_bytes_per_launch"] + res["render_kernel"]["FETCH_SIZE"]["bench_scene_bytes_per_launch"],
        "upper": res["render_kernel"]["WRITE_SIZE"]["bench_scene_bytes_per_launch"] + 2 * res["render_kernel"]["FETCH_SIZE"]["bench_scene_bytes_per_launch"],
    },
    "note": "render_dense_kernel's WRITE_SIZE when it runs empty (first 8 frames after a scene change) is scratch spill "
            "traffic of its prologue (9 VGPRs x 4096 waves) and write-back of earlier kernels' dirty L2 lines",
}
try:  # optional third pass: --pmc SQ_INSTS_VALU ... (instruction counts per launch)
    rows = list(csv.DictReader(open(f"{out}/pmc_SQ/pmc_counter_collection.csv")))
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        name = r["Kernel_Name"]
        key = ("render_dense_kernel" if "render_dense" in name else "render_kernel" if "render_kernel" in name
               else "build_tile_lists_kernel" if "build_tile" in name else None)
        if key:
            per[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    summary["sq_counters_bench_scene_per_launch"] = {k: {c: sum(v[:8]) / 8 for c, v in d.items()} for k, d in per.items()}
except FileNotFoundError:
    pass
print(json.dumps(summary, indent=1))
