"""bench.py's contract on one GPU: ONE JSON line with the driver's keys, `value` = rays of exactly K timed steps over their
wall time, the roofline and CPU-baseline objects, every frame buffer of the timed loop equal to a reference render."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("steps,in_flight", [(20, 2), (40, 2)])
def test_bench_line(tmp_path, steps, in_flight):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", "3", "--setup-ms", "5"],
                       cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{")                   # one line, nothing else on stdout
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == steps and d["warmup"] == 3 and d["higher_is_better"] is True
    assert d["unit"] == "Mrays/s" and d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "2048" in d["metric"] and "64x64" in d["metric"]
    cfg = d["config"]
    assert "workload" in cfg and "-g 64 -w 2048" in cfg["workload"] and "model" not in cfg
    assert cfg["frames_in_flight"] == in_flight and cfg["frame_equals_single_gpu_frame"] is True
    # value is whole-job throughput of the timed steps: rays per step / time per step
    assert abs(d["value"] - 2048 * 2048 / (d["ms_per_step"] * 1e-3) / 1e6) <= 1e-6 * d["value"]
    assert 10e-3 < d["ms_per_step"] < 80e-3 and 20e-3 < d["ms_per_frame"] < 100e-3
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["traffic"] is None or r["traffic"] > 0
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0
    # the dominant kernel's duration is that of the serial launch: a component of what one caller waits for a frame
    assert 0 < r["kernel_ms"] <= d["ms_per_frame"] and r["kernel_ms_with_frames_in_flight"] > 0
    v = d["valu"]
    assert 0 < v["algorithmic"]["frac_of_frame_time"] < 1
    # a frame loop whose camera moves: nothing memoised; slower than the static view, same images as fresh renders
    m = d["moving_camera"]
    assert m["frames_equal_reference"] is True and m["frames_in_flight"] == in_flight
    assert d["ms_per_step"] * 0.8 < m["ms_per_step"] < 0.2 and d["ms_per_frame"] * 0.8 < m["serial_ms_per_frame"] < 0.3
    assert abs(m["value"] - 2048 * 2048 / (m["ms_per_step"] * 1e-3) / 1e6) <= 1e-6 * m["value"]
    sw = d["in_flight_sweep_ms_per_step"]
    ret = sw.pop("retained_frame_buffers")
    assert sorted(sw) == [str(k) for k in range(1, 5)] and all(0 < t < 0.2 for t in sw.values())   # 1..4 in flight, whatever the timed region used
    assert 0 < ret["ms_per_step"] < 0.2 and 0 < ret["serial_ms_per_frame"] < 0.2
    # what the speed costs in accuracy (round-3 verdict): the same loop at exact settings, and the oracle's verdict on the timed frame
    ex = d["exact_settings"]
    assert ex["frames_in_flight"] == in_flight and 0 < ex["ms_per_step"] < 0.2 and 0 < ex["ms_per_frame"] < 0.2
    assert abs(ex["value"] - 2048 * 2048 / (ex["ms_per_step"] * 1e-3) / 1e6) <= 1e-6 * ex["value"]
    assert ex["ms_per_step"] > 0.9 * d["ms_per_step"]            # the exact settings are not the faster ones
    par = d["parity"]
    assert par["pixels"] >= 256 and par["tolerance"] == 1e-4 and par["oracle_peak_radiance"] > 5e-3
    assert 0 < par["max_abs"] <= par["bound"] <= 5e-5, par
    assert par["timed_vs_exact_settings_max_abs"] <= par["bound"]
    assert d["collective"] is None                                # N = 1: no collective
