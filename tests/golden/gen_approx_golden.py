"""Generates tests/golden/approx_ref.npz from the REAL reference approximations
(/root/reference/src/vrt/approx.{h,cpp} compiled in place into oracle/_ref by oracle/Makefile).

Run in the build container (where /root/reference exists):  python tests/golden/gen_approx_golden.py
Grids: the reference's own accuracy experiment (tests/accuracy.cpp:19, 35-39: erf on [-6,6] step .1,
exp on [-16,0] step .1) plus a denser/wider sweep.  Only numbers are stored -- no reference source.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import oracle as O  # noqa: E402

O.build()
assert O.ref_lib() is not None, "oracle/_ref missing: needs /root/reference"

erf_x = np.unique(np.concatenate([np.arange(-6, 6.0001, 0.1), np.linspace(-9, 9, 1153), [0.0, -0.0, 1e-8, -1e-8]])).astype(np.float32)
exp_x = np.unique(np.concatenate([np.arange(-16, 0.0001, 0.1), np.linspace(-87, 0.5, 1401)])).astype(np.float32)
out = dict(erf_x=erf_x, exp_x=exp_x)
for key, fn in [("as_erf", "ref_as_erf"), ("simd_as_erf", "ref_simd_as_erf"), ("spline_erf", "ref_spline_erf"),
                ("spline_erf_mirror", "ref_spline_erf_mirror"), ("taylor_erf", "ref_taylor_erf"),
                ("simd_spline_erf", "ref_simd_spline_erf"), ("simd_spline_erf_mirror", "ref_simd_spline_erf_mirror"),
                ("simd_taylor_erf", "ref_simd_taylor_erf")]:
    out[key] = O.ref_map(fn, erf_x)
for key, fn in [("vcl_exp", "ref_simd_vcl_exp"), ("fast_exp", "ref_fast_exp"), ("spline_exp", "ref_spline_exp"),
                ("simd_fast_exp", "ref_simd_fast_exp"), ("simd_spline_exp", "ref_simd_spline_exp")]:
    out[key] = O.ref_map(fn, exp_x)
# libm columns (the template defaults expf/erff of rt.h:32), float64 math rounded to float
import math
out["libm_erf"] = np.array([math.erf(float(v)) for v in erf_x], np.float32)
out["libm_exp"] = np.exp(exp_x.astype(np.float64)).astype(np.float32)
np.savez_compressed(os.path.join(HERE, "approx_ref.npz"), **out)
print("wrote approx_ref.npz:", {k: v.shape for k, v in out.items()})
