"""Golden tables the reference itself holds for erf / exp: the output of its accuracy experiment
(/root/reference/src/volumetric-ray-tracer/tests/accuracy.cpp:9-58), kept in the reference tree as the plot data of the thesis:

    thesis/plots/cmp_erf_approx.tex   x in [-6, 6] step .1f (float-accumulated: -5.7000003 ...), 121 points per series:
                                      spline, spline mirror, taylor, abramowitz stegun, svml, std::erf
    thesis/plots/cmp_erf_err.tex      the same series minus std::erf (what the thesis plots as the error)
    thesis/plots/cmp_exp_approx.tex   x in [-16, 0) in steps of .1f as the experiment's loops produce them, 160 points per series:
                                      spline, fast, vcl, svml, std::exp
    thesis/plots/cmp_exp_err.tex      the same minus std::exp
    thesis/plots/taylor_erf.tex       erf and its Taylor polynomials of 4 / 8 / 10 / 16 terms on [-3, 3]

The values are float32 printed in shortest round-trip form, so parsing them back to float32 recovers the reference's bits.
Only NUMBERS are taken (the pgfplots `table` rows); no text of the files is kept.  Run in the build container (needs /root/reference):

    python tests/golden/gen_thesis_plot_golden.py        ->  tests/golden/thesis_plots.npz

Keys: "<file>/<series>/x", "<file>/<series>/y" (float32) with the series named by the plot's legend entry
(spaces and "::" replaced by "_").
"""
import os
import re
import sys

import numpy as np

PLOTS = "/root/reference/thesis/plots"
FILES = ("cmp_erf_approx", "cmp_erf_err", "cmp_exp_approx", "cmp_exp_err", "taylor_erf")
HERE = os.path.dirname(os.path.abspath(__file__))

ROW = re.compile(r"^\s*(-?[0-9.eE+-]+|NaN|Inf|-Inf)\s+(-?[0-9.eE+-]+|NaN|Inf|-Inf)\s*\\\\\s*$")
LEGEND = re.compile(r"\\addlegendentry\s*\{(.*?)\}")


def parse(path):
    """[(legend, x float32[], y float32[])] in file order: the rows between an \\addplot and its \\addlegendentry."""
    out, xs, ys = [], [], []
    for line in open(path, encoding="utf-8"):
        m = ROW.match(line)
        if m:
            xs.append(np.float32(m.group(1))); ys.append(np.float32(m.group(2)))
            continue
        m = LEGEND.search(line)
        if m:
            out.append((m.group(1).strip(), np.array(xs, np.float32), np.array(ys, np.float32)))
            xs, ys = [], []
    return out


def main():
    if not os.path.isdir(PLOTS):
        sys.exit(f"{PLOTS} not found: run this in the build container")
    arrays = {}
    for f in FILES:
        for name, x, y in parse(os.path.join(PLOTS, f + ".tex")):
            key = f + "/" + name.replace("::", "_").replace(" ", "_")
            assert len(x) and len(x) == len(y), key
            arrays[key + "/x"] = x
            arrays[key + "/y"] = y
            print(f"{key:45s} {len(x):4d} points, x {x[0]:+.7g} .. {x[-1]:+.7g}")
    np.savez_compressed(os.path.join(HERE, "thesis_plots.npz"), **arrays)
    print("wrote", os.path.join(HERE, "thesis_plots.npz"), f"({len(arrays) // 2} series)")


if __name__ == "__main__":
    main()
