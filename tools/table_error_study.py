"""Interpolation error of the table kernel (csrc/vrt_kernels.hip, render_table_body): a unit Abramowitz-Stegun erf
(approx.cpp:90-110 of the reference) tabulated at node spacing u and read off by 4-point Lagrange interpolation.

The A&S erf is the odd extension of a rational function: its second derivative jumps by 0.586 at 0 (a "kink").  For a
kink at position kappa (in node units, relative to the left node of the interval the sample lies in) the script scans the
largest error over that interval for all phases and checks the constants the kernel uses:

    kink in the interval's own bin      (kappa = theta)      : err <= W_own(theta) u^2 + 0.36 u^4
    kink in the bin to the left         (kappa = theta - 1)  : err <= W_left(theta) u^2 + 0.36 u^4,   W_left  = 0.0212 theta^2
    kink in the bin to the right        (kappa = theta + 1)  : err <= W_right(theta) u^2 + 0.36 u^4,  W_right = 0.0212 (1 - theta)^2
    kink outside the stencil                                 : err <= 0.36 u^4
    W_own(theta) = 0.0212 min(1, 0.28 + 2.58 |theta - 1/2|)

Writes the table that DESIGN.md section 4 quotes.  numpy only; runs in a few seconds.
    python tools/table_error_study.py > profiles/r03_table_error_study.md
"""
import numpy as np

a = [0.278393, 0.230389, 0.000972, 0.078108]

def erf_as(x):
    t = np.abs(x)
    p = 1 + t * (a[0] + t * (a[1] + t * (a[2] + t * a[3])))
    return np.sign(x) * (1 - 1 / p ** 4)

def lagw(t):
    return np.stack([t * (t - 1) * (t - 2) / -6, (t + 1) * (t - 1) * (t - 2) / 2, (t + 1) * t * (t - 2) / -2, (t + 1) * t * (t - 1) / 6])

T = np.linspace(0, 1, 201)
WT = lagw(T)

def err(u, kappa):
    """largest |interpolant - erf| over the interval [0, 1) (node units) for a kink at kappa"""
    nodes = np.arange(-1, 3)
    f = erf_as((nodes - kappa) * u)
    return np.abs((WT * f[:, None]).sum(0) - erf_as((T - kappa) * u)).max()

W0 = 0.0212
def w_own(th): return W0 * np.minimum(1.0, 0.28 + 2.58 * np.abs(th - 0.5))
def w_left(th): return W0 * th ** 2
def w_right(th): return W0 * (1 - th) ** 2

print("# Table kernel: interpolation error of a unit A&S erf (`tools/table_error_study.py`)\n")
print("Largest error over a node interval, all phases of the kink; E_in: the 4-point stencil contains the kink, E_out: it does not.\n")
print("| u | E_in | E_in / u^2 | E_out | E_out / u^4 | worst excess over the kernel's per-kink bound W(kappa) u^2 + 0.36 u^4 |")
print("|---|---|---|---|---|---|")
ok = True
thetas = np.linspace(0, 1, 81)[:-1]
for u in [0.005, 0.01, 0.02, 0.03, 0.04, 0.05, 0.06, 0.08, 0.1, 0.125, 0.15, 0.2, 0.25, 0.3]:
    e_in = e_out = 0.0
    worst = -1e9
    for th in thetas:
        for k, wf in ((0, w_own), (-1, w_left), (1, w_right)):
            e = err(u, k + th)
            e_in = max(e_in, e)
            worst = max(worst, e - (wf(th) * u * u + 0.36 * u ** 4))
        for k in list(range(-int(8 / u) - 2, -1)) + list(range(2, int(8 / u) + 3)):
            if abs(k) > 40 and k % 7: continue   # far kinks: every 7th
            e_out = max(e_out, err(u, k + th))
    worst = max(worst, e_out - 0.36 * u ** 4)
    ok = ok and worst <= 0 and e_in <= 0.0212 * u * u
    print(f"| {u} | {e_in:.3e} | {e_in/u**2:.4f} | {e_out:.3e} | {e_out/u**4:.3f} | {worst:+.2e} |")
print()
print("All excesses negative and E_in <= 0.0212 u^2: the kernel's constants hold on this grid." if ok else "A CONSTANT OF THE KERNEL IS VIOLATED.")
