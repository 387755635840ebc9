"""Where does a kernel's register-spill code sit?  Compiles csrc/vrt_kernels.hip to gfx950 assembly, finds the loops of one
kernel (backward branches) and lists, per loop that holds erf terms (v_rcp_f32) or scratch accesses, its size, its VALU
instruction count and its scratch loads / stores.  Round-1 verdict: "render_dense_kernel spills 9 VGPRs to scratch --
nobody checked whether the spill code sits in the absorber loop".

    python tools/spill_sites.py [mangled-name substring] [translation unit] > profiles/rNN_dense_spills.txt
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
want = sys.argv[1] if len(sys.argv) > 1 else "render_dense_kernelILi1ELi1ELi6ELi16ELb1"
# the kernel's translation unit: vrt_kernels.hip (exact dense kernel, list kernels), vrt_table_kernel.hip, vrt_block_kernel.hip
unit = sys.argv[2] if len(sys.argv) > 2 else "vrt_kernels.hip"
src = os.path.join(ROOT, "simd-gaussian-ray-tracing_amd", "csrc", unit)
extra = ["-DVRT_RENDER_ECMAX=6", "-DVRT_RENDER_WPE=3", "-mllvm", "-amdgpu-sched-strategy=max-ilp"] if "block" in unit else []
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "k.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-slp-vectorize", "-Wno-pass-failed",
                    "-S", "--cuda-device-only", "-o", out, src] + extra, check=True, stderr=subprocess.DEVNULL)
    s = open(out).read()
m = re.search(r"^(\S*%s\S*):\s*(;.*)?$" % re.escape(want), s, re.M)
name = m.group(1)
i = s.index(name + ":")
body = s[i:s.index(".Lfunc_end", i)].split("\n")
labels = {mm.group(1): n for n, l in enumerate(body) if (mm := re.match(r"^(\.LBB\d+_\d+):", l))}
loops = set()
for n, l in enumerate(body):
    mm = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
    if mm and mm.group(1) in labels and labels[mm.group(1)] < n:
        loops.add((labels[mm.group(1)], n))
meta = re.search(r"\.amdhsa_kernel %s(.*?)\.end_amdhsa_kernel" % re.escape(name), s, re.S).group(1)
print("kernel", name)
for key in ("next_free_vgpr", "private_segment_fixed_size"):
    mm = re.search(r"\.amdhsa_%s\s+(\d+)" % key, meta)
    print(f"  {key}: {mm.group(1) if mm else '?'}")
print(f"  {len(body)} lines of assembly, {sum('scratch_store' in l for l in body)} scratch stores, {sum('scratch_load' in l for l in body)} scratch loads")
print("loops that hold erf terms (v_rcp_f32) or scratch accesses, outermost first:")
for a, b in sorted(loops, key=lambda ab: (ab[0], -ab[1])):
    seg = body[a:b + 1]
    rcp = sum("v_rcp_f32" in l for l in seg)
    scr = sum("scratch_" in l for l in seg)
    if rcp or scr:
        print(f"  lines {a:5d}-{b:5d}: {sum(l.strip().startswith('v_') for l in seg):5d} VALU, {rcp:3d} v_rcp_f32, {scr:2d} scratch accesses")
