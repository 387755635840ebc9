// approx_cycles.hip -- GPU analogue of the reference's tests/approx_cycles.cpp:57-115 (cycles per value of every erf / exp
// approximation; its figures for AVX-512 are in thesis/main.tex:1810-1830).  Every variant of csrc/vrt_device_math.h is
// evaluated in a register loop (8 independent values per lane and iteration, inputs like the reference's: erf on [-6, 6],
// exp on [-10, 0]), 4 waves per SIMD, one workgroup per CU; cycles from s_memtime, median over the CUs.  A "value" is one
// lane's evaluation: a wave64 instruction serves 64 of them, so cycles per value = cycles per wave-evaluation / 64.
//   hipcc --offload-arch=gfx950 -O3 -I simd-gaussian-ray-tracing_amd/csrc tools/ubench/approx_cycles.hip -o /tmp/approx_cycles && /tmp/approx_cycles
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#include "vrt_device_math.h"
using namespace vrtk;

template <int KIND, bool IS_ERF>
__global__ void k(float *out, unsigned long long *stamps, float lo, float hi, int iters)
{
    float x[8], acc[8];
    const float span = hi - lo;
    for (int i = 0; i < 8; ++i) { x[i] = lo + span * (float)((threadIdx.x * 8 + i) % 509) / 509.f; acc[i] = 0.f; }
    const float step = span * 0.0137f;
    const unsigned long long r0 = wall_clock64(); // 100 MHz: calibrates the s_memtime tick
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            acc[i] += IS_ERF ? verf<KIND>(x[i]) : vexp<KIND>(x[i]);
            x[i] += step;
            x[i] = x[i] > hi ? x[i] - span : x[i];
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) { stamps[2 * gridDim.x] = t1 - t0; stamps[2 * gridDim.x + 1] = r1 - r0; }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { atomicMin(&stamps[2 * blockIdx.x], t0); atomicMax(&stamps[2 * blockIdx.x + 1], t1); }
}

template <int KIND, bool IS_ERF>
static void run(const char *name, int cus, float *out, unsigned long long *stamps)
{
    const int iters = 4096, wps = 4, threads = wps * 4 * 64;
    const float lo = IS_ERF ? -6.f : -10.f, hi = IS_ERF ? 6.f : 0.f;
    std::vector<unsigned long long> init(2 * cus);
    for (int i = 0; i < cus; ++i) { init[2 * i] = ~0ull; init[2 * i + 1] = 0; }
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL((k<KIND, IS_ERF>), dim3(cus), dim3(threads), 0, 0, out, stamps, lo, hi, iters);
    (void)hipMemcpy(stamps, init.data(), init.size() * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL((k<KIND, IS_ERF>), dim3(cus), dim3(threads), 0, 0, out, stamps, lo, hi, iters);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(2 * cus + 2);
    (void)hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> cyc(cus);
    for (int i = 0; i < cus; ++i) cyc[i] = (double)(h[2 * i + 1] - h[2 * i]) / ((double)iters * 8 * wps); // per wave-evaluation per SIMD
    std::nth_element(cyc.begin(), cyc.begin() + cus / 2, cyc.end());
    const double c = cyc[cus / 2] - 2 * 3.1; // the loop's own add + wrap (two full-rate instructions and a compare/select ~ 3 more) is left in: see note
    const double mhz = (double)h[2 * cus] / ((double)h[2 * cus + 1] * 0.01); // s_memtime ticks per microsecond
    printf("| %-28s | %6.1f | %6.3f | %.0f MHz |\n", name, cyc[cus / 2], cyc[cus / 2] / 64.0, mhz);
    (void)c;
}

int main()
{
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    float *out; unsigned long long *stamps;
    (void)hipMalloc(&out, sizeof(float) * 1024 * cus); (void)hipMalloc(&stamps, 16 * cus + 16);
    printf("| variant (csrc/vrt_device_math.h) | cycles per wave64 evaluation per SIMD (4 waves per SIMD; includes the loop's 4 bookkeeping instructions, ~13 cycles) | cycles per value | s_memtime tick rate (against the 100-MHz s_memrealtime) |\n|---|---|---|---|\n");
    run<VRT_ERF_LIBM, true>("erf: erff (libm)", cus, out, stamps);
    run<VRT_ERF_AS, true>("erf: Abramowitz-Stegun", cus, out, stamps);
    run<VRT_ERF_SPLINE, true>("erf: spline", cus, out, stamps);
    run<VRT_ERF_SPLINE_MIRROR, true>("erf: mirrored spline", cus, out, stamps);
    run<VRT_ERF_TAYLOR, true>("erf: Taylor-10", cus, out, stamps);
    run<VRT_EXP_LIBM, false>("exp: accurate (expf stand-in)", cus, out, stamps);
    run<VRT_EXP_VCL, false>("exp: vcl_exp stand-in", cus, out, stamps);
    run<VRT_EXP_FAST, false>("exp: fast_exp (Schraudolph)", cus, out, stamps);
    run<VRT_EXP_SPLINE, false>("exp: spline", cus, out, stamps);
    return 0;
}
