// valu_clock.hip -- what does a pure fp32 VALU loop sustain on this part, and at which clock?
//
// bench.py prices the render kernel's VALU work against the spec issue peak (a wave64 VALU instruction issues over 2
// cycles per SIMD, 2.4 GHz: MI355X_MICROARCH.md).  This micro-benchmark measures the same quantity with in-kernel
// stamps (guide, "DVFS give-back" item 6): cycles per instruction = d(s_memtime) / instructions, clock =
// d(s_memtime) / d(s_memrealtime) x 100 MHz, median over workgroups, after the chip has been kept busy for ~1.5 s.
// Long loops (ms per launch), so launch gaps do not enter; events give the wall-clock rate next to it.
//
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_clock.hip -o /tmp/valu_clock && /tmp/valu_clock > profiles/rNN_valu_ubench.json
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define N_IT (1 << 16)

__global__ void fma_loop(float *out, unsigned long long *stamps, float a, float b)
{
    float x0 = threadIdx.x * 1e-3f + a, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < N_IT; ++i)
        asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                     "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    // the SIMD arbiter favours its oldest wave: wave 0 alone would read its own, shortest, run time -- take the
    // workgroup's span (earliest start to latest end over its waves)
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&stamps[4 * blockIdx.x], c0); atomicMax(&stamps[4 * blockIdx.x + 1], c1);
        atomicMin(&stamps[4 * blockIdx.x + 2], r0); atomicMax(&stamps[4 * blockIdx.x + 3], r1);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

int main()
{
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    float *out;
    unsigned long long *stamps;
    hipMalloc(&out, sizeof(float) * 1024 * cus);
    hipMalloc(&stamps, sizeof(unsigned long long) * 4 * cus);
    std::vector<unsigned long long> init(4 * cus);
    for (int i = 0; i < cus; ++i) { init[4 * i] = init[4 * i + 2] = ~0ull; init[4 * i + 1] = init[4 * i + 3] = 0; }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    printf("{\"what\": \"v_fma_f32 x8 independent, %d iterations, one workgroup per CU (tools/ubench/valu_clock.hip)\", \"cus\": %d, \"runs\": [", N_IT, cus);
    bool first = true;
    for (int wps : {1, 2, 3, 4}) {
        const int threads = wps * 4 * 64;
        // keep the chip busy first: the clock under load is what a frame loop sees
        for (int r = 0; r < (wps == 1 ? 600 : 300) / wps; ++r) hipLaunchKernelGGL(fma_loop, dim3(cus), dim3(threads), 0, 0, out, stamps, 0.999f, 0.5f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        const int reps = 20;
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(fma_loop, dim3(cus), dim3(threads), 0, 0, out, stamps, 0.999f, 0.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        // one more launch, stamped from clean min/max slots
        hipMemcpy(stamps, init.data(), init.size() * sizeof(unsigned long long), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(fma_loop, dim3(cus), dim3(threads), 0, 0, out, stamps, 0.999f, 0.5f);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(4 * cus);
        hipMemcpy(h.data(), stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        std::vector<double> cyc(cus), clk(cus);
        for (int i = 0; i < cus; ++i) {
            const double dc = (double)(h[4 * i + 1] - h[4 * i]), dr = (double)(h[4 * i + 3] - h[4 * i + 2]);
            cyc[i] = dc / ((double)N_IT * 8 * wps); // shader cycles per wave-instruction per SIMD
            clk[i] = dc / dr * 0.1;                  // GHz (s_memrealtime ticks at 100 MHz)
        }
        std::nth_element(cyc.begin(), cyc.begin() + cus / 2, cyc.end());
        std::nth_element(clk.begin(), clk.begin() + cus / 2, clk.end());
        const double instr_per_simd = (double)N_IT * 8 * wps * reps;
        const double ns = ms * 1e6 / instr_per_simd;
        printf("%s{\"waves_per_simd\": %d, \"cycles_per_wave_instr_per_simd\": %.3f, \"in_kernel_clock_ghz\": %.3f, "
               "\"ns_per_wave_instr_per_simd_by_events\": %.4f, \"wave_instr_per_s_chip\": %.4e}",
               first ? "" : ", ", wps, cyc[cus / 2], clk[cus / 2], ns, cus * 4 / (ns * 1e-9));
        first = false;
    }
    printf("]}\n");
    return 0;
}
