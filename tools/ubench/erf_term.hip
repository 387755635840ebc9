// erf_term.hip -- what does the table kernel's node loop sustain in isolation?  NT independent chains of the ten-instruction
// A&S term  acc[t] = fma(A, Em1 + 1 / p(|x0 + t hr|)^4, acc[t])  per "absorber", nothing else in the loop (no LDS, no branches).
// Cycles per VALU instruction per SIMD from s_memtime, for 1 / 2 / 4 waves per SIMD and NT = 8 / 12 / 24.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/erf_term.hip -o /tmp/erf_term && /tmp/erf_term
//   (add -mllvm -amdgpu-sched-strategy=max-ilp for the other scheduler)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

__device__ __forceinline__ float pin(float c) { asm("" : "+v"(c)); return c; }

template <int NT, bool GENERAL>
__global__ void k(float *out, unsigned long long *stamps, float a, float b, int n_abs)
{
    const float c3 = pin(0.078108f), c2 = pin(0.000972f), c1 = pin(0.230389f), c0 = pin(0.278393f);
    float acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = 0.f;
    float A = a + threadIdx.x * 1e-6f, x0 = b + threadIdx.x * 1e-3f, hr = 0.05f + a * 1e-3f, E = -1.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int j = 0; j < n_abs; ++j) {
        const float Em1 = E - 1.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float x = __builtin_fmaf((float)t, hr, x0);
            const float tt = __builtin_fabsf(x);
            float p = __builtin_fmaf(c3, tt, c2);
            p = __builtin_fmaf(p, tt, c1);
            p = __builtin_fmaf(p, tt, c0);
            p = __builtin_fmaf(p, tt, 1.0f);
            const float p2 = p * p;
            const float R = __builtin_amdgcn_rcpf(p2 * p2);
            if (GENERAL) acc[t] = __builtin_fmaf(A, E - __builtin_copysignf(1.0f - R, x), acc[t]);
            else acc[t] = __builtin_fmaf(A, Em1 + R, acc[t]);
        }
        A = A * 1.0001f; x0 += 0.01f; E = -E; // the next absorber (keeps the compiler from hoisting anything)
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) s += acc[t];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { atomicMin(&stamps[2 * blockIdx.x], t0); atomicMax(&stamps[2 * blockIdx.x + 1], t1); }
}

template <int NT, bool GENERAL>
static void run(int cus, float *out, unsigned long long *stamps, const char *name)
{
    const int n_abs = 4096;
    const int per_term = GENERAL ? 12 : 10;
    for (int wps : {1, 2, 4}) {
        const int threads = wps * 4 * 64;
        std::vector<unsigned long long> init(2 * cus);
        for (int i = 0; i < cus; ++i) { init[2 * i] = ~0ull; init[2 * i + 1] = 0; }
        for (int r = 0; r < 20; ++r) hipLaunchKernelGGL((k<NT, GENERAL>), dim3(cus), dim3(threads), 0, 0, out, stamps, 0.01f, -0.3f, n_abs);
        hipMemcpy(stamps, init.data(), init.size() * 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL((k<NT, GENERAL>), dim3(cus), dim3(threads), 0, 0, out, stamps, 0.01f, -0.3f, n_abs);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(2 * cus);
        hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> cyc(cus);
        for (int i = 0; i < cus; ++i) cyc[i] = (double)(h[2 * i + 1] - h[2 * i]) / ((double)n_abs * NT * per_term * wps);
        std::nth_element(cyc.begin(), cyc.begin() + cus / 2, cyc.end());
        printf("%-34s NT %2d  waves/SIMD %d : %.2f cycles per VALU instruction per SIMD (%d-instruction term: %.1f cycles)\n", name, NT, wps,
               cyc[cus / 2], per_term, cyc[cus / 2] * per_term);
    }
}

int main()
{
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    float *out; unsigned long long *stamps;
    hipMalloc(&out, sizeof(float) * 1024 * cus); hipMalloc(&stamps, 16 * cus);
    run<8, false>(cus, out, stamps, "sign-uniform term (E-1)+R");
    run<12, false>(cus, out, stamps, "sign-uniform term (E-1)+R");
    run<24, false>(cus, out, stamps, "sign-uniform term (E-1)+R");
    run<12, true>(cus, out, stamps, "general term E-copysign(1-R,x)");
    return 0;
}
