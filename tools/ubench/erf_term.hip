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

// The same term two nodes at a time on the packed fp32 pipe (v_pk_fma_f32 / v_pk_mul_f32: 4.96 / 5.04 cycles per PAIR against 2 x 3.25,
// profiles/r02_valu_ops.txt).  Only where the sign of x is the same for the whole wave: |x| is then x or -x, formed by negating x0 and hr
// once per absorber (VOP3P has no |abs| modifier).  SPLIT: acc[t] += A R only; A (E -+ 1) goes to one sum per absorber (8 packed ops +
// 2 v_rcp_f32 per pair of terms instead of 9 + 2).
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
template <int NT, bool SPLIT>
__global__ void kp(float *out, unsigned long long *stamps, float a, float b, int n_abs)
{
    const float c3 = pin(0.078108f), c2 = pin(0.000972f), c1 = pin(0.230389f), c0 = pin(0.278393f);
    const v2f C3 = { c3, c3 }, C2 = { c2, c2 }, C1 = { c1, c1 }, C0 = { c0, c0 }, ONE = { 1.f, 1.f };
    v2f acc[NT / 2];
#pragma unroll
    for (int t = 0; t < NT / 2; ++t) acc[t] = (v2f){ 0.f, 0.f };
    float A = a + threadIdx.x * 1e-6f, x0 = b + threadIdx.x * 1e-3f, hr = 0.05f + a * 1e-3f, E = -1.f, common = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int j = 0; j < n_abs; ++j) {
        const float Em1 = E - 1.f;
        const v2f A2 = { A, A }, X0 = { x0, x0 + hr }, HR2 = { 2.f * hr, 2.f * hr }, EM = { Em1, Em1 };
#pragma unroll
        for (int t = 0; t < NT / 2; ++t) {
            const v2f T = { (float)t, (float)t };
            const v2f tt = fma2(T, HR2, X0);
            v2f p = fma2(C3, tt, C2);
            p = fma2(p, tt, C1);
            p = fma2(p, tt, C0);
            p = fma2(p, tt, ONE);
            const v2f p2 = p * p, p4 = p2 * p2;
            const v2f R = { __builtin_amdgcn_rcpf(p4.x), __builtin_amdgcn_rcpf(p4.y) };
            if (SPLIT) acc[t] = fma2(A2, R, acc[t]);
            else acc[t] = fma2(A2, EM + R, acc[t]);
        }
        if (SPLIT) common = __builtin_fmaf(A, Em1, common);
        A = A * 1.0001f; x0 += 0.01f; E = -E;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = common;
#pragma unroll
    for (int t = 0; t < NT / 2; ++t) s += acc[t].x + acc[t].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { atomicMin(&stamps[2 * blockIdx.x], t0); atomicMax(&stamps[2 * blockIdx.x + 1], t1); }
}
// scalar, split sums: nine instructions per term
template <int NT>
__global__ void ks(float *out, unsigned long long *stamps, float a, float b, int n_abs)
{
    const float c3 = pin(0.078108f), c2 = pin(0.000972f), c1 = pin(0.230389f), c0 = pin(0.278393f);
    float acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = 0.f;
    float A = a + threadIdx.x * 1e-6f, x0 = b + threadIdx.x * 1e-3f, hr = 0.05f + a * 1e-3f, E = -1.f, common = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int j = 0; j < n_abs; ++j) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float tt = __builtin_fabsf(__builtin_fmaf((float)t, hr, x0));
            float p = __builtin_fmaf(c3, tt, c2);
            p = __builtin_fmaf(p, tt, c1);
            p = __builtin_fmaf(p, tt, c0);
            p = __builtin_fmaf(p, tt, 1.0f);
            const float p2 = p * p;
            acc[t] = __builtin_fmaf(A, __builtin_amdgcn_rcpf(p2 * p2), acc[t]);
        }
        common = __builtin_fmaf(A, E - 1.f, common);
        A = A * 1.0001f; x0 += 0.01f; E = -E;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = common;
#pragma unroll
    for (int t = 0; t < NT; ++t) s += acc[t];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { atomicMin(&stamps[2 * blockIdx.x], t0); atomicMax(&stamps[2 * blockIdx.x + 1], t1); }
}

template <typename K>
static void run_terms(K kernel, int NT, int cus, float *out, unsigned long long *stamps, const char *name)
{
    const int n_abs = 16384;
    for (int wps : {1, 2, 4}) {
        const int threads = wps * 4 * 64;
        std::vector<unsigned long long> init(2 * cus);
        for (int i = 0; i < cus; ++i) { init[2 * i] = ~0ull; init[2 * i + 1] = 0; }
        for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(kernel, dim3(cus), dim3(threads), 0, 0, out, stamps, 0.01f, 0.3f, n_abs);
        hipMemcpy(stamps, init.data(), init.size() * 8, hipMemcpyHostToDevice);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kernel, dim3(cus), dim3(threads), 0, 0, out, stamps, 0.01f, 0.3f, n_abs);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(2 * cus);
        hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> cyc(cus);
        for (int i = 0; i < cus; ++i) cyc[i] = (double)(h[2 * i + 1] - h[2 * i]) / ((double)n_abs * NT * wps);
        std::nth_element(cyc.begin(), cyc.begin() + cus / 2, cyc.end());
        // wall time of the launch (events; ~10 us of launch overhead in 1-4 ms) against the cycles: the clock the chip sustained
        const double ns_term = (double)ms * 1e6 / ((double)n_abs * NT * wps);
        printf("%-36s NT %2d  waves/SIMD %d : %.1f cycles per TERM per SIMD, %.2f ns per term per SIMD (launch %.3f ms) -> %.2f GHz\n", name, NT, wps,
               cyc[cus / 2], ns_term, ms, cyc[cus / 2] / ns_term);
    }
}

template <int NT, bool GENERAL>
static void run(int cus, float *out, unsigned long long *stamps, const char *name)
{
    const int n_abs = 16384;
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
    run_terms(k<12, false>, 12, cus, out, stamps, "scalar (E-1)+R (10 instr)");
    run_terms(k<24, false>, 24, cus, out, stamps, "scalar (E-1)+R (10 instr)");
    run_terms(ks<8>, 8, cus, out, stamps, "scalar, split sums (9 instr)");
    run_terms(ks<12>, 12, cus, out, stamps, "scalar, split sums (9 instr)");
    run_terms(kp<8, false>, 8, cus, out, stamps, "packed pairs (E-1)+R (9 pk + 2 rcp)");
    run_terms(kp<12, false>, 12, cus, out, stamps, "packed pairs (E-1)+R (9 pk + 2 rcp)");
    run_terms(kp<24, false>, 24, cus, out, stamps, "packed pairs (E-1)+R (9 pk + 2 rcp)");
    run_terms(kp<8, true>, 8, cus, out, stamps, "packed pairs, split (8 pk + 2 rcp)");
    run_terms(kp<12, true>, 12, cus, out, stamps, "packed pairs, split (8 pk + 2 rcp)");
    run_terms(kp<24, true>, 24, cus, out, stamps, "packed pairs, split (8 pk + 2 rcp)");
    return 0;
}
