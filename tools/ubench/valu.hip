// VALU issue-rate microbenchmark for gfx950: wave-instructions per cycle per SIMD for a few opcodes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N_IT 32768
typedef float v2f __attribute__((ext_vector_type(2)));

template <int OP>
__global__ void k(float *out, float a, float b)
{
    float x0 = threadIdx.x * 1e-3f + a, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    v2f p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, p4 = {x1, x2}, p5 = {x3, x4}, p6 = {x5, x6}, p7 = {x7, x0};
    v2f pa = {a, a}, pb = {b, b};
    for (int i = 0; i < N_IT; ++i) {
        if (OP == 0) { // v_fma_f32 x8 independent
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        } else if (OP == 1) { // v_pk_fma_f32 x8
            asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                         "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pa), "v"(pb));
        } else if (OP == 2) { // v_rcp_f32 x8
            asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
        } else if (OP == 3) { // 6 fma + 2 rcp mixed (the erf chain's ratio is ~10:1)
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_rcp_f32 %3, %3\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_rcp_f32 %7, %7\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        } else if (OP == 4) { // dependent fma chain x8 (one accumulator)
            asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         : "+v"(x0) : "v"(a), "v"(b));
        } else if (OP == 5) { // v_exp_f32 x8
            asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
        } else if (OP == 6) { // v_mul_f32 x8
            asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
        } else if (OP == 7) { // v_bfi_b32 x8
            asm volatile("v_bfi_b32 %0, %8, %0, %9\n v_bfi_b32 %1, %8, %1, %9\n v_bfi_b32 %2, %8, %2, %9\n v_bfi_b32 %3, %8, %3, %9\n v_bfi_b32 %4, %8, %4, %9\n v_bfi_b32 %5, %8, %5, %9\n v_bfi_b32 %6, %8, %6, %9\n v_bfi_b32 %7, %8, %7, %9\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        } else if (OP == 8) { // v_med3_f32 x8
            asm volatile("v_med3_f32 %0, %0, %8, %9\n v_med3_f32 %1, %1, %8, %9\n v_med3_f32 %2, %2, %8, %9\n v_med3_f32 %3, %3, %8, %9\n v_med3_f32 %4, %4, %8, %9\n v_med3_f32 %5, %5, %8, %9\n v_med3_f32 %6, %6, %8, %9\n v_med3_f32 %7, %7, %8, %9\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        } else if (OP == 9) { // v_and_or_b32 x8
            asm volatile("v_and_or_b32 %0, %8, %0, %9\n v_and_or_b32 %1, %8, %1, %9\n v_and_or_b32 %2, %8, %2, %9\n v_and_or_b32 %3, %8, %3, %9\n v_and_or_b32 %4, %8, %4, %9\n v_and_or_b32 %5, %8, %5, %9\n v_and_or_b32 %6, %8, %6, %9\n v_and_or_b32 %7, %8, %7, %9\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        } else if (OP == 10) { // v_sub_f32 (VOP2) x8
            asm volatile("v_sub_f32 %0, %8, %0\n v_sub_f32 %1, %8, %1\n v_sub_f32 %2, %8, %2\n v_sub_f32 %3, %8, %3\n v_sub_f32 %4, %8, %4\n v_sub_f32 %5, %8, %5\n v_sub_f32 %6, %8, %6\n v_sub_f32 %7, %8, %7\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
        } else if (OP == 11) { // v_xor_b32 (VOP2) x8
            asm volatile("v_xor_b32 %0, %8, %0\n v_xor_b32 %1, %8, %1\n v_xor_b32 %2, %8, %2\n v_xor_b32 %3, %8, %3\n v_xor_b32 %4, %8, %4\n v_xor_b32 %5, %8, %5\n v_xor_b32 %6, %8, %6\n v_xor_b32 %7, %8, %7\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
        } else if (OP == 13) { // v_fma_f32 with |abs| modifier and an SGPR operand (the erf polynomial's form) x8
            asm volatile("v_fma_f32 %0, |%0|, %8, %9\n v_fma_f32 %1, |%1|, %8, %9\n v_fma_f32 %2, |%2|, %8, %9\n v_fma_f32 %3, |%3|, %8, %9\n"
                         "v_fma_f32 %4, |%4|, %8, %9\n v_fma_f32 %5, |%5|, %8, %9\n v_fma_f32 %6, |%6|, %8, %9\n v_fma_f32 %7, |%7|, %8, %9\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(a), "v"(b));
        } else if (OP == 14) { // v_fmamk_f32 (32-bit literal) x8
            asm volatile("v_fmamk_f32 %0, %0, 0x3f7fbe77, %8\n v_fmamk_f32 %1, %1, 0x3f7fbe77, %8\n v_fmamk_f32 %2, %2, 0x3f7fbe77, %8\n v_fmamk_f32 %3, %3, 0x3f7fbe77, %8\n"
                         "v_fmamk_f32 %4, %4, 0x3f7fbe77, %8\n v_fmamk_f32 %5, %5, 0x3f7fbe77, %8\n v_fmamk_f32 %6, %6, 0x3f7fbe77, %8\n v_fmamk_f32 %7, %7, 0x3f7fbe77, %8\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b));
        } else if (OP == 15) { // v_mov_b32 x8
            asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
        } else if (OP == 16) { // v_cmp_lt_f32 + v_cndmask_b32 pairs x4
            asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %9, vcc\n v_cmp_lt_f32 vcc, %1, %8\n v_cndmask_b32 %1, %1, %9, vcc\n"
                         "v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %9, vcc\n v_cmp_lt_f32 vcc, %3, %8\n v_cndmask_b32 %3, %3, %9, vcc\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
        } else if (OP == 17) { // v_add_f32 (VOP2) x8
            asm volatile("v_add_f32 %0, %8, %0\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n v_add_f32 %4, %8, %4\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
        } else if (OP == 18) { // v_fma_f32, all three sources in ONE register bank (v0, v4, v8 style: operands 4 apart)
            asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                         "v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
        } else if (OP == 19) { // v_fma_f32 with |abs| modifier, all VGPR x8
            asm volatile("v_fma_f32 %0, |%0|, %8, %9\n v_fma_f32 %1, |%1|, %8, %9\n v_fma_f32 %2, |%2|, %8, %9\n v_fma_f32 %3, |%3|, %8, %9\n"
                         "v_fma_f32 %4, |%4|, %8, %9\n v_fma_f32 %5, |%5|, %8, %9\n v_fma_f32 %6, |%6|, %8, %9\n v_fma_f32 %7, |%7|, %8, %9\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        } else if (OP == 20) { // v_fma_f32 with an SGPR operand, no modifier x8
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(a), "v"(b));
        } else if (OP == 21) { // v_pk_fma_f32 with an SGPR-pair operand x8
            asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                         "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "s"(pa), "v"(pb));
        } else if (OP == 22) { // v_pk_add_f32 x8
            asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pa));
        } else if (OP == 23) { // v_pk_fma_f32, sources 1 and 2 broadcast from their low halves (op_sel_hi) x8
            asm volatile("v_pk_fma_f32 %0, %0, %8, %9 op_sel_hi:[1,0,0]\n v_pk_fma_f32 %1, %1, %8, %9 op_sel_hi:[1,0,0]\n v_pk_fma_f32 %2, %2, %8, %9 op_sel_hi:[1,0,0]\n v_pk_fma_f32 %3, %3, %8, %9 op_sel_hi:[1,0,0]\n"
                         "v_pk_fma_f32 %4, %4, %8, %9 op_sel_hi:[1,0,0]\n v_pk_fma_f32 %5, %5, %8, %9 op_sel_hi:[1,0,0]\n v_pk_fma_f32 %6, %6, %8, %9 op_sel_hi:[1,0,0]\n v_pk_fma_f32 %7, %7, %8, %9 op_sel_hi:[1,0,0]\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pa), "v"(pb));
        } else if (OP == 24) { // v_and_b32 (VOP2) x8
            asm volatile("v_and_b32 %0, %8, %0\n v_and_b32 %1, %8, %1\n v_and_b32 %2, %8, %2\n v_and_b32 %3, %8, %3\n v_and_b32 %4, %8, %4\n v_and_b32 %5, %8, %5\n v_and_b32 %6, %8, %6\n v_and_b32 %7, %8, %7\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
        } else if (OP == 25) { // v_fmaak_f32 (VOP2: a*b + literal) x8
            asm volatile("v_fmaak_f32 %0, %0, %8, 0x3e6bee59\n v_fmaak_f32 %1, %1, %8, 0x3e6bee59\n v_fmaak_f32 %2, %2, %8, 0x3e6bee59\n v_fmaak_f32 %3, %3, %8, 0x3e6bee59\n"
                         "v_fmaak_f32 %4, %4, %8, 0x3e6bee59\n v_fmaak_f32 %5, %5, %8, 0x3e6bee59\n v_fmaak_f32 %6, %6, %8, 0x3e6bee59\n v_fmaak_f32 %7, %7, %8, 0x3e6bee59\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
        } else if (OP == 26) { // v_fmac_f32 (VOP2: d += a*b) x8
            asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        } else if (OP == 12) { // v_pk_mul_f32 x8
            asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pa));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}

template <int OP>
void run(const char *name, int waves_per_simd)
{
    // one workgroup per CU holding waves_per_simd*4 waves
    const int threads = waves_per_simd * 4 * 64, blocks = 256;
    float *out;
    hipMalloc(&out, sizeof(float) * threads * blocks);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, 0.999f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, 0.999f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    const double instrs_per_simd = (double)N_IT * 8 * waves_per_simd; // wave-instructions each SIMD issued
    printf("%-22s waves/SIMD %d: %8.3f ms  -> %.2f ns per wave-instr per SIMD (= %.2f cycles @2.4GHz)\n", name, waves_per_simd, ms,
           ms * 1e6 / instrs_per_simd, ms * 1e6 / instrs_per_simd * 2.4);
    hipFree(out);
}

int main()
{
    for (int w : {3}) {
        run<0>("v_fma_f32 indep", w);
        run<4>("v_fma_f32 dependent", w);
        run<1>("v_pk_fma_f32", w);
        run<6>("v_mul_f32", w);
        run<7>("v_bfi_b32", w);
        run<2>("v_rcp_f32", w);
        run<5>("v_exp_f32", w);
        run<3>("6 fma + 2 rcp", w);
        run<8>("v_med3_f32", w);
        run<9>("v_and_or_b32", w);
        run<10>("v_sub_f32", w);
        run<11>("v_xor_b32", w);
        run<12>("v_pk_mul_f32", w);
        run<13>("v_fma |abs| + sgpr", w);
        run<14>("v_fmamk literal", w);
        run<15>("v_mov_b32", w);
        run<16>("v_cmp + v_cndmask (x4)", w);
        run<17>("v_add_f32", w);
        run<18>("v_fma same-register srcs", w);
        run<19>("v_fma |abs| all VGPR", w);
        run<20>("v_fma with SGPR src", w);
        run<21>("v_pk_fma SGPR-pair src", w);
        run<22>("v_pk_add_f32", w);
        run<23>("v_pk_fma op_sel bcast", w);
        run<24>("v_and_b32", w);
        run<25>("v_fmaak literal", w);
        run<26>("v_fmac_f32 (VOP2)", w);
    }
    return 0;
}
