// Work-queue microbenchmark for gfx950: cost of returning atomicAdd pulls when G single-wave workgroups pull ITEMS work
// items from NQ counters (item i lives in queue i % NQ), each item followed by `work` dependent fmas per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(64) void pull(unsigned *ctr, unsigned nq, unsigned items_per_q, unsigned stride, int work, float *out)
{
    const unsigned q0 = blockIdx.x % nq;
    float x = threadIdx.x * 1e-3f;
    unsigned done = 0;
    for (unsigned dq = 0; dq < nq; ++dq) { // own queue first, then steal from the others
        unsigned *c = ctr + ((q0 + dq) % nq) * stride;
        for (;;) {
            unsigned it = 0;
            if (threadIdx.x == 0) it = atomicAdd(c, 1u);
            it = __builtin_amdgcn_readfirstlane(it);
            if (it >= items_per_q) break;
            for (int i = 0; i < work; ++i) x = __builtin_fmaf(x, 1.0001f, 0.5f);
            ++done;
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = x + done;
}

int main()
{
    unsigned *ctr; float *out;
    hipMalloc(&ctr, 1 << 20); hipMalloc(&out, 8192 * 64 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const unsigned items = 3136;
    for (int work : {0, 2000}) for (unsigned G : {1024u, 2048u, 4096u}) for (unsigned nq : {1u, 8u, 32u, 128u}) {
        const unsigned per = (items + nq - 1) / nq;
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            hipMemset(ctr, 0, 1 << 20);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(pull, dim3(G), dim3(64), 0, 0, ctr, nq, per, 64u /* 256 B apart */, work, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("work %5d  G %5u  queues %4u : %8.2f us\n", work, G, nq, best * 1e3f);
    }
    return 0;
}
