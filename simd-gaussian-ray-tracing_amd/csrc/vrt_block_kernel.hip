// vrt_block_kernel.hip -- the "block kernel": persistent one-wave workgroups, one 8x8 pixel block per work item, per-ray
// candidate lists (sparse scenes: sigma of a pixel or two).  A translation unit of its own, compiled with
// -mllvm -amdgpu-sched-strategy=max-ilp (csrc/Makefile): the default scheduler chains the 20 independent erf terms of an absorber
// one after the other through two registers to save VGPRs, so a wave that is alone on its SIMD (the tail of the kernel) crawls;
// max-ilp interleaves them (about 150 VGPRs, three waves per SIMD, which is what the persistent grid uses anyway).
#include "vrt_kernels_common.hpp"

namespace vrtk {

// ---------------------------------------------------------------------------------------------
// Fast path of the image kernel: the block's candidates sit in LDS (index + the two parameter rows the
// inner loop needs) and every lane walks ITS OWN list of them -- the candidates whose sigma*mag*exp(-x)
// reaches cull_eps on that lane's ray.  In sparse scenes (sigma of a pixel or two) a ray meets a third of
// its block's candidates, and the pair loop is quadratic in the list length.  All lanes run the loops to
// the longest lane list; a lane past the end of its list adds exact zeros (A = 0).
// ---------------------------------------------------------------------------------------------
// One chunk of EC emitters (list positions i0 .. i0+EC-1 of every lane) against the lane's whole list.
template <int EXP, int ERF, int EC>
__device__ __forceinline__ void shade_chunk(const float4 *s_A, const float4 *s_B, const float4 *s_M, const float4 *s_C,
                                            const float *s_q, const uint8_t *s_lane /*[k*64 + lane]*/, uint32_t nl,
                                            uint32_t nmax, uint32_t lane, const LaneRay &ray, uint32_t i0, float &Lr,
                                            float &Lg, float &Lb, float &La)
{
    const ErfEval<ERF> erf;
    float e_mubar[EC], e_sigma[EC];
    uint32_t e_li[EC];
#pragma unroll
    for (int e = 0; e < EC; ++e) {
        const bool ve = i0 + e < nl;
        e_li[e] = ve ? s_lane[(i0 + e) * 64 + lane] : 0u;
        const float4 a = s_A[e_li[e]];
        e_mubar[e] = dot3_ref(a.x, a.y, a.z, ray.nx, ray.ny, ray.nz);
        e_sigma[e] = s_M[e_li[e]].w;
    }
    float acc[EC][5];
#pragma unroll
    for (int e = 0; e < EC; ++e)
#pragma unroll
        for (int k = 0; k < 5; ++k) acc[e][k] = 0.f;

    // absorber stream over this lane's list; next entry's LDS rows are fetched one iteration ahead
    uint32_t lj = nl ? s_lane[lane] : 0u;
    float4 a = s_A[lj], b = s_B[lj];
    for (uint32_t j = 0; j < nmax; ++j) {
        const float4 ca = a, cb = b;
        const bool vj = j < nl;
        if (j + 1 < nmax) {
            lj = (j + 1 < nl) ? s_lane[(j + 1) * 64 + lane] : 0u;
            a = s_A[lj]; b = s_B[lj];
        }
        const float mubar = dot3_ref(ca.x, ca.y, ca.z, ray.nx, ray.ny, ray.nz);
        const float d2 = sub_ref(ca.w, mul_ref(mubar, mubar));
        const float A = vj ? cb.z * vexp<EXP>(-(d2 * cb.y)) : 0.f;
        const float m = mubar * cb.x;
        const float E = erf(-m);
#pragma unroll
        for (int e = 0; e < EC; ++e) {
            const float base = __builtin_fmaf(e_mubar[e], cb.x, -m);
            const float step = e_sigma[e] * cb.x;
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const float x = __builtin_fmaf((float)(k - 4), step, base);
                acc[e][k] = __builtin_fmaf(A, E - erf(x), acc[e][k]);
            }
        }
    }

    // emission (see shade_list)
#pragma unroll
    for (int e = 0; e < EC; ++e) {
        if (i0 + e < nl) {
            const float4 ms = s_M[e_li[e]];
            const float inv2s2 = s_B[e_li[e]].y;
            const float q = s_q[e_li[e]];
            float inner = 0.f;
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const float sk = madd_ref((float)(k - 4), ms.w, e_mubar[e]);
                const float px = sub_ref(madd_ref(ray.nx, sk, ray.ox), ms.x);
                const float py = sub_ref(madd_ref(ray.ny, sk, ray.oy), ms.y);
                const float pz = sub_ref(madd_ref(ray.nz, sk, ray.oz), ms.z);
                const float dd = dot3_ref(px, py, pz, px, py, pz);
                inner += emission_term<EXP>(q, dd * inv2s2, acc[e][k]);
            }
            const float4 alb = s_C[e_li[e]];
            Lr = __builtin_fmaf(alb.x, inner, Lr);
            Lg = __builtin_fmaf(alb.y, inner, Lg);
            Lb = __builtin_fmaf(alb.z, inner, Lb);
            La = __builtin_fmaf(alb.w, inner, La);
        }
    }
}

// Chunks of EC emitters, then one chunk of exactly the remaining 1..EC-1: the pair loop costs nmax^2, not
// nmax * (nmax rounded up to a multiple of EC).
template <int EXP, int ERF, int EC>
__device__ __forceinline__ void shade_lanes(const float4 *s_A, const float4 *s_B, const float4 *s_M, const float4 *s_C,
                                            const float *s_q, const uint8_t *s_lane /*[k*64 + lane]*/, uint32_t nl,
                                            uint32_t nmax, uint32_t lane, const LaneRay &ray, float &Lr, float &Lg,
                                            float &Lb, float &La, uint32_t i_start = 0, uint32_t i_step = EC)
{
    Lr = Lg = Lb = La = 0.f;
    for (uint32_t i0 = i_start; i0 < nmax; i0 += i_step) {
        const uint32_t rem = nmax - i0;
        if (rem >= (uint32_t)EC) shade_chunk<EXP, ERF, EC>(s_A, s_B, s_M, s_C, s_q, s_lane, nl, nmax, lane, ray, i0, Lr, Lg, Lb, La);
        else if (EC > 3 && rem == 3) shade_chunk<EXP, ERF, 3>(s_A, s_B, s_M, s_C, s_q, s_lane, nl, nmax, lane, ray, i0, Lr, Lg, Lb, La);
        else if (EC > 2 && rem == 2) shade_chunk<EXP, ERF, 2>(s_A, s_B, s_M, s_C, s_q, s_lane, nl, nmax, lane, ray, i0, Lr, Lg, Lb, La);
        else shade_chunk<EXP, ERF, 1>(s_A, s_B, s_M, s_C, s_q, s_lane, nl, nmax, lane, ray, i0, Lr, Lg, Lb, La);
    }
}

// The same with balanced chunks of at most ECMAX emitters: ceil(nmax / ECMAX) chunks of nearly equal size, so that a list
// of 5 is ONE pass over the absorbers (not 4 + 1) and a list of 9 is 5 + 4 (not 4 + 4 + 1).  Emitters and absorbers are
// visited in the same order as before: the sums are bit-identical.
#ifndef VRT_RENDER_ECMAX
#define VRT_RENDER_ECMAX 4
#endif
template <int EXP, int ERF, int ECMAX>
__device__ __forceinline__ void shade_lanes_balanced(const float4 *s_A, const float4 *s_B, const float4 *s_M, const float4 *s_C,
                                                     const float *s_q, const uint8_t *s_lane, uint32_t nl, uint32_t nmax, uint32_t lane,
                                                     const LaneRay &ray, float &Lr, float &Lg, float &Lb, float &La)
{
    Lr = Lg = Lb = La = 0.f;
    uint32_t chunks = (nmax + ECMAX - 1) / ECMAX;
    for (uint32_t i0 = 0; i0 < nmax; --chunks) {
        const uint32_t size = (nmax - i0 + chunks - 1) / chunks;
        if (ECMAX >= 6 && size == 6) shade_chunk<EXP, ERF, (ECMAX >= 6 ? 6 : 1)>(s_A, s_B, s_M, s_C, s_q, s_lane, nl, nmax, lane, ray, i0, Lr, Lg, Lb, La);
        else if (ECMAX >= 5 && size == 5) shade_chunk<EXP, ERF, (ECMAX >= 5 ? 5 : 1)>(s_A, s_B, s_M, s_C, s_q, s_lane, nl, nmax, lane, ray, i0, Lr, Lg, Lb, La);
        else if (size == 4) shade_chunk<EXP, ERF, 4>(s_A, s_B, s_M, s_C, s_q, s_lane, nl, nmax, lane, ray, i0, Lr, Lg, Lb, La);
        else if (size == 3) shade_chunk<EXP, ERF, 3>(s_A, s_B, s_M, s_C, s_q, s_lane, nl, nmax, lane, ray, i0, Lr, Lg, Lb, La);
        else if (size == 2) shade_chunk<EXP, ERF, 2>(s_A, s_B, s_M, s_C, s_q, s_lane, nl, nmax, lane, ray, i0, Lr, Lg, Lb, La);
        else shade_chunk<EXP, ERF, 1>(s_A, s_B, s_M, s_C, s_q, s_lane, nl, nmax, lane, ray, i0, Lr, Lg, Lb, La);
        i0 += size;
    }
}

// ---------------------------------------------------------------------------------------------
// Image kernel: persistent one-wave workgroups; a work item is one 8x8 pixel block (64 rays, lane = ray) of a
// 32x32 pixel cell with a non-empty candidate list.  A wave's first block is static (item = wave); frames with
// more blocks than waves hand the rest out through eight work counters (CellGrid::rq).  Empty cells are cleared
// by the fused list kernel; when that did not run for this target (unfused lists, re-render) they are cleared here.
// ---------------------------------------------------------------------------------------------
// occupancy experiment knob (csrc/Makefile EXTRA=-DVRT_RENDER_WPE=4): cap the registers for N waves per SIMD
#ifdef VRT_RENDER_WPE
#define VRT_RENDER_ATTR __attribute__((amdgpu_waves_per_eu(VRT_RENDER_WPE, VRT_RENDER_WPE)))
#else
#define VRT_RENDER_ATTR
#endif

// Budgeted ray-level cull (round 3).  The level-wise thresholds above are worst-case counting: a level that n candidates enter drops
// below eps * 1365 / n, as if all n sat just under it.  A ray's own list knows better: what it lost is the SUM of sigma*mag*exp(-x) over
// what it dropped, and on a grid scene one or two of a ray's five entries carry 1e-7 .. 1e-6 while the worst case reserves room for
// dozens.  So after the unconditional pass the lane looks at the entries it kept (e_k = sigma*mag*exp(-x_k) in units of the tile
// level's eps; the pass leaves ln e_k = cull_x - x_k in LDS, as fp16 rounded up) and drops the smallest ones as long as their sum stays inside `budget` (CellGrid::prune_budget = kappa * 1365 eps:
// 3 * that is what the ray's radiance can change by): smallest first, exactly -- entry k goes iff the sum of all
// entries not larger than it fits.  Entries whose threshold sits at the Exp floor (cull_eps = 0, huge magnitudes) are never dropped.
// A function of the block's survivors alone, so every path that must give identical bits still does.  `-g 64 -w 2048`: per-ray lists
// 3.9 -> 2.5, the block's longest 5.4 -> 3.6; the pair loops are quadratic in that.
constexpr uint32_t PRUNE_PL = 16; // lists up to this long are pruned; s_t holds ln(e_k) of their entries (fp16, rounded up)
template <int N>
__device__ __forceinline__ uint32_t prune_list(const __half *s_t, uint8_t *s_lane, uint32_t nl, uint32_t lane, float budget)
{
    float e[N];
#pragma unroll
    for (int k = 0; k < N; ++k) e[k] = (uint32_t)k < nl ? __expf(__half2float(s_t[k * 64 + lane])) : INFINITY;
    float least = e[0];
#pragma unroll
    for (int k = 1; k < N; ++k) least = fminf(least, e[k]);
    if (__ballot(least <= budget) == 0ull) return nl; // nothing in this block is small enough
    uint32_t drop = 0;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        float below = 0.f; // the sum of everything not larger than entry k, itself included
#pragma unroll
        for (int l = 0; l < N; ++l) below += e[l] <= e[k] ? e[l] : 0.f;
        if (below <= budget) drop |= 1u << k;
    }
    uint32_t w = 0;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        if ((uint32_t)k < nl) {
            const uint8_t v = s_lane[k * 64 + lane];
            if (!((drop >> k) & 1u)) { s_lane[w * 64 + lane] = v; ++w; }
        }
    }
    return w;
}

// One wavefront shades a block on its own.  (Rounds 1-3 carried a two-waves-per-block variant -- shared cull and lists, half of the
// emitters each -- measured slower every time, profiles/r02_experiments.md: the cull and list phases are latency-bound and do not
// shrink when a block gets two waves.  Retired in round 4.)
template <int EXP, int ERF, int EC, bool CLAIM = false>
__device__ __forceinline__ void render_body(const SceneTables &S, const TileLists &T, const CellGrid &C, const RayGen &R, const RenderTarget &O)
{
    // every row a kept candidate needs later (absorber: A, B; emitter: mu/sigma, albedo, sigma*mag) is fetched in the one
    // round trip of the block cull: the shading loops then run out of LDS only
    __shared__ float4 s_A[PCAP], s_B[PCAP], s_M[PCAP], s_C[PCAP];
    __shared__ float s_q[PCAP];
    __shared__ uint8_t s_lane[PL * 64];
    __shared__ __half s_t[PRUNE_PL * 64]; // ln(sigma*mag*exp(-x) / eps) of the first PRUNE_PL entries of every lane's list (prune_list)
    const uint32_t lane = threadIdx.x & 63u, wave = blockIdx.x, G = gridDim.x;
    const bool first = threadIdx.x == 0;
    const uint64_t npix = (uint64_t)R.width * R.height;
    // A block's way to its candidates is a chain of dependent loads (queue entry -> list -> parameter rows), and at the start of a launch
    // every wave walks it at the same time, with nothing to hide it behind.  Two links are taken out: an entry of the active queue carries
    // its cell's list length in its top byte (ACTIVE_COUNT_SHIFT), and the entry of a wave's static first block is fetched together with
    // the counters it is checked against (stale beyond n_active: used only below it).
    const uint32_t spec_entry = C.n_cells ? C.active[min(wave >> 4, C.n_cells - 1u)] : 0u;
    const uint32_t n_active = *C.n_active, n_dense_cells = *C.n_dense;
    if (C.feedback && wave == 0 && first) {
        __hip_atomic_store(&C.feedback[0], n_dense_cells, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const uint32_t n_light = C.light_threshold ? *C.n_light : 0u; // cells filed from the back of the queue: shaded last
    const uint32_t n_shade = (n_active + n_light) * 16u; // the dense cells belong to the 16-waves-per-block kernel behind this one
    if (C.feedback && wave == 0 && first) __hip_atomic_store(&C.feedback[1], n_shade, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); // for the host's choice of the CLAIM variant
    if (O.sparse_hdr && wave == 0 && first) { // sparse shard header; the counts are final: the list kernel is done
        O.sparse_hdr[0] = n_active + n_dense_cells; O.sparse_hdr[1] = O.sparse_cap;
        O.sparse_hdr[2] = C.cells_x * C.cells_y; O.sparse_hdr[3] = 0;
    }

    // ---- clear the cells nothing can reach (4 B per ray: the only HBM traffic of most of the frame) ----
    const uint32_t zero_px = (O.pack_flags & VRT_ALPHA_COMPUTED) ? 0u : 0xFF000000u;
    // (only when the list kernel of this frame did not do it: unfused lists, or a re-render from unchanged lists)
    for (uint32_t cell = wave; cell < C.n_cells && !O.cleared; cell += G) { // one whole cell per item: 16 x (2 rows of 32 px)
        if (C.count[cell] != 0u) continue;
        const uint32_t cpt = C.cells_x * C.cells_y;
        const uint32_t lt = cell / cpt, ci = cell % cpt;
        const uint32_t t = O.tile_map ? O.tile_map[lt] : lt;
        const uint32_t tx = t % T.tiles_w, ty = t / T.tiles_w;
        const uint32_t pxt = (ci % C.cells_x) * CELL + (lane & 31);
#pragma unroll 4
        for (uint32_t pass = 0; pass < CELL / 2; ++pass) {
            const uint32_t pyt = (ci / C.cells_x) * CELL + pass * 2 + (lane >> 5);
            const uint64_t pix = (uint64_t)(tx * T.tile_w + pxt) + (uint64_t)T.stride * (ty * T.tile_h + pyt);
            if (pxt < T.tile_w && pyt < T.tile_h && pix < npix) {
                const uint64_t out = O.compact ? ((uint64_t)lt * T.tile_h + pyt) * T.tile_w + pxt : pix;
                if (O.image) O.image[out] = zero_px;
                if (O.radiance) O.radiance[out] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }

    // ---- shade ----
    if (wave == 0 && threadIdx.x < RQ_N) C.rq_next[threadIdx.x * RQ_STRIDE] = 0;
    const uint32_t n_dyn = n_shade > G ? n_shade - G : 0u;
    static_assert(RQ_N <= 64, "one bit of rq_dead and one lane of the poll per queue");
    uint64_t rq_dead = 0; // queues this wave has seen run out (bit l = the l-th from its own; RQ_N = 64 of them: 64 bits)
    // next block: blocks cost between ~1 and ~30 units (the pair loops are quadratic in the per-ray list length), so
    // after its static first block a workgroup pulls more from the queues, its own first, until all are empty
    auto next_item = [&]() -> uint32_t {
        // all RQ_N counters are looked at in ONE round trip (lane l reads the l-th queue from the wave's own on): a wave that is done
        // leaves after one load instead of after RQ_N dependent ones -- at the end of a launch that was 5 us of every wave's exit
        while (true) {
            const uint32_t q = (wave + lane) % RQ_N;
            const uint32_t per = n_dyn > q ? (n_dyn - q + RQ_N - 1) / RQ_N : 0u;
            bool have = false;
            if (lane < RQ_N && per && !((rq_dead >> lane) & 1ull)) have = __hip_atomic_load(C.rq + q * RQ_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < per;
            const unsigned long long mask = __ballot(have);
            if (!mask) return 0xFFFFFFFFu;
            const uint32_t l = (uint32_t)__builtin_ctzll(mask);
            const uint32_t ql = (wave + l) % RQ_N, perl = (n_dyn - ql + RQ_N - 1) / RQ_N;
            uint32_t m = 0xFFFFFFFFu;
            if (first) m = atomicAdd(C.rq + ql * RQ_STRIDE, 1u);
            m = __builtin_amdgcn_readfirstlane(m);
            if (m < perl) return G + ql + RQ_N * m;
            rq_dead |= 1ull << l; // lost the race for its last entry: that queue is empty for good, so at most RQ_N rounds
        }
    };
    auto write_block = [&](float Lr, float Lg, float Lb, float La, bool valid, uint64_t out) {
        if (valid) {
            if (O.image) O.image[out] = pack_pixel(Lr, Lg, Lb, La, O.pack_flags);
            if (O.radiance) O.radiance[out] = make_float4(Lr, Lg, Lb, La);
        }
    };
    // The next entry of the wave's own queue is claimed BEFORE the block is shaded: the atomic's round trip (device scope: microseconds
    // when thousands of waves ask) passes behind the pair loops instead of in front of the next block's chain of loads.  A claimed entry
    // is always worked off by the wave that claimed it.  (Issued after the cull's loads have been consumed: returns are in order.)
    const uint32_t q_own = wave % RQ_N, per_own = n_dyn > q_own ? (n_dyn - q_own + RQ_N - 1) / RQ_N : 0u;
    // (only where most claims succeed: with few entries beyond the static ones thousands of failing claims would queue up on 8 counters)
    // (a kernel variant of its own, CLAIM: with the claim compiled in, the kernel that never claims -- the headline frame has no queue entry
    // at all -- ran 6 % longer: 10 more VGPRs, 20 more spilled SGPRs; the host picks the variant from what earlier frames reported)
    const bool claim_early = CLAIM && C.claim_early > 0 && per_own && (uint64_t)n_dyn * (uint32_t)C.claim_early >= G;
    uint32_t claim = 0xFFFFFFFFu;
    bool claimed = false;
    auto advance = [&]() -> uint32_t {
        if constexpr (CLAIM) {
            if (claimed) {
                claimed = false;
                const uint32_t m = __builtin_amdgcn_readfirstlane(claim);
                if (m < per_own) return G + q_own + RQ_N * m;
                rq_dead |= 1ull; // the own queue is empty for good
            }
        }
        return next_item();
    };
    for (uint32_t item = wave; item < n_shade; item = advance()) {
        const unsigned long long tl0 = O.timeline ? wall_clock64() : 0ull; // diagnostics (VRT_HIP_TIMELINE runs only)
        const uint32_t ci = item >> 4;
        const uint32_t entry = (item == wave && ci < n_active) ? spec_entry : C.active[ci < n_active ? ci : C.n_cells - 1u - (ci - n_active)];
        const uint32_t cell = entry & ACTIVE_CELL_MASK;
        const uint32_t bi = item & 15u;
        const BlockPos p = block_of(T, C, O, cell, bi, lane);
        if (!p.inside) continue;
        const uint32_t tx = p.t % T.tiles_w, ty = p.t / T.tiles_w;
        bool valid = p.pxt < T.tile_w && p.pyt < T.tile_h;
        const uint32_t pxc = min(p.pxt, T.tile_w - 1), pyc = min(p.pyt, T.tile_h - 1);
        uint64_t pix = (uint64_t)(tx * T.tile_w + pxc) + (uint64_t)T.stride * (ty * T.tile_h + pyc);
        if (pix >= npix) { valid = false; pix = npix - 1; }
        const uint64_t out = out_index(T, C, O, cell, bi, lane, p, pix, n_active);

        // the cell's candidate list (or, if it overflowed its slot, the tile's)
        uint32_t n_list = entry >> ACTIVE_COUNT_SHIFT;
        if (n_list == 255u) n_list = C.count[cell]; // a list length that does not fit the byte
        const uint32_t *list = C.indices + (size_t)cell * C.cstride;
        if (n_list == 0xFFFFFFFFu) { n_list = T.count[p.t]; list = T.indices + T.start[p.t]; }

        LaneRay ray = pixel_ray(R, pix);
        // the origin is wave-uniform (SGPRs): as a VGPR operand the 15 adds per emitter of the emission issue at full rate
        ray.ox = pin_vgpr(ray.ox); ray.oy = pin_vgpr(ray.oy); ray.oz = pin_vgpr(ray.oz);

        // ---- block cone: axis = mean of the four centre rays, angle = farthest lane ----
        float cx = lane_value(ray.nx, 27) + lane_value(ray.nx, 28) + lane_value(ray.nx, 35) + lane_value(ray.nx, 36);
        float cy = lane_value(ray.ny, 27) + lane_value(ray.ny, 28) + lane_value(ray.ny, 35) + lane_value(ray.ny, 36);
        float cz = lane_value(ray.nz, 27) + lane_value(ray.nz, 28) + lane_value(ray.nz, 35) + lane_value(ray.nz, 36);
        {
            const float inv = __builtin_amdgcn_rsqf(cx * cx + cy * cy + cz * cz);
            cx *= inv; cy *= inv; cz *= inv;
        }
        float co, si;
        cos_sin(ray.nx, ray.ny, ray.nz, cx, cy, cz, co, si);
        const Cone cone = make_cone(cx, cy, cz, wave_min(co), wave_max(si));

        // ---- block cull over the cell's list (ballot compaction, order preserving) ----
        __syncthreads(); // previous item's LDS reads are done
        const float slack = level_slack(T.cull_ref_n, n_list);
        uint32_t cnt = 0;
        for (uint32_t base = 0; base < n_list; base += 64) {
            const uint32_t k = base + lane;
            bool keep = false;
            float4 a, bq, ms, alb;
            float q;
            if (k < n_list) {
                const uint32_t idx = list[k];
                bq = S.gB[idx]; ms = S.mu_sig[idx]; alb = S.gC[idx]; q = S.gD[idx].y;
                // the per-origin row (centre - origin, |.|^2) from the centre itself, in prep_frame_kernel's arithmetic: one 16-B gather
                // of five less per candidate (round 4; the table S.gA stays for the dense-path kernels)
                {
                    const float ocx = ms.x - R.origin[0], ocy = ms.y - R.origin[1], ocz = ms.z - R.origin[2];
                    a = make_float4(ocx, ocy, ocz, dot3_ref(ocx, ocy, ocz, ocx, ocy, ocz));
                }
                keep = cone_keeps(cone, a, make_float4(bq.x, bq.y, bq.z, slack_cull_x(bq.w, slack, T.floor_x)));
            }
            const unsigned long long mask = __ballot(keep);
            const uint32_t pos = cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
            if (keep && pos < PCAP) { s_A[pos] = a; s_B[pos] = bq; s_M[pos] = ms; s_C[pos] = alb; s_q[pos] = q; }
            cnt += (uint32_t)__popcll(mask);
        }
        __syncthreads();

        const unsigned long long tl1 = O.timeline ? wall_clock64() : 0ull;
        // ---- lane cull: this ray's own candidates (exact per-ray criterion x > cull_x, no margin needed) ----
        uint32_t nl = 0;
        bool fast = cnt <= PCAP;
        if (fast) {
            const float slack_r = level_slack(T.cull_ref_n, cnt); // ray level: the block's survivors enter
            for (uint32_t j = 0; j < cnt; ++j) {
                const float4 a = s_A[j], bq = s_B[j];
                const float mubar = dot3_ref(a.x, a.y, a.z, ray.nx, ray.ny, ray.nz);
                const float x = sub_ref(a.w, mul_ref(mubar, mubar)) * bq.y;
                if (!(x > slack_cull_x(bq.w, slack_r, T.floor_x))) {
                    if (nl < PL) s_lane[nl * 64 + lane] = (uint8_t)j;
                    if (nl < PRUNE_PL) { // fp16 rounds to nearest within 2^-11: the bias keeps the stored value above the true one
                        const float t = bq.w - x;
                        s_t[nl * 64 + lane] = __float2half(bq.w < T.floor_x ? t + 0.001f * fabsf(t) + 1e-4f : INFINITY);
                    }
                    ++nl;
                }
            }
            fast = __ballot(nl > PL) == 0ull;
        }
        __syncthreads();
        if (!fast) {
            // Per-ray lists that outgrow LDS: the block goes to the 16-waves-per-block kernel, which runs after this one
            // (the host leaves that launch out only when a frame of exactly this state has reported that nothing is
            // handed over and no cell is dense).  Which kernel shades a block is a function of the block alone (the two
            // kernels sum in different orders), so the image does not depend on launch heuristics: the host only
            // chooses how LARGE the dense launch is (vrt_hip_api.cpp, render_common).
            if (first) C.overflow[atomicAdd(C.n_overflow, 1u)] = (cell << 4) | bi;
            continue;
        }
        if (O.stats && first) {
            atomicAdd(&O.stats[0], (unsigned long long)cnt);
            atomicAdd(&O.stats[1], (unsigned long long)n_list);
            atomicAdd(&O.stats[5], 1ull);
        }
        uint32_t nmax = wave_max_u32(nl);
        if (C.prune_budget > 0.f && nmax <= PRUNE_PL) {
            switch (nmax) { // exactly as many entries as the block's longest list has, while that is cheap
            case 0: break;
            case 1: nl = prune_list<1>(s_t, s_lane, nl, lane, C.prune_budget); break;
            case 2: nl = prune_list<2>(s_t, s_lane, nl, lane, C.prune_budget); break;
            case 3: nl = prune_list<3>(s_t, s_lane, nl, lane, C.prune_budget); break;
            case 4: nl = prune_list<4>(s_t, s_lane, nl, lane, C.prune_budget); break;
            case 5: nl = prune_list<5>(s_t, s_lane, nl, lane, C.prune_budget); break;
            case 6: nl = prune_list<6>(s_t, s_lane, nl, lane, C.prune_budget); break;
            case 7: nl = prune_list<7>(s_t, s_lane, nl, lane, C.prune_budget); break;
            case 8: nl = prune_list<8>(s_t, s_lane, nl, lane, C.prune_budget); break;
            case 9: case 10: case 11: case 12: nl = prune_list<12>(s_t, s_lane, nl, lane, C.prune_budget); break;
            default: nl = prune_list<16>(s_t, s_lane, nl, lane, C.prune_budget); break;
            }
            nmax = wave_max_u32(nl);
        }
        if (O.stats) {
            unsigned long long tot = nl;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor((int)tot, off, 64);
            unsigned long long sq = (unsigned long long)nl * nl; // this ray's (emitter, absorber) pairs: 5 erf terms each
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) sq += (unsigned long long)(uint32_t)__shfl_xor((int)sq, off, 64);
            if (lane == 0) { atomicAdd(&O.stats[3], tot); atomicAdd(&O.stats[4], (unsigned long long)nmax); atomicAdd(&O.stats[12], sq); }
        }
        const unsigned long long tl2 = O.timeline ? wall_clock64() : 0ull;
        if constexpr (CLAIM) {
            if (claim_early && !(rq_dead & 1ull)) {
                claimed = true;
                if (first) claim = atomicAdd(C.rq + q_own * RQ_STRIDE, 1u);
            }
        }
        float Lr, Lg, Lb, La;
        if constexpr (VRT_RENDER_ECMAX > 4) shade_lanes_balanced<EXP, ERF, VRT_RENDER_ECMAX>(s_A, s_B, s_M, s_C, s_q, s_lane, nl, nmax, lane, ray, Lr, Lg, Lb, La);
        else shade_lanes<EXP, ERF, EC>(s_A, s_B, s_M, s_C, s_q, s_lane, nl, nmax, lane, ray, Lr, Lg, Lb, La);
        write_block(Lr, Lg, Lb, La, valid, out);
        if (O.timeline && first) {
            unsigned long long *tl = O.timeline + 5 * (size_t)item;
            tl[0] = tl0; tl[1] = tl1; tl[2] = tl2; tl[3] = wall_clock64();
            const uint32_t hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_REG_HW_ID, 32 bits
            const uint32_t xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)); // HW_REG_XCC_ID
            tl[4] = ((unsigned long long)nmax << 48) | ((unsigned long long)(xcc & 0xFFFFu) << 32) | hw;
        }
    }
}

template <int EXP, int ERF, int EC, bool CLAIM = false>
__global__ __launch_bounds__(64) VRT_RENDER_ATTR void render_kernel(RenderArgs) // read through kernel_args<>: vrt_kernels_common.hpp
{
    const RenderArgs &a = kernel_args<RenderArgs>();
    render_body<EXP, ERF, EC, CLAIM>(a.S, a.T, a.C, a.R, a.O);
}
// several frames per launch: blockIdx.y is the frame (FrameArgs)
template <int EXP, int ERF, int EC, bool CLAIM = false>
__global__ __launch_bounds__(64) VRT_RENDER_ATTR void render_batch_kernel(const FrameArgs *__restrict__ frames)
{
    const FrameArgs &a = frames[blockIdx.y];
    render_body<EXP, ERF, EC, CLAIM>(a.S, a.T, a.C, a.R, a.O);
}

template <int EXP, int ERF>
static void launch_render_t(const SceneTables &s, const TileLists &t, const CellGrid &c, const RayGen &r,
                            const RenderTarget &o, uint32_t grid, hipStream_t st)
{
    if (grid == 0) return;
    if (c.claim_early > 0 && EXP == VRT_EXP_VCL && ERF == VRT_ERF_AS) // frames with many more blocks than waves (the default pair only: compile time)
        hipLaunchKernelGGL((render_kernel<VRT_EXP_VCL, VRT_ERF_AS, 4, true>), dim3(grid), dim3(64), 0, st, RenderArgs{ s, t, c, r, o });
    else hipLaunchKernelGGL((render_kernel<EXP, ERF, 4>), dim3(grid), dim3(64), 0, st, RenderArgs{ s, t, c, r, o });
}

template <int EXP, int ERF>
static void launch_render_batch_t(const FrameArgs *d_frames, uint32_t nframes, uint32_t grid, bool claim, hipStream_t st)
{
    if (grid == 0 || nframes == 0) return;
    if (claim && EXP == VRT_EXP_VCL && ERF == VRT_ERF_AS) hipLaunchKernelGGL((render_batch_kernel<VRT_EXP_VCL, VRT_ERF_AS, 4, true>), dim3(grid, nframes), dim3(64), 0, st, d_frames);
    else hipLaunchKernelGGL((render_batch_kernel<EXP, ERF, 4>), dim3(grid, nframes), dim3(64), 0, st, d_frames);
}


void launch_render(const SceneTables &s, const TileLists &t, const CellGrid &c, const RayGen &r, const RenderTarget &o,
                   uint32_t grid, int exp_kind, int erf_kind, hipStream_t st)
{
    VRT_DISPATCH_EXP_ERF(launch_render_t, s, t, c, r, o, grid, st);
}
void launch_render_batch(const FrameArgs *d_frames, uint32_t nframes, uint32_t grid, bool claim, int exp_kind, int erf_kind, hipStream_t st)
{
    VRT_DISPATCH_EXP_ERF(launch_render_batch_t, d_frames, nframes, grid, claim, st);
}

} // namespace vrtk
