// vrt_dense_block.hpp -- one dense 8x8 block shaded exactly by a workgroup of DW waves: the body of the exact dense kernel
// (vrt_kernels.hip) and, since round 4, the fallback INSIDE the table kernel for the blocks it declines (vrt_table_kernel.hip) --
// a frame then needs one dense-path launch, not two.  Which arithmetic shades a block is a function of the block alone either way.
#pragma once
#include "vrt_kernels_common.hpp"

namespace vrtk {

// ---------------------------------------------------------------------------------------------
// Dense blocks (hundreds of candidates per 8x8 block: sigma of many pixels, rays of a block see the
// same Gaussians).  One 16-wave workgroup per block: the block's candidates are culled cooperatively
// into LDS once, the EMITTERS are dealt to the 16 waves (wave w takes chunks w, w+16, ...), every
// wave streams all absorbers for its emitters out of LDS (wave-uniform broadcast reads), and the 16
// partial radiances are summed in wave order -- deterministic, no float atomics.  Blocks are pulled
// from a queue with one atomic per block (a block is >= 1e5 instructions; the counter is cold).
// ---------------------------------------------------------------------------------------------
// One block of LDS carved by hand: the two rows the absorber loop reads sit in the first 64 KB, where a DS
// instruction's 16-bit offset field reaches them (arrays placed beyond cost a VALU address add per read: +4 %).
template <int DW>
struct DenseLds {
    float4 A[DCAP], B[DCAP];          // sorted by depth
    float4 L[DW][64];
    uint32_t idx0[DCAP], idx[DCAP];   // "0": in list order; idx: sorted
    float key[DCAP];
    uint32_t wave_cnt[DW];
};
// what the saturation tests decide, per (emitter chunk, absorber) visit of a wave (wave-uniform: scalar adds beside the vector
// work; written out only when statistics are on)
struct DenseVisits { uint32_t full = 0, zero = 0, common = 0; };

// Block `bi` of `cell`, all DW * 64 threads of the workgroup; `scratch`: this workgroup's slot of C.scratch; `file_key`: the block
// is the first of a dense cell of a sparse shard (its key is filed here).  The caller's barrier separates it from the previous use
// of `lds`.
template <int EXP, int ERF, int EC, int DW, bool SKIP>
__device__ __forceinline__ void dense_shade_block(const SceneTables &S, const TileLists &T, const CellGrid &C, const RayGen &R, const RenderTarget &O,
                                                  DenseLds<DW> &lds, uint32_t *scratch, uint32_t cell, uint32_t bi, bool file_key, DenseVisits &visits)
{
    float4(&s_A)[DCAP] = lds.A;
    float4(&s_B)[DCAP] = lds.B;
    float4(&s_L)[DW][64] = lds.L;
    uint32_t(&s_idx0)[DCAP] = lds.idx0;
    uint32_t(&s_idx)[DCAP] = lds.idx;
    float(&s_key)[DCAP] = lds.key;
    uint32_t(&s_wave_cnt)[DW] = lds.wave_cnt;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t npix = (uint64_t)R.width * R.height;
    constexpr float SAT = erf_saturation<ERF>();
    constexpr float SAT_M = SAT + 1e-3f; // the range bounds are re-associated forms of the arguments: keep a margin
    const ErfEval<ERF> erf;
    uint32_t &n_visit_full = visits.full, &n_visit_zero = visits.zero, &n_visit_common = visits.common;
    {
        const BlockPos p = block_of(T, C, O, cell, bi, lane);
        if (!p.inside) return;
        const uint32_t tx = p.t % T.tiles_w, ty = p.t / T.tiles_w;
        bool valid = p.pxt < T.tile_w && p.pyt < T.tile_h;
        const uint32_t pxc = min(p.pxt, T.tile_w - 1), pyc = min(p.pyt, T.tile_h - 1);
        uint64_t pix = (uint64_t)(tx * T.tile_w + pxc) + (uint64_t)T.stride * (ty * T.tile_h + pyc);
        if (pix >= npix) { valid = false; pix = npix - 1; }
        const uint32_t n_active_cells = *C.n_active;
        const uint64_t out = out_index(T, C, O, cell, bi, lane, p, pix, n_active_cells);
        if (O.sparse && file_key && bi == 0 && tid == 0) // a dense cell's key (the active cells' are filed by the list kernel)
            O.keys[n_active_cells + (C.slot[cell] & 0x7FFFFFFFu)] = p.t * (C.cells_x * C.cells_y) + cell % (C.cells_x * C.cells_y);

        uint32_t n_list = C.count[cell];
        const uint32_t *list = C.indices + (size_t)cell * C.cstride;
        if (n_list == 0xFFFFFFFFu) { n_list = T.count[p.t]; list = T.indices + T.start[p.t]; }

        const LaneRay ray = pixel_ray(R, pix); // every wave holds the same 64 rays
        float cx = __shfl(ray.nx, 27, 64) + __shfl(ray.nx, 28, 64) + __shfl(ray.nx, 35, 64) + __shfl(ray.nx, 36, 64);
        float cy = __shfl(ray.ny, 27, 64) + __shfl(ray.ny, 28, 64) + __shfl(ray.ny, 35, 64) + __shfl(ray.ny, 36, 64);
        float cz = __shfl(ray.nz, 27, 64) + __shfl(ray.nz, 28, 64) + __shfl(ray.nz, 35, 64) + __shfl(ray.nz, 36, 64);
        {
            const float inv = __builtin_amdgcn_rsqf(cx * cx + cy * cy + cz * cz);
            cx *= inv; cy *= inv; cz *= inv;
        }
        float co, si;
        cos_sin(ray.nx, ray.ny, ray.nz, cx, cy, cz, co, si);
        const Cone cone = make_cone(cx, cy, cz, wave_min_bpermute(co), wave_max_bpermute(si)); // (not the DPP forms: vrt_kernels_common.hpp)

        // ---- cooperative block cull, order preserving across the 16 waves ----
        uint32_t cnt = 0;
        for (uint32_t base = 0; base < n_list; base += DW * 64) {
            const uint32_t k = base + tid;
            bool keep = false;
            uint32_t idx = 0;
            float4 a, bq;
            if (k < n_list) {
                idx = list[k];
                a = S.gA[idx]; bq = S.gB[idx];
                bq.w = slack_cull_x(bq.w, level_slack(T.cull_ref_n, n_list), T.floor_x);
                keep = cone_keeps(cone, a, bq);
            }
            const unsigned long long mask = __ballot(keep);
            if (lane == 0) s_wave_cnt[wave] = (uint32_t)__popcll(mask);
            __syncthreads();
            uint32_t before = 0, chunk = 0;
#pragma unroll
            for (uint32_t wv = 0; wv < DW; ++wv) {
                const uint32_t c = s_wave_cnt[wv];
                before += (wv < wave) ? c : 0;
                chunk += c;
            }
            const uint32_t pos = cnt + before + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
            if (keep && pos < DCAP) {
                s_idx0[pos] = idx;
                s_key[pos] = a.x * cone.cx + a.y * cone.cy + a.z * cone.cz; // depth along the block's axis
            }
            // the survivors also go to this workgroup's slot of global scratch: should they outgrow LDS, the fallback
            // below streams THEM (cnt^2 pairs) and not the whole cell list (n_list^2)
            if (keep && pos < C.cstride) scratch[pos] = idx;
            cnt += chunk;
            __syncthreads();
        }
        // ---- sort the candidates by depth (rank sort: every thread ranks one candidate against all keys) so
        //      that the emitters of a chunk are neighbours in depth and whole absorbers saturate for them ----
        if (cnt <= DCAP) {
            for (uint32_t i = tid; i < cnt; i += DW * 64) {
                const float ki = s_key[i];
                uint32_t r = 0;
                for (uint32_t k = 0; k < cnt; ++k) {
                    const float kk = s_key[k];
                    r += (kk < ki || (kk == ki && k < i)) ? 1u : 0u;
                }
                const uint32_t idx = s_idx0[i];
                s_idx[r] = idx; s_A[r] = S.gA[idx]; s_B[r] = S.gB[idx]; // rows come back from L2 (read a moment ago)
            }
        }
        __syncthreads();
        if (O.stats && tid == 0) {
            atomicAdd(&O.stats[0], (unsigned long long)(cnt <= C.cstride ? cnt : n_list));
            atomicAdd(&O.stats[1], (unsigned long long)n_list);
            if (cnt > DCAP) atomicAdd(&O.stats[2], 1ull);
            atomicAdd(&O.stats[6], 1ull);
        }

        float Lr = 0.f, Lg = 0.f, Lb = 0.f, La = 0.f;
        if (cnt > DCAP) {
            // does not fit LDS: every wave streams the block's survivors (written above by this workgroup: visible to
            // all its waves after the barrier + fence) for its share of the emitters
            __threadfence_block();
            if (cnt <= C.cstride) shade_list<EXP, ERF, 4, true>(S, scratch, cnt, ray, Lr, Lg, Lb, La, wave * 4, DW * 4);
            else shade_list<EXP, ERF, 4, true>(S, list, n_list, ray, Lr, Lg, Lb, La, wave * 4, DW * 4);
        } else {
            for (uint32_t i0 = wave * EC; i0 < cnt; i0 += DW * EC) {
                float e_mubar[EC], e_sigma[EC];
                uint32_t e_idx[EC];
#pragma unroll
                for (int e = 0; e < EC; ++e) {
                    const uint32_t ii = (i0 + e < cnt) ? (i0 + e) : i0;
                    e_idx[e] = __builtin_amdgcn_readfirstlane(s_idx[ii]);
                    const float4 a = s_A[ii];
                    e_mubar[e] = dot3_ref(a.x, a.y, a.z, ray.nx, ray.ny, ray.nz);
                    e_sigma[e] = uload(S.gD, e_idx[e]).x;
                }
                float acc[EC][5];
#pragma unroll
                for (int e = 0; e < EC; ++e)
#pragma unroll
                    for (int k = 0; k < 5; ++k) acc[e][k] = 0.f;

                // Saturation: Erf(x) is EXACTLY +-1 in fp32 for |x| >= SAT.  An absorber j in front of the camera
                // (m_j >= SAT => E_j = -1) whose Erf argument is <= -SAT at every sample of every emitter of the chunk
                // on every ray of the block adds exactly A_j*(-1 - -1) = 0: skipped before even forming A_j.  One whose
                // arguments are all >= SAT adds exactly -2 A_j to all 5*EC sums: one fma into `common`.
                float common = 0.f;
                float s_max = -INFINITY, s_min = INFINITY; // this ray's sample range over the chunk's emitters
#pragma unroll
                for (int e = 0; e < EC; ++e) {
                    s_max = fmaxf(s_max, e_mubar[e]);
                    s_min = fminf(s_min, __builtin_fmaf(-4.f, e_sigma[e], e_mubar[e]));
                }
                float4 a = s_A[0], b = s_B[0];
                for (uint32_t j = 0; j < cnt; ++j) {
                    const float4 ca = a, cb = b;
                    if (j + 1 < cnt) { a = s_A[j + 1]; b = s_B[j + 1]; }
                    const float mubar = dot3_ref(ca.x, ca.y, ca.z, ray.nx, ray.ny, ray.nz);
                    const float m = mubar * cb.x;
                    // argument range over the chunk's samples on this ray: [(s_min - mubar_j) r_j, (s_max - mubar_j) r_j]
                    const float hi = __builtin_fmaf(s_max, cb.x, -m), lo = __builtin_fmaf(s_min, cb.x, -m);
                    const bool front = m >= SAT;
                    if (SKIP && __all(front && hi <= -SAT_M)) { ++n_visit_zero; continue; }
                    const float d2 = sub_ref(ca.w, mul_ref(mubar, mubar));
                    const float A = cb.z * vexp<EXP>(-(d2 * cb.y));
                    if (SKIP && __all(front && lo >= SAT_M)) { common = __builtin_fmaf(A, -2.f, common); ++n_visit_common; continue; }
                    ++n_visit_full;
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
#pragma unroll
                for (int e = 0; e < EC; ++e) {
                    if (i0 + e < cnt) {
                        const float4 ms = uload(S.mu_sig, e_idx[e]);
                        const float inv2s2 = uload(S.gB, e_idx[e]).y;
                        const float q = uload(S.gD, e_idx[e]).y;
                        float inner = 0.f;
#pragma unroll
                        for (int k = 0; k < 5; ++k) {
                            const float sk = madd_ref((float)(k - 4), ms.w, e_mubar[e]);
                            const float px = sub_ref(madd_ref(ray.nx, sk, ray.ox), ms.x);
                            const float py = sub_ref(madd_ref(ray.ny, sk, ray.oy), ms.y);
                            const float pz = sub_ref(madd_ref(ray.nz, sk, ray.oz), ms.z);
                            const float dd = dot3_ref(px, py, pz, px, py, pz);
                            inner += emission_term<EXP>(q, dd * inv2s2, acc[e][k] + common);
                        }
                        const float4 alb = uload(S.gC, e_idx[e]);
                        Lr = __builtin_fmaf(alb.x, inner, Lr);
                        Lg = __builtin_fmaf(alb.y, inner, Lg);
                        Lb = __builtin_fmaf(alb.z, inner, Lb);
                        La = __builtin_fmaf(alb.w, inner, La);
                    }
                }
            }
        }
        // ---- sum the waves' partial radiances in wave order ----
        s_L[wave][lane] = make_float4(Lr, Lg, Lb, La);
        __syncthreads();
        if (wave == 0 && valid) {
            float4 sum = s_L[0][lane];
#pragma unroll
            for (int w = 1; w < DW; ++w) {
                const float4 v = s_L[w][lane];
                sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
            }
            if (O.image) O.image[out] = pack_pixel(sum.x, sum.y, sum.z, sum.w, O.pack_flags);
            if (O.radiance) O.radiance[out] = sum;
        }
    }
}

} // namespace vrtk
