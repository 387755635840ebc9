// vrt_table_kernel.hip -- the table kernel: dense blocks through a per-ray table of the transmittance exponent.  A translation unit
// of its own (the three kernel units compile side by side); the default scheduler (csrc/Makefile has the measurement).
// gfx950 only: TableLds takes 145 KB of the CU's 160 KB of LDS (static_assert below).
// Diagnostic builds (never the product): -DVRT_TABLE_FORCE=1|2|3 sends every (wave, absorber) visit down one form of the term (timing only,
// wrong images), -DVRT_TABLE_FINE moves the phase clock's slots 1-4 inside the kink pass, -DVRT_TABLE_COUNT_CLASSES counts the one-sign and
// the general visits into the statistics words of VRT_HIP_TABLE_DIAG (profiles/r04_experiments.md).
#include "vrt_dense_block.hpp"

namespace vrtk {

// ---------------------------------------------------------------------------------------------
// Table mode (default; vrt_hip_set_table_step(0) = the exact kernels only): dense blocks through a per-ray TABLE of the
// transmittance exponent.  Along one ray  X(s) = sum_j A_j (E_j - Erf(s r_j - m_j))  is ONE function of s, and the radiance
// needs it at 5 n points (five samples per emitter).  The exact kernel evaluates every one of them term by term: 5 n^2 erf
// terms per ray.  Here X is evaluated at G equidistant nodes of the ray's sample range (n G terms) and the 5 n samples are
// read off by 4-point Lagrange interpolation.
//
// Error control.  The Abramowitz-Stegun erf has a jump of 0.586 in its second derivative at 0 (it is an odd extension of a
// rational function), so X has a kink at every mubar_j.  For a unit erf tabulated with node spacing u (in units of 1/r_j) the
// 4-point interpolant is off by at most 0.0212 u^2 where the stencil contains the kink -- less, by a known factor w <= 1, depending
// on where in the stencil it lies (TB_W0 below) -- and by at most 0.36 u^4 where it does not (tools/table_error_study.py, all phases,
// u <= 0.3).  So for a sample s in node interval g
//     |dX(s)| <= 0.0212 u^2 * K(g) + 0.36 u^4 * S_all,   K(g) = sum of w_j |A_j| over the kinks in intervals g-1 .. g+1,  S_all = sum_j |A_j|,
// and a ray's radiance moves by at most  sum_ik |albedo_i|max * |term_ik| * |dX(s_ik)|  (term_ik = the emission sample).
// The kernel accumulates exactly this sum per ray (K from a per-ray pass over the A_j, kept as one byte per node) and
// keeps a block only if every ray stays below the budget (CellGrid::table_budget, default 2.5e-5: with the cull thresholds'
// 2.5e-5 the frame's worst case is 5e-5, half the 1e-4 tolerance).  A block that fails is tried once more at 0.6 of the
// spacing and then shaded exactly by the same workgroup (dense_shade_block, vrt_dense_block.hpp), as are blocks with more than 2048
// survivors or a sample range of more than 8 x 376 nodes.  The bound is a worst case (every kink at its worst phase,
// all errors aligned): measured deviations are 20-30 x smaller.
//
// Work split: 16 waves hold the same 64 rays (lane = ray).  Wave w owns the contiguous nodes [w NT, (w+1) NT): an absorber
// whose erf is saturated (-1 or +1, erf_saturation<>) over that whole range on all 64 rays costs one add, one whose argument keeps its
// sign there runs on the packed fp32 pipe, two nodes per instruction; which of the three it is is asked once per wave and 64 absorbers
// with lane = absorber (table_nodes).  The per-(ray, absorber) set-up (A_j, m_j, E_j: a dot product, an Exp, an Erf) is made ONCE per
// block instead of by every wave for its own nodes: up to 128 absorbers at a time are staged in the memory of the table itself (wave w
// stages absorbers w, w + 16, ...), one barrier, and every wave then walks all of them without another one.
// ---------------------------------------------------------------------------------------------
// Sizes per number of waves DW of a workgroup (= of a block: every wave holds the block's 64 rays).  DW = 16: one workgroup per CU, a
// table of 384 nodes; DW = 8 (round 4): two workgroups per CU, tables of 192 nodes -- blocks with few survivors and short sample
// ranges spend a third of their time at barriers and in loops of a handful of iterations, which a second workgroup on the CU fills.
template <int DW>
struct TableCfg { static constexpr int TC = 2048, GMAX = 24 * DW, STAGE = 8 * DW; };
// Per-kink error bound of the 4-point interpolant, in units of u^2 |A_j| (tools/table_error_study.py verifies the constants
// for u <= 0.3, all phases): a kink at offset theta in [0, 1) of its node interval moves the interpolant by at most
//   TB_W0 min(1, 0.28 + 2.58 |theta - 1/2|) u^2 in that interval, TB_W0 theta^2 u^2 in the interval to its right,
//   TB_W0 (1 - theta)^2 u^2 in the one to its left, and by at most TB_COUT u^4 anywhere else.
constexpr float TB_W0 = 0.0212f, TB_COUT = 0.36f;
// Where the table treats an Erf as +-1 (on all of a wave's nodes and all 64 rays): the Abramowitz-Stegun form reaches exactly +-1 only at
// |x| = 5.46 -- its tail is 1 / p(|x|)^4, not a Gaussian's -- so an absorber stays "live" over +-5.5 / u nodes.  Cut at 4.5 the tail left
// out is R(4.5) = 4.31e-7 |A_j| per absorber (libm: erfc(3.5) = 7.43e-7), a step of that size where a saturated wave's nodes meet a live
// one's (4-point interpolation across a step: at most 1.25 of it): |dX| <= 1.25 R(4.5) S_all everywhere, a term of the bound like the
// smooth part's 0.36 u^4 S_all -- and a fifth of that at the finest spacing the kernel uses.  18 % fewer live (absorber, node) pairs.
template <int ERF> __host__ __device__ constexpr float table_saturation() { return ERF == VRT_ERF_AS ? 4.5f : 3.5f; }
template <int ERF> __host__ __device__ constexpr float table_saturation_eps() { return ERF == VRT_ERF_AS ? 1.25f * 4.31e-7f : 1.25f * 7.44e-7f; }
template <int DW>
struct TableLds {
    static constexpr int TB_TC = TableCfg<DW>::TC, TB_GMAX = TableCfg<DW>::GMAX, TB_STAGE = TableCfg<DW>::STAGE, TB_DW = DW;
    uint32_t idx[TB_TC];                       //  8 KB: the block's survivors, in list order (every pass below needs a survivor's rows
                                               //        once per wave: WaveRows)
    union {                                    // 96 KB: X at node g of lane l; before that the kink weights per interval (fixed point);
        float tab[TB_GMAX][64];                //        after the emission pass the partial error sums
        uint32_t hist[TB_GMAX][64];
    };
    uint8_t s3[TB_GMAX][64];                   // 24 KB: kink weight of interval g / S_all in 1/255, rounded up
    union {                                    // 16 KB
        float4 L[TB_DW][64];                                     // partial radiances
        float red[4][TB_DW][64];                                 // partial sums of the range and histogram passes
    };
    float st_r[TB_STAGE], st_cmin[TB_STAGE], st_cmax[TB_STAGE]; // per staged absorber (table_nodes): r_j and the range of its argument offset over the 64 rays
};
// 145 KB of the 160 KB of LDS a gfx950 CU has (one workgroup per CU): this translation unit is gfx950-only; another ARCH needs a smaller
// TB_GMAX (or a build whose host keeps table_on() false)
static_assert(sizeof(TableLds<16>) + 512 <= 160 * 1024, "TableLds must fit the 160 KB of LDS of a gfx950 CU");
static_assert(sizeof(TableLds<8>) + 512 <= 80 * 1024, "two 8-wave workgroups share the 160 KB of LDS of a gfx950 CU");

// The rows (gA, gB) of the survivors a wave works on in one pass -- survivors first, first + 16, ... (at most 64 of them) -- fetched with ONE
// gather, lane i loading the rows of survivor first + 16 i, and handed to the loop iteration by v_readlane_b32 (the passes used to fetch a
// survivor's rows by scalar loads one iteration ahead).  Measured neutral (profiles/r04_experiments.md): the fixed phases are not waiting
// for those loads, they issue ~90 instructions per survivor on four waves per SIMD; kept because the loops lose their lgkmcnt waits.
template <int TB_DW>
struct WaveRows {
    float4 a, b;
    __device__ __forceinline__ void load(const SceneTables &S, const uint32_t *idx /* LDS */, uint32_t first, uint32_t end, uint32_t lane)
    {
        a = make_float4(0.f, 0.f, 0.f, 0.f); b = a;
        const uint32_t j = first + (uint32_t)TB_DW * lane;
        if (j < end) { const uint32_t id = idx[j]; a = S.gA[id]; b = S.gB[id]; }
    }
    static __device__ __forceinline__ float rl(float v, uint32_t i) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), (int)i)); }
    __device__ __forceinline__ float4 A(uint32_t i) const { return make_float4(rl(a.x, i), rl(a.y, i), rl(a.z, i), rl(a.w, i)); }
    __device__ __forceinline__ float4 B(uint32_t i) const { return make_float4(rl(b.x, i), rl(b.y, i), rl(b.z, i), rl(b.w, i)); }
    // iterations of the loop over survivors first, first + 16, ... < end (wave-uniform)
    static __device__ __forceinline__ uint32_t count(uint32_t first, uint32_t end)
    {
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)(first < end ? min(64u, (end - first + (uint32_t)TB_DW - 1u) / (uint32_t)TB_DW) : 0u));
    }
};

// acc[k] += A2 * R(X0 + k HR2) for the NT / 2 node pairs of a wave, written THREE pairs abreast: the chains of one pair (argument, four
// fmas, two squarings, two reciprocals, the sum) are each other's operands, and the scheduler, short of registers at four waves per SIMD,
// otherwise lays them out one after the other with a wait state between any two packed instructions
template <int NT, typename Erf>
__device__ __forceinline__ void pair_terms(const Erf &erf, v2f (&acc)[NT / 2], v2f A2, v2f HR2, v2f X0)
{
    constexpr int W = 3;
#pragma unroll
    for (int k0 = 0; k0 < NT / 2; k0 += W) {
        v2f t[W], p[W];
#pragma unroll
        for (int i = 0; i < W; ++i)
            if (k0 + i < NT / 2) { const v2f K = { (float)(k0 + i), (float)(k0 + i) }; t[i] = fma2(K, HR2, X0); }
#pragma unroll
        for (int i = 0; i < W; ++i) if (k0 + i < NT / 2) p[i] = fma2((v2f){ erf.c3, erf.c3 }, t[i], (v2f){ erf.c2, erf.c2 });
#pragma unroll
        for (int i = 0; i < W; ++i) if (k0 + i < NT / 2) p[i] = fma2(p[i], t[i], (v2f){ erf.c1, erf.c1 });
#pragma unroll
        for (int i = 0; i < W; ++i) if (k0 + i < NT / 2) p[i] = fma2(p[i], t[i], (v2f){ erf.c0, erf.c0 });
#pragma unroll
        for (int i = 0; i < W; ++i) if (k0 + i < NT / 2) p[i] = fma2(p[i], t[i], (v2f){ 1.0f, 1.0f });
#pragma unroll
        for (int i = 0; i < W; ++i) if (k0 + i < NT / 2) p[i] = p[i] * p[i];
#pragma unroll
        for (int i = 0; i < W; ++i) if (k0 + i < NT / 2) p[i] = p[i] * p[i];
#pragma unroll
        for (int i = 0; i < W; ++i) if (k0 + i < NT / 2) p[i] = (v2f){ __builtin_amdgcn_rcpf(p[i].x), __builtin_amdgcn_rcpf(p[i].y) };
#pragma unroll
        for (int i = 0; i < W; ++i) if (k0 + i < NT / 2) acc[k0 + i] = fma2(A2, p[i], acc[k0 + i]);
    }
}

// one pass over the survivors for the NT nodes [g0, g0 + NT) of this wave; tab[g] = sum_j A_j (E_j - Erf(x_gj))
//
// Round 4, second half: what a wave does with an absorber -- nothing but one add (its Erf is saturated on all of the wave's nodes on all
// 64 rays), the one-sign form of the term, or the general form -- used to be asked absorber by absorber inside the node loop: four vector
// compares, their way to the scalar unit, a three-way branch whose arms the register allocator joined with a copy of every accumulator.
// With the arms compiled in unconditionally the same loop ran 15-25 % faster than with the questions in it (VRT_TABLE_FORCE experiments,
// profiles/r04_experiments.md).  Now the questions are asked ONCE per wave and group of 64 staged absorbers, with LANE = ABSORBER, from
// two wave-uniform numbers per absorber that the staging pass leaves in LDS (the smallest and the largest argument offset over the 64
// rays): five ballots give five bit masks, and five tight loops walk their set bits -- no question, no branch between forms, every
// accumulator updated in place.  The one-sign loops run on the packed fp32 pipe (v_pk_fma_f32 / v_pk_mul_f32: two nodes per
// instruction; 26.5 cycles per term against 36.3, tools/ubench/erf_term.hip, profiles/r04_erf_term_packed.txt) and keep
// A (E -+ 1) out of the node sums (one fma per absorber instead of one add per term).
//
// The sums: tab[g] = [sum over the saturated and one-sign absorbers of A (E -+ 1)] + [sum of +-A R(|x_g|)] + [general terms]; the first
// bracket is common to the wave's nodes.  Rounding is at the magnitude of sum |A_j| (< 60, or the block is shaded exactly) times 2^-24 per
// addition -- the same order as the rounding of E and Erf themselves (values near 1 with an ulp of 6e-8) in the term-by-term form.
template <int EXP, int ERF, int NT, int TB_DW>
__device__ __forceinline__ void table_nodes(const SceneTables &S, TableLds<TB_DW> &lds, uint32_t cnt, const LaneRay &ray,
                                            float s_seg /* node 0 of this segment's table on this lane's ray */, float h, uint32_t g0,
                                            uint32_t wave, uint32_t lane, uint32_t &n_skip,
                                            unsigned long long *diag /* nullable: statistics runs */)
{
    // diagnostics (statistics runs, every wave's clock; stats words 22, 23): ticks in the node loops and waiting at the group barriers
    unsigned long long d_stage = 0, d_loop = 0, d_wait = 0, d_t = diag ? wall_clock64() : 0ull;
#ifdef VRT_TABLE_COUNT_CLASSES
    auto lap = [&](unsigned long long &) {};
#else
    auto lap = [&](unsigned long long &acc) { if (diag) { const unsigned long long t = wall_clock64(); acc += t - d_t; d_t = t; } };
#endif
    constexpr float SAT_M = table_saturation<ERF>() + 1e-3f;
    constexpr bool AS = ERF == VRT_ERF_AS;
    constexpr int TB_GMAX = TableCfg<TB_DW>::GMAX, TB_STAGE = TableCfg<TB_DW>::STAGE;
    const ErfEval<ERF> erf;
    static_assert(NT % 2 == 0, "the nodes of a wave are worked on in pairs");
    // the wave's nodes in PAIRS (2k, 2k + 1), arguments x_2k = fma(k, 2 h r, x0), x_2k+1 = fma(k, 2 h r, x0 + h r): one definition for the
    // packed and the scalar form of the term
    v2f acc[NT / 2];
#pragma unroll
    for (int k = 0; k < NT / 2; ++k) acc[k] = (v2f){ 0.f, 0.f };
    float common = 0.f;
    const float g0h = (float)g0 * h, g1h = g0h + (float)(NT - 1) * h; // the wave's first and last node, from the segment's node 0
    const float s_first = __builtin_fmaf((float)g0, h, s_seg);        // this lane's position of node g0
    // Staging: the per-(ray, absorber) values (A, m, E) of up to TB_STAGE = 128 absorbers at a time go into the memory of the TABLE itself
    // -- it is written only at the end of this function, and until then its 96 KB are free: three [128][64] float arrays.  Wave w stages
    // absorbers w, w + 16, ... of the group, one barrier, then every wave walks the whole group without another barrier (rounds 1-3
    // staged 16 at a time, a barrier per 16: the waves of a SIMD do not run in step -- the arbiter favours the oldest -- so every barrier
    // had early finishers idling).  Per absorber also r and the range [cmin, cmax] over the 64 rays of c = s_seg r - m: the argument of
    // node g on a ray is c + g h r.
    static_assert(TB_GMAX * 64 == 3 * TB_STAGE * 64, "three staging arrays fill the table exactly");
    float(*stA)[64] = reinterpret_cast<float(*)[64]>(&lds.tab[0][0]);
    float(*stM)[64] = stA + TB_STAGE;
    float(*stE)[64] = stM + TB_STAGE;
    for (uint32_t base = 0; base < cnt; base += TB_STAGE) {
        const uint32_t nb = min((uint32_t)TB_STAGE, cnt - base);
        WaveRows<TB_DW> rows;
        rows.load(S, lds.idx + base, wave, nb, lane);
        const uint32_t n_it = WaveRows<TB_DW>::count(wave, nb);
        for (uint32_t i = 0; i < n_it; ++i) {
            const uint32_t jj = wave + (uint32_t)TB_DW * i;
            const float4 pa = rows.A(i), pb = rows.B(i);
            const float mubar = dot3_ref(pa.x, pa.y, pa.z, ray.nx, ray.ny, ray.nz);
            const float d2 = sub_ref(pa.w, mul_ref(mubar, mubar));
            const float m = mubar * pb.x;
            stA[jj][lane] = pb.z * vexp<EXP>(-(d2 * pb.y)); stM[jj][lane] = m; stE[jj][lane] = erf(-m);
            const float c = __builtin_fmaf(s_seg, pb.x, -m);
            const float c_lo = wave_min(c), c_hi = wave_max(c);
            if (lane == 0) { lds.st_r[jj] = pb.x; lds.st_cmin[jj] = c_lo; lds.st_cmax[jj] = c_hi; }
        }
        lap(d_stage);
        __syncthreads();
        lap(d_wait);
        for (uint32_t half = 0; half < nb; half += 64) {
            // ---- the questions, lane = absorber: is Erf saturated (-1: the absorber lies behind the wave's nodes, +1: in front) on all
            //      nodes and rays?  Is the argument of one sign on all of them (all but the absorbers whose kink lies inside the range)?
            //      The answers only choose between forms of the same term: an argument that the rounding of the two formulas (here
            //      c + g h r, below fma(s, r, -m)) puts on the other side of 0 or of the saturation point moves the term by less
            //      than 1e-7 |A| (Erf is odd and smooth to first order at 0; R(5.5) = 3e-8). ----
            unsigned long long m_lo, m_hi, m_pos, m_neg, m_gen;
            {
                const uint32_t j = half + lane;
                const bool in = j < nb;
                const uint32_t jc = in ? j : 0u;
                const float r_j = lds.st_r[jc], x_lo = __builtin_fmaf(g0h, r_j, lds.st_cmin[jc]), x_hi = __builtin_fmaf(g1h, r_j, lds.st_cmax[jc]);
#if defined(VRT_TABLE_FORCE)   /* timing experiments only (wrong images): every visit on one form */
                const bool b_lo = in && VRT_TABLE_FORCE == 3, b_hi = false, b_pos = in && VRT_TABLE_FORCE == 1, b_neg = false;
#else
                const bool b_lo = in && x_hi <= -SAT_M, b_hi = in && x_lo >= SAT_M;
                const bool b_pos = AS && in && !b_hi && x_lo >= 0.f, b_neg = AS && in && !b_lo && !b_pos && x_hi <= 0.f;
#endif
                m_lo = __ballot(b_lo); m_hi = __ballot(b_hi); m_pos = __ballot(b_pos); m_neg = __ballot(b_neg);
                m_gen = __ballot(in && !(b_lo | b_hi | b_pos | b_neg));
            }
            n_skip += (uint32_t)__popcll(m_lo) + (uint32_t)__popcll(m_hi);
#ifdef VRT_TABLE_COUNT_CLASSES   /* diagnostic build: stats words 22 / 23 = one-sign / general (wave, absorber) visits instead of the clocks */
            d_loop += (unsigned long long)__popcll(m_pos) + (unsigned long long)__popcll(m_neg);
            d_wait += (unsigned long long)__popcll(m_gen);
#endif
            // ---- saturated: E - Erf = E + 1 or E - 1 on every node ----
            auto walk_AE = [&](unsigned long long mask, auto &&body) {
                if (!mask) return;
                uint32_t jn = half + (uint32_t)__builtin_ctzll(mask);
                float nA = stA[jn][lane], nE = stE[jn][lane];
                while (mask) {
                    const float A = nA, E = nE;
                    mask &= mask - 1ull;
                    if (mask) { jn = half + (uint32_t)__builtin_ctzll(mask); nA = stA[jn][lane]; nE = stE[jn][lane]; }
                    body(A, E);
                }
            };
            walk_AE(m_lo, [&](float A, float E) { common = __builtin_fmaf(A, E + 1.f, common); });
            walk_AE(m_hi, [&](float A, float E) { common = __builtin_fmaf(A, E - 1.f, common); });
            // ---- the other forms: the next absorber's staged values are requested one iteration ahead (LDS latency behind the terms) ----
            auto walk = [&](unsigned long long mask, auto &&body) {
                if (!mask) return;
                uint32_t jn = half + (uint32_t)__builtin_ctzll(mask);
                float nA = stA[jn][lane], nM = stM[jn][lane], nE = stE[jn][lane], nR = lds.st_r[jn];
                while (mask) {
                    const float A = nA, m = nM, E = nE, r = nR;
                    mask &= mask - 1ull;
                    if (mask) { jn = half + (uint32_t)__builtin_ctzll(mask); nA = stA[jn][lane]; nM = stM[jn][lane]; nE = stE[jn][lane]; nR = lds.st_r[jn]; }
                    body(A, m, E, r);
                }
            };
            if constexpr (AS) {
                // one sign, x >= 0: Erf = 1 - R(x), E - Erf = (E - 1) + R; x <= 0: Erf = -(1 - R(-x)), E - Erf = (E + 1) - R.  |x| without an
                // |abs| modifier (the packed instructions have none): fma(k, -a, -b) = -fma(k, a, b) exactly, the negations are modifiers.
                walk(m_pos, [&](float A, float m, float E, float r) {
                    const float hr = h * r, hr2 = hr + hr, x0 = __builtin_fmaf(s_first, r, -m), x0b = x0 + hr;
                    const v2f X0 = { x0, x0b }, HR2 = { hr2, hr2 }, A2 = { A, A };
                    common = __builtin_fmaf(A, E - 1.f, common);
                    pair_terms<NT>(erf, acc, A2, HR2, X0);
                });
                walk(m_neg, [&](float A, float m, float E, float r) {
                    const float hr = h * r, hr2 = hr + hr, x0 = __builtin_fmaf(s_first, r, -m), x0b = x0 + hr;
                    const v2f X0 = { -x0, -x0b }, HR2 = { -hr2, -hr2 }, A2 = { -A, -A };
                    common = __builtin_fmaf(A, E + 1.f, common);
                    pair_terms<NT>(erf, acc, A2, HR2, X0);
                });
            }
            walk(m_gen, [&](float A, float m, float E, float r) {
                const float hr = h * r, hr2 = hr + hr, x0 = __builtin_fmaf(s_first, r, -m), x0b = x0 + hr;
#pragma unroll
                for (int k = 0; k < NT / 2; ++k) {
                    acc[k].x = __builtin_fmaf(A, E - erf(__builtin_fmaf((float)k, hr2, x0)), acc[k].x);
                    acc[k].y = __builtin_fmaf(A, E - erf(__builtin_fmaf((float)k, hr2, x0b)), acc[k].y);
                }
            });
        }
        lap(d_loop);
        __syncthreads(); // everyone is done with the staged values: the next group, or the table itself, goes into their memory
        lap(d_wait);
    }
    if (diag && lane == 0) { atomicAdd(&diag[0], d_loop); atomicAdd(&diag[1], d_wait); } // summed over the 16 waves (staging = the rest of the table phase)
#pragma unroll
    for (int t = 0; t < NT; ++t)
        if (g0 + t < (uint32_t)TB_GMAX) lds.tab[g0 + t][lane] = ((t & 1) ? acc[t / 2].y : acc[t / 2].x) + common;
}

template <int EXP, int ERF, int DW>
__device__ __forceinline__ void render_table_body(const SceneTables &S, const TileLists &T, const CellGrid &C, const RayGen &R,
                                                  const RenderTarget &O)
{
    constexpr int TC = TableCfg<DW>::TC, TB_GMAX = TableCfg<DW>::GMAX, TB_DW = DW;
    __shared__ TableLds<DW> lds;
    __shared__ uint32_t s_wave_cnt[DW];
    __shared__ float s_rmax[DW];
    __shared__ uint32_t s_item, s_flag;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t npix = (uint64_t)R.width * R.height;
    const uint32_t n_dense16 = *C.n_dense * 16u, n_items = n_dense16 + *C.n_overflow;
    if (C.feedback && blockIdx.x == 0 && tid == 0) {
        __hip_atomic_store(&C.feedback[2], n_items, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&C.feedback[3], C.frame_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const uint32_t *dense_queue = C.dense_is_sorted ? C.dense_sorted : C.dense;
    uint32_t n_skip = 0; // (absorber, wave) visits the saturation test settled with one add (statistics)
    DenseVisits visits;  // ... and what the exact fallback's saturation tests decided

    for (;;) {
        __syncthreads(); // everyone is done with the previous item's LDS
        if (tid == 0) s_item = atomicAdd(C.dense_next, 1u);
        __syncthreads();
        const uint32_t item = s_item;
        if (item >= n_items) break;
        // phase clock (statistics runs only): thread 0 adds the time since the last stamp to stats[24 + phase]
        unsigned long long t_last = (O.stats && tid == 0) ? wall_clock64() : 0ull;
        auto stamp = [&](int phase) {
#ifdef VRT_TABLE_FINE   /* diagnostic build: slots 1-4 belong to the inside of kink_pass */
            if (phase >= 1 && phase <= 4) phase = 0;
#endif
            if (O.stats && tid == 0) { const unsigned long long t = wall_clock64(); atomicAdd(&O.stats[24 + phase], t - t_last); t_last = t; }
        };
        uint32_t cell, bi;
        if (item < n_dense16) { cell = dense_queue[item >> 4]; bi = item & 15u; }
        else { const uint32_t packed = C.overflow[item - n_dense16]; cell = packed >> 4; bi = packed & 15u; }
        const BlockPos p = block_of(T, C, O, cell, bi, lane);
        if (!p.inside) continue;
        const uint32_t tx = p.t % T.tiles_w, ty = p.t / T.tiles_w;
        bool valid = p.pxt < T.tile_w && p.pyt < T.tile_h;
        const uint32_t pxc = min(p.pxt, T.tile_w - 1), pyc = min(p.pyt, T.tile_h - 1);
        uint64_t pix = (uint64_t)(tx * T.tile_w + pxc) + (uint64_t)T.stride * (ty * T.tile_h + pyc);
        if (pix >= npix) { valid = false; pix = npix - 1; }
        const uint32_t n_active_cells = *C.n_active;
        const uint64_t out = out_index(T, C, O, cell, bi, lane, p, pix, n_active_cells);
        if (O.sparse && item < n_dense16 && bi == 0 && tid == 0) // a dense cell's key (the active cells' are filed by the list kernel)
            O.keys[n_active_cells + (C.slot[cell] & 0x7FFFFFFFu)] = p.t * (C.cells_x * C.cells_y) + cell % (C.cells_x * C.cells_y);

        uint32_t n_list = C.count[cell];
        const uint32_t *list = C.indices + (size_t)cell * C.cstride;
        if (n_list == 0xFFFFFFFFu) { n_list = T.count[p.t]; list = T.indices + T.start[p.t]; }

        const LaneRay ray = pixel_ray(R, pix); // every wave holds the same 64 rays
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
        stamp(0);

        // ---- cooperative block cull, order preserving across the 16 waves (as in the exact kernel) ----
        uint32_t cnt = 0;
        for (uint32_t base = 0; base < n_list; base += DW * 64) {
            const uint32_t k = base + tid;
            bool keep = false;
            uint32_t idx = 0;
            if (k < n_list) {
                idx = list[k];
                float4 bq = S.gB[idx];
                bq.w = slack_cull_x(bq.w, level_slack(T.cull_ref_n, n_list), T.floor_x);
                keep = cone_keeps(cone, S.gA[idx], bq);
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
            if (keep && pos < TC) lds.idx[pos] = idx;
            cnt += chunk;
            __syncthreads();
        }

        stamp(1);
        if (cnt == 0) { // nothing reaches this block (the rim of a dense cell): background
            if (wave == 0 && valid) {
                if (O.image) O.image[out] = pack_pixel(0.f, 0.f, 0.f, 0.f, O.pack_flags);
                if (O.radiance) O.radiance[out] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (O.stats && tid == 0) { atomicAdd(&O.stats[1], (unsigned long long)n_list); atomicAdd(&O.stats[6], 1ull); atomicAdd(&O.stats[7], 1ull); atomicAdd(&O.stats[21], 1ull); }
            continue;
        }
        // ---- every ray's sample range (wave w looks at survivors w, w + 16, ...), the block's node spacing ----
        bool ok = cnt <= (uint32_t)TC;
        float s_lo = INFINITY, s_hi = -INFINITY, r_max = 0.f;
        if (ok) {
            // (the wave's rows by one gather per 1024 survivors: WaveRows; here and in the passes below)
            for (uint32_t first = wave; first < cnt; first += 64u * DW) {
                WaveRows<DW> rows;
                rows.load(S, lds.idx, first, cnt, lane);
                const uint32_t n_it = WaveRows<DW>::count(first, cnt);
                for (uint32_t i = 0; i < n_it; ++i) {
                    const float4 a = rows.A(i);
                    const float r = WaveRows<DW>::rl(rows.b.x, i);
                    const float mubar = dot3_ref(a.x, a.y, a.z, ray.nx, ray.ny, ray.nz);
                    s_hi = fmaxf(s_hi, mubar);
                    s_lo = fminf(s_lo, mubar - 2.8285f / r); // mubar - 4 sigma, sigma = 1/(sqrt2 r), rounded outwards
                    r_max = fmaxf(r_max, r);
                }
            }
            lds.red[0][wave][lane] = s_hi; lds.red[1][wave][lane] = s_lo;
            if (lane == 0) s_rmax[wave] = r_max;
            __syncthreads();
#pragma unroll
            for (int w = 0; w < DW; ++w) {
                s_hi = fmaxf(s_hi, lds.red[0][w][lane]); s_lo = fminf(s_lo, lds.red[1][w][lane]);
                r_max = fmaxf(r_max, s_rmax[w]);
            }
            __syncthreads();
        }
        stamp(2);
        const float range = wave_max(s_hi - s_lo); // the block's longest sample range
        const float h_req = C.table_hx / r_max;    // requested spacing: table_hx in units of 1/r of the narrowest Gaussian
        ok = ok && range >= 0.f && h_req > 0.f && range < INFINITY; // (false for NaN)
        bool done = false;
        float h_target = h_req;
        for (int attempt = 0; ok && !done; ++attempt) {
            // Every ray has its own grid of Gtot nodes from its first sample on (two nodes of margin at either end, so that
            // every sample has its four neighbours), all with the block's spacing.  The table holds TB_GMAX nodes: a deeper
            // range is worked off in segments of SL intervals (+ the margins); a sample belongs to the segment its interval
            // lies in.  Segments in empty space cost next to nothing: every absorber is saturated there.  The waves evaluate
            // NT nodes each, NT from a short menu: the spacing is then REDUCED until the segments fill 16 NT nodes exactly.
            float h = 0.f, u = 0.f, lo = 0.f, inv_h = 0.f;
            uint32_t nseg = 0, SL = 0, G = 0, NTsel = 0, Gtot = 0;
            auto plan = [&](float ht) -> bool {
                const float need = ceilf(range / ht); // intervals the samples span
                if (!(need < 8.f * (float)(TB_GMAX - 8))) return false;
                nseg = max(1u, ((uint32_t)need + (uint32_t)(TB_GMAX - 8) - 1u) / (uint32_t)(TB_GMAX - 8));
                const uint32_t sl_need = max(1u, ((uint32_t)need + nseg - 1u) / nseg);
                const uint32_t nt = (sl_need + 8u + DW - 1) / DW;
                NTsel = nt <= 4 ? 4 : nt <= 6 ? 6 : nt <= 8 ? 8 : nt <= 12 ? 12 : nt <= 16 ? 16 : nt <= 20 ? 20 : 24;
                // intervals per segment; nodes in the table: the segment's intervals, two nodes before them, and up to six behind
                // the last one (the samples' intervals start at 2 and end at Gtot - 4 <= nseg SL + 2, whose stencil ends at nseg SL + 4)
                G = NTsel * DW; SL = G - 8u;
                h = fminf(ht, range / (float)(nseg * SL) * 1.00001f);
                if (!(h > 0.f)) h = ht; // range == 0: one sample point per ray
                u = h * r_max;
                Gtot = nseg * SL + 6u;
                lo = s_lo - 2.f * h; inv_h = 1.f / h;
                return u <= 0.3f;
            };
            if (!plan(h_target)) { ok = false; break; }

            float S_all = 0.f;
            // ---- kink pass for one segment: wave w takes absorbers w, w + 16, ...: the weight of the kink of j (TB_W0 units,
            //      fixed point, rounded up; integer adds: the order of the atomics does not matter) into the interval of mubar_j
            //      and its two neighbours; with `sums` also S_all = sum |A_j| ----
            auto kink_pass = [&](uint32_t seg, bool sums) {
                const float node0 = (float)(seg * SL) - 2.f; // the segment's first node on the ray's grid
#ifdef VRT_TABLE_FINE
                auto fine = [&](int k) { if (O.stats && tid == 0) { const unsigned long long t = wall_clock64(); atomicAdd(&O.stats[24 + k], t - t_last); t_last = t; } };
#else
                auto fine = [&](int) {};
#endif
                fine(0);
                for (uint32_t g = wave; g < G; g += DW) lds.hist[g][lane] = 0u;
                __syncthreads();
                fine(1);
                float s_part = 0.f;
                for (uint32_t first = wave; first < cnt; first += 64u * DW) {
                    WaveRows<DW> rows;
                    rows.load(S, lds.idx, first, cnt, lane);
                    const uint32_t n_it = WaveRows<DW>::count(first, cnt);
                    for (uint32_t i = 0; i < n_it; ++i) {
                        const float4 ca = rows.A(i), cb = rows.B(i);
                        const float mubar = dot3_ref(ca.x, ca.y, ca.z, ray.nx, ray.ny, ray.nz);
                        const float d2 = sub_ref(ca.w, mul_ref(mubar, mubar));
                        const float A = cb.z * vexp<EXP>(-(d2 * cb.y));
                        if (sums) s_part += fabsf(A);
                        const float pos = (mubar - lo) * inv_h;
                        const float gb = floorf(pos);
                        const float th = fminf(fmaxf(pos - gb, 0.f), 1.f);
                        const float a16 = fminf(fabsf(A), 60.f) * 65536.f;
                        const float gl = gb - node0; // interval of the kink in this segment's table
                        if (gl >= 0.f && gl < (float)G) atomicAdd(&lds.hist[(uint32_t)gl][lane], (uint32_t)ceilf(a16 * fminf(1.f, 0.28f + 2.58f * fabsf(th - 0.5f))));
                        if (gl + 1.f >= 0.f && gl + 1.f < (float)G) atomicAdd(&lds.hist[(uint32_t)(gl + 1.f)][lane], (uint32_t)ceilf(a16 * th * th));
                        if (gl - 1.f >= 0.f && gl - 1.f < (float)G) atomicAdd(&lds.hist[(uint32_t)(gl - 1.f)][lane], (uint32_t)ceilf(a16 * (1.f - th) * (1.f - th)));
                    }
                }
                fine(2);
                if (sums) lds.red[3][wave][lane] = s_part;
                __syncthreads();
                fine(3);
                if (sums) {
                    S_all = 0.f;
#pragma unroll
                    for (int w = 0; w < DW; ++w) S_all += lds.red[3][w][lane];
                }
            };

            // ---- first attempt: how much coarser than requested may the nodes be?  An ESTIMATE of the bound the emission pass
            //      will find, from the kink weights: emission of interval g ~ 1.4 D_g T_g with D_g = K_g / 1.35 the absorber mass of
            //      the interval and T_g = exp(-2 sum of the mass before it); the estimate scales with the spacing like rho^3 (kink
            //      part) and rho^4 (smooth part).  It only picks the spacing: the bound itself is checked below, and a block that
            //      fails it is redone at 0.6 of the spacing.
            //      The weights are taken at the COARSEST spacing the settings and the menu of table sizes allow (rounds 3-4 took them
            //      at the requested spacing -- two or three segments' passes for a teapot block -- and then again at the chosen one):
            //      44-48 % of the blocks of the OBJ scenes keep that spacing, and for them these weights are the final ones; the
            //      others scale the estimate DOWN to the finer candidates. ----
            bool have_kinks = false; // the kink weights of segment 0 at the final spacing are in LDS
            bool have_sums = false;  // S_all is known (it does not depend on the spacing)
            if (attempt == 0 && C.table_adapt > 1.f) {
                const float h_r = h, u_r = u; // the requested spacing, as the menu of table sizes rounds it
                const uint32_t nodes_r = nseg * NTsel;
                auto menu = [](int c) { return c == 0 ? 3.f : c == 1 ? 2.5f : c == 2 ? 2.f : c == 3 ? 1.6f : 1.3f; };
                float k_max = 1.f;
#pragma unroll
                for (int c = 4; c >= 0; --c)
                    if (menu(c) <= C.table_adapt && menu(c) * u_r <= 0.3f) k_max = menu(c);
                if (k_max > 1.f && plan(h_r * k_max) && nseg * NTsel < nodes_r) {
                    const float h_c = h;
                    float P = 0.f, pin = 0.f, pout = 0.f;
                    for (uint32_t seg = 0; seg < nseg; ++seg) {
                        kink_pass(seg, seg == 0);
                        const uint32_t ga = seg ? 2u : 0u, gz = min(G, SL + 2u); // the segment's own intervals
                        const uint32_t wa = min(gz, ga + wave * NTsel), wz = min(gz, wa + NTsel);
                        float mass = 0.f;
                        for (uint32_t g = wa; g < wz; ++g) mass += (float)lds.hist[g][lane];
                        lds.red[0][wave][lane] = mass * (1.f / (1.35f * 65536.f));
                        __syncthreads();
                        float before = P, total = 0.f;
#pragma unroll
                        for (int w = 0; w < DW; ++w) {
                            const float mw = lds.red[0][w][lane];
                            before += (uint32_t)w < wave ? mw : 0.f;
                            total += mw;
                        }
                        for (uint32_t g = wa; g < wz; ++g) {
                            const float Kg = (float)lds.hist[g][lane] * (1.f / 65536.f), D = Kg * (1.f / 1.35f);
                            const float e = 1.4f * D * __expf(-2.f * before);
                            pin = __builtin_fmaf(e, Kg, pin); pout += e;
                            before += D;
                        }
                        P += total;
                        __syncthreads(); // red[0] is rewritten by the next segment
                    }
                    have_sums = true;
                    lds.red[0][wave][lane] = pin; lds.red[1][wave][lane] = pout;
                    __syncthreads();
                    float pin_t = 0.f, pout_t = 0.f;
#pragma unroll
                    for (int w = 0; w < DW; ++w) { pin_t += lds.red[0][w][lane]; pout_t += lds.red[1][w][lane]; }
                    const float est_in = 1.01f * TB_W0 * u * u * pin_t, est_out = 1.01f * TB_COUT * (u * u) * (u * u) * S_all * pout_t, est_sat = 1.01f * table_saturation_eps<ERF>() * S_all * pout_t;
                    const float room = C.table_room * C.table_budget;
                    // the coarsest candidate whose estimate leaves room: k_max itself with the weights as they are (rho = 1), the finer
                    // ones by scaling; none: the requested spacing
                    float kappa = 1.f;
#pragma unroll
                    for (int c = 0; c < 5; ++c) {
                        const float k = menu(c), rho = k == k_max ? 1.f : h_r * k / h_c;
                        if (kappa == 1.f && k <= k_max && rho * rho * rho * (est_in + rho * est_out) + est_sat <= room) kappa = k;
                    }
                    kappa = wave_min(valid ? kappa : k_max); // the same in every wave: they hold the same rays
#ifdef VRT_TABLE_FINE   /* diagnostic build: slot 4 = (sum of 10 kappa) << 32 | blocks that took the coarsest candidate */
                    if (O.stats && tid == 0) atomicAdd(&O.stats[24 + 4], ((unsigned long long)(kappa * 10.f + 0.5f) << 32) | (kappa == k_max ? 1ull : 0ull));
#endif
                    __syncthreads();
                    if (kappa == k_max) have_kinks = nseg == 1u; // segment 0's weights at this spacing are still in LDS
                    else if (kappa == 1.f || !plan(h_r * kappa) || nseg * NTsel >= nodes_r) { // (the menu has no smaller table: as requested)
                        if (!plan(h_target)) { ok = false; break; }
                    }
                } else if (!plan(h_target)) { ok = false; break; } // no coarser table to be had: no estimate either
                if (O.stats && tid == 0 && h != h_r) atomicAdd(&O.stats[20], 1ull);
            }
            stamp(3);
            const uint32_t g0 = wave * NTsel;
            float Lr = 0.f, Lg = 0.f, Lb = 0.f, La = 0.f, b_in = 0.f, b_out = 0.f;

            for (uint32_t seg = 0; seg < nseg; ++seg) {
                const float node0 = (float)(seg * SL) - 2.f; // the segment's first node on the ray's grid
                if (!(seg == 0 && have_kinks)) kink_pass(seg, seg == 0 && !have_sums);
                const float s3_scale = 255.f / (fmaxf(S_all, 1e-30f) * 65536.f);
                for (uint32_t g = wave; g < G; g += DW)
                    lds.s3[g][lane] = (uint8_t)fminf(floorf((float)lds.hist[g][lane] * s3_scale) + 1.f, 255.f);
                __syncthreads(); // the weights are read: their memory becomes the table; the partial sums' memory the staging buffers
                stamp(4);

                // ---- table: wave w evaluates the nodes [w NT, (w+1) NT) of the segment against all survivors ----
                const float s_seg = __builtin_fmaf(node0, h, lo); // this lane's position of the segment's node 0
                switch (NTsel) {
                case 4: table_nodes<EXP, ERF, 4, DW>(S, lds, cnt, ray, s_seg, h, g0, wave, lane, n_skip, O.stats ? O.stats + 22 : nullptr); break;
                case 6: table_nodes<EXP, ERF, 6, DW>(S, lds, cnt, ray, s_seg, h, g0, wave, lane, n_skip, O.stats ? O.stats + 22 : nullptr); break;
                case 8: table_nodes<EXP, ERF, 8, DW>(S, lds, cnt, ray, s_seg, h, g0, wave, lane, n_skip, O.stats ? O.stats + 22 : nullptr); break;
                case 12: table_nodes<EXP, ERF, 12, DW>(S, lds, cnt, ray, s_seg, h, g0, wave, lane, n_skip, O.stats ? O.stats + 22 : nullptr); break;
                case 16: table_nodes<EXP, ERF, 16, DW>(S, lds, cnt, ray, s_seg, h, g0, wave, lane, n_skip, O.stats ? O.stats + 22 : nullptr); break;
                case 20: table_nodes<EXP, ERF, 20, DW>(S, lds, cnt, ray, s_seg, h, g0, wave, lane, n_skip, O.stats ? O.stats + 22 : nullptr); break;
                default: table_nodes<EXP, ERF, 24, DW>(S, lds, cnt, ray, s_seg, h, g0, wave, lane, n_skip, O.stats ? O.stats + 22 : nullptr); break;
                }
                __syncthreads();
                stamp(5);

                // ---- emission: the emitters are dealt to the waves; X(s_ik) by 4-point Lagrange interpolation for the samples
                //      of this segment; the error bound is accumulated beside the radiance ----
                const float seg_lo = (float)(seg * SL), seg_hi = seg + 1 == nseg ? INFINITY : (float)((seg + 1) * SL);
                // (emitter i of wave w: w, w + 16, ...; the five rows of the NEXT emitter are requested before this one's samples)
                struct Rows { float4 a, ms, alb; float inv2s2, q; } nx = {};
                auto fetch_rows = [&](uint32_t i) {
                    const uint32_t idx = __builtin_amdgcn_readfirstlane(lds.idx[i]);
                    nx.a = uload(S.gA, idx); nx.ms = uload(S.mu_sig, idx); nx.alb = uload(S.gC, idx);
                    nx.inv2s2 = uload(S.gB, idx).y; nx.q = uload(S.gD, idx).y;
                };
                if (wave < cnt) fetch_rows(wave);
                for (uint32_t i = wave; i < cnt; i += DW) {
                    {
                        const Rows cur = nx;
                        if (i + DW < cnt) fetch_rows(i + DW);
                        const float4 a = cur.a, ms = cur.ms, alb = cur.alb;
                        const float inv2s2 = cur.inv2s2, q = cur.q;
                        const float e_mubar = dot3_ref(a.x, a.y, a.z, ray.nx, ray.ny, ray.nz);
                        if (nseg > 1u) { // an emitter none of whose samples lies in this segment on any ray adds exact zeros: skipped
                            const float g_a = floorf((madd_ref(-4.f, ms.w, e_mubar) - lo) * inv_h), g_b = floorf((e_mubar - lo) * inv_h);
                            const float gi_a = fminf(fmaxf(g_a, 2.f), (float)(Gtot - 4)), gi_b = fminf(fmaxf(g_b, 2.f), (float)(Gtot - 4));
                            if (!__any(gi_b >= seg_lo && gi_a < seg_hi)) continue;
                        }
                        float inner = 0.f, inner_abs = 0.f, inner_s3 = 0.f;
#pragma unroll
                        for (int k = 0; k < 5; ++k) {
                            const float sk = madd_ref((float)(k - 4), ms.w, e_mubar);
                            const float uu = (sk - lo) * inv_h;
                            const float gi = fminf(fmaxf(floorf(uu), 2.f), (float)(Gtot - 4)); // interval on the ray's grid
                            const bool mine = nseg == 1 || (gi >= seg_lo && gi < seg_hi);
                            const float gfl = mine ? gi - node0 : 2.f; // ... and in the table
                            const float t = uu - (gfl + node0);
                            const uint32_t g = (uint32_t)gfl;
                            const float px = sub_ref(madd_ref(ray.nx, sk, ray.ox), ms.x);
                            const float py = sub_ref(madd_ref(ray.ny, sk, ray.oy), ms.y);
                            const float pz = sub_ref(madd_ref(ray.nz, sk, ray.oz), ms.z);
                            const float dd = dot3_ref(px, py, pz, px, py, pz);
                            const float tm1 = t - 1.f, tm2 = t - 2.f, tp1 = t + 1.f;
                            const float w0 = t * tm1 * tm2 * (-1.f / 6.f), w1 = tp1 * tm1 * tm2 * 0.5f;
                            const float w2 = tp1 * t * tm2 * -0.5f, w3 = tp1 * t * tm1 * (1.f / 6.f);
                            const float X = w0 * lds.tab[g - 1][lane] + w1 * lds.tab[g][lane] + w2 * lds.tab[g + 1][lane] + w3 * lds.tab[g + 2][lane];
                            const float term = mine ? emission_term<EXP>(q, dd * inv2s2, X) : 0.f;
                            inner += term;
                            inner_abs += fabsf(term);
                            inner_s3 = __builtin_fmaf(fabsf(term), (float)lds.s3[g][lane], inner_s3);
                        }
                        Lr = __builtin_fmaf(alb.x, inner, Lr);
                        Lg = __builtin_fmaf(alb.y, inner, Lg);
                        Lb = __builtin_fmaf(alb.z, inner, Lb);
                        La = __builtin_fmaf(alb.w, inner, La);
                        const float amax = fmaxf(fmaxf(fabsf(alb.x), fabsf(alb.y)), fmaxf(fabsf(alb.z), fabsf(alb.w)));
                        b_in = __builtin_fmaf(amax, inner_s3, b_in);
                        b_out = __builtin_fmaf(amax, inner_abs, b_out);
                    }
                }
                __syncthreads(); // nobody reads the staging buffers or the table any more
                stamp(6);
            }
            lds.L[wave][lane] = make_float4(Lr, Lg, Lb, La);
            float2 *bparts = reinterpret_cast<float2 *>(&lds.tab[0][0]); // [DW][64]
            bparts[wave * 64 + lane] = make_float2(b_in, b_out);
            __syncthreads();
            if (wave == 0) {
                float4 sum = lds.L[0][lane];
                float2 bs = bparts[lane];
#pragma unroll
                for (int w = 1; w < DW; ++w) {
                    const float4 v = lds.L[w][lane];
                    sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
                    const float2 bv = bparts[w * 64 + lane];
                    bs.x += bv.x; bs.y += bv.y;
                }
                // worst-case change of this ray's radiance (header comment); e^dX - 1 <= 1.01 dX for the dX in question
                const float e_in = TB_W0 * u * u, e_out = TB_COUT * (u * u) * (u * u) + table_saturation_eps<ERF>();
                const float bound = 1.01f * S_all * (e_in * (1.f / 255.f) * bs.x + e_out * bs.y);
                // (false for NaN; S_all beyond the fixed-point range of the weights: no bound)
                const bool good = !valid || (bound <= C.table_budget && S_all < 60.f);
                const bool all_good = __all(good);
                if (all_good && valid) {
                    if (O.image) O.image[out] = pack_pixel(sum.x, sum.y, sum.z, sum.w, O.pack_flags);
                    if (O.radiance) O.radiance[out] = sum;
                }
                if (lane == 0) s_flag = all_good ? 1u : 0u;
            }
            __syncthreads();
            stamp(7);
            done = s_flag != 0u;
            if (!done) {
                if (attempt >= 1) { ok = false; break; }
                h_target = 0.6f * h;
            } else if (O.stats && tid == 0) {
                atomicAdd(&O.stats[0], (unsigned long long)cnt);
                atomicAdd(&O.stats[1], (unsigned long long)n_list);
                atomicAdd(&O.stats[6], 1ull);
                atomicAdd(&O.stats[7], 1ull);
                atomicAdd(&O.stats[16], (unsigned long long)Gtot);
                if (attempt) atomicAdd(&O.stats[17], 1ull);
            }
        }
        if (!ok) { // wave-uniform and the same in every wave
            // Declined: the block is shaded EXACTLY by this very workgroup, with the exact dense kernel's body in the table's LDS (round 4;
            // rounds 1-3 queued it for an exact launch behind this one -- a launch every frame of a moving camera paid although nothing
            // was ever in that queue).  Which arithmetic shades a block is still a function of the block alone.
            if (O.stats && tid == 0) atomicAdd(&O.stats[19], 1ull);
            __syncthreads(); // everyone is done with the table's LDS
            static_assert(sizeof(DenseLds<TB_DW>) <= sizeof(TableLds<DW>), "the exact fallback works in the table's LDS");
            dense_shade_block<EXP, ERF, 6, TB_DW, true>(S, T, C, R, O, *reinterpret_cast<DenseLds<TB_DW> *>(&lds),
                                                        C.scratch + (size_t)blockIdx.x * C.cstride, cell, bi, false, visits);
            continue;
        }
    }
    if (O.stats && lane == 0) {
        atomicAdd(&O.stats[18], (unsigned long long)n_skip);
        atomicAdd(&O.stats[13], (unsigned long long)visits.full); atomicAdd(&O.stats[14], (unsigned long long)visits.zero);
        atomicAdd(&O.stats[15], (unsigned long long)visits.common);
    }
}

// (second launch-bounds argument: waves per SIMD the register budget must allow -- 4 = 128 VGPRs, one 16-wave or two 8-wave workgroups per CU)
template <int EXP, int ERF, int DW>
__global__ __launch_bounds__(DW * 64, 4) void render_table_kernel(RenderArgs) // read through kernel_args<>: vrt_kernels_common.hpp
{
    const RenderArgs &a = kernel_args<RenderArgs>();
    render_table_body<EXP, ERF, DW>(a.S, a.T, a.C, a.R, a.O);
}
// several frames per launch: blockIdx.y is the frame
template <int EXP, int ERF, int DW>
__global__ __launch_bounds__(DW * 64, 4) void render_table_batch_kernel(const FrameArgs *__restrict__ frames)
{
    const FrameArgs &a = frames[blockIdx.y];
    render_table_body<EXP, ERF, DW>(a.S, a.T, a.C, a.R, a.O);
}

// Waves per block: 8 -- two workgroups per CU -- for frames of at least 2^22 rays (2048^2), 16 for smaller ones.  What decides between
// them is a block's depth: the 8-wave table holds 184 intervals, so the teapot's blocks (two thin walls far apart: ~260 nodes) need two
// segments there -- `-f teapot.obj -w 1024` 2.53 against 1.54 ms, at 2048^2 4.57 against 4.73 (-i 90: 6.95 against 6.13) -- while the
// monkey's (~100-150 nodes) fit: 1.89 against 2.27 ms at 1024^2, 6.25 against 7.77 at 2048^2 from the far side, 21.2 against 22.4 at
// 4096^2.  The two variants sum in different orders, so the choice must not depend on anything a batch, a shard or a rank sees
// differently; a block-wise choice would need both shapes in one launch.  It is a function of the FRAME's size alone:
// small frames are few blocks, and 16 waves finish a block in half the time.  VRT_HIP_TABLE_WAVES = 8 | 16 (read once) forces one variant.
static int table_waves(uint64_t npix)
{
    static const int forced = [] { const char *e = getenv("VRT_HIP_TABLE_WAVES"); const int v = e ? atoi(e) : 0; return v == 8 || v == 16 ? v : 0; }();
    return forced ? forced : (npix >= (1ull << 22) ? 8 : 16);
}

// The table kernel's error bound is that of the Abramowitz-Stegun erf (its kink) or of a smoother one (libm); the Exp must
// be an accurate one (Exp(a)Exp(b) = Exp(a + b)): four pairs are instantiated, the host keeps every other pair exact.
#define VRT_DISPATCH_TABLE_W(FN, E, F, ...)                                                        \
    if (table_waves(npix) == 8) FN<E, F, 8>(__VA_ARGS__);                                          \
    else FN<E, F, 16>(__VA_ARGS__)
#define VRT_DISPATCH_TABLE(FN, ...)                                                                \
    switch (exp_kind * 8 + erf_kind) {                                                             \
    case VRT_EXP_LIBM * 8 + VRT_ERF_LIBM: VRT_DISPATCH_TABLE_W(FN, VRT_EXP_LIBM, VRT_ERF_LIBM, __VA_ARGS__); break; \
    case VRT_EXP_LIBM * 8 + VRT_ERF_AS: VRT_DISPATCH_TABLE_W(FN, VRT_EXP_LIBM, VRT_ERF_AS, __VA_ARGS__); break;     \
    case VRT_EXP_VCL * 8 + VRT_ERF_LIBM: VRT_DISPATCH_TABLE_W(FN, VRT_EXP_VCL, VRT_ERF_LIBM, __VA_ARGS__); break;   \
    default: VRT_DISPATCH_TABLE_W(FN, VRT_EXP_VCL, VRT_ERF_AS, __VA_ARGS__); break;                \
    }
// `grid` = the number of 16-wave workgroups the host wants (at most one per CU)
template <int EXP, int ERF, int DW>
static void launch_render_table_t(const SceneTables &s, const TileLists &t, const CellGrid &c, const RayGen &r,
                                  const RenderTarget &o, uint32_t grid, hipStream_t st)
{
    if (grid == 0) return;
    hipLaunchKernelGGL((render_table_kernel<EXP, ERF, DW>), dim3(grid * (16 / DW)), dim3(DW * 64), 0, st, RenderArgs{ s, t, c, r, o });
}
void launch_render_table(const SceneTables &s, const TileLists &t, const CellGrid &c, const RayGen &r,
                         const RenderTarget &o, uint32_t grid, int exp_kind, int erf_kind, hipStream_t st)
{
    const uint64_t npix = (uint64_t)r.width * r.height;
    VRT_DISPATCH_TABLE(launch_render_table_t, s, t, c, r, o, grid, st);
}
template <int EXP, int ERF, int DW>
static void launch_render_table_only_batch_t(const FrameArgs *d_frames, uint32_t nframes, uint32_t grid, hipStream_t st)
{
    if (nframes && grid) hipLaunchKernelGGL((render_table_batch_kernel<EXP, ERF, DW>), dim3(grid * (16 / DW), nframes), dim3(DW * 64), 0, st, d_frames);
}
void launch_render_table_batch(const FrameArgs *d_frames, uint32_t nframes, uint32_t grid, uint64_t npix, int exp_kind, int erf_kind, hipStream_t st)
{
    VRT_DISPATCH_TABLE(launch_render_table_only_batch_t, d_frames, nframes, grid, st);
}

} // namespace vrtk
