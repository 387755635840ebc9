// vrt_kernels.hip -- exact dense kernel, list kernels (tile binning rt.cpp:29-69 + cell lists), scene tables, frame assembly
// (multi-GPU) and the point queries.  The block kernel and the table kernel live in vrt_block_kernel.hip / vrt_table_kernel.hip;
// what all three share is in vrt_kernels_common.hpp.
#include "vrt_dense_block.hpp"

namespace vrtk {

// The exact dense kernel: persistent DW-wave workgroups pull blocks (the cells of the dense queue, then what the block kernel handed
// over) with one atomic per block and shade them with dense_shade_block (vrt_dense_block.hpp).
template <int EXP, int ERF, int EC, int DW, bool SKIP = true>
__device__ __forceinline__ void render_dense_body(const SceneTables &S, const TileLists &T, const CellGrid &C, const RayGen &R,
                                                  const RenderTarget &O)
{
    __shared__ DenseLds<DW> lds;
    __shared__ uint32_t s_item;
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t n_dense16 = *C.n_dense * 16u, n_items = n_dense16 + *C.n_overflow;
    if (C.feedback && blockIdx.x == 0 && tid == 0) { // launch feedback: how much this frame had for this kernel
        __hip_atomic_store(&C.feedback[2], n_items, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&C.feedback[3], C.frame_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const uint32_t *dense_queue = C.dense_is_sorted ? C.dense_sorted : C.dense;
    uint32_t *scratch = C.scratch + (size_t)blockIdx.x * C.cstride;
    const unsigned long long t_start = O.stats ? wall_clock64() : 0ull;
    DenseVisits visits;

    for (;;) {
        __syncthreads(); // everyone is done with the previous item's LDS
        if (tid == 0) s_item = atomicAdd(C.dense_next, 1u);
        __syncthreads();
        const uint32_t item = s_item;
        if (item >= n_items) {
            if (O.stats && lane == 0) {
                atomicAdd(&O.stats[13], (unsigned long long)visits.full); atomicAdd(&O.stats[14], (unsigned long long)visits.zero);
                atomicAdd(&O.stats[15], (unsigned long long)visits.common);
            }
            if (O.stats && tid == 0) { // workgroup timeline: how long the queue kept this workgroup busy
                const unsigned long long t_end = wall_clock64();
                atomicMin(&O.stats[8], t_start); atomicMax(&O.stats[9], t_end);
                atomicAdd(&O.stats[10], t_end - t_start); atomicAdd(&O.stats[11], 1ull);
            }
            break;
        }
        uint32_t cell, bi;
        if (item < n_dense16) { cell = dense_queue[item >> 4]; bi = item & 15u; }
        else { const uint32_t packed = C.overflow[item - n_dense16]; cell = packed >> 4; bi = packed & 15u; }
        dense_shade_block<EXP, ERF, EC, DW, SKIP>(S, T, C, R, O, lds, scratch, cell, bi, item < n_dense16, visits);
    }
}
template <int EXP, int ERF, int EC, int DW, bool SKIP = true>
__global__ __launch_bounds__(DW * 64, 4) void render_dense_kernel(RenderArgs) // read through kernel_args<>: vrt_kernels_common.hpp
{
    const RenderArgs &a = kernel_args<RenderArgs>();
    render_dense_body<EXP, ERF, EC, DW, SKIP>(a.S, a.T, a.C, a.R, a.O);
}
template <int EXP, int ERF, int EC, int DW, bool SKIP = true>
__global__ __launch_bounds__(DW * 64, 4) void render_dense_batch_kernel(const FrameArgs *__restrict__ frames)
{
    const FrameArgs &a = frames[blockIdx.y];
    render_dense_body<EXP, ERF, EC, DW, SKIP>(a.S, a.T, a.C, a.R, a.O);
}

// Queue order of the dense kernel: cells by descending candidate count (a block costs ~ count^2), so that the
// blocks still running when the queue empties are the cheapest ones.  Counting sort, one workgroup.
__device__ __forceinline__ void order_dense_body(const CellGrid &C)
{
    __shared__ uint32_t s_hist[1024], s_scan[1024];
    const uint32_t n = *C.n_dense, tid = threadIdx.x;
    s_hist[tid] = 0;
    __syncthreads();
    for (uint32_t i = tid; i < n; i += 1024) atomicAdd(&s_hist[1023u - (min(C.count[C.dense[i]], 4095u) >> 2)], 1u);
    __syncthreads();
    // exclusive prefix over the buckets (bucket 0 = longest lists)
    uint32_t v = s_hist[tid];
    s_scan[tid] = v;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
        const uint32_t add = tid >= off ? s_scan[tid - off] : 0u;
        __syncthreads();
        s_scan[tid] += add;
        __syncthreads();
    }
    s_hist[tid] = s_scan[tid] - v;
    __syncthreads();
    for (uint32_t i = tid; i < n; i += 1024) {
        const uint32_t cell = C.dense[i];
        C.dense_sorted[atomicAdd(&s_hist[1023u - (min(C.count[cell], 4095u) >> 2)], 1u)] = cell;
    }
}

__global__ __launch_bounds__(1024) void order_dense_kernel(CellGrid C) { order_dense_body(C); }
__global__ __launch_bounds__(1024) void order_dense_batch_kernel(const FrameArgs *__restrict__ frames)
{
    const FrameArgs &a = frames[blockIdx.x];
    if (a.do_order) order_dense_body(a.C);
}
void launch_order_dense(const CellGrid &c, hipStream_t st)
{
    hipLaunchKernelGGL(order_dense_kernel, dim3(1), dim3(1024), 0, st, c);
}
void launch_order_dense_batch(const FrameArgs *d_frames, const FrameArgs *h_frames, uint32_t nframes, hipStream_t st)
{
    bool any = false;
    for (uint32_t f = 0; f < nframes; ++f) any = any || h_frames[f].do_order;
    if (any) hipLaunchKernelGGL(order_dense_batch_kernel, dim3(nframes), dim3(1024), 0, st, d_frames);
}



template <int EXP, int ERF>
static void launch_render_dense_t(const SceneTables &s, const TileLists &t, const CellGrid &c, const RayGen &r,
                                  const RenderTarget &o, uint32_t grid, int dw, hipStream_t st)
{
    if (grid == 0) return;
    if (dw == 17) hipLaunchKernelGGL((render_dense_kernel<EXP, ERF, 6, 16, false>), dim3(grid), dim3(1024), 0, st, RenderArgs{ s, t, c, r, o });
    else if (dw == 16) hipLaunchKernelGGL((render_dense_kernel<EXP, ERF, 6, 16>), dim3(grid), dim3(1024), 0, st, RenderArgs{ s, t, c, r, o });
    else if (dw == 8) hipLaunchKernelGGL((render_dense_kernel<EXP, ERF, 6, 8>), dim3(grid), dim3(512), 0, st, RenderArgs{ s, t, c, r, o });
    else hipLaunchKernelGGL((render_dense_kernel<EXP, ERF, 6, 4>), dim3(grid), dim3(256), 0, st, RenderArgs{ s, t, c, r, o });
}
void launch_render_dense(const SceneTables &s, const TileLists &t, const CellGrid &c, const RayGen &r,
                         const RenderTarget &o, uint32_t grid, int dw, int exp_kind, int erf_kind, hipStream_t st)
{
    VRT_DISPATCH_EXP_ERF(launch_render_dense_t, s, t, c, r, o, grid, dw, st);
}
template <int EXP, int ERF>
static void launch_render_dense_batch_t(const FrameArgs *d_frames, uint32_t nframes, uint32_t grid, int dw, hipStream_t st)
{
    if (grid == 0 || nframes == 0) return;
    const dim3 g(grid, nframes);
    if (dw == 17) hipLaunchKernelGGL((render_dense_batch_kernel<EXP, ERF, 6, 16, false>), g, dim3(1024), 0, st, d_frames);
    else if (dw == 16) hipLaunchKernelGGL((render_dense_batch_kernel<EXP, ERF, 6, 16>), g, dim3(1024), 0, st, d_frames);
    else if (dw == 8) hipLaunchKernelGGL((render_dense_batch_kernel<EXP, ERF, 6, 8>), g, dim3(512), 0, st, d_frames);
    else hipLaunchKernelGGL((render_dense_batch_kernel<EXP, ERF, 6, 4>), g, dim3(256), 0, st, d_frames);
}
void launch_render_dense_batch(const FrameArgs *d_frames, uint32_t nframes, uint32_t grid, int dw, int exp_kind, int erf_kind,
                               hipStream_t st)
{
    VRT_DISPATCH_EXP_ERF(launch_render_dense_batch_t, d_frames, nframes, grid, dw, st);
}

// ---------------------------------------------------------------------------------------------
// Scene tables
// ---------------------------------------------------------------------------------------------
__global__ void build_static_kernel(uint32_t n, const float *mu_x, const float *mu_y, const float *mu_z,
                                    const float *ar, const float *ag, const float *ab, const float *aa,
                                    const float *sigma, const float *mag, float cull_eps, float exp_floor_x,
                                    float4 *mu_sig, float4 *gB, float4 *gC, float4 *gD)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float s = sigma[i], m = mag[i];
    mu_sig[i] = make_float4(mu_x[i], mu_y[i], mu_z[i], s);
    const float q = s * m;
    // cull_x: drop when d^2/(2 sigma^2) > ln(|q|/eps); never keep what Exp flushes to zero anyway
    float cull_x = exp_floor_x;
    if (q == 0.f) cull_x = -INFINITY;
    else if (cull_eps > 0.f) cull_x = fminf(cull_x, logf(fabsf(q) / cull_eps));
    gB[i] = make_float4(1.f / (SQRT_2 * s), 1.f / (2.f * s * s), q * INV_SQRT_2_PI, cull_x);
    gC[i] = make_float4(ar[i], ag[i], ab[i], aa ? aa[i] : 1.f);
    gD[i] = make_float4(s, q, m, 0.f);
}

void launch_build_static(uint32_t n, const float *mu_x, const float *mu_y, const float *mu_z, const float *ar,
                         const float *ag, const float *ab, const float *aa, const float *sigma, const float *mag,
                         float cull_eps, float exp_floor_x, float4 *mu_sig, float4 *gB, float4 *gC, float4 *gD,
                         hipStream_t st)
{
    if (!n) return;
    hipLaunchKernelGGL(build_static_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, mu_x, mu_y, mu_z, ar, ag, ab,
                       aa, sigma, mag, cull_eps, exp_floor_x, mu_sig, gB, gC, gD);
}

__global__ void prep_frame_kernel(uint32_t n, const float4 *mu_sig, float4 *gA, float ox, float oy, float oz)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 m = mu_sig[i];
    const float cx = m.x - ox, cy = m.y - oy, cz = m.z - oz;
    gA[i] = make_float4(cx, cy, cz, dot3_ref(cx, cy, cz, cx, cy, cz)); // vec4f_t::sqnorm order (types.h:69-72)
}

__global__ void prep_frame_batch_kernel(const FrameArgs *__restrict__ frames)
{
    const FrameArgs &a = frames[blockIdx.y];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (!a.do_prep || i >= a.S.n) return;
    const float4 m = a.S.mu_sig[i];
    const float cx = m.x - a.prep_origin[0], cy = m.y - a.prep_origin[1], cz = m.z - a.prep_origin[2];
    a.prep_gA[i] = make_float4(cx, cy, cz, dot3_ref(cx, cy, cz, cx, cy, cz));
}
void launch_prep_frame(const SceneTables &s, float4 *gA_out, const float origin[3], hipStream_t st)
{
    if (!s.n) return;
    hipLaunchKernelGGL(prep_frame_kernel, dim3((s.n + 255) / 256), dim3(256), 0, st, s.n, s.mu_sig, gA_out, origin[0],
                       origin[1], origin[2]);
}

__global__ void iota_kernel(uint32_t *p, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = i;
}
void launch_iota(uint32_t *p, uint32_t n, hipStream_t st)
{
    if (n) hipLaunchKernelGGL(iota_kernel, dim3((n + 255) / 256), dim3(256), 0, st, p, n);
}

// ---------------------------------------------------------------------------------------------
// Per-tile Gaussian lists.  One 1024-thread workgroup per reference tile builds, in one pass,
//   (a) the reference's tile set: vrt/rt.cpp:29-69 on device (or takes a caller-made list), and
//   (b) optionally ("refine") drops from it every Gaussian that is below cull_eps for the whole tile
//       (cone through the tile's corner rays), so that the 8x8-block cull of the render kernel
//       scans tens of candidates instead of the reference's ~1600.
// The reference-set arithmetic is kept unfused and in the reference's order so that inclusion
// decisions (a "<=" on floats) reproduce the host algorithm; order of indices is preserved.
// ---------------------------------------------------------------------------------------------
// With F.enabled the same workgroup goes on to the second level: the tile's surviving candidates stay in LDS
// (index + the two parameter rows the cone test reads) and each of its 16 waves filters them for the tile's
// 32x32-pixel cells, files every non-empty cell as active or dense (at most two atomics per TILE) and clears the
// pixels of empty cells on the spot -- no second kernel, no global round trip, no idle clear phase later.
// cone of a whole reference tile, from the centre and the four corner pixels (pinhole rays: the farthest ray of a rectangle
// on the image plane from its centre ray is a corner ray)
__device__ __forceinline__ Cone tile_cone(const BinArgs &P, uint32_t tx, uint32_t ty, uint32_t lane)
{
    const uint64_t npix0 = (uint64_t)P.R.width * P.R.height;
    auto at = [&](uint32_t x, uint32_t y) {
        uint64_t pix = (uint64_t)(tx * P.tile_w + x) + (uint64_t)P.stride * (ty * P.tile_h + y);
        if (pix >= npix0) pix = npix0 - 1;
        return cone_ray(P.R, pix);
    };
    return rect_cone(at, 0, 0, P.tile_w - 1, P.tile_h - 1, lane);
}
// cone of cell ci of a tile (32x32 px, clipped to the tile), as the second level builds it
__device__ __forceinline__ Cone cell_cone(const BinArgs &P, uint32_t tx, uint32_t ty, uint32_t ci, uint32_t cells_x, uint32_t lane)
{
    const uint64_t npix = (uint64_t)P.R.width * P.R.height;
    const uint32_t x0 = (ci % cells_x) * CELL, y0 = (ci / cells_x) * CELL;
    const uint32_t x1 = min(x0 + CELL, P.tile_w) - 1, y1 = min(y0 + CELL, P.tile_h) - 1;
    auto at = [&](uint32_t x, uint32_t y) {
        uint64_t pix = (uint64_t)(tx * P.tile_w + x) + (uint64_t)P.stride * (ty * P.tile_h + y);
        if (pix >= npix) pix = npix - 1;
        return cone_ray(P.R, pix);
    };
    return rect_cone(at, x0, y0, x1, y1, lane);
}
// A row of the cone table: (axis, tag), (cos, sin, tag, -).  The tag (BinArgs::cone_gen) says which camera the row was made for; it sits in
// BOTH halves, so a reader that catches a row between the writer's two stores sees two different tags and takes the row for missing.
__device__ __forceinline__ void cone_row_write(float4 *table, size_t row, const Cone &k, uint32_t gen)
{
    table[2 * row] = make_float4(k.cx, k.cy, k.cz, __uint_as_float(gen));
    table[2 * row + 1] = make_float4(k.cos_t, k.sin_t, __uint_as_float(gen), 0.f);
}
__device__ __forceinline__ bool cone_row_read(const float4 *table, size_t row, uint32_t gen, Cone &k)
{
    const float4 c0 = table[2 * row], c1 = table[2 * row + 1];
    if (__float_as_uint(c0.w) != gen || __float_as_uint(c1.z) != gen) return false;
    k.cx = c0.x; k.cy = c0.y; k.cz = c0.z; k.cos_t = c1.x; k.sin_t = c1.y;
    return true;
}
// One wave per cone: per tile id its own cone (slot 0) and the cones of its cells (slots 1 .. cells per tile) -- what the
// list kernel's workgroups build (and file) themselves when they do not find them; frames of a batch get them from this one launch.
__global__ __launch_bounds__(256) void tile_cones_batch_kernel(const FrameArgs *__restrict__ frames)
{
    const FrameArgs &a = frames[blockIdx.y];
    if (!a.do_cones) return;
    const BinArgs &P = a.bin;
    const uint32_t per = 1 + a.cones_cx * a.cones_cy;
    const uint32_t k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (k >= a.cones_tiles * per) return;
    const uint32_t t = k / per, c = k % per;
    const Cone cn = c ? cell_cone(P, t % P.tiles_w, t / P.tiles_w, c - 1, a.cones_cx, lane) : tile_cone(P, t % P.tiles_w, t / P.tiles_w, lane);
    if (lane == 0) cone_row_write(a.cones_out, k, cn, P.cone_gen);
}
void launch_frame_setup_batch(const FrameArgs *d_frames, const FrameArgs *h_frames, uint32_t nframes, hipStream_t st)
{
    uint32_t n_prep = 0, n_cones = 0;
    for (uint32_t f = 0; f < nframes; ++f) {
        if (h_frames[f].do_prep) n_prep = std::max(n_prep, h_frames[f].S.n);
        if (h_frames[f].do_cones) n_cones = std::max(n_cones, h_frames[f].cones_tiles * (1 + h_frames[f].cones_cx * h_frames[f].cones_cy));
    }
    if (n_prep) hipLaunchKernelGGL(prep_frame_batch_kernel, dim3((n_prep + 255) / 256, nframes), dim3(256), 0, st, d_frames);
    if (n_cones) hipLaunchKernelGGL(tile_cones_batch_kernel, dim3((n_cones + 3) / 4, nframes), dim3(256), 0, st, d_frames);
}

// ---------------------------------------------------------------------------------------------
// Chunk table (round 3): every 64 consecutive Gaussians get a bounding sphere whose radius includes the members' reach -- the distance
// beyond which cone_keeps drops them: x = d^2 / (2 sigma^2) with 0.999 x - 1e-3 > cull_x.  cone_keeps' lower bound of the distance
// between a point and the rays of a cone is 1-Lipschitz in the point, so a cone farther than the radius from the chunk's centre keeps
// none of its members: the tile level then tests N / 64 spheres and the members of the chunks that are left instead of all N Gaussians
// (256 tiles x 4096 x 32 B = 33 MB of L2 reads per `-g 64 -w 2048` frame before: the level ran at L2 bandwidth).  Index order is kept (chunks in
// order, members in order), so the lists are the same lists.  Depends on the scene and cull_eps only: built with the static tables.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void build_chunks_kernel(uint32_t n, const float4 *__restrict__ mu_sig, const float4 *__restrict__ gB, float4 *__restrict__ chunks)
{
    const uint32_t lane = threadIdx.x & 63u, chunk = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (chunk * 64u >= n) return;
    const uint32_t i = chunk * 64u + lane;
    const bool valid = i < n;
    const float4 p = valid ? mu_sig[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 bq = valid ? gB[i] : make_float4(0.f, 1.f, 0.f, -INFINITY);
    const float lo_x = wave_min(valid ? p.x : INFINITY), hi_x = wave_max(valid ? p.x : -INFINITY);
    const float lo_y = wave_min(valid ? p.y : INFINITY), hi_y = wave_max(valid ? p.y : -INFINITY);
    const float lo_z = wave_min(valid ? p.z : INFINITY), hi_z = wave_max(valid ? p.z : -INFINITY);
    const float mx = 0.5f * (lo_x + hi_x), my = 0.5f * (lo_y + hi_y), mz = 0.5f * (lo_z + hi_z);
    const float dx = p.x - mx, dy = p.y - my, dz = p.z - mz;
    const float xr = bq.w + 1e-3f; // kept iff 0.999 d^2 bq.y - 1e-3 <= cull_x
    const float reach = xr > 0.f ? sqrtf(xr / (0.999f * bq.y)) : 0.f;
    float rho = wave_max(valid ? sqrtf(dx * dx + dy * dy + dz * dz) + reach : 0.f);
    rho = rho * 1.0001f + 1e-6f * (1.f + fabsf(mx) + fabsf(my) + fabsf(mz));
    if (lane == 0) chunks[chunk] = make_float4(mx, my, mz, rho);
}
void launch_build_chunks(uint32_t n, const float4 *mu_sig, const float4 *gB, float4 *chunks, hipStream_t st)
{
    const uint32_t nch = (n + 63u) / 64u;
    if (nch) hipLaunchKernelGGL(build_chunks_kernel, dim3((nch + 3u) / 4u), dim3(256), 0, st, n, mu_sig, gB, chunks);
}
__device__ __forceinline__ bool chunk_keeps(const Cone &k, float4 ch, float ox, float oy, float oz)
{
    const float ax = ch.x - ox, ay = ch.y - oy, az = ch.z - oz;
    const float d2 = ax * ax + ay * ay + az * az;
    const float tc = ax * k.cx + ay * k.cy + az * k.cz;
    const float dperp = __builtin_amdgcn_sqrtf(fmaxf(0.f, d2 - tc * tc - 8e-6f * d2)); // the cancellation's rounding, on the keeping side
    const float dmin = fmaxf(0.f, dperp * k.cos_t - fabsf(tc) * k.sin_t);
    return !(dmin * 0.9999f > ch.w);
}

template <bool FROM_LIST, bool CHUNKS>
__device__ __forceinline__ void build_tile_lists_body(const BinArgs &P, const FuseArgs &F)
{
    static_assert(!(FROM_LIST && CHUNKS), "chunks are runs of consecutive indices: device binning only");
    __shared__ uint32_t s_wave_cnt[64];
    __shared__ uint32_t s_idx[TCAP];
    __shared__ float4 s_A[TCAP], s_B[TCAP];
    __shared__ uint32_t s_flag[MAX_FUSED_CELLS], s_inact[MAX_FUSED_CELLS];
    __shared__ uint32_t s_base[4];
    __shared__ uint32_t s_chunk[CHUNKS ? CH_CAP : 1];
    const uint32_t lt = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t t = F.tile_map ? F.tile_map[lt] : lt;
    const uint32_t tx = t % P.tiles_w, ty = t / P.tiles_w;
    // the queue counters are cleared by the PREVIOUS frame's last kernel or a memset when fusing (this kernel
    // adds to them); the unfused pipeline clears them here for the cell kernel that follows
    if (P.zero8 && !F.enabled && lt == 0 && tid < 8) P.zero8[tid] = 0;
    if (P.next_zero8 && lt == 0 && tid < 8) P.next_zero8[tid] = 0;
    unsigned long long *tl = (F.timeline && tid == 0) ? F.timeline + 8 * (size_t)lt : nullptr;
    if (tl) tl[0] = wall_clock64();
    const float org_x = P.R.origin[0], org_y = P.R.origin[1], org_z = P.R.origin[2];
    // the per-origin row of a Gaussian: centre - origin and its squared norm in vec4f_t::sqnorm order (types.h:69-72) -- prep_frame_kernel's arithmetic
    auto rel = [&](const float4 &m) {
        const float cx = m.x - org_x, cy = m.y - org_y, cz = m.z - org_z;
        return make_float4(cx, cy, cz, dot3_ref(cx, cy, cz, cx, cy, cz));
    };

    float x = 0.f, y = 0.f, ax = 0.f, ay = 0.f;
    uint32_t n_in;
    const uint32_t *in_list = nullptr;
    if constexpr (FROM_LIST) {
        n_in = P.in_count[t];
        in_list = P.in_indices + P.in_start[t];
    } else {
#pragma clang fp contract(off)
        n_in = P.n;
        x = P.xc[tx]; y = P.yc[ty];
        ax = fabsf(x) + P.tw / 2; ay = fabsf(y) + P.th / 2; // rt.cpp:58-59 (left-to-right sums)
    }
    uint32_t *out = P.out_indices + P.out_start[t];
    uint32_t total = 0;
    // Four candidates per thread and pass: their rows are requested together (one memory round trip instead of four)
    // and the order-preserving compaction needs one barrier pair per 4096 candidates.  Order = index order:
    // (sub-pass u, wave, lane) lexicographic.
    bool keep[4];
    uint32_t idx[4];
    float4 gb[4], gm[4]; // gm: centre and sigma -- the per-origin row is computed from it (rel), and the reference's tile test needs it anyway
    // chunked: the candidates are the members of the chunks that passed the chunk test, one chunk per (sub-pass, wave) -- the same
    // lexicographic (sub-pass, wave, lane) order as below, which is index order again
    bool chunked = CHUNKS && P.refine && P.chunks != nullptr;
    uint32_t n_slots = 0;
    auto fetch = [&](uint32_t base) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (CHUNKS && chunked) {
                const uint32_t slot = base / 64u + u * 16u + wave;
                idx[u] = slot < n_slots ? s_chunk[slot] * 64u + lane : 0xFFFFFFFFu;
                keep[u] = idx[u] < n_in;
                if (!keep[u]) idx[u] = 0u;
            } else {
                const uint32_t k = base + u * 1024 + tid;
                keep[u] = k < n_in;
                idx[u] = keep[u] ? (FROM_LIST ? in_list[k] : k) : 0u;
            }
            if (keep[u] && (P.refine || F.enabled)) gb[u] = P.gB[idx[u]];
            if (keep[u] && (P.refine || F.enabled || !FROM_LIST)) gm[u] = P.mu_sig[idx[u]];
        }
    };
    if (!CHUNKS || !chunked) fetch(0); // in flight while the cone is set up
    // (b) tile cone from the centre and the four corner pixels of the tile (pinhole rays: the
    // farthest ray of a rectangle on the image plane from its centre ray is a corner ray)
    Cone cone = {};
    if (P.refine) {
        bool have = false;
        const size_t row = (size_t)t * (1 + P.cones_cells);
        if (P.tile_cones && P.cones_known) have = cone_row_read(P.tile_cones, row, P.cone_gen, cone); // filed by an earlier frame with this camera (or by tile_cones_kernel)
        if (!have) {
            cone = tile_cone(P, tx, ty, lane);
            if (P.tile_cones && tid == 0) cone_row_write(P.tile_cones, row, cone, P.cone_gen);
        }
    }

    if constexpr (CHUNKS) {
        if (chunked) { // ---- chunk test: one sphere per thread and pass, order-preserving compaction of the chunk ids ----
            const uint32_t nch = (n_in + 63u) / 64u;
            const float ox = P.R.origin[0], oy = P.R.origin[1], oz = P.R.origin[2];
            for (uint32_t cb = 0; cb < nch; cb += 1024) {
                const uint32_t c = cb + tid;
                const bool kc = c < nch && chunk_keeps(cone, P.chunks[c], ox, oy, oz);
                const unsigned long long m = __ballot(kc);
                if (lane == 0) s_wave_cnt[wave] = (uint32_t)__popcll(m);
                __syncthreads();
                const uint32_t v = lane < 16 ? s_wave_cnt[lane] : 0u;
                const uint32_t incl = wave_inclusive_sum(v);
                const uint32_t before = lane_value_u32(incl - v, wave);
                if (kc) {
                    const uint32_t pos = n_slots + before + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
                    if (pos < CH_CAP) s_chunk[pos] = c;
                }
                n_slots += lane_value_u32(incl, 15u);
                __syncthreads();
            }
            if (n_slots > CH_CAP) chunked = false; // (wave-uniform) too many chunks for LDS: every Gaussian is a candidate
            fetch(0);
        }
    }
    if (tl) tl[1] = wall_clock64();
    const uint32_t n_cand = (CHUNKS && chunked) ? n_slots * 64u : n_in;
    for (uint32_t base = 0; base < n_cand; base += 4096) {
        if (base) fetch(base);
        unsigned long long mask[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            // The result is (reference tile test) AND (cone test).  The cone test goes first: it is the cheaper
            // one (no divisions) and drops >95 % of the pairs in sparse scenes.
            if (keep[u] && P.refine) keep[u] = cone_keeps(cone, rel(gm[u]), gb[u]);
            if constexpr (!FROM_LIST) {
                if (keep[u]) {
#pragma clang fp contract(off)
                    keep[u] = false;
                    const float4 g = gm[u];
                    // glm mat4*vec4: (m0*v0 + m1*v1) + (m2*v2 + m3*v3), v = (mu, 1)   (rt.cpp:37)
                    const float vx = (P.V.m[0] * g.x + P.V.m[4] * g.y) + (P.V.m[8] * g.z + P.V.m[12] * 1.f);
                    const float vy = (P.V.m[1] * g.x + P.V.m[5] * g.y) + (P.V.m[9] * g.z + P.V.m[13] * 1.f);
                    const float vz = (P.V.m[2] * g.x + P.V.m[6] * g.y) + (P.V.m[10] * g.z + P.V.m[14] * 1.f);
                    if (!(vz < 1.f)) {                       // rt.cpp:38
                        const float sig = g.w / vz;          // rt.cpp:40
                        if (!(sig < 1e-5f)) {                // rt.cpp:41
                            const float dx = fabsf(x - vx / vz), dy = fabsf(y - vy / vz);
                            const float s33 = 3.3f * sig;
                            keep[u] = (dx <= ax + s33) && (dy <= ay + s33); // rt.cpp:58-59
                        }
                    }
                }
            }
            mask[u] = __ballot(keep[u]);
            if (lane == 0) s_wave_cnt[u * 16 + wave] = (uint32_t)__popcll(mask[u]);
        }
        __syncthreads();
        // exclusive prefix over the 64 (sub-pass, wave) counts: one count per lane, wave-level scan
        const uint32_t v = s_wave_cnt[lane];
        const uint32_t incl = wave_inclusive_sum(v);
        const uint32_t excl = incl - v;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t before = lane_value_u32(excl, (uint32_t)u * 16u + wave);
            if (keep[u]) {
                const uint32_t pos = total + before + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask[u] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask[u], 0));
                out[pos] = idx[u];
                if (F.enabled && pos < TCAP) { s_idx[pos] = idx[u]; s_A[pos] = rel(gm[u]); s_B[pos] = gb[u]; }
            }
        }
        total += lane_value_u32(incl, 63u);
        __syncthreads();
    }
    const float slack = level_slack(P.cull_ref_n, total); // cell level: the tile's work list enters
    if (tid == 0) P.out_count[t] = total;
    if (tl) tl[2] = wall_clock64();
    // the per-origin table for the render kernels of this frame (this kernel reads none of it): behind everything the frame waits for
    auto write_prep = [&]() {
        if (P.prep_gA)
            for (uint32_t i = blockIdx.x * 1024u + tid; i < P.n; i += gridDim.x * 1024u) P.prep_gA[i] = rel(P.mu_sig[i]);
    };
    if (!F.enabled) { write_prep(); return; }

    // ---------------- second level, fused ----------------
    const CellGrid &C = F.C;
    const uint32_t cpt = C.cells_x * C.cells_y;
    const uint64_t npix = (uint64_t)P.R.width * P.R.height;
    for (uint32_t ci = wave; ci < cpt; ci += 16) {
        const uint32_t cell = lt * cpt + ci;
        uint32_t ctotal = 0;
        if (total > TCAP) {
            ctotal = 0xFFFFFFFFu; // the tile's list did not fit LDS: its cells use the tile list itself
        } else if (total) {
            Cone cc = {};
            if (P.refine) {
                bool have = false;
                const size_t row = (size_t)t * (1 + cpt) + 1 + ci;
                const bool tabled = P.tile_cones && P.cones_cells == cpt;
                if (tabled && P.cones_known) have = cone_row_read(P.tile_cones, row, P.cone_gen, cc);
                if (!have) {
                    cc = cell_cone(P, tx, ty, ci, C.cells_x, lane);
                    if (tabled && lane == 0) cone_row_write(P.tile_cones, row, cc, P.cone_gen);
                }
            }
            uint32_t *cout = C.indices + (size_t)cell * C.cstride;
            for (uint32_t base = 0; base < total; base += 64) {
                const uint32_t k = base + lane;
                bool keep = k < total;
                if (keep && P.refine) {
                    float4 bq = s_B[k];
                    bq.w = slack_cull_x(bq.w, slack, P.floor_x);
                    keep = cone_keeps(cc, s_A[k], bq);
                }
                const unsigned long long mask = __ballot(keep);
                const uint32_t pos = ctotal + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
                if (keep && pos < C.cstride) cout[pos] = s_idx[k];
                ctotal += (uint32_t)__popcll(mask);
            }
            if (ctotal > C.cstride) ctotal = 0xFFFFFFFFu;
        }
        if (lane == 0) {
            C.count[cell] = ctotal;
            s_flag[ci] = (ctotal ? (ctotal > C.dense_threshold ? 3u : (ctotal <= C.light_threshold ? 5u : 1u)) : 0u) | (min(ctotal, 255u) << 8);
        }
    }
    __syncthreads();
    if (tl) tl[3] = wall_clock64();
    if (wave == 0) { // cpt <= 64: one flag per lane
        const uint32_t flag = lane < cpt ? s_flag[lane] : 2u;
        const uint32_t mine = flag & 0xFFu, packed_count = (flag >> 8) << ACTIVE_COUNT_SHIFT; // the list length rides in the queue entry
        const unsigned long long m_act = __ballot(mine == 1u), m_dense = __ballot(mine == 3u);
        const unsigned long long m_light = __ballot(mine == 5u);
        // Retained frame buffer (RenderTarget::stamp): the buffer still holds this context's previous frame, so an empty cell
        // needs its 4 KB of background only if it was lit then; a lit cell notes the frame it was lit in.
        bool clear_me = mine == 0u;
        if (F.O.stamp && lane < cpt) {
            uint32_t *stamp = F.O.stamp + (size_t)t * cpt + lane;
            if (mine == 0u) clear_me = *stamp == F.O.stamp_seq - 1u;
            else *stamp = F.O.stamp_seq;
        }
        const unsigned long long m_empty = __ballot(clear_me);
        const uint32_t na = (uint32_t)__popcll(m_act), nd = (uint32_t)__popcll(m_dense), nl = (uint32_t)__popcll(m_light);
        // empty cells are not queued (count == 0 says it): most tiles of a sparse frame then add to no counter at all
        uint32_t base_a = 0, base_d = 0, base_l = 0;
        if (lane == 0) {
            if (na) base_a = atomicAdd(C.n_active, na);
            if (nd) base_d = atomicAdd(C.n_dense, nd);
            if (nl) base_l = atomicAdd(C.n_light, nl);
            s_base[3] = (uint32_t)__popcll(m_empty);
        }
        base_a = lane_value_u32(base_a, 0u); base_d = lane_value_u32(base_d, 0u);
        base_l = lane_value_u32(base_l, 0u);
        const unsigned long long below = (1ull << lane) - 1ull;
        const uint32_t cell = lt * cpt + lane;
        if (mine == 1u) {
            const uint32_t pos = base_a + (uint32_t)__popcll(m_act & below);
            C.active[pos] = cell | packed_count;
            if (C.slot) C.slot[cell] = pos;
            if (F.O.sparse) F.O.keys[pos] = t * cpt + lane;
        } else if (mine == 5u) { // light: from the back of the active queue (raster frames only: no slot, no key)
            C.active[C.n_cells - 1u - (base_l + (uint32_t)__popcll(m_light & below))] = cell | packed_count;
        } else if (mine == 3u) {
            const uint32_t pos = base_d + (uint32_t)__popcll(m_dense & below);
            C.dense[pos] = cell;
            if (C.slot) C.slot[cell] = pos | 0x80000000u;
        } else if (clear_me) {
            s_inact[(uint32_t)__popcll(m_empty & below)] = lane;
        }
    }
    __syncthreads();

    // ---- clear the cells nothing can reach: 4 B per ray, most of the frame's HBM traffic.  All 1024 threads,
    //      16-byte stores (4 pixels per lane, 512 B per row segment) when the geometry is 4-pixel aligned ----
    if (tl) tl[4] = wall_clock64();
    if (!F.do_clear) { write_prep(); return; }
    const uint32_t zero_px = (F.O.pack_flags & VRT_ALPHA_COMPUTED) ? 0u : 0xFF000000u;
    const bool wide = F.O.image && !F.O.radiance && (P.tile_w % 4 == 0) && (P.stride % 4 == 0) &&
                      ((uintptr_t)F.O.image % 16 == 0) && (!F.O.compact || (P.tile_w * P.tile_h) % 4 == 0);
    const uint32_t n_inact = s_base[3];
    if (wide) {
        for (uint32_t it = tid; it < n_inact * (CELL * CELL / 4); it += 1024) { // one 4-pixel quad per item
            const uint32_t ci = s_inact[it / (CELL * CELL / 4)], q = it % (CELL * CELL / 4);
            const uint32_t pxt = (ci % C.cells_x) * CELL + (q % (CELL / 4)) * 4, pyt = (ci / C.cells_x) * CELL + q / (CELL / 4);
            const uint64_t pix = (uint64_t)(tx * P.tile_w + pxt) + (uint64_t)P.stride * (ty * P.tile_h + pyt);
            if (pxt < P.tile_w && pyt < P.tile_h && pix + 3 < npix) {
                const uint64_t o = F.O.compact ? ((uint64_t)lt * P.tile_h + pyt) * P.tile_w + pxt : pix;
                {   // non-temporal: 16 MB of background per 2048^2 frame that nobody reads again before the host does -- streamed past the
                    // L2 instead of ending up as its dirty lines (two frames in flight 20.7 -> 20.1 us, serial frame 31.2 -> 30.9)
                    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                    __builtin_nontemporal_store((u32x4){ zero_px, zero_px, zero_px, zero_px }, reinterpret_cast<u32x4 *>(F.O.image + o));
                }
            }
        }
    } else {
        for (uint32_t it = tid; it < n_inact * (CELL * CELL); it += 1024) {
            const uint32_t ci = s_inact[it / (CELL * CELL)], q = it % (CELL * CELL);
            const uint32_t pxt = (ci % C.cells_x) * CELL + q % CELL, pyt = (ci / C.cells_x) * CELL + q / CELL;
            const uint64_t pix = (uint64_t)(tx * P.tile_w + pxt) + (uint64_t)P.stride * (ty * P.tile_h + pyt);
            if (pxt < P.tile_w && pyt < P.tile_h && pix < npix) {
                const uint64_t o = F.O.compact ? ((uint64_t)lt * P.tile_h + pyt) * P.tile_w + pxt : pix;
                if (F.O.image) F.O.image[o] = zero_px;
                if (F.O.radiance) F.O.radiance[o] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
    if (tl) tl[5] = wall_clock64();
    write_prep();
}
struct ListArgs { BinArgs P; FuseArgs F; };
template <bool FROM_LIST, bool CHUNKS = false>
__global__ __launch_bounds__(1024) void build_tile_lists_kernel(ListArgs) // read through kernel_args<>: vrt_kernels_common.hpp
{
    const ListArgs &a = kernel_args<ListArgs>();
    build_tile_lists_body<FROM_LIST, CHUNKS>(a.P, a.F);
}
template <bool FROM_LIST, bool CHUNKS = false>
__global__ __launch_bounds__(1024) void build_tile_lists_batch_kernel(const FrameArgs *__restrict__ frames)
{
    const FrameArgs &a = frames[blockIdx.y];
    build_tile_lists_body<FROM_LIST, CHUNKS>(a.bin, a.fuse);
}

void launch_build_tile_lists(const BinArgs &a, const FuseArgs &f, bool from_list, uint32_t ntiles, hipStream_t st)
{
    if (!ntiles) return;
    if (from_list) hipLaunchKernelGGL(build_tile_lists_kernel<true>, dim3(ntiles), dim3(1024), 0, st, ListArgs{ a, f });
    else if (a.chunks && a.refine) hipLaunchKernelGGL((build_tile_lists_kernel<false, true>), dim3(ntiles), dim3(1024), 0, st, ListArgs{ a, f });
    else hipLaunchKernelGGL(build_tile_lists_kernel<false>, dim3(ntiles), dim3(1024), 0, st, ListArgs{ a, f });
}
void launch_build_tile_lists_batch(const FrameArgs *d_frames, uint32_t nframes, bool from_list, bool chunks, uint32_t ntiles, hipStream_t st)
{
    if (!ntiles || !nframes) return;
    if (from_list) hipLaunchKernelGGL(build_tile_lists_batch_kernel<true>, dim3(ntiles, nframes), dim3(1024), 0, st, d_frames);
    else if (chunks) hipLaunchKernelGGL((build_tile_lists_batch_kernel<false, true>), dim3(ntiles, nframes), dim3(1024), 0, st, d_frames);
    else hipLaunchKernelGGL(build_tile_lists_batch_kernel<false>, dim3(ntiles, nframes), dim3(1024), 0, st, d_frames);
}

// ---------------------------------------------------------------------------------------------
// Second level: one wavefront per 32x32 pixel cell filters its tile's list with the cell's cone.
// 16 cells share a 1024-thread workgroup so that filing the non-empty cells as active or dense costs
// at most two atomics per 16 cells -- one hot counter word serialises at ~11 ns per returning atomic,
// which 4096 single-cell atomics would turn into the longest kernel.  Empty cells: count == 0, no queue.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void build_cell_lists_kernel(SceneTables S, TileLists T, CellGrid C, RayGen R,
                                                                 const uint32_t *tile_map, uint32_t n_cells, int refine,
                                                                 uint32_t *keys /* sparse shard keys, nullable */)
{
    __shared__ uint32_t s_flag[16];  // 0 = empty, 1 = active (sparse), 3 = active (dense), 2 = no such cell
    __shared__ uint32_t s_base[3];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t cell = blockIdx.x * 16 + wave;
    uint32_t total = 0;
    if (cell < n_cells) {
        const uint32_t cpt = C.cells_x * C.cells_y;
        const uint32_t lt = cell / cpt, ci = cell % cpt;
        const uint32_t t = tile_map ? tile_map[lt] : lt;
        const uint32_t n_in = T.count[t];
        if (n_in) {
            const uint32_t tx = t % T.tiles_w, ty = t / T.tiles_w;
            const uint32_t x0 = (ci % C.cells_x) * CELL, y0 = (ci / C.cells_x) * CELL;
            const uint32_t x1 = min(x0 + CELL, T.tile_w) - 1, y1 = min(y0 + CELL, T.tile_h) - 1;
            Cone cone = {};
            if (refine) {
                const uint64_t npix = (uint64_t)R.width * R.height;
                auto at = [&](uint32_t x, uint32_t y) {
                    uint64_t pix = (uint64_t)(tx * T.tile_w + x) + (uint64_t)T.stride * (ty * T.tile_h + y);
                    if (pix >= npix) pix = npix - 1;
                    return cone_ray(R, pix);
                };
                cone = rect_cone(at, x0, y0, x1, y1, lane);
            }
            const uint32_t *in_list = T.indices + T.start[t];
            uint32_t *out = C.indices + (size_t)cell * C.cstride;
            for (uint32_t base = 0; base < n_in; base += 64) {
                const uint32_t k = base + lane;
                bool keep = false;
                uint32_t idx = 0;
                if (k < n_in) {
                    idx = in_list[k];
                    if (refine) {
                        float4 bq = S.gB[idx];
                        bq.w = slack_cull_x(bq.w, level_slack(T.cull_ref_n, n_in), T.floor_x);
                        keep = cone_keeps(cone, S.gA[idx], bq);
                    } else {
                        keep = true;
                    }
                }
                const unsigned long long mask = __ballot(keep);
                const uint32_t pos = total + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
                if (keep && pos < C.cstride) out[pos] = idx;
                total += (uint32_t)__popcll(mask);
            }
        }
        if (lane == 0) C.count[cell] = total > C.cstride ? 0xFFFFFFFFu : total;
    }
    if (lane == 0) s_flag[wave] = cell < n_cells ? (total ? (total > C.dense_threshold ? 3u : 1u) : 0u) : 2u;
    const uint32_t packed_count = min(total, 255u) << ACTIVE_COUNT_SHIFT;
    __syncthreads();
    if (tid == 0) {
        uint32_t na = 0, nd = 0;
        for (int w = 0; w < 16; ++w) { na += s_flag[w] == 1u; nd += s_flag[w] == 3u; }
        s_base[0] = na ? atomicAdd(C.n_active, na) : 0u;
        s_base[2] = nd ? atomicAdd(C.n_dense, nd) : 0u;
    }
    __syncthreads();
    if (lane == 0 && cell < n_cells) {
        uint32_t before = 0;
        const uint32_t mine = s_flag[wave];
        for (uint32_t w = 0; w < wave; ++w) before += s_flag[w] == mine;
        if (mine == 1u) {
            C.active[s_base[0] + before] = cell | packed_count;
            if (C.slot) C.slot[cell] = s_base[0] + before;
            if (keys) {
                const uint32_t cpt = C.cells_x * C.cells_y, lt = cell / cpt;
                keys[s_base[0] + before] = (tile_map ? tile_map[lt] : lt) * cpt + cell % cpt;
            }
        }
        else if (mine == 3u) { C.dense[s_base[2] + before] = cell; if (C.slot) C.slot[cell] = (s_base[2] + before) | 0x80000000u; }
    }
}

void launch_build_cell_lists(const SceneTables &s, const TileLists &t, const CellGrid &c, const RayGen &r,
                             const uint32_t *tile_map, uint32_t n_cells, int refine, uint32_t *keys, hipStream_t st)
{
    if (!n_cells) return;
    hipLaunchKernelGGL(build_cell_lists_kernel, dim3((n_cells + 15) / 16), dim3(1024), 0, st, s, t, c, r, tile_map,
                       n_cells, refine, keys);
}

// scatter rank-major shard buffers [rank](stride rank_stride)[slot][tile_h][tile_w] into the raster image (rt.h:388-399)
__global__ void assemble_kernel(const uint32_t *gathered, uint32_t *image, const uint32_t *tile_of_slot, TileLists T,
                                uint32_t width, uint32_t height, uint32_t slots_per_rank, uint64_t rank_stride)
{
    const uint32_t slot = blockIdx.y;
    const uint32_t t = tile_of_slot[slot];
    if (t == 0xFFFFFFFFu) return;
    const uint32_t *src = gathered + (slot / slots_per_rank) * rank_stride + (size_t)(slot % slots_per_rank) * (T.tile_w * T.tile_h);
    const uint32_t tx = t % T.tiles_w, ty = t / T.tiles_w;
    const uint32_t per_tile = T.tile_w * T.tile_h;
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < per_tile; p += gridDim.x * blockDim.x) {
        const uint32_t lx = p % T.tile_w, ly = p / T.tile_w;
        const uint64_t pix = (uint64_t)(tx * T.tile_w + lx) + (uint64_t)T.stride * (ty * T.tile_h + ly);
        if (pix < (uint64_t)width * height) image[pix] = src[p];
    }
}

void launch_assemble(const uint32_t *gathered, uint32_t *image, const uint32_t *tile_of_slot, uint32_t slots_per_rank,
                     uint32_t world, uint64_t rank_stride, const TileLists &t, uint32_t width, uint32_t height, hipStream_t st)
{
    const uint32_t n_slots = slots_per_rank * world;
    if (!n_slots) return;
    const uint32_t per_tile = t.tile_w * t.tile_h;
    const uint32_t gx = min((per_tile + 255u) / 256u, 64u);
    hipLaunchKernelGGL(assemble_kernel, dim3(gx ? gx : 1, n_slots), dim3(256), 0, st, gathered, image, tile_of_slot, t,
                       width, height, slots_per_rank, rank_stride);
}

// Frame assembly from sparse shards (multi-GPU): one workgroup per (shard, slot).  The shards may live in another
// GPU's memory (peer access over xGMI): they are read once, 16 B per lane, and only the stored cells travel.
// `stamp` (nullable): per cell of the FRAME (key order), the sequence number of the last assembly that stored it -- see
// clear_stale_cells_kernel.
__device__ __forceinline__ void scatter_sparse_body(const uint32_t *sh, uint32_t max_cells, uint32_t *image, const TileLists &T,
                                                    uint32_t cells_x, uint32_t cells_y, uint32_t width, uint32_t height,
                                                    uint32_t *stamp, uint32_t seq)
{
    const uint32_t slot = blockIdx.x, n = sh[0], cap = sh[1];
    if (cap != max_cells || sh[2] != cells_x * cells_y) return; // not a shard of this job's geometry: touch nothing
    if (slot >= n || slot >= max_cells) return;
    const uint32_t cpt = cells_x * cells_y;
    const uint32_t key = sh[SPARSE_HDR_WORDS + slot];
    const uint32_t t = key / cpt, ci = key % cpt;
    if (t >= T.tiles_w * T.tiles_h) return;
    const uint32_t tx = t % T.tiles_w, ty = t / T.tiles_w;
    if (stamp && threadIdx.x == 0) stamp[key] = seq;
    const uint4 *src = reinterpret_cast<const uint4 *>(sh + sparse_pixel_offset(cap) + (size_t)slot * (CELL * CELL));
    const uint64_t npix = (uint64_t)width * height;
    for (uint32_t q = threadIdx.x; q < CELL * CELL / 4; q += blockDim.x) { // one 4-pixel quad per lane and pass
        const uint4 v = src[q];
        const uint32_t cx = (q % (CELL / 4)) * 4, cy = q / (CELL / 4);
        const uint32_t pxt = (ci % cells_x) * CELL + cx, pyt = (ci / cells_x) * CELL + cy;
        if (pyt >= T.tile_h) continue;
        const uint64_t pix = (uint64_t)(tx * T.tile_w + pxt) + (uint64_t)T.stride * (ty * T.tile_h + pyt);
        const uint32_t px[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k)
            if (pxt + k < T.tile_w && pix + k < npix) image[pix + k] = px[k];
    }
}

__global__ __launch_bounds__(256) void scatter_sparse_kernel(ShardPtrs shards, uint32_t max_cells, uint32_t *image, TileLists T,
                                                             uint32_t cells_x, uint32_t cells_y, uint32_t width, uint32_t height,
                                                             uint32_t *stamp, uint32_t seq)
{
    scatter_sparse_body(shards.p[blockIdx.y], max_cells, image, T, cells_x, cells_y, width, height, stamp, seq);
}
// several frames per launch (blockIdx.z): frame f's shard s starts frame_stride words behind frame f-1's
__global__ __launch_bounds__(256) void scatter_sparse_batch_kernel(ShardPtrs shards, size_t frame_stride, AssemblyFrames F,
                                                                   uint32_t max_cells, TileLists T, uint32_t cells_x, uint32_t cells_y,
                                                                   uint32_t width, uint32_t height)
{
    const uint32_t f = blockIdx.z;
    scatter_sparse_body(shards.p[blockIdx.y] + f * frame_stride, max_cells, F.image[f], T, cells_x, cells_y, width, height,
                        F.stamp[f], F.seq[f]);
}

// Retained frames: an image buffer that still holds the previous assembly needs no background fill -- only the cells that
// were stored last time and are not stored now go back to background.  One wave per cell of the frame: stamp == seq - 1
// means "stored by the previous assembly, not by this one" (the scatter kernel of this assembly ran before this kernel).
__device__ __forceinline__ void clear_stale_cells_body(const uint32_t *stamp, uint32_t seq, uint32_t n_cells, uint32_t *image,
                                                       const TileLists &T, uint32_t cells_x, uint32_t cells_y, uint32_t width,
                                                       uint32_t height, uint32_t background)
{
    // one cell per LANE to look at (a frame has thousands of cells and a handful of stale ones), the wave then clears the
    // stale ones of its 64 one after the other
    const uint32_t lane = threadIdx.x & 63, key0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
    if (!stamp || key0 >= n_cells) return;
    const uint32_t mine = key0 + lane;
    unsigned long long stale = __ballot(mine < n_cells && stamp[mine] == seq - 1);
    const uint32_t cpt = cells_x * cells_y;
    const uint64_t npix = (uint64_t)width * height;
    while (stale) {
        const uint32_t key = key0 + (uint32_t)__builtin_ctzll(stale);
        stale &= stale - 1;
        const uint32_t t = key / cpt, ci = key % cpt;
        const uint32_t tx = t % T.tiles_w, ty = t / T.tiles_w;
        for (uint32_t q = lane; q < CELL * CELL; q += 64) {
            const uint32_t pxt = (ci % cells_x) * CELL + q % CELL, pyt = (ci / cells_x) * CELL + q / CELL;
            const uint64_t pix = (uint64_t)(tx * T.tile_w + pxt) + (uint64_t)T.stride * (ty * T.tile_h + pyt);
            if (pxt < T.tile_w && pyt < T.tile_h && pix < npix) image[pix] = background;
        }
    }
}
__global__ __launch_bounds__(256) void clear_stale_cells_kernel(const uint32_t *stamp, uint32_t seq, uint32_t n_cells, uint32_t *image,
                                                                TileLists T, uint32_t cells_x, uint32_t cells_y, uint32_t width,
                                                                uint32_t height, uint32_t background)
{
    clear_stale_cells_body(stamp, seq, n_cells, image, T, cells_x, cells_y, width, height, background);
}
// blockIdx.y = frame; frames whose buffer got the full fill this time carry clear[f] = 0
__global__ __launch_bounds__(256) void clear_stale_cells_batch_kernel(AssemblyFrames F, uint32_t n_cells, TileLists T, uint32_t cells_x,
                                                                      uint32_t cells_y, uint32_t width, uint32_t height,
                                                                      uint32_t background)
{
    const uint32_t f = blockIdx.y;
    if (!F.clear[f]) return;
    clear_stale_cells_body(F.stamp[f], F.seq[f], n_cells, F.image[f], T, cells_x, cells_y, width, height, background);
}

void launch_scatter_sparse(const ShardPtrs &shards, int nshards, uint32_t max_cells, uint32_t *image, const TileLists &t,
                           uint32_t cells_x, uint32_t cells_y, uint32_t width, uint32_t height, uint32_t *stamp, uint32_t seq,
                           hipStream_t st)
{
    if (nshards <= 0 || !max_cells) return;
    hipLaunchKernelGGL(scatter_sparse_kernel, dim3(max_cells, (uint32_t)nshards), dim3(256), 0, st, shards, max_cells, image, t,
                       cells_x, cells_y, width, height, stamp, seq);
}
void launch_assemble_sparse_batch(const ShardPtrs &shards, int nshards, size_t frame_stride, const AssemblyFrames &frames, int nframes,
                                  uint32_t max_cells, uint32_t n_cells, const TileLists &t, uint32_t cells_x, uint32_t cells_y,
                                  uint32_t width, uint32_t height, uint32_t background, hipStream_t st)
{
    if (nshards <= 0 || nframes <= 0 || !max_cells) return;
    // a frame stride shorter than a whole shard (a gathered prefix) cannot hold more cells than fit in it: no workgroups for
    // slots that did not travel
    uint32_t slots = max_cells;
    const size_t hdr = sparse_pixel_offset(max_cells);
    if (frame_stride > hdr && frame_stride < hdr + (size_t)max_cells * (CELL * CELL))
        slots = (uint32_t)std::max<size_t>(1, (frame_stride - hdr) / (CELL * CELL));
    hipLaunchKernelGGL(scatter_sparse_batch_kernel, dim3(slots, (uint32_t)nshards, (uint32_t)nframes), dim3(256), 0, st, shards,
                       frame_stride, frames, max_cells, t, cells_x, cells_y, width, height);
    bool any = false;
    for (int f = 0; f < nframes; ++f) any = any || frames.clear[f];
    if (any && n_cells)
        hipLaunchKernelGGL(clear_stale_cells_batch_kernel, dim3((n_cells + 255) / 256, (uint32_t)nframes), dim3(256), 0, st, frames, n_cells, t,
                           cells_x, cells_y, width, height, background);
}
void launch_clear_stale_cells(const uint32_t *stamp, uint32_t seq, uint32_t n_cells, uint32_t *image, const TileLists &t,
                              uint32_t cells_x, uint32_t cells_y, uint32_t width, uint32_t height, uint32_t background, hipStream_t st)
{
    if (!n_cells) return;
    hipLaunchKernelGGL(clear_stale_cells_kernel, dim3((n_cells + 255) / 256), dim3(256), 0, st, stamp, seq, n_cells, image, t, cells_x,
                       cells_y, width, height, background);
}

// ---------------------------------------------------------------------------------------------
// Point queries (API parity with rt.h:32-54, rt.cpp:8-27, rt.h:146-223); not performance paths.
// ---------------------------------------------------------------------------------------------
template <int EXP, int ERF>
__global__ void transmittance_kernel(SceneTables S, float ox, float oy, float oz, float nx, float ny, float nz,
                                     const float *s_in, size_t ns, float *T_out)
{
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ns) return;
    const float s = s_in[k];
    float T = 0.f;
    for (uint32_t q = 0; q < S.n; ++q) { // rt.h:36-52, same operations in the same order, unfused (see dot3_ref)
        const float4 g = S.mu_sig[q];
        const float mag = S.gD[q].z;
        const float cx = sub_ref(g.x, ox), cy = sub_ref(g.y, oy), cz = sub_ref(g.z, oz);
        const float mu_bar = dot3_ref(cx, cy, cz, nx, ny, nz);
        const float oc_sq = dot3_ref(cx, cy, cz, cx, cy, cz);
        const float inv_2_sigma2 = 1.f / mul_ref(mul_ref(2.f, g.w), g.w);
        const float c_bar = mul_ref(mag, vexp<EXP>(-mul_ref(sub_ref(oc_sq, mul_ref(mu_bar, mu_bar)), inv_2_sigma2)));
        const float sqrt_2_sig = mul_ref(SQRT_2, g.w);
        const float mu_bar_n = mu_bar / sqrt_2_sig;
        const float s_n = s / sqrt_2_sig;
        const float term = mul_ref(mul_ref(mul_ref(g.w, c_bar), INV_SQRT_2_PI), sub_ref(verf<ERF>(-mu_bar_n), verf<ERF>(sub_ref(s_n, mu_bar_n))));
        T = add_ref(T, term);
    }
    T_out[k] = vexp<EXP>(T);
}
// The same per ray: ray k has its own origin, direction and sample point -- broadcast_transmittance (rt.h:102-127),
// lane = ray.  The arithmetic is transmittance_kernel's (exact divides; the reference's rcp14 estimates are not
// reproduced).
template <int EXP, int ERF>
__global__ void transmittance_rays_kernel(SceneTables S, const float *origins, const float *dirs, const float *s_in,
                                          size_t nrays, float *T_out)
{
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nrays) return;
    const float ox = origins[3 * k], oy = origins[3 * k + 1], oz = origins[3 * k + 2];
    const float nx = dirs[3 * k], ny = dirs[3 * k + 1], nz = dirs[3 * k + 2];
    const float s = s_in[k];
    float T = 0.f;
    for (uint32_t q = 0; q < S.n; ++q) {
        const float4 g = S.mu_sig[q];
        const float mag = S.gD[q].z;
        const float cx = sub_ref(g.x, ox), cy = sub_ref(g.y, oy), cz = sub_ref(g.z, oz);
        const float mu_bar = dot3_ref(cx, cy, cz, nx, ny, nz);
        const float oc_sq = dot3_ref(cx, cy, cz, cx, cy, cz);
        const float inv_2_sigma2 = 1.f / mul_ref(mul_ref(2.f, g.w), g.w);
        const float c_bar = mul_ref(mag, vexp<EXP>(-mul_ref(sub_ref(oc_sq, mul_ref(mu_bar, mu_bar)), inv_2_sigma2)));
        const float sqrt_2_sig = mul_ref(SQRT_2, g.w);
        const float mu_bar_n = mu_bar / sqrt_2_sig;
        const float s_n = s / sqrt_2_sig;
        const float term = mul_ref(mul_ref(mul_ref(g.w, c_bar), INV_SQRT_2_PI), sub_ref(verf<ERF>(-mu_bar_n), verf<ERF>(sub_ref(s_n, mu_bar_n))));
        T = add_ref(T, term);
    }
    T_out[k] = vexp<EXP>(T);
}
template <int EXP, int ERF>
static void launch_transmittance_rays_t(const SceneTables &s, const float *d_o, const float *d_n, const float *d_s, size_t nrays,
                                        float *d_T, hipStream_t st)
{
    hipLaunchKernelGGL((transmittance_rays_kernel<EXP, ERF>), dim3((uint32_t)((nrays + 63) / 64)), dim3(64), 0, st, s, d_o,
                       d_n, d_s, nrays, d_T);
}
void launch_transmittance_rays(const SceneTables &s, const float *d_o, const float *d_n, const float *d_s, size_t nrays,
                               float *d_T, int exp_kind, int erf_kind, hipStream_t st)
{
    if (!nrays) return;
    VRT_DISPATCH_EXP_ERF(launch_transmittance_rays_t, s, d_o, d_n, d_s, nrays, d_T, st);
}

template <int EXP, int ERF>
static void launch_transmittance_t(const SceneTables &s, const float o[3], const float n[3], const float *d_s, size_t ns,
                                   float *d_T, hipStream_t st)
{
    hipLaunchKernelGGL((transmittance_kernel<EXP, ERF>), dim3((uint32_t)((ns + 63) / 64)), dim3(64), 0, st, s, o[0],
                       o[1], o[2], n[0], n[1], n[2], d_s, ns, d_T);
}
void launch_transmittance(const SceneTables &s, const float o[3], const float n[3], const float *d_s, size_t ns,
                          float *d_T, int exp_kind, int erf_kind, hipStream_t st)
{
    if (!ns) return;
    VRT_DISPATCH_EXP_ERF(launch_transmittance_t, s, o, n, d_s, ns, d_T, st);
}

// rt.cpp:8-17: Riemann sum with step delta, fast_exp of the negated sum
__global__ void transmittance_step_kernel(SceneTables S, float ox, float oy, float oz, float nx, float ny, float nz,
                                          const float *s_in, size_t ns, float delta, float *T_out)
{
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ns) return;
    const float s = s_in[k];
    float T = 0.f;
    for (float t = 0; t <= s; t += delta)
        for (uint32_t q = 0; q < S.n; ++q) {
            const float4 g = S.mu_sig[q];
            const float dx = ox + nx * t - g.x, dy = oy + ny * t - g.y, dz = oz + nz * t - g.z;
            T += delta * (S.gD[q].z * exp_accurate(-(dx * dx + dy * dy + dz * dz) / (2 * g.w * g.w)));
        }
    T_out[k] = exp_fast(-T);
}
void launch_transmittance_step(const SceneTables &s, const float o[3], const float n[3], const float *d_s, size_t ns,
                               float delta, float *d_T, hipStream_t st)
{
    if (!ns) return;
    hipLaunchKernelGGL(transmittance_step_kernel, dim3((uint32_t)((ns + 63) / 64)), dim3(64), 0, st, s, o[0], o[1],
                       o[2], n[0], n[1], n[2], d_s, ns, delta, d_T);
}

// rt.cpp:19-27
__global__ void density_kernel(SceneTables S, const float *pts, size_t npts, float *D)
{
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= npts) return;
    const float x = pts[3 * k], y = pts[3 * k + 1], z = pts[3 * k + 2];
    float acc = 0.f;
    for (uint32_t q = 0; q < S.n; ++q) {
        const float4 g = S.mu_sig[q];
        const float dx = x - g.x, dy = y - g.y, dz = z - g.z;
        acc += S.gD[q].z * exp_accurate(-(dx * dx + dy * dy + dz * dz) / (2 * g.w * g.w));
    }
    D[k] = acc;
}
void launch_density(const SceneTables &s, const float *d_pts, size_t npts, float *d_D, hipStream_t st)
{
    if (!npts) return;
    hipLaunchKernelGGL(density_kernel, dim3((uint32_t)((npts + 63) / 64)), dim3(64), 0, st, s, d_pts, npts, d_D);
}

// arbitrary rays: lane = ray, every Gaussian of the scene, per-lane origin
template <int EXP, int ERF>
__global__ __launch_bounds__(64) void radiance_kernel(SceneTables S, const float *origins, const float *dirs,
                                                       size_t nrays, const uint32_t *iota, float4 *out)
{
    const size_t r = (size_t)blockIdx.x * 64 + threadIdx.x;
    const size_t rc = r < nrays ? r : nrays - 1;
    LaneRay ray;
    ray.ox = origins[3 * rc]; ray.oy = origins[3 * rc + 1]; ray.oz = origins[3 * rc + 2];
    ray.nx = dirs[3 * rc]; ray.ny = dirs[3 * rc + 1]; ray.nz = dirs[3 * rc + 2];
    float Lr, Lg, Lb, La;
    shade_list<EXP, ERF, 4, false>(S, iota, S.n, ray, Lr, Lg, Lb, La);
    if (r < nrays) out[r] = make_float4(Lr, Lg, Lb, La);
}
template <int EXP, int ERF>
static void launch_radiance_t(const SceneTables &s, const float *d_origins, const float *d_dirs, size_t nrays,
                              const uint32_t *iota, float4 *d_out, hipStream_t st)
{
    hipLaunchKernelGGL((radiance_kernel<EXP, ERF>), dim3((uint32_t)((nrays + 63) / 64)), dim3(64), 0, st, s, d_origins,
                       d_dirs, nrays, iota, d_out);
}
void launch_radiance(const SceneTables &s, const float *d_origins, const float *d_dirs, size_t nrays,
                     const uint32_t *iota, float4 *d_out, int exp_kind, int erf_kind, hipStream_t st)
{
    if (!nrays) return;
    VRT_DISPATCH_EXP_ERF(launch_radiance_t, s, d_origins, d_dirs, nrays, iota, d_out, st);
}

template <int K>
__global__ void eval_erf_kernel(const float *x, size_t n, float *y)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = verf<K>(x[i]);
}
template <int K>
__global__ void eval_exp_kernel(const float *x, size_t n, float *y)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = vexp<K>(x[i]);
}
void launch_eval_erf(int kind, const float *x, size_t n, float *y, hipStream_t st)
{
    if (!n) return;
    const dim3 g((uint32_t)((n + 255) / 256)), b(256);
    switch (kind) {
    case VRT_ERF_AS: hipLaunchKernelGGL(eval_erf_kernel<VRT_ERF_AS>, g, b, 0, st, x, n, y); break;
    case VRT_ERF_SPLINE: hipLaunchKernelGGL(eval_erf_kernel<VRT_ERF_SPLINE>, g, b, 0, st, x, n, y); break;
    case VRT_ERF_SPLINE_MIRROR: hipLaunchKernelGGL(eval_erf_kernel<VRT_ERF_SPLINE_MIRROR>, g, b, 0, st, x, n, y); break;
    case VRT_ERF_TAYLOR: hipLaunchKernelGGL(eval_erf_kernel<VRT_ERF_TAYLOR>, g, b, 0, st, x, n, y); break;
    default: hipLaunchKernelGGL(eval_erf_kernel<VRT_ERF_LIBM>, g, b, 0, st, x, n, y); break;
    }
}
void launch_eval_exp(int kind, const float *x, size_t n, float *y, hipStream_t st)
{
    if (!n) return;
    const dim3 g((uint32_t)((n + 255) / 256)), b(256);
    switch (kind) {
    case VRT_EXP_VCL: hipLaunchKernelGGL(eval_exp_kernel<VRT_EXP_VCL>, g, b, 0, st, x, n, y); break;
    case VRT_EXP_FAST: hipLaunchKernelGGL(eval_exp_kernel<VRT_EXP_FAST>, g, b, 0, st, x, n, y); break;
    case VRT_EXP_SPLINE: hipLaunchKernelGGL(eval_exp_kernel<VRT_EXP_SPLINE>, g, b, 0, st, x, n, y); break;
    default: hipLaunchKernelGGL(eval_exp_kernel<VRT_EXP_LIBM>, g, b, 0, st, x, n, y); break;
    }
}


} // namespace vrtk
