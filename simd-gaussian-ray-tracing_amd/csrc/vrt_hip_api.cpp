// vrt_hip_api.cpp -- the C ABI of libvrt_hip.so (see include/vrt_hip.h): context, device-resident
// scene / tile / ray state, launches.  Compiled with hipcc for gfx950; links only libamdhip64.
// There is no CPU fallback anywhere in this file: without a GPU vrt_hip_create() fails.
//
// Frame pipeline (all on the caller's stream for the *_device entry points):
//   prep_frame_kernel        only when the origin or the scene changed: oc = mu - origin, |oc|^2
//   build_tile_lists_kernel  only when tiles / rays / origin / options changed: the reference's tile sets
//                            (rt.cpp:29-69, or the caller's lists) intersected with a tile-level cull
//   render_kernel            one wavefront per 8x8 pixel block
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/vrt_hip.h"
#include "vrt_kernels.h"

using namespace vrtk;

namespace {

std::string g_create_error;

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t n)
    {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        hipError_t e = hipMalloc((void **)&p, (n ? n : 1) * sizeof(T));
        if (e == hipSuccess) cap = n ? n : 1;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

enum TileMode { TILES_NONE = 0, TILES_HOST = 1, TILES_DEVICE = 2 };

} // namespace

struct vrt_hip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    // the caller's stream the last *_device call enqueued on (may differ from `stream`): state-changing calls wait for
    // it before they touch buffers its kernels may still be reading (quiesce)
    hipStream_t last_stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::string err;

    // counts every change of scene, options or table settings (vrt_hip_state_generation): a holder of mirrored contexts
    // (vrt_hip_group's batch lanes) sees when they are out of date
    uint64_t state_gen = 1;
    // scene (static SoA copy kept so options can be re-applied)
    uint32_t n = 0;
    DevBuf<float> soa[9]; // mu_x mu_y mu_z ar ag ab aa sigma mag
    bool has_alpha = false;
    DevBuf<float4> mu_sig, gA, gB, gC, gD;
    DevBuf<float4> gChunk;       // bounding spheres of every 64 consecutive Gaussians (launch_build_chunks): the tile level tests these first
    int use_chunks = 1;          // VRT_HIP_CHUNKS: 0 every tile tests every Gaussian (rounds 1-2); 1 (default) chunks first for scenes beyond 8192 Gaussians; 2 always
    DevBuf<uint32_t> iota;
    bool tables_dirty = true;
    bool gA_valid = false;
    float gA_origin[3] = { 0, 0, 0 };

    // tiles.  "ref" lists carry the reference semantics (tiles_t): uploaded by the caller (TILES_HOST), one
    // tile holding everything (TILES_NONE), or produced on demand for queries (TILES_DEVICE).  "work" lists
    // are what the render kernel scans: ref lists intersected with the tile-level cull.
    TileMode tile_mode = TILES_NONE;
    float tw = 2.f, th = 2.f;
    uint32_t tiles_w = 1, tiles_h = 1;
    float view[16] = { 0 };
    DevBuf<uint32_t> ref_start, ref_count, ref_indices;
    bool ref_valid = false;
    DevBuf<uint32_t> w_start, w_count, w_indices;
    // frame batches (vrt_hip_frame_batch_device): while `defer` is set the three per-frame launches are recorded, not made
    FrameArgs *defer = nullptr;
    struct Deferred {
        bool lists = false, from_list = false, render = false, order = false;
        uint32_t list_grid = 0, render_grid = 0, dense_grid = 0;
    } deferred;
    // the context a batch is issued through keeps the argument rows: a ring of pinned host slots and device slots
    static constexpr int BATCH_SLOTS = 4;
    FrameArgs *batch_host = nullptr, *batch_dev = nullptr;
    size_t batch_cap = 0; // frames per slot
    hipEvent_t batch_copied[BATCH_SLOTS] = {};
    uint32_t batch_seq = 0;
    // retained assembly (vrt_hip_scatter_sparse_retained_device): which cells the buffer's last assembly stored
    struct Retained {
        uint32_t *image = nullptr;
        uint64_t sig = 0;
        uint32_t bg = 0, seq = 0;
        DevBuf<uint32_t> stamp;
    };
    std::vector<Retained> retained; // one history per frame buffer (at most MAX_ASSEMBLY_FRAMES, oldest dropped)
    // tile cones of the list kernel: a function of the rays and the tile geometry only, kept across frames (cone_key =
    // what they were made for)
    DevBuf<float4> tile_cones;
    uint32_t cone_gen = 0;     // tag of the rows made for cone_key's camera (BinArgs::cone_gen)
    std::string cone_key;
    uint32_t plane_gen = 0;
    // cells with at most this many candidates are shaded last (CellGrid::light_threshold; 32 and more file too many cells as light, profiles/r02_experiments.md);
    // lists_light: what the lists now in the buffers were built with (0 for sparse shards and the two-kernel list path)
    static constexpr uint32_t light_cells = 24;
    uint32_t lists_light = 0;
    int claim_early = 8;         // CellGrid::claim_early (measured: 2 leaves `-g 16 -w 2048` at 67 us, 8 takes it to 43, "always" costs a 12-waves-per-CU grid 8 % in flight); VRT_HIP_CLAIM_EARLY=0: the block kernel's waves ask for their next block only when they are done with the current one
    bool skip_idle_dense = true; // VRT_HIP_DENSE_SKIP=0: the dense kernel is launched behind every block kernel
    float albedo_scale = 1.f;    // max(1, largest |albedo| of the scene): divides the prune budget
    float cull_prune = 6.f;      // vrt_hip_set_cull_prune(): a block-kernel ray may drop the smallest entries of its list while their sum stays below
                                 // cull_prune * cull_ref_n * cull_eps (prune_list; 0 = off).  6: 3 * 6 * 1365 * 1e-9 = 2.46e-5 -- DESIGN.md section 4
    float cull_ref_n = 4096.f / 3.f; // TileLists::cull_ref_n; VRT_HIP_CULL_REF_N=0: one threshold at every level (round 1)
    // second level: 32x32-pixel cells of the local tiles + the active / dense queues of the render kernels
    DevBuf<uint32_t> c_count, c_indices, c_active, c_dense, c_dense_sorted, c_scratch, c_overflow, c_counters, c_rq, c_slot;
    uint32_t rq_gen = 0;      // render launches: selects the work-queue counter set (CellGrid::rq)
    int render_waves_per_cu = 13; // persistent one-wave workgroups per CU: what LDS allows (VGPRs: three per SIMD run at a time; the
                                  // 13th starts when the first retires); VRT_HIP_RENDER_WAVES overrides
    uint32_t render_grid_override = 0; // VRT_HIP_RENDER_GRID (tests): exactly this many block-kernel workgroups, e.g. ONE wave that drains all work queues
    uint32_t cells_x = 1, cells_y = 1, cstride = 1, n_cells = 0;
    int lists_for_shard = -1; // sharding mode the cell lists were built for
    bool prep_pending = false; // the per-origin table (gA) of gA_origin is still to be written: by the next list kernel, or by flush_prep()
    bool lists_fresh = false; // the queue counters were zeroed by the list build of this very call
    uint32_t list_gen = 0;    // list generation: selects the counter set (see cell_grid)
    // dense-launch feedback (CellGrid::feedback): host-mapped, read frames later
    volatile uint32_t *h_fb = nullptr;
    uint32_t *d_fb = nullptr;
    // dense-launch sizing: frame_seq counts render launches; a report in h_fb[3] (the sequence number of the frame
    // that wrote it) newer than reset_seq comes from the current scene / camera / options
    // A camera that moved keeps the reports (an orbit changes the picture gradually) but widens the idle launch until a
    // report from the new pose has arrived (cam_seq): a jump to a pose with dense cells costs one frame at a quarter of
    // the GPU, not one frame on one workgroup.
    uint32_t frame_seq = 0, reset_seq = 0, cam_seq = 0;
    int num_cus = 256;
    float table_hx = 0.05f;      // vrt_hip_set_table_step(): requested node spacing of the table kernel; 0 = the exact kernels only
    float table_budget = 2.5e-5f; // vrt_hip_set_table_budget(): worst-case change of a ray's radiance the table kernel may cause
    float table_room = 0.9f;      // share of the budget the kernel's ESTIMATE of its bound may fill when it coarsens the spacing (VRT_HIP_TABLE_ROOM)
    float table_adapt = 3.f;      // the table kernel may coarsen the requested spacing by up to this factor where its estimate of the
                                  // bound leaves room (VRT_HIP_TABLE_ADAPT; 1 = never)
    static constexpr int dense_idle_grid = 1; // workgroups of the dense launch when nothing is expected for it: one
                             // 1024-thread workgroup finds a CU with 61 KB of LDS free sooner than eight do (-2 % with frames in flight)
    int dense_waves = 16; // waves per block in the dense kernel (tuning knob: VRT_HIP_DENSE_WAVES = 4 | 8 | 16)
    bool work_is_ref = false; // render straight from the ref lists (no tile-level cull possible)
    bool lists_dirty = true;
    DevBuf<float> xc, yc;
    float grid_tw = 0.f, grid_th = 0.f;
    uint32_t grid_n = 0xFFFFFFFFu;

    // rays
    uint32_t w = 0, h = 0;
    bool plane_mode = false;
    bool view_mode = false;   // rays from inverse(view) (vrt_hip_set_camera_view)
    float inv_view[16] = { 0 };
    bool plane_affine = false; // plane arrays are a pinhole pattern: corner rays bound a tile's cone
    DevBuf<float> xs, ys, zs;
    float cam_pos[3] = { 0, 0, 0 }, cam_right[3] = { 1, 0, 0 }, cam_up[3] = { 0, 1, 0 }, cam_front[3] = { 0, 0, -1 };
    float focal = 1.f;
    bool rays_set = false;

    // options
    int exp_kind = VRT_EXP_VCL, erf_kind = VRT_ERF_AS;
    float cull_eps = 1e-9f;

    // sharding
    int rank = 0, world = 1;
    DevBuf<uint32_t> tile_map, slot_tiles;
    uint32_t n_local = 0, n_slots = 0;
    bool shard_dirty = true;

    // scratch + statistics
    // d_image: the library's own frame buffer (vrt_hip_frame, vrt_hip_render).  Retained between vrt_hip_frame calls: own_stamp[cell]
    // = own_seq of the last frame that lit the cell, valid while own_sig (image size, tile grid, background) stays and nothing else
    // wrote the buffer (own_seq = 0: the next frame clears everything and starts a new history)
    DevBuf<uint32_t> own_stamp;
    uint32_t own_seq = 0;
    struct OwnGeometry { // what a retained history is valid for: compared field by field (a hash of overlapping fields let two tile grids collide)
        uint32_t w = 0, h = 0, tiles_w = 0, tiles_h = 0, tile_w = 0, tile_h = 0, background = 0;
        const uint32_t *image = nullptr;
        bool operator==(const OwnGeometry &o) const
        {
            return w == o.w && h == o.h && tiles_w == o.tiles_w && tiles_h == o.tiles_h && tile_w == o.tile_w && tile_h == o.tile_h &&
                   background == o.background && image == o.image;
        }
    } own_sig;
    bool retain_next = false; // set by vrt_hip_frame around its render_common call
    DevBuf<uint32_t> d_image;
    DevBuf<float4> d_rad;
    DevBuf<unsigned long long> d_stats, d_timeline; // d_timeline: VRT_HIP_TIMELINE=1 diagnostics
    size_t timeline_items = 0, timeline_tiles = 0;
    DevBuf<unsigned long long> d_timeline_lists;
    bool stats_on = false;
    vrt_hip_stats last{};
    // kernel timing ring (vrt_hip_enable_kernel_timing)
    static constexpr int TIMING_RING = 512;
    bool timing_on = false;
    std::vector<hipEvent_t> tev; // 4 per slot: before lists, before render, after render, after dense
    bool timing_full = true;
    uint32_t timing_period = 1, timing_frame = 0;
    uint64_t timing_count = 0;
};

static void print_timeline(vrt_hip_ctx *c);
namespace {

int fail(vrt_hip_ctx *c, int code, const std::string &msg)
{
    if (c) c->err = msg; else g_create_error = msg;
    return code;
}

#define HIPCHK(c, call)                                                                            \
    do {                                                                                           \
        hipError_t _e = (call);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return fail((c), VRT_HIP_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(_e)); \
    } while (0)

// Frames enqueued through the *_device entry points run on the CALLER's stream and read the context's tables, lists
// and plane arrays.  Every call that rewrites one of those waits here first -- for the context's own stream and for the
// stream of the last enqueued frame -- so a caller may change state right after enqueueing frames without a
// synchronisation of its own (include/vrt_hip.h, "Streams").  An error from the caller's stream (it may have been
// destroyed, which completes its work) is not this call's error.
int quiesce(vrt_hip_ctx *c)
{
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->last_stream && c->last_stream != c->stream) {
        if (hipStreamSynchronize(c->last_stream) != hipSuccess) (void)hipGetLastError();
    }
    c->last_stream = nullptr;
    return VRT_HIP_OK;
}

float exp_floor_x(int exp_kind)
{
    // Exp(-x) is exactly 0 past this point for the chosen Exp, so such Gaussians contribute nothing:
    // vcl_exp flushes below -87.3 (vectormath_exp.h:393); expf reaches 0 below ~-103.98.
    switch (exp_kind) {
    case VRT_EXP_VCL: return 87.3f;
    case VRT_EXP_LIBM: return 104.f;
    case VRT_EXP_FAST: return 88.f;   // clamped fast_exp returns 0 for x < -87.3 (a*x+b < 2^23)
    case VRT_EXP_SPLINE: return 9.0f; // spline_exp(x) = 0 for x <= -9 (approx.cpp:143)
    default: return INFINITY;
    }
}

// table mode applies to the Exp / Erf pairs its error bound covers (vrt_kernels.hip, VRT_DISPATCH_TABLE)
bool table_on(const vrt_hip_ctx *c)
{
    return c->table_hx > 0.f && (c->erf_kind == VRT_ERF_AS || c->erf_kind == VRT_ERF_LIBM) &&
           (c->exp_kind == VRT_EXP_VCL || c->exp_kind == VRT_EXP_LIBM);
}

int rebuild_tables(vrt_hip_ctx *c)
{
    if (!c->tables_dirty) return VRT_HIP_OK;
    { int rc = quiesce(c); if (rc) return rc; } // frames in flight read the tables this rewrites
    HIPCHK(c, c->mu_sig.reserve(c->n)); HIPCHK(c, c->gA.reserve(c->n)); HIPCHK(c, c->gB.reserve(c->n));
    HIPCHK(c, c->gC.reserve(c->n)); HIPCHK(c, c->gD.reserve(c->n)); HIPCHK(c, c->iota.reserve(c->n));
    // cull_eps bounds what ONE Gaussian dropped at the tile level could have contributed; a ray can lose all N of them, so
    // for scenes beyond 4096 Gaussians the threshold shrinks with N: 3 * eps_eff * N stays at the 1.2e-5 of N = 4096 and the
    // frame's worst case at 2.5e-5 whatever N is (the lower levels already scale with their list lengths)
    const float eps_eff = c->cull_eps * std::min(1.f, 4096.f / (float)std::max(c->n, 1u));
    launch_build_static(c->n, c->soa[0].p, c->soa[1].p, c->soa[2].p, c->soa[3].p, c->soa[4].p, c->soa[5].p,
                        c->has_alpha ? c->soa[6].p : nullptr, c->soa[7].p, c->soa[8].p, eps_eff,
                        exp_floor_x(c->exp_kind), c->mu_sig.p, c->gB.p, c->gC.p, c->gD.p, c->stream);
    HIPCHK(c, c->gChunk.reserve((c->n + 63u) / 64u));
    launch_build_chunks(c->n, c->mu_sig.p, c->gB.p, c->gChunk.p, c->stream);
    launch_iota(c->iota.p, c->n, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream)); // later launches may use a caller's stream
    c->tables_dirty = false;
    c->gA_valid = false;
    c->lists_dirty = true;
    return VRT_HIP_OK;
}

SceneTables tables(const vrt_hip_ctx *c)
{
    SceneTables s;
    s.mu_sig = c->mu_sig.p; s.gA = c->gA.p; s.gB = c->gB.p; s.gC = c->gC.p; s.gD = c->gD.p; s.n = c->n;
    return s;
}

int prep_frame(vrt_hip_ctx *c, const float origin[3], hipStream_t st)
{
    int rc = rebuild_tables(c);
    if (rc) return rc;
    if (c->gA_valid && !memcmp(c->gA_origin, origin, 3 * sizeof(float))) return VRT_HIP_OK;
    if (c->defer) { // a frame of a batch: one prep launch for all frames (launch_frame_setup_batch)
        c->defer->do_prep = 1; c->defer->prep_gA = c->gA.p;
        memcpy(c->defer->prep_origin, origin, 3 * sizeof(float));
    } else {
        c->prep_pending = true; // the list kernel of this frame writes the table (BinArgs::prep_gA); flush_prep() launches it if none does
    }
    memcpy(c->gA_origin, origin, 3 * sizeof(float));
    c->cam_seq = c->frame_seq; // the camera moved
    c->gA_valid = true;
    c->lists_dirty = true; // the tile-level cull depends on the origin
    return VRT_HIP_OK;
}

int flush_prep(vrt_hip_ctx *c, hipStream_t st)
{
    if (!c->prep_pending) return VRT_HIP_OK;
    launch_prep_frame(tables(c), c->gA.p, c->gA_origin, st);
    HIPCHK(c, hipGetLastError());
    c->prep_pending = false;
    return VRT_HIP_OK;
}

// tile geometry for the current image size (rt.h:348-349, 364-365); list pointers filled by the caller
TileLists tile_geometry(const vrt_hip_ctx *c)
{
    TileLists t{};
    if (c->tile_mode != TILES_NONE) {
        t.tiles_w = c->tiles_w; t.tiles_h = c->tiles_h;
        t.tile_w = (uint32_t)(uint64_t)(c->w * c->tw / 2.f);
        t.tile_h = (uint32_t)(uint64_t)(c->h * c->th / 2.f);
    } else {
        t.tiles_w = t.tiles_h = 1;
        t.tile_w = c->w; t.tile_h = c->h;
    }
    t.stride = t.tile_w * t.tiles_w;
    return t;
}

RayGen ray_gen(const vrt_hip_ctx *c, const float origin[3])
{
    RayGen r;
    r.xs = c->plane_mode ? c->xs.p : nullptr; r.ys = c->plane_mode ? c->ys.p : nullptr;
    r.zs = c->plane_mode ? c->zs.p : nullptr;
    for (int i = 0; i < 3; ++i) {
        r.origin[i] = origin[i]; r.pos[i] = c->cam_pos[i]; r.right[i] = c->cam_right[i]; r.up[i] = c->cam_up[i];
        r.front[i] = c->cam_front[i];
    }
    r.focal = c->focal;
    r.inv_half_w = 1.f / (c->w / 2.f); r.inv_half_h = 1.f / (c->h / 2.f);
    r.width = c->w; r.height = c->h;
    r.view_mode = (c->view_mode && !c->plane_mode) ? 1 : 0;
    for (int i = 0; i < 3; ++i) { r.m0[i] = c->inv_view[i]; r.m1[i] = c->inv_view[4 + i]; r.m3[i] = c->inv_view[12 + i]; }
    r.half_w = c->w / 2.f; r.half_h = c->h / 2.f;
    return r;
}

// (Re)builds the tile-centre arrays when the tile grid changes (the reference's float loops, rt.cpp:47-49).
int prepare_tile_grid(vrt_hip_ctx *c, float tw, float th)
{
    { int rc = quiesce(c); if (rc) return rc; } // xc / yc / w_start are read by list kernels in flight
    std::vector<float> xc, yc;
    for (float x = -1.f + tw / 2; x < 1.f; x += tw) { xc.push_back(x); if (xc.size() > 4096) break; }
    for (float y = -1.f + th / 2; y < 1.f; y += th) { yc.push_back(y); if (yc.size() > 4096) break; }
    const uint32_t tiles_w = (uint32_t)std::ceil(2.f / tw), tiles_h = (uint32_t)std::ceil(2.f / th); // types.h:280
    if (xc.size() > 4096 || yc.size() > 4096 || tiles_w > 4096 || tiles_h > 4096)
        return fail(c, VRT_HIP_ERR_INVALID, "tile_gaussians: more than 4096 tiles per axis");
    // The reference indexes tiles.gaussians[ty*tiles.w + tx] for ty < tiles.h, tx < tiles.w (rt.h:356) while the
    // float loops produced xc.size() tiles per row; they agree unless 2/t is not representable.  Keep tiles.w x
    // tiles.h tiles, each tested against the centre the loops would have produced for that row/column.
    while (xc.size() < tiles_w) xc.push_back(xc.empty() ? -1.f + tw / 2 : xc.back() + tw);
    while (yc.size() < tiles_h) yc.push_back(yc.empty() ? -1.f + th / 2 : yc.back() + th);
    const size_t nt = (size_t)tiles_w * tiles_h;
    if (nt * (size_t)std::max(c->n, 1u) > (size_t)1 << 31)
        return fail(c, VRT_HIP_ERR_NOMEM, "tile_gaussians: tiles x gaussians too large");
    HIPCHK(c, c->xc.reserve(tiles_w)); HIPCHK(c, c->yc.reserve(tiles_h));
    HIPCHK(c, c->w_start.reserve(nt)); HIPCHK(c, c->w_count.reserve(nt)); HIPCHK(c, c->w_indices.reserve(nt * c->n));
    std::vector<uint32_t> start(nt);
    for (size_t t = 0; t < nt; ++t) start[t] = (uint32_t)(t * c->n);
    HIPCHK(c, hipMemcpy(c->xc.p, xc.data(), tiles_w * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->yc.p, yc.data(), tiles_h * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->w_start.p, start.data(), nt * 4, hipMemcpyHostToDevice));
    c->grid_tw = tw; c->grid_th = th; c->grid_n = c->n;
    if (c->tile_mode != TILES_DEVICE || c->tiles_w != tiles_w || c->tiles_h != tiles_h) c->shard_dirty = true;
    // another tile grid = other cell lists, other cells beyond dense_threshold: launch reports of the old grid say nothing
    // about this one (round-2 advisor finding: the dense launch was left out on the first frame of a new grid)
    c->reset_seq = c->frame_seq;
    c->tile_mode = TILES_DEVICE; c->tw = tw; c->th = th; c->tiles_w = tiles_w; c->tiles_h = tiles_h;
    return VRT_HIP_OK;
}

BinArgs bin_args(const vrt_hip_ctx *c)
{
    BinArgs a{};
    a.mu_sig = c->mu_sig.p; a.gA = c->gA.p; a.gB = c->gB.p; a.n = c->n;
    // the chunk test costs a round trip of its own (the rows can only be asked for after it): worth it where the per-Gaussian pass is long
    a.chunks = (c->use_chunks == 2 || (c->use_chunks == 1 && c->n > 8192u)) ? c->gChunk.p : nullptr;
    for (int i = 0; i < 16; ++i) a.V.m[i] = c->view[i];
    a.xc = c->xc.p; a.yc = c->yc.p; a.tw = c->tw; a.th = c->th; a.tiles_w = c->tiles_w;
    return a;
}

// The single-tile "everything" list of the untiled overloads (rt.h:227-228, 315-316).
int ensure_none_ref_lists(vrt_hip_ctx *c)
{
    if (c->tile_mode != TILES_NONE || c->ref_valid) return VRT_HIP_OK;
    HIPCHK(c, c->ref_start.reserve(1)); HIPCHK(c, c->ref_count.reserve(1));
    const uint32_t zero = 0, n = c->n;
    HIPCHK(c, hipMemcpy(c->ref_start.p, &zero, 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->ref_count.p, &n, 4, hipMemcpyHostToDevice));
    c->ref_valid = true;
    return VRT_HIP_OK;
}

// Reference-semantics lists of the device binning, for queries only (get_tile_counts / get_tile_indices).
int ensure_device_ref_lists(vrt_hip_ctx *c)
{
    if (c->tile_mode != TILES_DEVICE || c->ref_valid) return VRT_HIP_OK;
    const size_t nt = (size_t)c->tiles_w * c->tiles_h;
    HIPCHK(c, c->ref_count.reserve(nt)); HIPCHK(c, c->ref_indices.reserve(nt * c->n));
    BinArgs a = bin_args(c);
    a.refine = 0;
    a.out_start = c->w_start.p; a.out_indices = c->ref_indices.p; a.out_count = c->ref_count.p;
    launch_build_tile_lists(a, FuseArgs{}, false, (uint32_t)nt, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->ref_valid = true;
    return VRT_HIP_OK;
}

TileLists work_lists(const vrt_hip_ctx *c);
int rebuild_shard(vrt_hip_ctx *c);

// Queue counters come in two sets used by alternate list generations: a fused list kernel ADDS to its set
// (cleared one generation earlier by its predecessor) and clears the other set for its successor -- no memset
// node on the per-frame path.
CellGrid cell_grid(const vrt_hip_ctx *c)
{
    CellGrid g{};
    uint32_t *cnt = c->c_counters.p + 8 * (c->list_gen & 1);
    g.cells_x = c->cells_x; g.cells_y = c->cells_y; g.cstride = c->cstride;
    g.count = c->c_count.p; g.indices = c->c_indices.p; g.active = c->c_active.p; g.n_cells = c->n_cells;
    g.dense = c->c_dense.p; g.dense_sorted = c->c_dense_sorted.p; g.scratch = c->c_scratch.p; g.slot = c->c_slot.p;
    g.n_active = cnt; g.n_dense = cnt + 2;
    g.n_light = cnt + 1; g.light_threshold = c->lists_light; // as the lists in the buffers were built
    g.dense_next = cnt + 3;
    g.overflow = c->c_overflow.p; g.n_overflow = cnt + 4;
    g.table_hx = table_on(c) ? c->table_hx : 0.f; g.table_budget = c->table_budget; g.table_adapt = c->table_adapt; g.table_room = c->table_room;
    g.claim_early = c->claim_early;
    // prune_list sums sigma*mag*exp(-x) in units of the TILE level's eps (cull_x = ln(sigma*mag / eps_eff), rebuild_tables)
    g.prune_budget = (c->cull_eps > 0.f) ? c->cull_prune * (c->cull_ref_n > 0.f ? c->cull_ref_n : 4096.f / 3.f) * std::max(1.f, (float)c->n / 4096.f) / c->albedo_scale : 0.f;
    g.dense_threshold = 96; // longer cell lists go straight to the 16-waves-per-block kernel (must be <= PCAP)
    g.feedback = c->d_fb;
    g.dense_is_sorted = 1;
    return g;
}

// `target`: where this frame is rendered to; when the fused list kernel runs it clears the cells nothing can reach.
int build_work_lists(vrt_hip_ctx *c, const float origin[3], hipStream_t st, bool use_shard, RenderTarget *target)
{
    if (!c->lists_dirty && c->lists_for_shard == (int)use_shard) return VRT_HIP_OK;
    int rc = ensure_none_ref_lists(c);
    if (rc) return rc;
    const TileLists geo = tile_geometry(c);
    const size_t nt = (size_t)geo.tiles_w * geo.tiles_h;
    // the tile cone is built from corner rays: needs pinhole rays (always true for in-kernel ray generation)
    // ... and tiles that are rectangles of the image: with the reference's truncated tile size the row stride
    // tile_w*tiles_w can differ from the width (rt.h:364-365), a tile's rows then drift sideways and wrap around the
    // image edge, and its rays are no cone around its corner rays (found by tests/fuzz_parity.py: 33x100, 5 tiles)
    const bool refine = (!c->plane_mode || c->plane_affine) && geo.stride == c->w;

    // geometry of the second level and its buffers
    uint32_t n_local = (uint32_t)nt;
    const uint32_t *tile_map = nullptr;
    if (use_shard) {
        if ((rc = rebuild_shard(c))) return rc;
        n_local = c->n_local; tile_map = c->tile_map.p;
    }
    c->cells_x = (geo.tile_w + CELL - 1) / CELL; c->cells_y = (geo.tile_h + CELL - 1) / CELL;
    c->n_cells = n_local * c->cells_x * c->cells_y;
    if (c->n_cells > ACTIVE_CELL_MASK) return fail(c, VRT_HIP_ERR_INVALID, "tile grid: more than 2^24 cells of 32 x 32 pixels on one device");
    c->cstride = std::max(1u, std::min(c->n, 4096u));
    HIPCHK(c, c->c_count.reserve(c->n_cells)); HIPCHK(c, c->c_active.reserve(c->n_cells));
    HIPCHK(c, c->c_dense.reserve(c->n_cells));
    HIPCHK(c, c->c_dense_sorted.reserve(c->n_cells));
    HIPCHK(c, c->c_slot.reserve(c->n_cells));
    HIPCHK(c, c->c_scratch.reserve((size_t)c->num_cus * 4 * c->cstride)); // one slot per dense workgroup (<= 4 per CU)
    HIPCHK(c, c->c_overflow.reserve((size_t)c->n_cells * 16));
    HIPCHK(c, c->c_indices.reserve((size_t)c->n_cells * c->cstride));
    if (!c->c_counters.p) {
        HIPCHK(c, c->c_counters.reserve(16));
        HIPCHK(c, hipMemsetAsync(c->c_counters.p, 0, 16 * sizeof(uint32_t), st));
    }
    ++c->list_gen; // this build fills counter set (list_gen & 1)

    BinArgs a = bin_args(c);
    a.refine = refine ? 1 : 0;
    a.cull_ref_n = c->cull_ref_n; a.floor_x = exp_floor_x(c->exp_kind);
    a.R = ray_gen(c, origin);
    a.tile_w = geo.tile_w; a.tile_h = geo.tile_h; a.stride = geo.stride;
    c->work_is_ref = false;
    if (c->tile_mode == TILES_DEVICE && c->grid_n != c->n) { // the scene was replaced after tile_gaussians()
        HIPCHK(c, hipStreamSynchronize(st));
        if ((rc = prepare_tile_grid(c, c->tw, c->th))) return rc;
        c->last_stream = st;
        a = bin_args(c);
        a.refine = refine ? 1 : 0; a.R = ray_gen(c, origin);
        a.cull_ref_n = c->cull_ref_n; a.floor_x = exp_floor_x(c->exp_kind);
        a.tile_w = geo.tile_w; a.tile_h = geo.tile_h; a.stride = geo.stride;
    }
    if (refine) {
        std::string key((const char *)&a.R, sizeof a.R);
        const uint32_t geo_key[6] = { geo.tile_w, geo.tile_h, geo.stride, geo.tiles_w, geo.tiles_h, c->plane_gen };
        key.append((const char *)geo_key, sizeof geo_key);
        a.tiles_w = geo.tiles_w;
        // with the cells' cones when the tile's cells are filtered by the same workgroup (the fused list kernel)
        const uint32_t cpt = c->cells_x * c->cells_y;
        const uint32_t cones_cells = cpt <= (uint32_t)MAX_FUSED_CELLS ? cpt : 0u;
        const size_t rows = nt * (1 + cones_cells);
        // The table fills itself: a row is valid if it carries the tag of this camera (cone_gen); the list kernel's workgroups build the cones
        // they do not find and file them (round 3: the table's own launch cost a frame whose camera moved 5 us).  Frames of a batch get
        // theirs from one launch for the whole batch, as before.
        bool known = true;
        if (c->tile_cones.cap < 2 * rows) {
            HIPCHK(c, c->tile_cones.reserve(2 * rows));
            HIPCHK(c, hipMemsetAsync(c->tile_cones.p, 0, c->tile_cones.cap * sizeof(float4), st)); // tag 0: no camera's
            c->cone_key.clear();
        }
        if (key != c->cone_key) {
            if (++c->cone_gen == 0u) c->cone_gen = 1u;
            a.cone_gen = c->cone_gen;
            known = false;
            if (c->defer) { // a frame of a batch: one cone launch for all frames (its BinArgs are the frame's bin row)
                c->defer->do_cones = 1; c->defer->cones_tiles = a.tiles_w * geo.tiles_h;
                c->defer->cones_cx = cones_cells ? c->cells_x : 0u; c->defer->cones_cy = cones_cells ? c->cells_y : 0u;
                c->defer->cones_out = c->tile_cones.p;
                known = true;
            }
            c->cone_key = key;
        }
        a.tile_cones = c->tile_cones.p; a.cones_cells = cones_cells; a.cone_gen = c->cone_gen; a.cones_known = known ? 1 : 0;
    }
    const bool device_bin = c->tile_mode == TILES_DEVICE;
    // one fused kernel when a tile's cells fit one workgroup's waves; otherwise tile kernel + one-wave-per-cell kernel
    const bool fuse = c->cells_x * c->cells_y <= (uint32_t)MAX_FUSED_CELLS && (device_bin || refine);
    c->lists_light = (fuse && !(target && target->sparse)) ? c->light_cells : 0u;
    FuseArgs f{};
    f.enabled = fuse ? 1 : 0;
    f.tile_map = fuse ? tile_map : nullptr;
    f.C = cell_grid(c);
    if (fuse && target) { f.O = *target; f.do_clear = target->sparse ? 0 : 1; }
    c->timeline_tiles = 0;
    if (fuse && getenv("VRT_HIP_TIMELINE")) {
        c->timeline_tiles = n_local;
        HIPCHK(c, c->d_timeline_lists.reserve((size_t)n_local * 8));
        HIPCHK(c, hipMemsetAsync(c->d_timeline_lists.p, 0, (size_t)n_local * 8 * sizeof(unsigned long long), st));
        f.timeline = c->d_timeline_lists.p;
    }
    uint32_t *other_set = c->c_counters.p + 8 * ((c->list_gen + 1) & 1);
    a.zero8 = fuse ? nullptr : c->c_counters.p + 8 * (c->list_gen & 1);
    if (device_bin) {
        a.out_start = c->w_start.p; a.out_indices = c->w_indices.p; a.out_count = c->w_count.p;
    } else if (refine) {
        const size_t total = c->tile_mode == TILES_NONE ? c->n : c->ref_indices.cap;
        HIPCHK(c, c->w_count.reserve(nt)); HIPCHK(c, c->w_indices.reserve(total));
        a.in_start = c->ref_start.p; a.in_count = c->ref_count.p;
        a.in_indices = c->tile_mode == TILES_NONE ? c->iota.p : c->ref_indices.p;
        a.tiles_w = geo.tiles_w;
        a.out_start = c->ref_start.p; a.out_indices = c->w_indices.p; a.out_count = c->w_count.p;
    } else {
        c->work_is_ref = true;
    }
    if (fuse) {
        a.next_zero8 = other_set; // cleared for the next generation by workgroup 0
        if (n_local && c->defer) {
            c->defer->bin = a; c->defer->fuse = f;
            c->deferred.lists = true; c->deferred.from_list = !device_bin; c->deferred.list_grid = n_local;
        } else if (n_local) {
            if (c->prep_pending) { a.prep_gA = c->gA.p; c->prep_pending = false; } // a.R.origin is the origin prep_frame() noted
            launch_build_tile_lists(a, f, !device_bin, n_local, st);
        } else {
            // a rank that owns no tile (more ranks than tiles) launches no list kernel: nobody adds to this generation's
            // counters and nobody clears the next one's -- do both here, or the next render would add to stale counts
            // (found by tests/fuzz_parity.py: 4 tiles on 5 and 8 ranks)
            HIPCHK(c, hipMemsetAsync(c->c_counters.p, 0, 16 * sizeof(uint32_t), st));
        }
        if (target) target->cleared = 1;
    } else {
        if (c->defer) return fail(c, VRT_HIP_ERR_INVALID, "frame batch: tiles of more than 64 cells (or rays that are no pinhole bundle) "
                                                          "need the two-kernel list path, which is not batched");
        if (!c->work_is_ref) {
            if (c->prep_pending) { a.prep_gA = c->gA.p; c->prep_pending = false; }
            launch_build_tile_lists(a, f, !device_bin, (uint32_t)nt, st);
        }
        else HIPCHK(c, hipMemsetAsync(c->c_counters.p + 8 * (c->list_gen & 1), 0, 8 * sizeof(uint32_t), st));
        HIPCHK(c, hipGetLastError());
        if ((rc = flush_prep(c, st))) return rc; // the cell kernel reads the per-origin table: if no tile kernel wrote it, now
        launch_build_cell_lists(tables(c), work_lists(c), cell_grid(c), a.R, tile_map, c->n_cells, refine ? 1 : 0,
                                (target && target->sparse) ? target->keys : nullptr, st);
        if (target && target->sparse) target->cleared = 1; // a sparse shard stores no empty cells: nothing to clear
        // the set the NEXT generation will add to (if it is a fused one) must be clear
        HIPCHK(c, hipMemsetAsync(other_set, 0, 8 * sizeof(uint32_t), st));
    }
    HIPCHK(c, hipGetLastError());
    c->lists_dirty = false;
    c->lists_fresh = true;
    c->lists_for_shard = (int)use_shard;
    return VRT_HIP_OK;
}

TileLists work_lists(const vrt_hip_ctx *c)
{
    TileLists t = tile_geometry(c);
    t.cull_ref_n = c->cull_ref_n;
    t.floor_x = exp_floor_x(c->exp_kind);
    if (c->tile_mode == TILES_DEVICE) {
        t.start = c->w_start.p; t.count = c->w_count.p; t.indices = c->w_indices.p;
    } else if (c->work_is_ref) {
        t.start = c->ref_start.p; t.count = c->ref_count.p;
        t.indices = c->tile_mode == TILES_NONE ? c->iota.p : c->ref_indices.p;
    } else {
        t.start = c->ref_start.p; t.count = c->w_count.p; t.indices = c->w_indices.p;
    }
    return t;
}

// Owner of tile t = (tx, ty): the ranks form an a x b brick (a * b = world, a >= b as square as the divisors allow) that
// tiles the tile grid, every row of bricks shifted by half a brick against the one above:
// owner = (tx + (a / 2) * (ty / b)) % a + a * (ty % b).  Every a x b window of tiles, wherever it lies, holds every rank once,
// so an object that covers a few tiles in the middle of the frame -- `-g 64 -w 2048` lights 4 x 4 of the 16 x 16 tiles, the
// outer ones partly -- is spread evenly: the busiest of 8 ranks gets 12.8 % of its cells (ideal 12.5 %; unshifted bricks
// 14.3 %, because a rank's two tiles then share a column; dealing tiles along diagonals, as round 1 did, 25 %).
// sharding.py mirrors this.
inline int shard_owner(uint32_t t, uint32_t tiles_w, int world)
{
    uint32_t b = 1;
    for (uint32_t d = 1; d * d <= (uint32_t)world; ++d)
        if ((uint32_t)world % d == 0) b = d;
    const uint32_t a = (uint32_t)world / b;
    const uint32_t tx = t % tiles_w, ty = t / tiles_w;
    return (int)((tx + (a / 2) * (ty / b)) % a + a * (ty % b));
}

int rebuild_shard(vrt_hip_ctx *c)
{
    if (!c->shard_dirty) return VRT_HIP_OK;
    const bool tiled = c->tile_mode != TILES_NONE;
    const uint32_t tiles_w = tiled ? c->tiles_w : 1, ntiles = tiled ? c->tiles_w * c->tiles_h : 1;
    std::vector<std::vector<uint32_t>> owned(c->world);
    for (uint32_t t = 0; t < ntiles; ++t) owned[shard_owner(t, tiles_w, c->world)].push_back(t);
    size_t slots = 0;
    for (auto &v : owned) slots = std::max(slots, v.size());
    std::vector<uint32_t> slot_tiles((size_t)c->world * slots, 0xFFFFFFFFu);
    for (int r = 0; r < c->world; ++r)
        for (size_t k = 0; k < owned[r].size(); ++k) slot_tiles[(size_t)r * slots + k] = owned[r][k];
    c->n_local = (uint32_t)owned[c->rank].size();
    c->n_slots = (uint32_t)slots;
    HIPCHK(c, c->tile_map.reserve(c->n_local)); HIPCHK(c, c->slot_tiles.reserve(slot_tiles.size()));
    if (c->n_local)
        HIPCHK(c, hipMemcpy(c->tile_map.p, owned[c->rank].data(), c->n_local * 4, hipMemcpyHostToDevice));
    if (!slot_tiles.empty())
        HIPCHK(c, hipMemcpy(c->slot_tiles.p, slot_tiles.data(), slot_tiles.size() * 4, hipMemcpyHostToDevice));
    c->shard_dirty = false;
    return VRT_HIP_OK;
}

int check_ready(vrt_hip_ctx *c)
{
    if (!c) return VRT_HIP_ERR_INVALID;
    if (!c->rays_set || !c->w || !c->h) return fail(c, VRT_HIP_ERR_INVALID, "render: call vrt_hip_set_plane/set_camera first");
    return VRT_HIP_OK;
}

enum OutMode { OUT_RASTER = 0, OUT_COMPACT = 1, OUT_SPARSE = 2 };
uint32_t sparse_capacity(vrt_hip_ctx *c); // cells a sparse shard of this context can hold (the same on every rank)

int render_common(vrt_hip_ctx *c, const float origin[3], int pack_flags, uint32_t *d_image, float4 *d_rad,
                  hipStream_t st, int out_mode)
{
    const bool shard_compact = out_mode != OUT_RASTER; // compact and sparse targets hold this rank's tiles only
    int rc = check_ready(c);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    const TileLists geo = tile_geometry(c);
    if (geo.tile_w == 0 || geo.tile_h == 0) return fail(c, VRT_HIP_ERR_INVALID, "render: tile size is 0 pixels");
    const bool use_shard = c->world > 1 || shard_compact;
    if (c->last_stream && c->last_stream != st) {
        // another stream than the last frame's: the list kernels of this frame rewrite lists and queue counters that
        // the previous frame's kernels may still be reading
        if (hipStreamSynchronize(c->last_stream) != hipSuccess) (void)hipGetLastError();
    }
    c->last_stream = st;
    hipEvent_t *tev = nullptr;
    const bool timed_frame = !c->defer && c->timing_on && (c->timing_frame++ % c->timing_period) == 0;
    if (timed_frame) {
        if (c->tev.empty()) {
            c->tev.resize(4 * vrt_hip_ctx::TIMING_RING);
            for (auto &e : c->tev) HIPCHK(c, hipEventCreate(&e));
        }
        tev = &c->tev[4 * (c->timing_count % vrt_hip_ctx::TIMING_RING)];
        if (c->timing_full) HIPCHK(c, hipEventRecord(tev[0], st));
    }
    if ((rc = prep_frame(c, origin, st))) return rc;
    RenderTarget o{};
    o.image = d_image; o.radiance = d_rad; o.pack_flags = pack_flags;
    o.stats = c->stats_on ? c->d_stats.p : nullptr;
    o.compact = out_mode == OUT_COMPACT ? 1 : 0;
    if (use_shard) {
        if ((rc = rebuild_shard(c))) return rc;
        o.tile_map = c->tile_map.p; o.n_local_tiles = c->n_local;
    } else {
        o.tile_map = nullptr; o.n_local_tiles = geo.tiles_w * geo.tiles_h;
    }
    uint32_t sparse_cap = 0;
    if (out_mode == OUT_SPARSE) {
        // d_image is a sparse shard buffer: header | keys | pixels of the stored cells (vrt_kernels.h, RenderTarget)
        sparse_cap = sparse_capacity(c);
        o.sparse = 1; o.sparse_hdr = d_image; o.keys = d_image + SPARSE_HDR_WORDS;
        o.image = d_image + sparse_pixel_offset(sparse_cap);
        o.cleared = 1;
        c->lists_dirty = true; // the list kernel files the cell keys into THIS buffer
        o.sparse_cap = sparse_cap;
    }
    if (c->retain_next && out_mode == OUT_RASTER && !use_shard && !c->defer) { o.stamp = c->own_stamp.p; o.stamp_seq = c->own_seq; }
    if ((rc = build_work_lists(c, origin, st, use_shard, &o))) return rc;
    if ((rc = flush_prep(c, st))) return rc; // no list kernel took the per-origin table along (lists unchanged, caller-made lists used as they are, a batch)
    if (o.stamp && !o.cleared) c->own_seq = 0; // the list kernel of this frame was not the fused one: nobody kept the stamps
    const TileLists t = work_lists(c);
    if (o.stats) {
        HIPCHK(c, hipMemsetAsync(c->d_stats.p, 0, 32 * sizeof(unsigned long long), st));
        HIPCHK(c, hipMemsetAsync(c->d_stats.p + 8, 0xFF, sizeof(unsigned long long), st)); // running minimum
    }
    c->timeline_items = 0;
    if (getenv("VRT_HIP_TIMELINE")) {
        c->timeline_items = (size_t)c->n_cells * 16;
        HIPCHK(c, c->d_timeline.reserve(c->timeline_items * 5));
        HIPCHK(c, hipMemsetAsync(c->d_timeline.p, 0, c->timeline_items * 5 * sizeof(unsigned long long), st));
        o.timeline = c->d_timeline.p;
    }
    const uint32_t bx = (t.tile_w + BLOCK_W - 1) / BLOCK_W, by = (t.tile_h + BLOCK_H - 1) / BLOCK_H;
    c->last.blocks = (uint64_t)o.n_local_tiles * bx * by;
    c->last.rays = (uint64_t)o.n_local_tiles * t.tile_w * t.tile_h;
    // persistent grid: 12 one-wave workgroups per CU (three per SIMD at 145 VGPRs), never more than there are blocks
    const uint32_t grid = (uint32_t)std::min<uint64_t>((uint64_t)c->n_cells * 16u,
                                                       c->render_grid_override ? (uint64_t)c->render_grid_override
                                                                               : (uint64_t)c->num_cus * std::max(1, c->render_waves_per_cu));
    if (out_mode == OUT_SPARSE && grid == 0) // a rank without cells launches no render kernel: nobody writes the header
        HIPCHK(c, hipMemsetAsync(d_image, 0, SPARSE_HDR_WORDS * sizeof(uint32_t), st));
    if (!c->lists_fresh) // a re-render from unchanged lists: only the dense kernel's work counters need a reset
        HIPCHK(c, hipMemsetAsync(c->c_counters.p + 8 * (c->list_gen & 1) + 3, 0, 4 * sizeof(uint32_t), st));
    c->lists_fresh = false;
    // How large a dense launch?  The 16-waves-per-block kernel always runs behind the one-wave kernel (which kernel
    // shades a block depends on the block alone, so the image never depends on this heuristic); but a full launch --
    // queue sort + one 1024-thread workgroup per CU -- costs ~12 us even with empty queues.  Frames report
    // (asynchronously, CellGrid::feedback) what their dense kernel found; once a report has arrived from a frame
    // launched at least two frames after the last change of scene, rays, camera or options, and it says "nothing",
    // the launch shrinks to one workgroup and skips the sort.  A wrong guess costs speed only.
    bool expect_dense = true;
    bool camera_moved = false;
    if (c->h_fb && !c->stats_on && (int32_t)(c->h_fb[3] - c->reset_seq) >= 2) {
        expect_dense = c->h_fb[0] > 0 || c->h_fb[2] > 0;
        camera_moved = (int32_t)(c->h_fb[3] - c->cam_seq) < 2;
    }
    // Not a guess: which blocks are dense is a function of scene, options, rays, camera, tile grid and shard.  A report
    // from a frame that was launched AFTER the last change of any of them (sequence number above reset_seq and cam_seq)
    // and that found no dense cell and no handed-over block says the same of this frame: the dense launch -- 4.7 us of a
    // 47-us serial frame even for one idle workgroup, which also waits for 61 KB of LDS while other frames' block kernels
    // fill the CUs -- is left out.  Any change brings it back until a frame of the new state has reported.
    bool no_dense_work = false;
    if (c->skip_idle_dense && c->h_fb && !c->stats_on) {
        const uint32_t seen = c->h_fb[3]; // read first: what is read after it is at least as new
        no_dense_work = (int32_t)(seen - c->reset_seq) >= 1 && (int32_t)(seen - c->cam_seq) >= 1 && c->h_fb[0] == 0 && c->h_fb[2] == 0;
    }
    if (no_dense_work) ++c->last.dense_launch_skips;
    CellGrid cg = cell_grid(c);
    // the block kernel's variant that claims its next queue entry early: for frames with (by an earlier frame's report, however old:
    // speed only) at least grid / claim_early more blocks than the grid has waves
    {
        const uint32_t seen_blocks = (c->h_fb && !c->stats_on) ? c->h_fb[1] : 0u;
        const bool many = c->claim_early > 0 && seen_blocks > grid && (uint64_t)(seen_blocks - grid) * (uint32_t)c->claim_early >= grid;
        cg.claim_early = many ? c->claim_early : 0;
    }
    cg.dense_is_sorted = expect_dense ? 1 : 0;
    cg.frame_seq = ++c->frame_seq;
    if (!c->c_rq.p) {
        HIPCHK(c, c->c_rq.reserve(2 * RQ_N * RQ_STRIDE));
        HIPCHK(c, hipMemsetAsync(c->c_rq.p, 0, 2 * RQ_N * RQ_STRIDE * sizeof(uint32_t), st));
    }
    if (grid) ++c->rq_gen; // a skipped launch clears nothing: the sets must not swap
    cg.rq = c->c_rq.p + (c->rq_gen & 1) * RQ_N * RQ_STRIDE;
    cg.rq_next = c->c_rq.p + ((c->rq_gen + 1) & 1) * RQ_N * RQ_STRIDE;
    if (c->defer) {
        // a frame of a batch: the launches are made once for all frames by vrt_hip_frame_batch_device
        uint32_t dense_grid = (uint32_t)std::min<uint64_t>((uint64_t)c->n_cells * 16u, (uint64_t)c->num_cus * (16 / std::min(c->dense_waves, 16)));
        if (!expect_dense) dense_grid = std::min(dense_grid, (uint32_t)(camera_moved ? std::max(c->dense_idle_grid, c->num_cus / 4) : c->dense_idle_grid));
        if (no_dense_work) dense_grid = 0;
        FrameArgs &fa = *c->defer;
        fa.S = tables(c); fa.T = t; fa.C = cg; fa.R = ray_gen(c, origin); fa.O = o;
        c->deferred.render = true; c->deferred.render_grid = grid; c->deferred.order = expect_dense && !no_dense_work; c->deferred.dense_grid = dense_grid;
        return VRT_HIP_OK;
    }
    if (tev) HIPCHK(c, hipEventRecord(tev[1], st));
    launch_render(tables(c), t, cg, ray_gen(c, origin), o, grid, c->exp_kind, c->erf_kind, st);
    if (tev) HIPCHK(c, hipEventRecord(tev[2], st));
    // dense queue: 16-wave workgroups pull blocks until the queue is empty (they exit at once if it is)
    if (expect_dense && !no_dense_work) launch_order_dense(cg, st);
    if (!no_dense_work) {
        uint32_t dense_grid = (uint32_t)std::min<uint64_t>((uint64_t)c->n_cells * 16u, (uint64_t)c->num_cus * (16 / std::min(c->dense_waves, 16)));
        if (!expect_dense) dense_grid = std::min(dense_grid, (uint32_t)(camera_moved ? std::max(c->dense_idle_grid, c->num_cus / 4) : c->dense_idle_grid));
        if (table_on(c)) {
            // table mode (the default): the table kernel takes the whole dense queue; a block it declines it shades exactly itself
            // (dense_shade_block in its own LDS): ONE dense-path launch per frame (rounds 1-3: an exact launch behind it, idle in
            // every frame of a moving camera)
            launch_render_table(tables(c), t, cg, ray_gen(c, origin), o, std::min<uint32_t>(dense_grid, (uint32_t)c->num_cus), c->exp_kind, c->erf_kind, st);
        } else {
            launch_render_dense(tables(c), t, cg, ray_gen(c, origin), o, dense_grid, c->dense_waves, c->exp_kind, c->erf_kind, st);
        }
    }
    if (tev) {
        if (c->timing_full) HIPCHK(c, hipEventRecord(tev[3], st));
        ++c->timing_count;
    }
    HIPCHK(c, hipGetLastError());
    return VRT_HIP_OK;
}

// Retained frame buffer (vrt_hip_frame's own buffer; vrt_hip_frame_retained_device for a caller's): `image` still holds this
// context's previous frame at this geometry, so the list kernel clears only the cells that went dark (RenderTarget::stamp).
// One history per context, for ONE buffer: another buffer, image size, tile grid or background starts a new one with a
// full clear.  Sets retain_next for the render_common call that follows.
int retained_begin(vrt_hip_ctx *c, uint32_t *image, float tw, float th, int pack_flags, hipStream_t st)
{
    const size_t npix = (size_t)c->w * c->h;
    const uint32_t tiles_w = (uint32_t)std::ceil(2.f / tw), tiles_h = (uint32_t)std::ceil(2.f / th);
    const uint32_t tile_w = (uint32_t)(uint64_t)(c->w * tw / 2.f), tile_h = (uint32_t)(uint64_t)(c->h * th / 2.f);
    const uint32_t cx = (tile_w + CELL - 1) / CELL, cy = (tile_h + CELL - 1) / CELL;
    const size_t cells = (size_t)tiles_w * tiles_h * cx * cy;
    vrt_hip_ctx::OwnGeometry sig;
    sig.w = c->w; sig.h = c->h; sig.tiles_w = tiles_w; sig.tiles_h = tiles_h; sig.tile_w = tile_w; sig.tile_h = tile_h;
    sig.background = (pack_flags & VRT_ALPHA_COMPUTED) ? 0u : 0xFF000000u; sig.image = image;
    static const bool retain_on = [] { const char *e = getenv("VRT_HIP_RETAIN_FRAME"); return !e || atoi(e) != 0; }();
    c->retain_next = retain_on && cells > 0 && cells < (1u << 28) && tiles_w <= 4096 && tiles_h <= 4096 && c->world == 1;
    if (!c->retain_next) { c->own_seq = 0; return VRT_HIP_OK; }
    if (c->last_stream && c->last_stream != st) {
        // the previous frame ran on another stream: its list and block kernels may still be writing the stamps and the image that the
        // memsets below reset on THIS stream (render_common's own hand-over comes after this function)
        if (hipStreamSynchronize(c->last_stream) != hipSuccess) (void)hipGetLastError();
        c->last_stream = st;
    }
    if (!(sig == c->own_sig) || c->own_seq == 0 || c->own_seq >= 0xFFFFFFF0u || c->own_stamp.cap < cells) {
        if (c->own_stamp.cap < cells) { int rc = quiesce(c); if (rc) return rc; } // frames in flight write the old stamp buffer
        HIPCHK(c, c->own_stamp.reserve(cells));
        HIPCHK(c, hipMemsetAsync(c->own_stamp.p, 0, cells * sizeof(uint32_t), st));
        if (!(sig == c->own_sig)) HIPCHK(c, hipMemsetAsync(image, 0, npix * 4, st)); // pixels no tile of the NEW grid covers read 0
        c->own_sig = sig;
        c->own_seq = 1; // stamps of 0 = "never lit": with seq 1 every empty cell compares against 0 = seq - 1 and is cleared
    } else {
        ++c->own_seq;
    }
    return VRT_HIP_OK;
}

// Are the plane arrays an affine function of (row, column)?  Then rays are a pinhole bundle and the
// corner rays of a tile bound its cone.  Anything else disables the tile-level cull (the per-block cull
// works from the actual lane rays and stays exact for arbitrary arrays).
bool plane_is_affine(uint32_t w, uint32_t h, const float *xs, const float *ys, const float *zs)
{
    const float *a[3] = { xs, ys, zs };
    for (int k = 0; k < 3; ++k) {
        const float *p = a[k];
        const double p00 = p[0];
        const double dx = w > 1 ? ((double)p[w - 1] - p00) / (w - 1) : 0.0;
        const double dy = h > 1 ? ((double)p[(size_t)(h - 1) * w] - p00) / (h - 1) : 0.0;
        double scale = 1.0;
        for (uint32_t i = 0; i < h; i += (h > 64 ? h / 64 : 1)) scale = std::max(scale, std::fabs((double)p[(size_t)i * w]));
        const double tol = 1e-5 * scale;
        for (uint32_t i = 0; i < h; ++i)
            for (uint32_t j = 0; j < w; ++j)
                if (std::fabs((double)p[(size_t)i * w + j] - (p00 + dx * j + dy * i)) > tol) return false;
    }
    return true;
}

} // namespace

extern "C" {

const char *vrt_hip_version(void) { return "vrt_hip 0.2 (gfx950)"; }

int vrt_hip_create(int device, vrt_hip_ctx **out)
{
    if (!out) return fail(nullptr, VRT_HIP_ERR_INVALID, "create: out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0)
        return fail(nullptr, VRT_HIP_ERR_NO_DEVICE, "create: no HIP device visible (libvrt_hip has no CPU fallback)");
    if (device < 0 || device >= count) return fail(nullptr, VRT_HIP_ERR_INVALID, "create: device index out of range");
    HIPCHK(nullptr, hipSetDevice(device));
    vrt_hip_ctx *c = new (std::nothrow) vrt_hip_ctx();
    if (!c) return fail(nullptr, VRT_HIP_ERR_NOMEM, "create: out of host memory");
    c->device = device;
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) c->num_cus = cus;
    {
        void *hp = nullptr, *dp = nullptr;
        if (hipHostMalloc(&hp, 4 * sizeof(uint32_t), hipHostMallocMapped) == hipSuccess &&
            hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess) {
            memset(hp, 0, 4 * sizeof(uint32_t));
            c->h_fb = (volatile uint32_t *)hp; c->d_fb = (uint32_t *)dp;
        } else if (hp) {
            (void)hipHostFree(hp);
        }
    }
    if (const char *e = getenv("VRT_HIP_RENDER_WAVES")) { // one-wave kernel: persistent waves per CU (tuning knob)
        const int v = atoi(e);
        if (v >= 1 && v <= 16) c->render_waves_per_cu = v;
    }
    if (const char *e = getenv("VRT_HIP_RENDER_GRID")) c->render_grid_override = (uint32_t)std::max(0, atoi(e));
    if (const char *e = getenv("VRT_HIP_CULL_REF_N")) c->cull_ref_n = fmaxf(0.f, (float)atof(e));
    if (const char *e = getenv("VRT_HIP_CHUNKS")) c->use_chunks = std::max(0, std::min(2, atoi(e)));
    if (const char *e = getenv("VRT_HIP_CULL_PRUNE")) c->cull_prune = fmaxf(0.f, (float)atof(e));
    if (const char *e = getenv("VRT_HIP_TABLE_STEP")) { const float v = (float)atof(e); if (v >= 0.f && v <= 1.f) c->table_hx = v; }
    if (const char *e = getenv("VRT_HIP_TABLE_ROOM")) { const float v = (float)atof(e); if (v > 0.f && v <= 10.f) c->table_room = v; }
    if (const char *e = getenv("VRT_HIP_TABLE_ADAPT")) { const float v = (float)atof(e); if (v >= 1.f && v <= 3.f) c->table_adapt = v; }
    if (const char *e = getenv("VRT_HIP_TABLE_BUDGET")) { const float v = (float)atof(e); if (v > 0.f) c->table_budget = v; }
    if (const char *e = getenv("VRT_HIP_CLAIM_EARLY")) c->claim_early = std::max(0, atoi(e));
    if (const char *e = getenv("VRT_HIP_DENSE_SKIP")) c->skip_idle_dense = atoi(e) != 0;
    if (const char *e = getenv("VRT_HIP_DENSE_WAVES")) {
        const int v = atoi(e);
        if (v == 4 || v == 8 || v == 16 || v == 17) c->dense_waves = v; // 17 = 16 waves without saturation skipping (A/B)
    }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess ||
        c->d_stats.reserve(32) != hipSuccess) {
        delete c;
        return fail(nullptr, VRT_HIP_ERR_HIP, "create: stream/event creation failed");
    }
    *out = c;
    return VRT_HIP_OK;
}

void vrt_hip_destroy(vrt_hip_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)quiesce(c); // frames still in flight on the context's stream or on the caller's last stream read these buffers
    for (auto &b : c->soa) b.release();
    c->mu_sig.release(); c->gA.release(); c->gB.release(); c->gC.release(); c->gD.release(); c->iota.release();
    c->ref_start.release(); c->ref_count.release(); c->ref_indices.release();
    c->w_start.release(); c->w_count.release(); c->w_indices.release(); c->xc.release(); c->yc.release();
    c->c_count.release(); c->c_indices.release(); c->c_active.release(); c->c_dense.release(); c->c_dense_sorted.release(); c->c_scratch.release(); c->c_overflow.release(); c->c_counters.release(); c->c_rq.release(); c->c_slot.release();
    c->xs.release(); c->ys.release(); c->zs.release(); c->tile_map.release(); c->slot_tiles.release();
    c->d_image.release(); c->own_stamp.release(); c->d_rad.release(); c->d_stats.release(); c->d_timeline.release(); c->d_timeline_lists.release();
    if (c->h_fb) (void)hipHostFree((void *)c->h_fb);
    for (auto &r : c->retained) r.stamp.release();
    c->tile_cones.release();
    if (c->batch_host) (void)hipHostFree(c->batch_host);
    if (c->batch_dev) (void)hipFree(c->batch_dev);
    for (auto &e : c->batch_copied) if (e) (void)hipEventDestroy(e);
    for (auto &e : c->tev) (void)hipEventDestroy(e);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char *vrt_hip_last_error(const vrt_hip_ctx *c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int vrt_hip_set_gaussians(vrt_hip_ctx *c, size_t n, const float *mu_x, const float *mu_y, const float *mu_z,
                          const float *ar, const float *ag, const float *ab, const float *aa, const float *sigma,
                          const float *mag)
{
    if (!c) return VRT_HIP_ERR_INVALID;
    if (n > 0xFFFFFFF0ull) return fail(c, VRT_HIP_ERR_INVALID, "set_gaussians: too many Gaussians");
    const float *src[9] = { mu_x, mu_y, mu_z, ar, ag, ab, aa, sigma, mag };
    for (int i = 0; i < 9; ++i)
        if (n && !src[i] && i != 6) return fail(c, VRT_HIP_ERR_INVALID, "set_gaussians: NULL array");
    HIPCHK(c, hipSetDevice(c->device));
    { int rc = quiesce(c); if (rc) return rc; }
    for (int i = 0; i < 9; ++i) {
        HIPCHK(c, c->soa[i].reserve(n));
        if (n && src[i]) HIPCHK(c, hipMemcpy(c->soa[i].p, src[i], n * sizeof(float), hipMemcpyHostToDevice));
    }
    c->has_alpha = aa != nullptr;
    // the cull bounds count a dropped Gaussian's emission at albedo <= 1 (the reference's range): brighter scenes shrink the prune's budget
    c->albedo_scale = 1.f;
    for (int ch = 3; ch <= 6; ++ch)
        if (src[ch])
            for (size_t i = 0; i < n; ++i) { const float v = fabsf(src[ch][i]); if (v > c->albedo_scale && v < INFINITY) c->albedo_scale = v; }
    c->n = (uint32_t)n;
    ++c->state_gen;
    c->reset_seq = c->frame_seq;
    c->tables_dirty = true;
    c->lists_dirty = true;
    if (c->tile_mode == TILES_HOST) {
        // caller-made lists index the scene they were validated against (set_tiles checks every index < n): they do
        // not carry over to another scene -- back to untiled until set_tiles / tile_gaussians is called again
        c->tile_mode = TILES_NONE; c->tw = c->th = 2.f; c->tiles_w = c->tiles_h = 1;
        c->shard_dirty = true;
    }
    c->ref_valid = false;
    return VRT_HIP_OK;
}

int vrt_hip_set_gaussians_aos(vrt_hip_ctx *c, size_t n, const void *gaussians)
{
    if (!c) return VRT_HIP_ERR_INVALID;
    if (n && !gaussians) return fail(c, VRT_HIP_ERR_INVALID, "set_gaussians_aos: NULL");
    // gaussian_t: albedo[4] mu[4] sigma magnitude (types.h:195-200)
    const float *g = (const float *)gaussians;
    std::vector<float> soa[9];
    for (auto &v : soa) v.resize(n ? n : 1);
    for (size_t i = 0; i < n; ++i) {
        const float *p = g + 10 * i;
        soa[3][i] = p[0]; soa[4][i] = p[1]; soa[5][i] = p[2]; soa[6][i] = p[3];
        soa[0][i] = p[4]; soa[1][i] = p[5]; soa[2][i] = p[6];
        soa[7][i] = p[8]; soa[8][i] = p[9];
    }
    return vrt_hip_set_gaussians(c, n, soa[0].data(), soa[1].data(), soa[2].data(), soa[3].data(), soa[4].data(),
                                 soa[5].data(), soa[6].data(), soa[7].data(), soa[8].data());
}

uint64_t vrt_hip_state_generation(const vrt_hip_ctx *c) { return c ? c->state_gen : 0; }

int vrt_hip_get_image_size(const vrt_hip_ctx *c, uint32_t *w, uint32_t *h)
{
    if (!c || !c->rays_set) return VRT_HIP_ERR_INVALID;
    if (w) *w = c->w;
    if (h) *h = c->h;
    return VRT_HIP_OK;
}

int vrt_hip_copy_state(vrt_hip_ctx *dst, const vrt_hip_ctx *src)
{
    if (!dst || !src || dst == src) return VRT_HIP_ERR_INVALID;
    if (dst->device != src->device) return fail(dst, VRT_HIP_ERR_INVALID, "copy_state: the contexts live on different devices");
    HIPCHK(dst, hipSetDevice(dst->device));
    { int rc = quiesce(dst); if (rc) return rc; }
    HIPCHK(dst, hipStreamSynchronize(src->stream)); // uploads of the source are synchronous, its table build runs on its stream
    for (int i = 0; i < 9; ++i) {
        HIPCHK(dst, dst->soa[i].reserve(src->n));
        if (src->n && src->soa[i].p) HIPCHK(dst, hipMemcpy(dst->soa[i].p, src->soa[i].p, (size_t)src->n * sizeof(float), hipMemcpyDeviceToDevice));
    }
    dst->has_alpha = src->has_alpha; dst->n = src->n; dst->albedo_scale = src->albedo_scale;
    dst->exp_kind = src->exp_kind; dst->erf_kind = src->erf_kind; dst->cull_eps = src->cull_eps; dst->cull_prune = src->cull_prune;
    dst->table_hx = src->table_hx; dst->table_budget = src->table_budget; dst->table_adapt = src->table_adapt; dst->table_room = src->table_room;
    dst->rank = src->rank; dst->world = src->world;
    dst->tables_dirty = true; dst->lists_dirty = true; dst->shard_dirty = true; dst->ref_valid = false;
    dst->reset_seq = dst->frame_seq;
    if (dst->tile_mode == TILES_HOST) { dst->tile_mode = TILES_NONE; dst->tw = dst->th = 2.f; dst->tiles_w = dst->tiles_h = 1; }
    ++dst->state_gen;
    return VRT_HIP_OK;
}

int vrt_hip_set_options(vrt_hip_ctx *c, int exp_kind, int erf_kind, float cull_eps)
{
    if (!c) return VRT_HIP_ERR_INVALID;
    if (exp_kind < 0 || exp_kind > VRT_EXP_SPLINE || erf_kind < 0 || erf_kind > VRT_ERF_TAYLOR || !(cull_eps >= 0.f))
        return fail(c, VRT_HIP_ERR_INVALID, "set_options: bad argument");
    const bool supported = (erf_kind == VRT_ERF_AS) || (exp_kind == VRT_EXP_VCL) ||
                           (exp_kind == VRT_EXP_LIBM && erf_kind == VRT_ERF_LIBM);
    if (!supported) return fail(c, VRT_HIP_ERR_INVALID, "set_options: this Exp/Erf pair is not instantiated");
    if (exp_kind != c->exp_kind || cull_eps != c->cull_eps) c->tables_dirty = true;
    if (exp_kind != c->exp_kind || erf_kind != c->erf_kind || cull_eps != c->cull_eps) c->reset_seq = c->frame_seq;
    c->exp_kind = exp_kind; c->erf_kind = erf_kind; c->cull_eps = cull_eps;
    ++c->state_gen;
    return VRT_HIP_OK;
}

int vrt_hip_set_table_step(vrt_hip_ctx *c, float hx)
{
    if (!c) return VRT_HIP_ERR_INVALID;
    if (!(hx >= 0.f) || hx > 1.f) return fail(c, VRT_HIP_ERR_INVALID, "set_table_step: step must be in [0, 1]");
    if (hx != c->table_hx) { c->reset_seq = c->frame_seq; c->lists_dirty = true; }
    c->table_hx = hx;
    ++c->state_gen;
    return VRT_HIP_OK;
}

int vrt_hip_set_table_budget(vrt_hip_ctx *c, float budget)
{
    if (!c) return VRT_HIP_ERR_INVALID;
    if (!(budget > 0.f)) return fail(c, VRT_HIP_ERR_INVALID, "set_table_budget: the budget must be positive (INFINITY = unchecked)");
    if (budget != c->table_budget) c->reset_seq = c->frame_seq;
    c->table_budget = budget;
    ++c->state_gen;
    return VRT_HIP_OK;
}

int vrt_hip_set_cull_prune(vrt_hip_ctx *c, float kappa)
{
    if (!c) return VRT_HIP_ERR_INVALID;
    if (!(kappa >= 0.f)) return fail(c, VRT_HIP_ERR_INVALID, "set_cull_prune: the factor must be >= 0 (0 = off)");
    if (kappa != c->cull_prune) c->reset_seq = c->frame_seq;
    c->cull_prune = kappa;
    ++c->state_gen;
    return VRT_HIP_OK;
}

int vrt_hip_clear_tiles(vrt_hip_ctx *c)
{
    if (!c) return VRT_HIP_ERR_INVALID;
    if (c->tile_mode != TILES_NONE) c->reset_seq = c->frame_seq;
    c->tile_mode = TILES_NONE; c->tw = c->th = 2.f; c->tiles_w = c->tiles_h = 1;
    c->shard_dirty = true; c->lists_dirty = true; c->ref_valid = false;
    return VRT_HIP_OK;
}

int vrt_hip_set_tiles(vrt_hip_ctx *c, float tw, float th, uint64_t tiles_w, uint64_t tiles_h, const uint32_t *offsets,
                      const uint32_t *indices)
{
    if (!c) return VRT_HIP_ERR_INVALID;
    if (!offsets || tiles_w == 0 || tiles_h == 0 || tiles_w * tiles_h > (1u << 24) || !(tw > 0.f) || !(th > 0.f))
        return fail(c, VRT_HIP_ERR_INVALID, "set_tiles: bad argument");
    const size_t nt = (size_t)(tiles_w * tiles_h);
    const size_t total = offsets[nt];
    if (total && !indices) return fail(c, VRT_HIP_ERR_INVALID, "set_tiles: NULL indices");
    std::vector<uint32_t> start(nt), count(nt);
    for (size_t t = 0; t < nt; ++t) {
        if (offsets[t + 1] < offsets[t]) return fail(c, VRT_HIP_ERR_INVALID, "set_tiles: offsets not monotone");
        start[t] = offsets[t]; count[t] = offsets[t + 1] - offsets[t];
    }
    for (size_t k = 0; k < total; ++k)
        if (indices[k] >= c->n) return fail(c, VRT_HIP_ERR_INVALID, "set_tiles: index out of range (upload the scene first)");
    HIPCHK(c, hipSetDevice(c->device));
    { int rc = quiesce(c); if (rc) return rc; }
    // exact-size index buffer: build_work_lists sizes its output from ref_indices.cap
    c->ref_indices.release();
    HIPCHK(c, c->ref_start.reserve(nt)); HIPCHK(c, c->ref_count.reserve(nt)); HIPCHK(c, c->ref_indices.reserve(total));
    HIPCHK(c, hipMemcpy(c->ref_start.p, start.data(), nt * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->ref_count.p, count.data(), nt * 4, hipMemcpyHostToDevice));
    if (total) HIPCHK(c, hipMemcpy(c->ref_indices.p, indices, total * 4, hipMemcpyHostToDevice));
    c->reset_seq = c->frame_seq;
    c->tile_mode = TILES_HOST; c->tw = tw; c->th = th; c->tiles_w = (uint32_t)tiles_w; c->tiles_h = (uint32_t)tiles_h;
    c->shard_dirty = true; c->lists_dirty = true; c->ref_valid = true;
    return VRT_HIP_OK;
}

int vrt_hip_tile_gaussians_device(vrt_hip_ctx *c, float tw, float th, const float view[16], void *hip_stream)
{
    (void)hip_stream; // the lists are built by the next render on ITS stream (they depend on its rays and origin)
    if (!c) return VRT_HIP_ERR_INVALID;
    if (!view || !(tw > 0.f) || !(th > 0.f)) return fail(c, VRT_HIP_ERR_INVALID, "tile_gaussians: bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = rebuild_tables(c);
    if (rc) return rc;
    if (c->tile_mode != TILES_DEVICE || c->grid_tw != tw || c->grid_th != th || c->grid_n != c->n) {
        if ((rc = prepare_tile_grid(c, tw, th))) return rc; // waits for frames in flight (quiesce)
    }
    if (memcmp(c->view, view, 16 * sizeof(float))) c->cam_seq = c->frame_seq;
    memcpy(c->view, view, 16 * sizeof(float));
    c->lists_dirty = true;
    c->ref_valid = false;
    return VRT_HIP_OK;
}

int vrt_hip_tile_gaussians(vrt_hip_ctx *c, float tw, float th, const float view[16])
{
    if (!c) return VRT_HIP_ERR_INVALID;
    int rc = vrt_hip_tile_gaussians_device(c, tw, th, view, nullptr);
    if (rc) return rc;
    // the synchronous form also materialises the reference-semantics lists (what tiles_t would hold), timed
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    const size_t nt = (size_t)c->tiles_w * c->tiles_h;
    HIPCHK(c, c->ref_count.reserve(nt)); HIPCHK(c, c->ref_indices.reserve(nt * c->n));
    BinArgs a = bin_args(c);
    a.refine = 0;
    a.out_start = c->w_start.p; a.out_indices = c->ref_indices.p; a.out_count = c->ref_count.p;
    launch_build_tile_lists(a, FuseArgs{}, false, (uint32_t)nt, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    c->last.tiling_ms = ms;
    c->ref_valid = true;
    return VRT_HIP_OK;
}

int vrt_hip_get_tile_counts(vrt_hip_ctx *c, uint32_t *counts, size_t cap, uint64_t *tiles_w, uint64_t *tiles_h)
{
    if (!c) return VRT_HIP_ERR_INVALID;
    if (c->tile_mode == TILES_NONE) return fail(c, VRT_HIP_ERR_INVALID, "get_tile_counts: no tiles set");
    const size_t nt = (size_t)c->tiles_w * c->tiles_h;
    if (tiles_w) *tiles_w = c->tiles_w;
    if (tiles_h) *tiles_h = c->tiles_h;
    if (counts) {
        if (cap < nt) return fail(c, VRT_HIP_ERR_INVALID, "get_tile_counts: buffer too small");
        HIPCHK(c, hipSetDevice(c->device));
        int rc = ensure_device_ref_lists(c);
        if (rc) return rc;
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipMemcpy(counts, c->ref_count.p, nt * 4, hipMemcpyDeviceToHost));
    }
    return VRT_HIP_OK;
}

int vrt_hip_get_tile_indices(vrt_hip_ctx *c, uint64_t t, uint32_t *indices, size_t cap, uint32_t *count)
{
    if (!c) return VRT_HIP_ERR_INVALID;
    if (c->tile_mode == TILES_NONE || t >= (uint64_t)c->tiles_w * c->tiles_h)
        return fail(c, VRT_HIP_ERR_INVALID, "get_tile_indices: bad tile");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = ensure_device_ref_lists(c);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const uint32_t *starts = c->tile_mode == TILES_DEVICE ? c->w_start.p : c->ref_start.p;
    uint32_t start = 0, cnt = 0;
    HIPCHK(c, hipMemcpy(&start, starts + t, 4, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(&cnt, c->ref_count.p + t, 4, hipMemcpyDeviceToHost));
    if (count) *count = cnt;
    if (indices) {
        if (cap < cnt) return fail(c, VRT_HIP_ERR_INVALID, "get_tile_indices: buffer too small");
        if (cnt) HIPCHK(c, hipMemcpy(indices, c->ref_indices.p + start, (size_t)cnt * 4, hipMemcpyDeviceToHost));
    }
    return VRT_HIP_OK;
}

int vrt_hip_set_plane(vrt_hip_ctx *c, uint32_t w, uint32_t h, const float *xs, const float *ys, const float *zs)
{
    if (!c) return VRT_HIP_ERR_INVALID;
    if (!w || !h || !xs || !ys || !zs) return fail(c, VRT_HIP_ERR_INVALID, "set_plane: bad argument");
    const size_t n = (size_t)w * h;
    HIPCHK(c, hipSetDevice(c->device));
    { int rc = quiesce(c); if (rc) return rc; }
    HIPCHK(c, c->xs.reserve(n)); HIPCHK(c, c->ys.reserve(n)); HIPCHK(c, c->zs.reserve(n));
    HIPCHK(c, hipMemcpy(c->xs.p, xs, n * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->ys.p, ys, n * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->zs.p, zs, n * 4, hipMemcpyHostToDevice));
    c->plane_affine = plane_is_affine(w, h, xs, ys, zs);
    c->reset_seq = c->frame_seq; // new rays
    c->w = w; c->h = h; c->plane_mode = true; c->rays_set = true; c->lists_dirty = true;
    ++c->plane_gen; // other plane arrays behind the same pointers: the tile cones are no longer theirs
    return VRT_HIP_OK;
}

int vrt_hip_set_camera(vrt_hip_ctx *c, uint32_t w, uint32_t h, const float pos[3], const float right[3],
                       const float up[3], const float front[3], float focal)
{
    if (!c) return VRT_HIP_ERR_INVALID;
    if (!w || !h || !pos || !right || !up || !front) return fail(c, VRT_HIP_ERR_INVALID, "set_camera: bad argument");
    if (c->w != w || c->h != h || c->plane_mode || c->focal != focal || memcmp(c->cam_pos, pos, 12) || memcmp(c->cam_right, right, 12) ||
        memcmp(c->cam_up, up, 12) || memcmp(c->cam_front, front, 12))
        (c->w != w || c->h != h ? c->reset_seq : c->cam_seq) = c->frame_seq; // another camera / another image size
    memcpy(c->cam_pos, pos, 12); memcpy(c->cam_right, right, 12); memcpy(c->cam_up, up, 12); memcpy(c->cam_front, front, 12);
    c->focal = focal; c->w = w; c->h = h; c->plane_mode = false; c->view_mode = false; c->rays_set = true; c->lists_dirty = true;
    return VRT_HIP_OK;
}

int vrt_hip_set_camera_view(vrt_hip_ctx *c, uint32_t w, uint32_t h, const float view[16])
{
    if (!c) return VRT_HIP_ERR_INVALID;
    if (!w || !h || !view) return fail(c, VRT_HIP_ERR_INVALID, "set_camera_view: bad argument");
    float inv[16];
    vrt_hip_mat4_inverse(view, inv); // glm::inverse in glm's order, unfused (csrc/vrt_host_camera.cpp)
    for (int i = 0; i < 16; ++i)
        if (!std::isfinite(inv[i])) return fail(c, VRT_HIP_ERR_INVALID, "set_camera_view: the view matrix is singular");
    if (c->w != w || c->h != h || c->plane_mode || !c->view_mode || memcmp(c->inv_view, inv, sizeof inv))
        (c->w != w || c->h != h ? c->reset_seq : c->cam_seq) = c->frame_seq; // another camera / another image size
    memcpy(c->inv_view, inv, sizeof inv);
    c->w = w; c->h = h; c->plane_mode = false; c->view_mode = true; c->rays_set = true; c->lists_dirty = true;
    return VRT_HIP_OK;
}

size_t vrt_hip_image_pixels(const vrt_hip_ctx *c) { return (c && c->rays_set) ? (size_t)c->w * c->h : 0; }

int vrt_hip_render_device(vrt_hip_ctx *c, const float origin[3], int pack_flags, uint32_t *d_image, float *d_radiance,
                          void *hip_stream)
{
    if (!c || !origin) return VRT_HIP_ERR_INVALID;
    return render_common(c, origin, pack_flags, d_image, (float4 *)d_radiance, (hipStream_t)hip_stream, OUT_RASTER);
}

int vrt_hip_frame_device(vrt_hip_ctx *c, float tw, float th, const float view[16], const float origin[3], int pack_flags,
                         uint32_t *d_out, int shard, void *hip_stream)
{
    if (!c || !origin || !d_out) return VRT_HIP_ERR_INVALID;
    int rc = vrt_hip_tile_gaussians_device(c, tw, th, view, hip_stream);
    if (rc) return rc;
    return render_common(c, origin, pack_flags, d_out, nullptr, (hipStream_t)hip_stream, shard ? OUT_COMPACT : OUT_RASTER);
}

int vrt_hip_frame_retained_device(vrt_hip_ctx *c, float tw, float th, const float view[16], const float origin[3], int pack_flags,
                                  uint32_t *d_out, void *hip_stream)
{
    if (!c || !origin || !d_out) return VRT_HIP_ERR_INVALID;
    int rc = check_ready(c);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    if ((rc = retained_begin(c, d_out, tw, th, pack_flags, (hipStream_t)hip_stream))) return rc;
    rc = vrt_hip_frame_device(c, tw, th, view, origin, pack_flags, d_out, 0, hip_stream);
    c->retain_next = false;
    if (rc) c->own_seq = 0;
    return rc;
}

int vrt_hip_frame(vrt_hip_ctx *c, float tw, float th, const float view[16], const float origin[3], int pack_flags,
                  uint32_t *image_out, int wait)
{
    if (!c || !origin || !view) return VRT_HIP_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = check_ready(c);
    if (rc) return rc;
    const size_t npix = (size_t)c->w * c->h;
    if (c->d_image.cap < npix) { // first frame at this size: pixels no tile covers read 0
        HIPCHK(c, c->d_image.reserve(npix));
        HIPCHK(c, hipMemsetAsync(c->d_image.p, 0, npix * 4, c->stream));
        c->own_seq = 0;
    }
    // Retained frame buffer: d_image is written by nothing but this call and vrt_hip_render (which ends the history), so an
    // empty cell that was empty in the previous frame already holds the background.
    if ((rc = retained_begin(c, c->d_image.p, tw, th, pack_flags, c->stream))) return rc;
    rc = vrt_hip_frame_device(c, tw, th, view, origin, pack_flags, c->d_image.p, 0, c->stream);
    c->retain_next = false;
    if (rc) { c->own_seq = 0; return rc; }
    if (image_out) HIPCHK(c, hipMemcpyAsync(image_out, c->d_image.p, npix * 4, hipMemcpyDeviceToHost, c->stream));
    if (image_out || wait) HIPCHK(c, hipStreamSynchronize(c->stream));
    return VRT_HIP_OK;
}

int vrt_hip_sync(vrt_hip_ctx *c)
{
    if (!c) return VRT_HIP_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return VRT_HIP_OK;
}

int vrt_hip_render(vrt_hip_ctx *c, const float origin[3], int pack_flags, uint32_t *image_out, float *radiance_out)
{
    if (!c || !origin) return VRT_HIP_ERR_INVALID;
    int rc = check_ready(c);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t npix = (size_t)c->w * c->h;
    HIPCHK(c, c->d_image.reserve(npix));
    c->own_seq = 0; // the library's frame buffer gets another image: vrt_hip_frame's retained history of it ends
    if (radiance_out) HIPCHK(c, c->d_rad.reserve(npix));
    HIPCHK(c, hipMemsetAsync(c->d_image.p, 0, npix * 4, c->stream));
    if (radiance_out) HIPCHK(c, hipMemsetAsync(c->d_rad.p, 0, npix * 16, c->stream));
    // tables / frame prep outside the timed window (list building is part of a frame, like the reference's tiling)
    if ((rc = prep_frame(c, origin, c->stream))) return rc;
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    rc = render_common(c, origin, pack_flags, c->d_image.p, radiance_out ? c->d_rad.p : nullptr, c->stream, OUT_RASTER);
    if (rc) return rc;
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    c->last.kernel_ms = ms;
    if (c->stats_on) {
        unsigned long long st[32];
        HIPCHK(c, hipMemcpy(st, c->d_stats.p, sizeof st, hipMemcpyDeviceToHost));
        c->last.table_nodes = st[16]; c->last.table_retries = st[17]; c->last.table_skips = st[18]; c->last.table_declined = st[19]; c->last.table_coarser = st[20]; c->last.table_empty = st[21];
        for (int k = 0; k < 8; ++k) c->last.table_phase_ticks[k] = st[24 + k];
        if (getenv("VRT_HIP_TABLE_DIAG") && st[7]) // the table phase split by the first wave's clock: staging | node loops | waiting at the chunk barriers (note: [21] is table_empty)
            fprintf(stderr, "[vrt_hip] table kernel, us per block of its table phase (%.1f): node loops %.1f, waiting at the chunk barriers %.1f, staging the rest (mean over the 16 waves)\n",
                    st[24 + 5] * 0.01 / st[7], st[22] * 0.01 / 16 / st[7], st[23] * 0.01 / 16 / st[7]);
        c->last.lane_pairs = st[12];
        c->last.dense_visits_full = st[13]; c->last.dense_visits_zero = st[14]; c->last.dense_visits_common = st[15];
        c->last.dense_busy_frac = (st[11] && st[9] > st[8]) ? (double)st[10] / ((double)st[11] * (double)(st[9] - st[8])) : 0.0;
        c->last.shaded_blocks = st[5] + st[6];
        c->last.dense_blocks = st[6];
        c->last.table_blocks = st[7];
        c->last.list_entries = st[0]; c->last.tile_entries = st[1]; c->last.overflow_blocks = st[2];
        c->last.lane_entries = st[3]; c->last.lane_max_entries = st[4];
    }
    if (c->timeline_items) print_timeline(c);
    if (image_out) HIPCHK(c, hipMemcpy(image_out, c->d_image.p, npix * 4, hipMemcpyDeviceToHost));
    if (radiance_out) HIPCHK(c, hipMemcpy(radiance_out, c->d_rad.p, npix * 16, hipMemcpyDeviceToHost));
    return VRT_HIP_OK;
}

// VRT_HIP_TIMELINE=1: where the one-wave kernel's time goes (wall_clock64 ticks are 10 ns), printed by vrt_hip_render()
static void print_timeline(vrt_hip_ctx *c)
{
    if (c->timeline_tiles) {
        std::vector<unsigned long long> tt(c->timeline_tiles * 8);
        if (hipMemcpy(tt.data(), c->d_timeline_lists.p, tt.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess) {
            unsigned long long a = ~0ull, b = 0;
            double ph[5] = {0, 0, 0, 0, 0}, s_start = 0, n = 0, worst = 0, wph[5] = {0, 0, 0, 0, 0};
            size_t worst_i = 0;
            for (size_t i = 0; i < c->timeline_tiles; ++i) {
                const unsigned long long *e = &tt[8 * i];
                if (!e[4]) continue;
                a = std::min(a, e[0]); b = std::max(b, e[5] ? e[5] : e[4]);
            }
            for (size_t i = 0; i < c->timeline_tiles; ++i) {
                const unsigned long long *e = &tt[8 * i];
                if (!e[4]) continue;
                n += 1; s_start += (double)(e[0] - a);
                for (int k = 0; k < 5; ++k) ph[k] += (e[k + 1] >= e[k] && e[k + 1]) ? (double)(e[k + 1] - e[k]) : 0.0;
                const double dur = (double)((e[5] ? e[5] : e[4]) - e[0]);
                if (dur > worst) {
                    worst = dur; worst_i = i;
                    for (int k = 0; k < 5; ++k) wph[k] = (e[k + 1] >= e[k] && e[k + 1]) ? (double)(e[k + 1] - e[k]) : 0.0;
                }
            }
            if (n > 0)
                fprintf(stderr, "[vrt_hip] list kernel timeline: %.0f tiles, span %.2f us, mean start %.2f us; per tile: cone %.2f us, "
                                "level 1 %.2f us, level 2 %.2f us, filing %.2f us, clear %.2f us\n", n, (b - a) * 0.01, s_start / n * 0.01,
                        ph[0] / n * 0.01, ph[1] / n * 0.01, ph[2] / n * 0.01, ph[3] / n * 0.01, ph[4] / n * 0.01);
            if (n > 0)
                fprintf(stderr, "[vrt_hip]   slowest tile %zu: %.2f us = %.2f + %.2f + %.2f + %.2f + %.2f\n", worst_i, worst * 0.01,
                        wph[0] * 0.01, wph[1] * 0.01, wph[2] * 0.01, wph[3] * 0.01, wph[4] * 0.01);
        }
    }
    std::vector<unsigned long long> tl(c->timeline_items * 5);
    if (hipMemcpy(tl.data(), c->d_timeline.p, tl.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return;
    unsigned long long t0 = ~0ull, t1 = 0;
    double n = 0, s_start = 0, s_end = 0, p0 = 0, p1 = 0, p2 = 0;
    for (size_t i = 0; i < c->timeline_items; ++i)
        if (tl[5 * i + 3]) { t0 = std::min(t0, tl[5 * i]); t1 = std::max(t1, tl[5 * i + 3]); }
    std::vector<unsigned> starts(64, 0), ends(64, 0);
    std::map<uint32_t, std::pair<int, double>> per_simd; // blocks, last end (us)
    std::map<uint32_t, int> per_cu;
    for (size_t i = 0; i < c->timeline_items; ++i) {
        const unsigned long long *e = &tl[5 * i];
        if (!e[3]) continue;
        {   // gfx9 HW_ID: simd [5:4], cu [11:8], sh [12], se [15:13]; XCC_ID [3:0]
            const uint32_t hw = (uint32_t)e[4], xcc = (uint32_t)(e[4] >> 32) & 15u;
            const uint32_t cu = (xcc << 8) | (((hw >> 13) & 7u) << 5) | (((hw >> 12) & 1u) << 4) | ((hw >> 8) & 15u);
            const uint32_t simd = (cu << 2) | ((hw >> 4) & 3u);
            auto &a = per_simd[simd]; a.first += 1; a.second = std::max(a.second, (double)(e[3] - t0) * 0.01);
            per_cu[cu] += 1;
        }
        n += 1; s_start += (double)(e[0] - t0); s_end += (double)(e[3] - t0);
        p0 += (double)(e[1] - e[0]); p1 += (double)(e[2] - e[1]); p2 += (double)(e[3] - e[2]);
        const double span = (double)(t1 - t0) + 1;
        ++starts[(size_t)((e[0] - t0) * 64.0 / span)]; ++ends[(size_t)((e[3] - t0) * 64.0 / span)];
    }
    if (n == 0) return;
    if (const char *path = getenv("VRT_HIP_TIMELINE"); path && strstr(path, ".csv")) { // raw stamps for offline analysis
        if (FILE *f = fopen(path, "w")) {
            fprintf(f, "item,t0,t1,t2,t3,hw_id,xcc_id,nmax\n");
            for (size_t i = 0; i < c->timeline_items; ++i) {
                const unsigned long long *e = &tl[5 * i];
                if (e[3]) fprintf(f, "%zu,%llu,%llu,%llu,%llu,%u,%u,%u\n", i, e[0] - t0, e[1] - t0, e[2] - t0, e[3] - t0, (unsigned)e[4], (unsigned)(e[4] >> 32) & 0xFFFFu, (unsigned)(e[4] >> 48));
            }
            fclose(f);
        }
    }
    fprintf(stderr, "[vrt_hip] one-wave kernel timeline: %.0f blocks, span %.2f us, mean start %.2f us, mean end %.2f us; per block: "
                    "block cull %.2f us, lane lists %.2f us, shade+store %.2f us\n[vrt_hip]   running blocks per 1/64 of the span:",
            n, (t1 - t0) * 0.01, s_start / n * 0.01, s_end / n * 0.01, p0 / n * 0.01, p1 / n * 0.01, p2 / n * 0.01);
    long running = 0;
    for (int b = 0; b < 64; ++b) { running += starts[b]; fprintf(stderr, " %ld", running); running -= ends[b]; }
    fprintf(stderr, "\n");
    std::map<int, std::pair<int, double>> by_count; // blocks on a SIMD -> (SIMDs, mean last end)
    for (auto &kv : per_simd) { auto &b = by_count[kv.second.first]; b.first += 1; b.second += kv.second.second; }
    fprintf(stderr, "[vrt_hip]   %zu CUs, %zu SIMDs seen; blocks per SIMD -> SIMDs (mean time of their last block end):", per_cu.size(), per_simd.size());
    for (auto &kv : by_count) fprintf(stderr, "  %d -> %d (%.1f us)", kv.first, kv.second.first, kv.second.second / kv.second.first);
    std::map<int, int> cu_hist;
    for (auto &kv : per_cu) cu_hist[kv.second] += 1;
    fprintf(stderr, "\n[vrt_hip]   blocks per CU -> CUs:");
    for (auto &kv : cu_hist) fprintf(stderr, "  %d -> %d", kv.first, kv.second);
    fprintf(stderr, "\n");
}

int vrt_hip_set_shard(vrt_hip_ctx *c, int rank, int world)
{
    if (!c) return VRT_HIP_ERR_INVALID;
    if (world < 1 || rank < 0 || rank >= world) return fail(c, VRT_HIP_ERR_INVALID, "set_shard: bad rank/world");
    if (rank != c->rank || world != c->world) ++c->state_gen; // vrt_hip_copy_state copies rank / world: holders of mirrors must see the change
    c->rank = rank; c->world = world; c->shard_dirty = true; c->lists_dirty = true; c->reset_seq = c->frame_seq;
    return VRT_HIP_OK;
}

size_t vrt_hip_shard_pixels(const vrt_hip_ctx *cc)
{
    vrt_hip_ctx *c = const_cast<vrt_hip_ctx *>(cc);
    if (!c || !c->rays_set) return 0;
    if (rebuild_shard(c)) return 0;
    const TileLists t = tile_geometry(c);
    return (size_t)c->n_slots * t.tile_w * t.tile_h;
}

int vrt_hip_render_shard_device(vrt_hip_ctx *c, const float origin[3], int pack_flags, uint32_t *d_shard, void *hip_stream)
{
    if (!c || !origin || !d_shard) return VRT_HIP_ERR_INVALID;
    return render_common(c, origin, pack_flags, d_shard, nullptr, (hipStream_t)hip_stream, OUT_COMPACT);
}

int vrt_hip_assemble_shards_strided_device(vrt_hip_ctx *c, const uint32_t *d_gathered, size_t rank_stride_px,
                                           uint32_t *d_image, void *hip_stream)
{
    if (!c || !d_gathered || !d_image) return VRT_HIP_ERR_INVALID;
    int rc = check_ready(c);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    if ((rc = rebuild_shard(c))) return rc;
    const TileLists t = tile_geometry(c);
    if (rank_stride_px < (size_t)c->n_slots * t.tile_w * t.tile_h) return fail(c, VRT_HIP_ERR_INVALID, "assemble: rank stride smaller than one shard");
    launch_assemble(d_gathered, d_image, c->slot_tiles.p, c->n_slots, (uint32_t)c->world, rank_stride_px, t, c->w, c->h,
                    (hipStream_t)hip_stream);
    HIPCHK(c, hipGetLastError());
    return VRT_HIP_OK;
}

int vrt_hip_assemble_shards_device(vrt_hip_ctx *c, const uint32_t *d_gathered, uint32_t *d_image, void *hip_stream)
{
    if (!c) return VRT_HIP_ERR_INVALID;
    return vrt_hip_assemble_shards_strided_device(c, d_gathered, vrt_hip_shard_pixels(c), d_image, hip_stream);
}

// ---- sparse shards: only the cells some Gaussian reaches travel (multi-GPU transport) ------------------------------
namespace {
uint32_t sparse_capacity(vrt_hip_ctx *c)
{
    if (rebuild_shard(c)) return 0;
    const TileLists t = tile_geometry(c);
    const uint32_t cx = (t.tile_w + CELL - 1) / CELL, cy = (t.tile_h + CELL - 1) / CELL;
    return c->n_slots * cx * cy;
}
} // namespace

size_t vrt_hip_sparse_shard_words(const vrt_hip_ctx *cc)
{
    vrt_hip_ctx *c = const_cast<vrt_hip_ctx *>(cc);
    if (!c || !c->rays_set) return 0;
    const uint32_t cap = sparse_capacity(c);
    return sparse_pixel_offset(cap) + (size_t)cap * CELL * CELL;
}

int vrt_hip_frame_sparse_device(vrt_hip_ctx *c, float tw, float th, const float view[16], const float origin[3], int pack_flags,
                                uint32_t *d_sparse, void *hip_stream)
{
    if (!c || !origin || !d_sparse) return VRT_HIP_ERR_INVALID;
    if ((uintptr_t)d_sparse % 16) return fail(c, VRT_HIP_ERR_INVALID, "frame_sparse: the shard buffer must be 16-byte aligned");
    int rc = vrt_hip_tile_gaussians_device(c, tw, th, view, hip_stream);
    if (rc) return rc;
    return render_common(c, origin, pack_flags, d_sparse, nullptr, (hipStream_t)hip_stream, OUT_SPARSE);
}

int vrt_hip_frame_batch_device(vrt_hip_ctx *const *ctxs, int n, float tw, float th, const float *views, const float *origins,
                               int pack_flags, uint32_t *const *d_out, int out_kind, void *hip_stream)
{
    if (!ctxs || n < 1 || !ctxs[0]) return VRT_HIP_ERR_INVALID;
    vrt_hip_ctx *c0 = ctxs[0];
    if (!views || !origins || !d_out) return fail(c0, VRT_HIP_ERR_INVALID, "frame_batch: null argument");
    if (n > 64) return fail(c0, VRT_HIP_ERR_INVALID, "frame_batch: at most 64 frames per batch");
    if (out_kind < OUT_RASTER || out_kind > OUT_SPARSE) return fail(c0, VRT_HIP_ERR_INVALID, "frame_batch: out_kind is 0 (frame), 1 (compact shard) or 2 (sparse shard)");
    hipStream_t st = (hipStream_t)hip_stream;
    for (int i = 0; i < n; ++i) {
        vrt_hip_ctx *c = ctxs[i];
        if (!c || !d_out[i]) return fail(c0, VRT_HIP_ERR_INVALID, "frame_batch: null context or output");
        for (int k = 0; k < i; ++k)
            if (ctxs[k] == c) return fail(c0, VRT_HIP_ERR_INVALID, "frame_batch: a context holds ONE frame's lists and queues -- every frame of a batch needs its own");
        if (c->device != c0->device || c->exp_kind != c0->exp_kind || c->erf_kind != c0->erf_kind || c->dense_waves != c0->dense_waves ||
            c->table_hx != c0->table_hx || c->table_budget != c0->table_budget || c->cull_prune != c0->cull_prune || c->cull_eps != c0->cull_eps)
            return fail(c0, VRT_HIP_ERR_INVALID, "frame_batch: the contexts differ in device or in Exp / Erf / dense-kernel / table options");
        if (c->w != c0->w || c->h != c0->h || c->n != c0->n || c->rank != c0->rank || c->world != c0->world)
            return fail(c0, VRT_HIP_ERR_INVALID, "frame_batch: the frames differ in image size, scene size or shard");
        if (out_kind == OUT_SPARSE && (uintptr_t)d_out[i] % 16) return fail(c0, VRT_HIP_ERR_INVALID, "frame_batch: sparse shard buffers must be 16-byte aligned");
    }
    HIPCHK(c0, hipSetDevice(c0->device));
    // argument rows: slot (batch_seq % BATCH_SLOTS) of the pinned ring, copied to the same slot of the device ring
    if ((size_t)n > c0->batch_cap) {
        int rc = quiesce(c0);
        if (rc) return rc;
        HIPCHK(c0, hipStreamSynchronize(st));
        if (c0->batch_host) (void)hipHostFree(c0->batch_host);
        if (c0->batch_dev) (void)hipFree(c0->batch_dev);
        c0->batch_host = c0->batch_dev = nullptr; c0->batch_cap = 0;
        const size_t cap = std::max<size_t>(16, (size_t)n);
        HIPCHK(c0, hipHostMalloc((void **)&c0->batch_host, cap * vrt_hip_ctx::BATCH_SLOTS * sizeof(FrameArgs), hipHostMallocDefault));
        HIPCHK(c0, hipMalloc((void **)&c0->batch_dev, cap * vrt_hip_ctx::BATCH_SLOTS * sizeof(FrameArgs)));
        c0->batch_cap = cap;
        for (auto &e : c0->batch_copied) if (!e) HIPCHK(c0, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    const uint32_t slot = c0->batch_seq++ % vrt_hip_ctx::BATCH_SLOTS;
    if (c0->batch_seq > (uint32_t)vrt_hip_ctx::BATCH_SLOTS) HIPCHK(c0, hipEventSynchronize(c0->batch_copied[slot])); // the copy that last read this slot
    FrameArgs *rows = c0->batch_host + (size_t)slot * c0->batch_cap;
    FrameArgs *d_rows = c0->batch_dev + (size_t)slot * c0->batch_cap;

    // every frame's host work, memsets and per-origin table kernel as for a single frame; its three launches recorded
    int failed = VRT_HIP_OK, touched = 0;
    for (int i = 0; i < n && !failed; ++i) {
        vrt_hip_ctx *c = ctxs[i];
        c->defer = &rows[i];
        rows[i].do_prep = rows[i].do_cones = rows[i].do_order = 0;
        c->deferred = vrt_hip_ctx::Deferred{};
        touched = i + 1;
        int rc = vrt_hip_tile_gaussians_device(c, tw, th, views + 16 * (size_t)i, hip_stream);
        if (!rc) rc = render_common(c, origins + 3 * (size_t)i, pack_flags, d_out[i], nullptr, st, out_kind);
        c->defer = nullptr;
        if (rc) {
            if (c != c0) fail(c0, rc, std::string("frame_batch: frame ") + std::to_string(i) + ": " + c->err);
            failed = rc;
            break;
        }
        const auto &d = c->deferred, &d0 = c0->deferred;
        if (!d.render || d.lists != d0.lists || d.from_list != d0.from_list || d.list_grid != d0.list_grid || d.render_grid != d0.render_grid)
            failed = fail(c0, VRT_HIP_ERR_INVALID, "frame_batch: the frames differ in image size, tile grid, shard or scene size");
    }
    if (failed) {
        // The contexts prepared so far have advanced their list and queue generations for kernels that will not run: the
        // counter sets those kernels would have cleared for the next frame are stale.  Clear them and make the next frame
        // rebuild its lists.
        for (int i = 0; i < touched; ++i) {
            vrt_hip_ctx *c = ctxs[i];
            if (c->c_counters.p && hipMemsetAsync(c->c_counters.p, 0, 16 * sizeof(uint32_t), st) != hipSuccess) (void)hipGetLastError();
            if (c->c_rq.p && hipMemsetAsync(c->c_rq.p, 0, 2 * RQ_N * RQ_STRIDE * sizeof(uint32_t), st) != hipSuccess) (void)hipGetLastError();
            c->lists_dirty = true; c->lists_fresh = false;
            c->gA_valid = false; c->cone_key.clear(); // their deferred set-up launches were never made
        }
        return failed;
    }
    for (int i = 0; i < n; ++i) rows[i].do_order = ctxs[i]->deferred.order ? 1 : 0;
    // one-wave kernel: the persistent grid of ONE frame fills the GPU; n frames share it -- so a frame of a batch has 1/n of the waves and
    // that many more queue entries: the variant that claims them early is chosen against the per-frame grid
    const uint32_t full_grid = c0->deferred.render_grid;
    const uint32_t rgrid = full_grid ? std::min(full_grid, std::max(1u, (full_grid + (uint32_t)n - 1) / (uint32_t)n)) : 0u;
    bool claim = false;
    for (int i = 0; i < n; ++i) {
        const vrt_hip_ctx *c = ctxs[i];
        const uint32_t seen_blocks = (c->h_fb && !c->stats_on) ? c->h_fb[1] : 0u;
        const bool many = c->claim_early > 0 && seen_blocks > rgrid && (uint64_t)(seen_blocks - rgrid) * (uint32_t)c->claim_early >= rgrid;
        rows[i].C.claim_early = many ? c->claim_early : 0;
        claim = claim || many;
    }
    HIPCHK(c0, hipMemcpyAsync(d_rows, rows, (size_t)n * sizeof(FrameArgs), hipMemcpyHostToDevice, st));
    HIPCHK(c0, hipEventRecord(c0->batch_copied[slot], st));
    launch_frame_setup_batch(d_rows, rows, (uint32_t)n, st); // per-origin tables and cone tables of the frames that need new ones
    const auto &d0 = c0->deferred;
    if (d0.lists) launch_build_tile_lists_batch(d_rows, (uint32_t)n, d0.from_list, !d0.from_list && rows[0].bin.chunks && rows[0].bin.refine, d0.list_grid, st); // (same scene size and geometry in every frame: checked above)
    launch_render_batch(d_rows, (uint32_t)n, rgrid, claim, c0->exp_kind, c0->erf_kind, st);
    uint32_t dgrid = 0;
    for (int i = 0; i < n; ++i) dgrid = std::max(dgrid, ctxs[i]->deferred.dense_grid);
    launch_order_dense_batch(d_rows, rows, (uint32_t)n, st);
    if (table_on(c0))
        launch_render_table_batch(d_rows, (uint32_t)n, std::min<uint32_t>(dgrid, (uint32_t)c0->num_cus), (uint64_t)rows[0].R.width * rows[0].R.height,
                                  c0->exp_kind, c0->erf_kind, st);
    else
        launch_render_dense_batch(d_rows, (uint32_t)n, dgrid, c0->dense_waves, c0->exp_kind, c0->erf_kind, st);
    HIPCHK(c0, hipGetLastError());
    return VRT_HIP_OK;
}

namespace {
struct AssemblyGeometry {
    TileLists t;
    uint32_t cx, cy, max_cells, frame_cells, bg;
    size_t npix, covered;
    uint64_t sig;
};
int assembly_geometry(vrt_hip_ctx *c, int pack_flags, AssemblyGeometry &g)
{
    int rc = check_ready(c);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    g.t = tile_geometry(c);
    if (g.t.tile_w == 0 || g.t.tile_h == 0) return fail(c, VRT_HIP_ERR_INVALID, "scatter_sparse: tile size is 0 pixels");
    g.bg = (pack_flags & VRT_ALPHA_COMPUTED) ? 0u : 0xFF000000u; // what the kernels write where nothing is lit
    // the tiles cover the linear pixel range [0, stride * tile_h * tiles_h) (rt.h:364-365: pix = x + stride * y with the
    // truncated tile size); what lies beyond is written by nobody in a single-GPU frame either and reads 0
    g.npix = (size_t)c->w * c->h;
    g.covered = std::min(g.npix, (size_t)g.t.stride * g.t.tile_h * g.t.tiles_h);
    g.cx = (g.t.tile_w + CELL - 1) / CELL; g.cy = (g.t.tile_h + CELL - 1) / CELL;
    // one workgroup per (shard, slot) up to the shard capacity -- the same on every rank of this context's job
    g.max_cells = sparse_capacity(c);
    g.frame_cells = g.t.tiles_w * g.t.tiles_h * g.cx * g.cy;
    g.sig = ((uint64_t)g.t.tiles_w << 48) ^ ((uint64_t)g.t.tile_w << 32) ^ ((uint64_t)g.t.tile_h << 16) ^ g.t.tiles_h ^ ((uint64_t)c->w << 24) ^
            ((uint64_t)c->h << 8);
    return VRT_HIP_OK;
}
// Background of one frame buffer before its cells are scattered.  Retained: the caller promises that d_image still holds
// what the previous retained assembly of this context left in it; then the 4 B per ray of background (16.8 MB per 2048^2
// frame: ~4 us of HBM writes, more than a rank's share of the rendering at 8 GPUs) shrink to the cells that were lit last
// time and are not now (clear_stale_cells_kernel, after the scatter).  A new buffer, image size, tile grid or background
// value gets the full fill and starts a new history.
// `batch` / `nbatch`: the frame buffers of the call this one belongs to -- their histories' stamp buffers are already part of
// the launch being prepared and must not be evicted to make room (round-2 advisor finding).
int assembly_background(vrt_hip_ctx *c, const AssemblyGeometry &g, uint32_t *d_image, bool retained, hipStream_t st, uint32_t **stamp,
                        uint32_t *seq, bool *incremental, uint32_t *const *batch = nullptr, int nbatch = 0)
{
    *stamp = nullptr; *seq = 0; *incremental = false;
    auto it = std::find_if(c->retained.begin(), c->retained.end(), [&](const vrt_hip_ctx::Retained &r) { return r.image == d_image; });
    if (!retained) {
        if (it != c->retained.end()) { // a plain assembly into a retained buffer ends its history
            // its stamps may still be read by an earlier retained assembly on ANY stream: wait for the device
            HIPCHK(c, hipDeviceSynchronize());
            it->stamp.release();
            c->retained.erase(it);
        }
    } else {
        if (it == c->retained.end()) {
            if (c->retained.size() >= (size_t)MAX_ASSEMBLY_FRAMES) {
                // the oldest history that is not one of this call's own buffers (a call has at most MAX_ASSEMBLY_FRAMES
                // distinct buffers and this one is new, so there is one); earlier assemblies of it may have run on another
                // stream than `st`: wait for the device (a rare path: more than 64 frame buffers in rotation)
                auto victim = std::find_if(c->retained.begin(), c->retained.end(), [&](const vrt_hip_ctx::Retained &r) {
                    for (int k = 0; k < nbatch; ++k) if (batch[k] == r.image) return false;
                    return true;
                });
                if (victim == c->retained.end()) return fail(c, VRT_HIP_ERR_INVALID, "scatter_sparse: no retained history can be dropped");
                HIPCHK(c, hipDeviceSynchronize());
                victim->stamp.release();
                c->retained.erase(victim);
            }
            c->retained.emplace_back();
            it = c->retained.end() - 1;
            it->image = d_image;
        }
        *incremental = it->sig == g.sig && it->bg == g.bg && it->stamp.cap >= g.frame_cells && it->seq != 0 && it->seq != 0xFFFFFFFFu;
        if (!*incremental) {
            HIPCHK(c, it->stamp.reserve(g.frame_cells));
            HIPCHK(c, hipMemsetAsync(it->stamp.p, 0, (size_t)g.frame_cells * sizeof(uint32_t), st));
            it->sig = g.sig; it->bg = g.bg; it->seq = 0;
        }
        *seq = ++it->seq;
        *stamp = it->stamp.p;
    }
    if (!*incremental) {
        HIPCHK(c, hipMemsetD32Async((hipDeviceptr_t)d_image, (int)g.bg, g.covered, st));
        if (g.covered < g.npix) HIPCHK(c, hipMemsetAsync(d_image + g.covered, 0, (g.npix - g.covered) * sizeof(uint32_t), st));
    }
    return VRT_HIP_OK;
}
int scatter_sparse(vrt_hip_ctx *c, const uint32_t *const *d_shards, int nshards, int pack_flags, uint32_t *d_image, void *hip_stream,
                   bool retained)
{
    if (!c || !d_shards || !d_image || nshards < 1 || nshards > MAX_SHARDS) return VRT_HIP_ERR_INVALID;
    AssemblyGeometry g;
    int rc = assembly_geometry(c, pack_flags, g);
    if (rc) return rc;
    ShardPtrs sp{};
    for (int i = 0; i < nshards; ++i) {
        if (!d_shards[i]) return fail(c, VRT_HIP_ERR_INVALID, "scatter_sparse: NULL shard");
        sp.p[i] = d_shards[i];
    }
    hipStream_t st = (hipStream_t)hip_stream;
    uint32_t *stamp, seq;
    bool incremental;
    if ((rc = assembly_background(c, g, d_image, retained, st, &stamp, &seq, &incremental))) return rc;
    launch_scatter_sparse(sp, nshards, g.max_cells, d_image, g.t, g.cx, g.cy, c->w, c->h, stamp, seq, st);
    if (incremental) launch_clear_stale_cells(stamp, seq, g.frame_cells, d_image, g.t, g.cx, g.cy, c->w, c->h, g.bg, st);
    HIPCHK(c, hipGetLastError());
    return VRT_HIP_OK;
}
} // namespace

int vrt_hip_scatter_sparse_device(vrt_hip_ctx *c, const uint32_t *const *d_shards, int nshards, int pack_flags, uint32_t *d_image,
                                  void *hip_stream)
{
    return scatter_sparse(c, d_shards, nshards, pack_flags, d_image, hip_stream, false);
}
int vrt_hip_scatter_sparse_retained_device(vrt_hip_ctx *c, const uint32_t *const *d_shards, int nshards, int pack_flags,
                                           uint32_t *d_image, void *hip_stream)
{
    return scatter_sparse(c, d_shards, nshards, pack_flags, d_image, hip_stream, true);
}
int vrt_hip_scatter_sparse_batch_device(vrt_hip_ctx *c, const uint32_t *const *d_shards, int nshards, size_t frame_stride_words,
                                        int nframes, int pack_flags, uint32_t *const *d_images, int retained, void *hip_stream)
{
    if (!c || !d_shards || !d_images || nshards < 1 || nshards > MAX_SHARDS) return VRT_HIP_ERR_INVALID;
    if (nframes < 1 || nframes > MAX_ASSEMBLY_FRAMES) return fail(c, VRT_HIP_ERR_INVALID, "scatter_sparse_batch: 1..64 frames per call");
    if (frame_stride_words % 4) return fail(c, VRT_HIP_ERR_INVALID, "scatter_sparse_batch: the frame stride must keep the shards 16-byte aligned");
    AssemblyGeometry g;
    int rc = assembly_geometry(c, pack_flags, g);
    if (rc) return rc;
    ShardPtrs sp{};
    for (int i = 0; i < nshards; ++i) {
        if (!d_shards[i]) return fail(c, VRT_HIP_ERR_INVALID, "scatter_sparse_batch: NULL shard");
        sp.p[i] = d_shards[i];
    }
    hipStream_t st = (hipStream_t)hip_stream;
    AssemblyFrames fr{};
    for (int f = 0; f < nframes; ++f) {
        if (!d_images[f]) return fail(c, VRT_HIP_ERR_INVALID, "scatter_sparse_batch: NULL image");
        for (int k = 0; k < f; ++k)
            if (d_images[k] == d_images[f]) return fail(c, VRT_HIP_ERR_INVALID, "scatter_sparse_batch: two frames of a batch into one buffer");
        bool incremental;
        if ((rc = assembly_background(c, g, d_images[f], retained != 0, st, &fr.stamp[f], &fr.seq[f], &incremental, d_images, nframes))) return rc;
        fr.image[f] = d_images[f];
        fr.clear[f] = incremental ? 1 : 0;
    }
    launch_assemble_sparse_batch(sp, nshards, frame_stride_words, fr, nframes, g.max_cells, g.frame_cells, g.t, g.cx, g.cy, c->w, c->h, g.bg, st);
    HIPCHK(c, hipGetLastError());
    return VRT_HIP_OK;
}

// ---- point queries ------------------------------------------------------------------------------
static int upload(vrt_hip_ctx *c, DevBuf<float> &b, const float *src, size_t n)
{
    HIPCHK(c, b.reserve(n));
    if (n) HIPCHK(c, hipMemcpy(b.p, src, n * 4, hipMemcpyHostToDevice));
    return VRT_HIP_OK;
}

int vrt_hip_transmittance(vrt_hip_ctx *c, const float o[3], const float n[3], const float *s, size_t ns, float *T_out)
{
    if (!c || !o || !n || (ns && (!s || !T_out))) return VRT_HIP_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = rebuild_tables(c);
    if (rc) return rc;
    DevBuf<float> ds, dT;
    if ((rc = upload(c, ds, s, ns))) return rc;
    HIPCHK(c, dT.reserve(ns));
    launch_transmittance(tables(c), o, n, ds.p, ns, dT.p, c->exp_kind, c->erf_kind, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (ns) HIPCHK(c, hipMemcpy(T_out, dT.p, ns * 4, hipMemcpyDeviceToHost));
    ds.release(); dT.release();
    return VRT_HIP_OK;
}

int vrt_hip_transmittance_rays(vrt_hip_ctx *c, size_t nrays, const float *origins, const float *dirs, const float *s,
                               float *T_out)
{
    if (!c || (nrays && (!origins || !dirs || !s || !T_out))) return VRT_HIP_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = rebuild_tables(c);
    if (rc) return rc;
    DevBuf<float> d_o, d_d, ds, dT;
    if ((rc = upload(c, d_o, origins, nrays * 3))) return rc;
    if ((rc = upload(c, d_d, dirs, nrays * 3))) return rc;
    if ((rc = upload(c, ds, s, nrays))) return rc;
    HIPCHK(c, dT.reserve(nrays));
    launch_transmittance_rays(tables(c), d_o.p, d_d.p, ds.p, nrays, dT.p, c->exp_kind, c->erf_kind, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (nrays) HIPCHK(c, hipMemcpy(T_out, dT.p, nrays * 4, hipMemcpyDeviceToHost));
    d_o.release(); d_d.release(); ds.release(); dT.release();
    return VRT_HIP_OK;
}

int vrt_hip_transmittance_step(vrt_hip_ctx *c, const float o[3], const float n[3], const float *s, size_t ns, float delta,
                               float *T_out)
{
    if (!c || !o || !n || (ns && (!s || !T_out)) || !(delta > 0.f)) return VRT_HIP_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = rebuild_tables(c);
    if (rc) return rc;
    DevBuf<float> ds, dT;
    if ((rc = upload(c, ds, s, ns))) return rc;
    HIPCHK(c, dT.reserve(ns));
    launch_transmittance_step(tables(c), o, n, ds.p, ns, delta, dT.p, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (ns) HIPCHK(c, hipMemcpy(T_out, dT.p, ns * 4, hipMemcpyDeviceToHost));
    ds.release(); dT.release();
    return VRT_HIP_OK;
}

int vrt_hip_density(vrt_hip_ctx *c, size_t npts, const float *pts, float *D_out)
{
    if (!c || (npts && (!pts || !D_out))) return VRT_HIP_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = rebuild_tables(c);
    if (rc) return rc;
    DevBuf<float> dp, dD;
    if ((rc = upload(c, dp, pts, npts * 3))) return rc;
    HIPCHK(c, dD.reserve(npts));
    launch_density(tables(c), dp.p, npts, dD.p, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (npts) HIPCHK(c, hipMemcpy(D_out, dD.p, npts * 4, hipMemcpyDeviceToHost));
    dp.release(); dD.release();
    return VRT_HIP_OK;
}

int vrt_hip_radiance(vrt_hip_ctx *c, size_t nrays, const float *origins, const float *dirs, float *out)
{
    if (!c || (nrays && (!origins || !dirs || !out))) return VRT_HIP_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = rebuild_tables(c);
    if (rc) return rc;
    DevBuf<float> d_o, d_d;
    DevBuf<float4> d_out;
    if ((rc = upload(c, d_o, origins, nrays * 3))) return rc;
    if ((rc = upload(c, d_d, dirs, nrays * 3))) return rc;
    HIPCHK(c, d_out.reserve(nrays));
    launch_radiance(tables(c), d_o.p, d_d.p, nrays, c->iota.p, d_out.p, c->exp_kind, c->erf_kind, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (nrays) HIPCHK(c, hipMemcpy(out, d_out.p, nrays * 16, hipMemcpyDeviceToHost));
    d_o.release(); d_d.release(); d_out.release();
    return VRT_HIP_OK;
}

static int eval_common(vrt_hip_ctx *c, bool is_erf, int kind, const float *x, size_t n, float *y)
{
    if (!c || (n && (!x || !y))) return VRT_HIP_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    DevBuf<float> dx, dy;
    int rc = upload(c, dx, x, n);
    if (rc) return rc;
    HIPCHK(c, dy.reserve(n));
    if (is_erf) launch_eval_erf(kind, dx.p, n, dy.p, c->stream); else launch_eval_exp(kind, dx.p, n, dy.p, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (n) HIPCHK(c, hipMemcpy(y, dy.p, n * 4, hipMemcpyDeviceToHost));
    dx.release(); dy.release();
    return VRT_HIP_OK;
}
int vrt_hip_eval_erf(vrt_hip_ctx *c, int kind, const float *x, size_t n, float *y) { return eval_common(c, true, kind, x, n, y); }
int vrt_hip_eval_exp(vrt_hip_ctx *c, int kind, const float *x, size_t n, float *y) { return eval_common(c, false, kind, x, n, y); }

int vrt_hip_enable_stats(vrt_hip_ctx *c, int on)
{
    if (!c) return VRT_HIP_ERR_INVALID;
    c->stats_on = on != 0;
    return VRT_HIP_OK;
}

int vrt_hip_enable_kernel_timing(vrt_hip_ctx *c, int on)
{
    if (!c) return VRT_HIP_ERR_INVALID;
    c->timing_on = on != 0;
    c->timing_full = on == 1; // 2, 3: events around the one-wave render kernel only (two per frame instead of four)
    c->timing_period = on == 3 ? 8 : 1; // 3: on every 8th frame only
    if (on) {
        c->timing_count = 0; c->timing_frame = 0;
        if (c->tev.empty()) { // here, not in the first timed frame
            HIPCHK(c, hipSetDevice(c->device));
            c->tev.resize(4 * vrt_hip_ctx::TIMING_RING);
            for (auto &e : c->tev) HIPCHK(c, hipEventCreate(&e));
        }
    }
    return VRT_HIP_OK;
}

int vrt_hip_get_kernel_timing(vrt_hip_ctx *c, double *render_ms, double *dense_ms, double *lists_ms, uint64_t *launches)
{
    if (!c) return VRT_HIP_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    const uint64_t n = std::min<uint64_t>(c->timing_count, vrt_hip_ctx::TIMING_RING);
    double sr = 0, sd = 0, sl = 0;
    for (uint64_t i = 0; i < n; ++i) {
        hipEvent_t *e = &c->tev[4 * i];
        HIPCHK(c, hipEventSynchronize(e[c->timing_full ? 3 : 2]));
        float a = 0, b = 0, d = 0;
        HIPCHK(c, hipEventElapsedTime(&b, e[1], e[2]));
        if (c->timing_full) {
            HIPCHK(c, hipEventElapsedTime(&a, e[0], e[1]));
            HIPCHK(c, hipEventElapsedTime(&d, e[2], e[3]));
        }
        sl += a; sr += b; sd += d;
    }
    if (render_ms) *render_ms = n ? sr / n : 0.0;
    if (dense_ms) *dense_ms = n ? sd / n : 0.0;
    if (lists_ms) *lists_ms = n ? sl / n : 0.0;
    if (launches) *launches = n;
    return VRT_HIP_OK;
}

int vrt_hip_get_stats(vrt_hip_ctx *c, vrt_hip_stats *out)
{
    if (!c || !out) return VRT_HIP_ERR_INVALID;
    *out = c->last;
    return VRT_HIP_OK;
}

} // extern "C"
