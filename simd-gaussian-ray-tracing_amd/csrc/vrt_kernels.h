// vrt_kernels.h -- launch interface between the C-ABI host code (vrt_hip_api.cpp) and the
// gfx950 kernels (vrt_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vrtk {

constexpr int BLOCK_W = 8;        // one wavefront = one 8x8 pixel block (64 rays)
constexpr int BLOCK_H = 8;
constexpr int CELL = 32;          // second-level cull region: 32x32 pixels = 4x4 blocks
constexpr int TCAP = 1024;        // a tile's candidates kept in LDS by the fused list kernel
constexpr int CH_CAP = 2048;       // chunks of 64 Gaussians a tile may keep after the chunk test (more: every Gaussian is tested, as before round 3)
constexpr int MAX_FUSED_CELLS = 64; // more cells per tile than this: separate cell kernel (one wave per cell)
#ifndef VRT_PCAP
#define VRT_PCAP 96
#endif
#ifndef VRT_PL
#define VRT_PL 24
#endif
constexpr int PCAP = VRT_PCAP;    // per-block candidates cached in LDS (four parameter rows + sigma*mag = 68 B each; 96 since round 3 = the dense threshold: a block of
                                  // this kernel never has more)
#ifndef VRT_DCAP
#define VRT_DCAP 1024
#endif
constexpr int DCAP = VRT_DCAP;        // per-block candidates the dense kernel keeps in LDS (61 KB with the rest; measured: no change below 64 KB, +4 % at 81 KB, +10 % at 104 KB)
constexpr uint32_t ACTIVE_COUNT_SHIFT = 24, ACTIVE_CELL_MASK = 0xFFFFFFu; // CellGrid::active entries: cell id | min(list length, 255) << 24
constexpr int PL = VRT_PL;        // per-lane list capacity (u8 positions into the block's candidates).  A block with a longer per-ray list goes to the
                                  // 16-wave kernels.  Round 4: 24, not 48 -- the pair loops are quadratic in the list length and a block is ONE wave here: a
                                  // block with lists of 40 is 10^5 instructions on a wave that ends up alone on its SIMD, and the launch waits for it
                                  // (`-f cube.obj` 0.59 -> 0.33 ms per frame, teapot 2048^2 5.83 -> 5.63 ms, monkey 4096^2 12.1 -> 11.8 ms; 16 gains nothing
                                  // more and costs the far side of the monkey orbit 2 %: gpurun_out/r04, profiles/r04_experiments.md)

// Device-resident scene tables, 16 B rows for 128-bit (scalar) loads.
struct SceneTables {
    const float4 *mu_sig;   // static: (mu.x, mu.y, mu.z, sigma)
    const float4 *gA;       // per frame: (oc.x, oc.y, oc.z, |oc|^2), oc = mu - origin
    const float4 *gB;       // static: (r = 1/(sqrt2 sigma), 1/(2 sigma^2), K*sigma*mag, cull_x)
    const float4 *gC;       // static: albedo rgba
    const float4 *gD;       // static: (sigma, sigma*mag, mag, 0)
    uint32_t n;
};

struct TileLists {
    const uint32_t *start;    // [ntiles] first entry of tile t in `indices`
    const uint32_t *count;    // [ntiles]
    const uint32_t *indices;
    uint32_t tiles_w, tiles_h;
    uint32_t tile_w, tile_h;  // pixels: (u64)(width*tw/2.f), rt.h:348-349
    uint32_t stride;          // tile_w * tiles_w: the reference's row stride, rt.h:364-365
    // Level-wise cull thresholds (cull_ref_n = 0: one threshold everywhere).  cull_eps bounds what ONE dropped Gaussian
    // could have contributed; what a ray loses is the sum over everything dropped on its way through the levels.  A level
    // that `n` candidates enter (cell level: the tile's work list; block level: the cell's list; ray level: the block's
    // survivors) drops a candidate when its best-case contribution is below cull_eps * cull_ref_n / n, so that the level as a
    // whole loses less than ~3 * cull_ref_n * cull_eps whatever n is: short lists are cut harder, long lists more carefully
    // than with one fixed threshold.  In the tables that is cull_x - ln(cull_ref_n / n).  Thresholds at the Exp floor
    // (`floor_x`: cull_eps = 0, or sigma*mag/cull_eps beyond it) mean "keep unless Exp gives exactly 0" and stay.
    float cull_ref_n;
    float floor_x;
};


// Second-level (cell) candidate lists and the cell queues of the render kernels (empty cells: count == 0, no queue).
struct CellGrid {
    uint32_t cells_x, cells_y;       // cells per tile
    uint32_t cstride;                // capacity of a cell's list slot
    uint32_t *count;                 // [cells]; 0xFFFFFFFF = overflowed its slot: use the tile's list
    uint32_t *indices;               // cell c at indices + c*cstride
    uint32_t n_cells;                // cells of all local tiles
    uint32_t *active;                // cells with short lists (shaded one wavefront per block): cell id | list length << 24 (ACTIVE_COUNT_SHIFT)
    uint32_t *dense;                 // cell ids with long lists: shaded one 16-wave workgroup per block
    uint32_t *scratch;               // dense kernel: cstride words per workgroup (a block's survivors when they outgrow LDS)
    uint32_t *dense_sorted;          // the same, longest list first (order_dense_kernel): the queue order of the dense kernel
    uint32_t *slot;                  // [cells] position of a non-empty cell in its queue (bit 31: the dense queue): its place in
                                     // a sparse shard (RenderTarget::sparse) -- written by the list kernels
    uint32_t *n_active, *n_dense;    // device counters, zeroed before the list kernels add to them
    // Cells with short lists ("light": at most light_threshold candidates -- the rim of what the scene covers) are filed from
    // the BACK of `active` (n_light of them: active[n_cells - 1 - k]) and shaded last: a frame rarely has a multiple of the
    // GPU's wave slots in blocks, and the blocks that run as a fourth wave on their SIMD should be the cheap ones.
    // Raster frames only: a sparse shard needs its cells in one contiguous run of slots (light_threshold = 0 there).
    // (Round 4 tried to queue only the BLOCKS of light cells that some candidate reaches: tools/experiments/light_block_masks.patch.)
    uint32_t *n_light;
    uint32_t light_threshold;
    uint32_t *dense_next;            // work counter of the dense kernel (zeroed with the others)
    uint32_t *overflow, *n_overflow; // blocks (cell*16 + block) whose per-ray lists outgrew the one-wave kernel's
                                     // LDS slots: it hands them to the dense kernel, which runs after it
    uint32_t dense_threshold;        // a cell whose list is longer than this goes to the dense queue
    float prune_budget;              // block kernel: a ray may drop the smallest entries of its list while their sum (units of the tile level's eps) stays below this (prune_list; 0 = off)
    int claim_early;                 // block kernel: a wave claims its next queue entry before it shades the current block when the queues hold at least
                                     // 1/claim_early of the grid size in entries (VRT_HIP_CLAIM_EARLY; 0: never -- after the block, as in rounds 1-2)
    // Launch feedback (host-mapped memory, nullable): [0] = dense cells of the frame, [1] = blocks of its active cells (written by the one-wave kernel),
    // [2] = items (blocks) the dense kernel found in its queues, [3] = sequence number of the frame that wrote [2]
    // The host reads it frames later to SIZE the dense launch: a frame that is expected to have nothing
    // for it gets a handful of workgroups instead of one per CU and no queue sort.  Which kernel shades a block never
    // depends on it.
    uint32_t *feedback;
    uint32_t frame_seq;
    int dense_is_sorted;             // the dense kernel reads dense_sorted (order_dense_kernel ran) or dense (it did not)
    // Work queues of the one-wave kernel: a wave's first block is static (item = wave), the blocks beyond the grid size
    // are pulled from RQ_N counters RQ_STRIDE words apart (item G + q + RQ_N*m is the m-th of queue q).  `rq` is this
    // launch's set (zero on entry), `rq_next` the other set, which this launch clears for the next one.
    uint32_t *rq, *rq_next;
    // table mode (render_table_kernel): requested node spacing in units of 1/r (0 = off) and the largest change of a ray's
    // radiance the table may cause (worst-case bound, checked per ray; a block that fails it is shaded exactly by the same workgroup)
    float table_hx, table_budget;
    float table_room;                // ... and the share of the budget its estimate may fill when it does (the estimate is not the bound)
    float table_adapt;               // the table kernel may choose a spacing up to this many times the requested one where its estimate
                                     // of the error bound leaves room (1 = always as requested)
};
#ifndef VRT_RQ_N
#define VRT_RQ_N 64
#endif
constexpr uint32_t RQ_N = VRT_RQ_N, RQ_STRIDE = 64; // RQ_N <= 64: a wave looks at all counters at once, one per lane

struct RayGen {
    const float *xs, *ys, *zs; // plane arrays (nullptr => basis mode)
    float origin[3];
    float pos[3], right[3], up[3], front[3];
    float focal, inv_half_w, inv_half_h;
    uint32_t width, height;
    // view mode (vrt_hip_set_camera_view): the reference's own plane points, inverse(view) * (x, y, 0, 1) with
    // x = -1 + j / (w/2), y = -1 + i / (h/2) (camera.cpp:60-69), evaluated in glm's order -- bit-identical rays
    int view_mode;
    float m0[3], m1[3], m3[3];  // columns 0, 1, 3 of inverse(view) (column 2 meets the 0 of the point)
    float half_w, half_h;
};

struct RenderTarget {
    uint32_t *image;          // nullable
    float4 *radiance;         // nullable
    int pack_flags;
    // sharding: tile_map[lt] = global tile id of local tile lt (nullptr = identity);
    // compact != 0 => image is the tile-major shard buffer [lt][tile_h][tile_w]
    const uint32_t *tile_map;
    uint32_t n_local_tiles;
    int compact;
    int cleared;              // empty cells were already cleared by the list kernel of this frame
    // Retained frame buffer (vrt_hip_frame's own buffer): `image` still holds the previous frame of this context at this
    // geometry.  stamp[tile * cells per tile + cell] = stamp_seq of the last frame the cell was lit in; the list kernel then
    // clears only the empty cells that were lit in frame stamp_seq - 1 (16 MB of a 2048^2 frame's 18.7 MB of HBM traffic are
    // background written over background otherwise).  nullptr: every empty cell is cleared.
    uint32_t *stamp;
    uint32_t stamp_seq;
    // Sparse shard (multi-GPU transport, vrt_hip_frame_sparse_device): only the 32x32-px cells some Gaussian reaches
    // are stored, cell-major: `image` points at the pixel region, pixel (cx, cy) of the cell in slot s at
    // image[s*1024 + cy*32 + cx]; slot = position in the active queue, or n_active + position in the dense queue.
    // keys[slot] = tile id * cells per tile + cell in tile; sparse_hdr[0] = number of cells.  Empty cells are neither
    // stored nor cleared: whoever assembles the frame fills the background.
    int sparse;
    uint32_t *keys, *sparse_hdr;
    uint32_t sparse_cap;      // capacity (cells) of the shard buffer: header word 1, locates the pixel region
    unsigned long long *stats; // nullable: [0]=block candidates [1]=tile entries [2]=dense blocks with more than DCAP survivors
                               // [3]=sum of lane list lengths [4]=sum over blocks of the longest lane list [5]=shaded blocks
                               // [6]=dense blocks [7]=table blocks [8..11]=dense workgroup timeline [12]=sum over rays of (lane list length)^2
                               // [13..15]=dense kernel: (emitter chunk, absorber) visits evaluated in full / exactly zero / exactly -2A
                               // [16]=table kernel: sum of node counts over its blocks [17]=blocks it did at the reduced spacing
                               // [18]=(absorber, wave) visits it settled by saturation [19]=blocks it declined [20]=blocks it did at a coarser spacing
                               // than requested [21]=blocks of its queue that no Gaussian reaches
                               // [24..31]=table kernel: 10-ns ticks per phase, summed over blocks   (32 words in all)
    unsigned long long *timeline; // nullable diagnostics: 4 wall_clock64 stamps + the hardware id per one-wave work item (5 words)
};

void launch_prep_frame(const SceneTables &s, float4 *gA_out, const float origin[3], hipStream_t st);
void launch_build_static(uint32_t n, const float *mu_x, const float *mu_y, const float *mu_z, const float *ar,
                         const float *ag, const float *ab, const float *aa, const float *sigma, const float *mag,
                         float cull_eps, float exp_floor_x, float4 *mu_sig, float4 *gB, float4 *gC, float4 *gD,
                         hipStream_t st);
void launch_render(const SceneTables &s, const TileLists &t, const CellGrid &c, const RayGen &r, const RenderTarget &o,
                   uint32_t grid /* one-wave workgroups */, int exp_kind, int erf_kind, hipStream_t st);
void launch_render_dense(const SceneTables &s, const TileLists &t, const CellGrid &c, const RayGen &r,
                         const RenderTarget &o, uint32_t grid, int dw /* waves per block: 4, 8 or 16 */, int exp_kind,
                         int erf_kind, hipStream_t st);
void launch_order_dense(const CellGrid &c, hipStream_t st);
void launch_render_table(const SceneTables &s, const TileLists &t, const CellGrid &c, const RayGen &r,
                         const RenderTarget &o, uint32_t grid, int exp_kind, int erf_kind, hipStream_t st);
void launch_build_cell_lists(const SceneTables &s, const TileLists &t, const CellGrid &c, const RayGen &r,
                             const uint32_t *tile_map, uint32_t n_cells, int refine, uint32_t *keys, hipStream_t st);

// per-tile list construction: tile binning (rt.cpp:29-69) and/or tile-level cull ("refine")
struct Mat4 { float m[16]; };
struct BinArgs {
    const float4 *mu_sig, *gA, *gB;
    uint32_t n;
    float4 *prep_gA;                 // nullable: the list kernel also writes the per-origin table (centre - origin, squared norm) of ALL Gaussians for the
                                     // kernels behind it -- what prep_frame_kernel does, without its launch; the list kernel itself works from mu_sig
    const float4 *chunks;            // nullable: bounding sphere (centre, radius incl. the members' reach) of every 64 consecutive Gaussians (launch_build_chunks)
    // input: caller-made lists (from_list) ...
    const uint32_t *in_start, *in_count, *in_indices;
    // ... or on-device binning with the reference's test
    Mat4 V;
    const float *xc, *yc;
    float tw, th;
    uint32_t tiles_w;
    // tile-level cull
    int refine;
    RayGen R;
    uint32_t tile_w, tile_h, stride;
    // output: list of tile t at out_indices[out_start[t] ...], length out_count[t]
    const uint32_t *out_start;
    uint32_t *out_indices, *out_count;
    float cull_ref_n, floor_x;       // TileLists::cull_ref_n, floor_x
    // nullable: the tile cones and their cells' cones, two float4 per cone (axis, cos | sin, -, -, -), made by launch_tile_cones for these rays
    // and this tile geometry -- they depend on the camera only, so a frame with the camera of the last one reuses them
    // (otherwise every wave of the workgroup builds the tile's cone itself)
    float4 *tile_cones;
    uint32_t cone_gen;               // a row of tile_cones is valid if it carries this tag (second float4, .y): rows are filled by whoever needs them first --
                                     // the list kernel's workgroups themselves (a camera that moves never pays a table launch) or tile_cones_kernel (batches)
    int cones_known;                 // 0: the camera of this frame is new, no row can carry its tag: do not even read them
    uint32_t cones_cells;            // cells per tile the table holds cones for (row of tile t: t * (1 + cones_cells); 0: tile cones only)
    uint32_t *zero8;                 // nullable: 8 queue counters this launch clears for the kernels after it
    uint32_t *next_zero8;            // nullable: the OTHER counter set, cleared for the next list generation
};
// optional fused second level (see build_tile_lists_kernel)
struct FuseArgs {
    int enabled;
    int do_clear;
    const uint32_t *tile_map;  // local tile -> tile id (nullptr = identity); the grid is over LOCAL tiles
    CellGrid C;
    RenderTarget O;
    unsigned long long *timeline; // nullable diagnostics: 8 wall_clock64 stamps per tile workgroup
};
void launch_build_tile_lists(const BinArgs &a, const FuseArgs &f, bool from_list, uint32_t ntiles, hipStream_t st);
void launch_build_chunks(uint32_t n, const float4 *mu_sig, const float4 *gB, float4 *chunks, hipStream_t st);

// Several frames per launch (vrt_hip_frame_batch_device): what one frame's three kernels take, as a row of a device array;
// the batch variants of the kernels are the same code with blockIdx.y choosing the row.  A frame of a sparse scene is a
// few tens of microseconds of dependent launches: n frames per launch pay the launch gaps and the tails once, and a rank
// that owns an eighth of the tiles still fills its GPU.
struct FrameArgs {
    BinArgs bin; FuseArgs fuse;                                   // list kernel
    SceneTables S; TileLists T; CellGrid C; RayGen R; RenderTarget O;   // one-wave kernel and dense kernel
    // the frame's small set-up kernels, batched too (a launch per frame and member costs more than the work: 512 launches of
    // 14-20 us in a 128-frame orbit on four members, profiles/r03_group.md)
    int do_prep;  float4 *prep_gA; float prep_origin[3];          // prep_frame_kernel: oc = mu - origin, |oc|^2
    int do_cones; uint32_t cones_tiles, cones_cx, cones_cy; float4 *cones_out; // tile_cones_kernel for this frame's camera (bin = its rays and tile geometry)
    int do_order;                                                 // order_dense_kernel
};
void launch_build_tile_lists_batch(const FrameArgs *d_frames, uint32_t nframes, bool from_list, bool chunks, uint32_t ntiles, hipStream_t st);
void launch_render_batch(const FrameArgs *d_frames, uint32_t nframes, uint32_t grid, bool claim, int exp_kind, int erf_kind, hipStream_t st);
void launch_render_dense_batch(const FrameArgs *d_frames, uint32_t nframes, uint32_t grid, int dw, int exp_kind, int erf_kind,
                               hipStream_t st);
// the per-frame set-up kernels of a batch, one launch each: prep_frame (grid.y = frame), tile_cones, order_dense
void launch_frame_setup_batch(const FrameArgs *d_frames, const FrameArgs *h_frames, uint32_t nframes, hipStream_t st);
void launch_order_dense_batch(const FrameArgs *d_frames, const FrameArgs *h_frames, uint32_t nframes, hipStream_t st);
void launch_render_table_batch(const FrameArgs *d_frames, uint32_t nframes, uint32_t grid, uint64_t npix /* of a frame: all the same size */,
                               int exp_kind, int erf_kind, hipStream_t st);
void launch_assemble(const uint32_t *gathered, uint32_t *image, const uint32_t *tile_of_slot, uint32_t slots_per_rank,
                     uint32_t world, uint64_t rank_stride, const TileLists &t, uint32_t width, uint32_t height, hipStream_t st);
void launch_iota(uint32_t *p, uint32_t n, hipStream_t st);
// frame assembly from sparse shards: image = background, then every stored cell of every shard to its place
constexpr int MAX_SHARDS = 64;
struct ShardPtrs { const uint32_t *p[MAX_SHARDS]; };
void launch_scatter_sparse(const ShardPtrs &shards, int nshards, uint32_t max_cells, uint32_t *image, const TileLists &t,
                           uint32_t cells_x, uint32_t cells_y, uint32_t width, uint32_t height, uint32_t *stamp /* nullable */,
                           uint32_t seq, hipStream_t st);
// several frames per assembly launch: per frame its image, its stamps (nullable), its sequence number, and whether the
// stale-cell pass applies (0: the buffer got the full background fill this time)
constexpr int MAX_ASSEMBLY_FRAMES = 64;
struct AssemblyFrames {
    uint32_t *image[MAX_ASSEMBLY_FRAMES];
    uint32_t *stamp[MAX_ASSEMBLY_FRAMES];
    uint32_t seq[MAX_ASSEMBLY_FRAMES];
    uint8_t clear[MAX_ASSEMBLY_FRAMES];
};
void launch_assemble_sparse_batch(const ShardPtrs &shards, int nshards, size_t frame_stride, const AssemblyFrames &frames, int nframes,
                                  uint32_t max_cells, uint32_t n_cells, const TileLists &t, uint32_t cells_x, uint32_t cells_y,
                                  uint32_t width, uint32_t height, uint32_t background, hipStream_t st);
void launch_clear_stale_cells(const uint32_t *stamp, uint32_t seq, uint32_t n_cells, uint32_t *image, const TileLists &t,
                              uint32_t cells_x, uint32_t cells_y, uint32_t width, uint32_t height, uint32_t background, hipStream_t st);
constexpr uint32_t SPARSE_HDR_WORDS = 4; // [0] cells stored, [1] capacity (cells), [2] cells per tile, [3] reserved
__host__ __device__ inline size_t sparse_pixel_offset(uint32_t cap) { return (SPARSE_HDR_WORDS + (size_t)cap + 3) / 4 * 4; }

// point queries
void launch_transmittance(const SceneTables &s, const float o[3], const float n[3], const float *d_s, size_t ns,
                          float *d_T, int exp_kind, int erf_kind, hipStream_t st);
void launch_transmittance_rays(const SceneTables &s, const float *d_origins, const float *d_dirs, const float *d_s,
                               size_t nrays, float *d_T, int exp_kind, int erf_kind, hipStream_t st);
void launch_transmittance_step(const SceneTables &s, const float o[3], const float n[3], const float *d_s, size_t ns,
                               float delta, float *d_T, hipStream_t st);
void launch_density(const SceneTables &s, const float *d_pts, size_t npts, float *d_D, hipStream_t st);
// iota: identity index list 0..n-1 on device
void launch_radiance(const SceneTables &s, const float *d_origins, const float *d_dirs, size_t nrays,
                     const uint32_t *iota, float4 *d_out, int exp_kind, int erf_kind, hipStream_t st);
void launch_eval_erf(int kind, const float *x, size_t n, float *y, hipStream_t st);
void launch_eval_exp(int kind, const float *x, size_t n, float *y, hipStream_t st);

} // namespace vrtk
