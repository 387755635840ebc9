/*
 * vrt_hip.h -- C ABI of libvrt_hip.so: the MI355X (gfx950) implementation of the
 * reference's render / radiance / transmittance path.
 *
 * The reference has no FFI layer: the path is a set of C++ header templates in
 * namespace vrt (src/vrt/rt.h:16-405) instantiated by its callers
 * (src/volumetric-ray-tracer/main.cpp:271-292, tests/transmittance.cpp:27-30,
 * tests/img-error.cpp:34-43).  Each entry point below names the reference
 * interface it replaces.  The C++ header include/vrt/vrt.hpp re-creates the
 * vrt:: signatures on top of this ABI; INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types.
 *   - every function returns VRT_HIP_OK (0) or a negative vrt_hip_status; nothing
 *     exits the process (the reference _exit(1)s, include/definitions.h:23-30).
 *     vrt_hip_last_error() gives the message of the last failure of a context.
 *   - the caller owns every host pointer; set_* calls copy; nothing is retained.
 *   - a context is bound to one device and is not thread-safe (the reference
 *     renders from one thread, main.cpp:257-296).
 *   - *_device variants take DEVICE pointers and a hipStream_t (as void*) and are
 *     asynchronous; the others take HOST pointers and return after completion.
 *   - Streams: frames enqueued through a *_device call run on the caller's stream and read
 *     the context's tables, lists and plane arrays.  Every call that rewrites one of those
 *     (set_gaussians*, set_plane, set_tiles, tile_gaussians with another tile size, a frame
 *     on another stream) first waits for the frames in flight, so no synchronisation by the
 *     caller is needed before changing state.  A stream passed to a *_device call has to
 *     stay valid until the context's next state change, vrt_hip_sync() or destroy.
 *
 * All paths are relative to /root/reference/src.
 */
#ifndef VRT_HIP_H
#define VRT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vrt_hip_ctx vrt_hip_ctx;

typedef enum {
    VRT_HIP_OK = 0,
    VRT_HIP_ERR_INVALID = -1,   /* bad argument / call order            */
    VRT_HIP_ERR_HIP = -2,       /* a HIP runtime call failed             */
    VRT_HIP_ERR_NO_DEVICE = -3, /* no gfx950 device / no GPU             */
    VRT_HIP_ERR_NOMEM = -4
} vrt_hip_status;

/* Exp / Erf template arguments of the reference (rt.h:32,61,102,205,315,344;
 * defaults approx.h:110-127).  LIBM = expf/erff (rt.h:32 defaults), VCL = vcl_exp
 * (approx.h:91-106; ~1 ulp, flushes to 0 below -87.3), AS = Abramowitz-Stegun
 * (approx.cpp:90-110).  The others are the alternates of approx.cpp:9-188. */
enum { VRT_EXP_LIBM = 0, VRT_EXP_VCL = 1, VRT_EXP_FAST = 2, VRT_EXP_SPLINE = 3 };
enum { VRT_ERF_LIBM = 0, VRT_ERF_AS = 1, VRT_ERF_SPLINE = 2, VRT_ERF_SPLINE_MIRROR = 3, VRT_ERF_TAYLOR = 4 };

/* Pixel packing (rt.h:239-243 / 329-333 / 373-377): u32 = A<<24 | R<<16 | G<<8 | B. */
enum {
    VRT_PACK_TRUNC = 0,     /* (u32)(min(c,1)*255): scalar render_image, rt.h:240-242, 280-282        */
    VRT_PACK_ROUND = 1,     /* round-to-nearest-even (cvts): simd_render_image, rt.h:330-332, 374-376  */
    VRT_ALPHA_OPAQUE = 0,   /* A = 0xFF: rt.h:239, 279, 329                                           */
    VRT_ALPHA_COMPUTED = 2  /* A = round(min(1, sum albedo.w*inner)*255): tiled simd_render_image, rt.h:373 */
};

/* -------- context ---------------------------------------------------------------- */
/* Creates a context on HIP device `device`.  Fails with VRT_HIP_ERR_NO_DEVICE when no
 * GPU is visible -- there is no CPU fallback. */
int vrt_hip_create(int device, vrt_hip_ctx **out);
void vrt_hip_destroy(vrt_hip_ctx *ctx);
const char *vrt_hip_last_error(const vrt_hip_ctx *ctx); /* ctx may be NULL: last create() error */
const char *vrt_hip_version(void);

/* -------- scene: replaces gaussians_t / gaussian_vec_t (types.h:232-270) ----------- */
/* SoA upload == gaussian_vec_t::from_gaussians / load_gaussians (types.cpp:8-76).
 * albedo_a may be NULL (treated as 1, like simd_radiance rt.h:176).  n may be 0. */
int vrt_hip_set_gaussians(vrt_hip_ctx *ctx, size_t n, const float *mu_x, const float *mu_y, const float *mu_z,
                          const float *albedo_r, const float *albedo_g, const float *albedo_b,
                          const float *albedo_a, const float *sigma, const float *magnitude);
/* AoS upload == std::vector<gaussian_t> (types.h:195-200: albedo[4], mu[4], sigma, magnitude; 40 B). */
int vrt_hip_set_gaussians_aos(vrt_hip_ctx *ctx, size_t n, const void *gaussians);

/* -------- tiling: replaces tile_gaussians + tiles_t (rt.cpp:29-69, types.h:272-287) - */
/* On-device binning with the reference's exact inclusion test, from the camera's view
 * matrix (column-major float[16], glm::mat4 layout).  tiles.w/h = ceil(2/tw), ceil(2/th). */
int vrt_hip_tile_gaussians(vrt_hip_ctx *ctx, float tw, float th, const float view[16]);
/* Same, asynchronous on `hip_stream` (no host synchronisation once tw/th have been seen before):
 * the per-frame form for animation loops (main.cpp:263 runs it every frame). */
int vrt_hip_tile_gaussians_device(vrt_hip_ctx *ctx, float tw, float th, const float view[16], void *hip_stream);
/* Caller-made tile sets as index lists into the uploaded scene: tile t (row-major,
 * tidx = ty*tiles_w + tx, rt.cpp:47-51) holds indices[offsets[t] .. offsets[t+1]). */
int vrt_hip_set_tiles(vrt_hip_ctx *ctx, float tw, float th, uint64_t tiles_w, uint64_t tiles_h,
                      const uint32_t *offsets, const uint32_t *indices);
/* Untiled rendering (the gaussians_t overloads, rt.h:227-228, 315-316): every ray sees every Gaussian. */
int vrt_hip_clear_tiles(vrt_hip_ctx *ctx);
/* Copies the current per-tile counts (tiles_w*tiles_h entries) to the host; for tests/statistics. */
int vrt_hip_get_tile_counts(vrt_hip_ctx *ctx, uint32_t *counts, size_t cap, uint64_t *tiles_w, uint64_t *tiles_h);
/* Copies tile t's index list (ascending scene order, like rt.cpp:52-62) to the host. */
int vrt_hip_get_tile_indices(vrt_hip_ctx *ctx, uint64_t t, uint32_t *indices, size_t cap, uint32_t *count);

/* -------- rays: replaces camera_t::projection_plane (camera.h:28-33, camera.cpp:50-71) */
/* Reference-style ray source: three w*h arrays of world-space plane points; ray i has
 * direction normalize(plane[i] - origin) (rt.h:231-237, 321-326, 366-371). */
int vrt_hip_set_plane(vrt_hip_ctx *ctx, uint32_t w, uint32_t h, const float *xs, const float *ys, const float *zs);
/* In-kernel ray generation from the camera basis: plane(i,j) = pos + x*right + y*up - focal*front,
 * x = -1 + j/(w/2), y = -1 + i/(h/2) -- the closed form of camera.cpp:52,60-69 (no 12 B/ray read). */
int vrt_hip_set_camera(vrt_hip_ctx *ctx, uint32_t w, uint32_t h, const float pos[3], const float right[3],
                       const float up[3], const float front[3], float focal);
/* In-kernel rays, bit for bit the reference's: the projection-plane point of pixel (i, j) is
 * inverse(view) * (-1 + j/(w/2), -1 + i/(h/2), 0, 1) (camera.cpp:60-69), evaluated per ray with glm's inverse and
 * glm's mat4*vec4 in their order of operations.  `view` = camera_t::view (column-major, glm layout).  Same image as
 * vrt_hip_set_plane() with the arrays camera_t would build, without the 12 B per ray.  (vrt_hip_set_camera above
 * takes the closed form pos + x right + y up - focal front instead: the same rays up to the rounding of the plane
 * points -- which the reference's |oc|^2 - mubar^2 can amplify to 1e-4 for small sigma.) */
int vrt_hip_set_camera_view(vrt_hip_ctx *ctx, uint32_t width, uint32_t height, const float view[16]);

/* width * height of the current ray set-up (0 before any set_plane / set_camera*). */
size_t vrt_hip_image_pixels(const vrt_hip_ctx *ctx);

/* -------- host camera: replaces camera_t (camera.h:20-44, camera.cpp:7-71) -------------------------------- */
/* Pure host code (no GPU needed, no context): the reference's yaw/pitch camera evaluated in glm's order of operations
 * (lookAtRH -> translate -> inverse -> mat4*vec4, unfused), so that view matrices and projection-plane points are the
 * reference's to the last bit whatever floating-point flags the CALLER is compiled with (csrc/vrt_host_camera.cpp).
 * Field meaning as in camera_t; `view` is camera_t::view_matrix, column-major (glm::mat4 layout). */
typedef struct {
    float position[3], front[3], up[3], world_up[3], right[3];
    float view[16];
    float focal_length;
    uint64_t w, h;
} vrt_hip_camera;
/* camera_t::camera_t (camera.cpp:25-36): stores the arguments, then turn(yaw, pitch). */
void vrt_hip_camera_init(vrt_hip_camera *cam, const float position[3], const float up[3], const float front[3],
                         float yaw, float pitch, uint64_t w, uint64_t h, float focal_length);
/* camera_t::turn (camera.cpp:7-23) + the view-matrix half of update() (camera.cpp:52); position is read as it is. */
void vrt_hip_camera_turn(vrt_hip_camera *cam, float yaw, float pitch, int constrain);
/* The view-matrix half of camera_t::update() alone (camera.cpp:52), from position / front / up / focal_length as stored. */
void vrt_hip_camera_refresh(vrt_hip_camera *cam);
/* The projection-plane half of camera_t::update() (camera.cpp:54-70): three w*h arrays, caller-allocated. */
void vrt_hip_camera_plane(const vrt_hip_camera *cam, float *xs, float *ys, float *zs);
/* The orbit step of the frame loop (main.cpp:252, 330): position = rotate(I, radians(deg), +Y) * (position, 1).
 * The caller follows it with `angle -= deg; turn(angle, 0)` like main.cpp:254-255, 332-333. */
void vrt_hip_camera_orbit(vrt_hip_camera *cam, float deg);
/* glm::inverse(mat4) in glm's order of operations (what camera.cpp:62 applies to view_matrix). */
void vrt_hip_mat4_inverse(const float m[16], float out[16]);

/* -------- options ---------------------------------------------------------------------- */
/* exp_kind / erf_kind: the reference's template arguments.  cull_eps: Gaussians whose
 * sigma*magnitude*exp(-d^2/(2 sigma^2)) is below cull_eps for every ray of an 8x8 pixel
 * block are skipped for that block (0 disables culling and reproduces the reference's
 * full O(5 N^2) sum).  Defaults: VRT_EXP_VCL, VRT_ERF_AS, cull_eps = 1e-9. */
int vrt_hip_set_options(vrt_hip_ctx *ctx, int exp_kind, int erf_kind, float cull_eps);
/* Mirror contexts (one frame needs one context: a batch of n frames needs n of them with the same scene): copies scene,
 * Exp / Erf / cull options, table settings and shard from src to dst (same device; device-to-device copies).  Rays, tiles
 * and statistics settings are not copied.  vrt_hip_state_generation() changes whenever one of the copied things changes. */
int vrt_hip_copy_state(vrt_hip_ctx *dst, const vrt_hip_ctx *src);
uint64_t vrt_hip_state_generation(const vrt_hip_ctx *ctx);
/* Image size of the rays last set (VRT_HIP_ERR_INVALID before vrt_hip_set_plane / set_camera / set_camera_view). */
int vrt_hip_get_image_size(const vrt_hip_ctx *ctx, uint32_t *w, uint32_t *h);

/* -------- render: replaces render_image / simd_render_image (rt.h:227-404) ------------- */
/* Renders the current scene/tiles/rays.  image_out: w*h u32 (nullable); radiance_out:
 * w*h*4 f32 = broadcast_radiance's vec4 per pixel before clamping (nullable).
 * Returns 0 when finished (the reference returns false = "not aborted"). */
int vrt_hip_render(vrt_hip_ctx *ctx, const float origin[3], int pack_flags, uint32_t *image_out, float *radiance_out);
int vrt_hip_render_device(vrt_hip_ctx *ctx, const float origin[3], int pack_flags, uint32_t *d_image,
                          float *d_radiance, void *hip_stream);

/* One animation frame in one call == the body of the reference's frame loop (main.cpp:259-296): tile_gaussians(view)
 * then render into d_image (raster order) or, when `shard` is non-zero, into this rank's compact shard buffer. */
int vrt_hip_frame_device(vrt_hip_ctx *ctx, float tw, float th, const float view[16], const float origin[3],
                         int pack_flags, uint32_t *d_out, int shard, void *hip_stream);
/* The same for a frame buffer that the CALLER promises still holds this context's previous frame, written by nothing else in
 * between (the reference's `image` is such a buffer: allocated once, written every frame, main.cpp:245): only the cells
 * that were lit in that frame and are empty now are reset to background, not all of them (16 MB of a 2048^2 frame's
 * 18.7 MB of HBM traffic).  The same frame, bit for bit.  One history per context: another buffer, image size, tile grid
 * or background value falls back to the full clear by itself; so does vrt_hip_frame (the library's own buffer gets this
 * treatment without being asked). */
int vrt_hip_frame_retained_device(vrt_hip_ctx *ctx, float tw, float th, const float view[16], const float origin[3],
                                  int pack_flags, uint32_t *d_out, void *hip_stream);

/* The same for callers without device memory of their own (the CLI): the frame goes to the context's own image
 * buffer on the context's stream.  image_out != NULL: copied out and waited for (a frame whose PNG is written);
 * image_out == NULL && !wait: returns as soon as the frame is enqueued -- an animation that only reports its
 * average frame time (main.cpp:310-315) keeps the GPU busy back to back; vrt_hip_sync() waits for the stream. */
int vrt_hip_frame(vrt_hip_ctx *ctx, float tw, float th, const float view[16], const float origin[3], int pack_flags,
                  uint32_t *image_out, int wait);
int vrt_hip_sync(vrt_hip_ctx *ctx);

/* Multi-GPU tile sharding: the context renders only tiles t with shard_of(t) == rank.
 * Owned tiles are written tile-major into a compact buffer of
 * vrt_hip_shard_pixels() u32s: [local tile][tile_h][tile_w].  assemble() scatters the
 * rank-major concatenation of all shards (e.g. an RCCL gather result) into raster order. */
int vrt_hip_set_shard(vrt_hip_ctx *ctx, int rank, int world);
size_t vrt_hip_shard_pixels(const vrt_hip_ctx *ctx);
int vrt_hip_render_shard_device(vrt_hip_ctx *ctx, const float origin[3], int pack_flags, uint32_t *d_shard,
                                void *hip_stream);
int vrt_hip_assemble_shards_device(vrt_hip_ctx *ctx, const uint32_t *d_gathered, uint32_t *d_image, void *hip_stream);
/* Same, for gathers that carry several frames per rank ([rank][frame][shard]): rank r's shard of the frame to
 * assemble starts at d_gathered + r * rank_stride_px (pass d_gathered already offset to the frame). */
int vrt_hip_assemble_shards_strided_device(vrt_hip_ctx *ctx, const uint32_t *d_gathered, size_t rank_stride_px,
                                           uint32_t *d_image, void *hip_stream);

/* Sparse shards: the transport format for small frames.  Most of a frame is background (95 % of `-g 64 -w 2048`), so a
 * rank ships only the 32x32-pixel cells that some Gaussian reaches.  Buffer layout, u32 words (16-byte aligned):
 *   [0] cells stored   [1] capacity C (cells)   [2] cells per tile   [3] 0
 *   [4 .. 4+C)         key of the cell in each slot: tile id * cells per tile + cell in tile (row-major 32-px cells)
 *   [P ..)             32*32 pixels per stored cell, P = 4 + C rounded up to a multiple of 4
 * C is the same on every rank of a job, so a fixed-size prefix [0, P + 1024 * max cells) can travel through a gather.
 * frame_sparse == vrt_hip_frame_device into such a buffer; scatter_sparse builds the raster frame on the assembling
 * rank: background everywhere, then every stored cell of every shard.  The shard pointers must be readable from this
 * context's device: own memory, a gathered copy, or another GPU's memory with peer access enabled (xGMI). */
/* words of a shard buffer for the CURRENT tile grid and shard (call after tile_gaussians / set_tiles / set_shard) */
size_t vrt_hip_sparse_shard_words(const vrt_hip_ctx *ctx);
int vrt_hip_frame_sparse_device(vrt_hip_ctx *ctx, float tw, float th, const float view[16], const float origin[3],
                                int pack_flags, uint32_t *d_sparse, void *hip_stream);
int vrt_hip_scatter_sparse_device(vrt_hip_ctx *ctx, const uint32_t *const *d_shards, int nshards, int pack_flags,
                                  uint32_t *d_image, void *hip_stream);
/* The same for a frame buffer that is reused frame after frame (the `image` of main.cpp:245, written every frame of the
 * loop): by calling this variant the caller promises that d_image still holds what the previous call of this variant on
 * this context left in it.  Then only the cells stored last time and not stored now are reset to background, instead
 * of all w * h pixels (16.8 MB per 2048^2 frame) -- the same frame, bit for bit.  A new buffer, image size, tile grid
 * or background falls back to the full fill by itself. */
int vrt_hip_scatter_sparse_retained_device(vrt_hip_ctx *ctx, const uint32_t *const *d_shards, int nshards, int pack_flags,
                                           uint32_t *d_image, void *hip_stream);
/* Several frames per assembly (the other end of vrt_hip_frame_batch_device): frame f is assembled into d_images[f] -- all
 * different buffers -- from shard s at d_shards[s] + f * frame_stride_words (what a gather of [rank][frame][prefix]
 * delivers), with one launch for all frames; retained != 0 as in the retained variant, per buffer (the context keeps a
 * history for up to 64 frame buffers). */
int vrt_hip_scatter_sparse_batch_device(vrt_hip_ctx *ctx, const uint32_t *const *d_shards, int nshards,
                                        size_t frame_stride_words, int nframes, int pack_flags, uint32_t *const *d_images,
                                        int retained, void *hip_stream);

/* Several frames per launch: the animation loop of main.cpp:257-335 (`--frames N`: the orbit's cameras are known in advance)
 * with n frames handed over at once.  Frame i is rendered by ctxs[i] -- every frame needs a context of its own (lists,
 * queues and per-origin tables are per frame; same scene, options, rays' image size, tile grid and shard in all of them) --
 * with view matrix views[16 i ..] and origin origins[3 i ..] into d_out[i], exactly as vrt_hip_frame_device (out_kind 0),
 * vrt_hip_frame_device(.., shard = 1) (out_kind 1) or vrt_hip_frame_sparse_device (out_kind 2) would, bit for bit; but
 * each of the three kernels of a frame is launched ONCE for all n frames (grid.y = frame).  A sparse frame is tens of
 * microseconds of dependent launches: a batch pays the launch gaps and the kernels' tails once, and a rank that owns an
 * eighth of the tiles still fills its GPU.  Everything is enqueued on hip_stream; errors are reported through ctxs[0].
 * Not batched (VRT_HIP_ERR_INVALID): tiles of more than 64 cells, plane arrays that are no pinhole bundle. */
int vrt_hip_frame_batch_device(vrt_hip_ctx *const *ctxs, int n, float tw, float th, const float *views,
                               const float *origins, int pack_flags, uint32_t *const *d_out, int out_kind,
                               void *hip_stream);

/* -------- several GPUs from one process: replaces the thread pool over tiles (rt.h:355-399) one level up -------------
 * A group holds one context per entry of `devices` (a device may be listed more than once: each entry is a member
 * with its own stream -- how the protocol is tested on a one-GPU box).  Scene, options and rays are set per member
 * through vrt_hip_group_ctx(); vrt_hip_group_frame renders ONE frame tile-sharded over the members: member i renders
 * the tiles of shard (i, n) as a sparse shard on its own device and stream, member 0 waits for all of them (events),
 * reads the shards -- directly over xGMI where peer access exists, through a copy otherwise -- and assembles the
 * raster frame (the copy loop of rt.h:388-399).  Animations that do not need every frame in one place should deal
 * whole frames to the members instead (vrt_hip_frame on vrt_hip_group_ctx(g, k % n)): no exchange at all. */
typedef struct vrt_hip_group vrt_hip_group;
int vrt_hip_group_create(const int *devices, int n, vrt_hip_group **out);
void vrt_hip_group_destroy(vrt_hip_group *g);
int vrt_hip_group_size(const vrt_hip_group *g);
vrt_hip_ctx *vrt_hip_group_ctx(vrt_hip_group *g, int member);
const char *vrt_hip_group_last_error(const vrt_hip_group *g);
/* image_out (host, w*h u32) may be NULL; wait != 0 or image_out != NULL: returns when the frame is complete. */
int vrt_hip_group_frame(vrt_hip_group *g, float tw, float th, const float view[16], const float origin[3], int pack_flags,
                        uint32_t *image_out, int wait);
/* The assembled frame on member 0's device (valid until the next group_frame), for callers that keep it on the GPU. */
const uint32_t *vrt_hip_group_image_device(const vrt_hip_group *g);
int vrt_hip_group_sync(vrt_hip_group *g);
/* n frames of an animation tile-sharded over the members with ONE launch of each kernel per member and ONE assembly launch
 * on member 0 (vrt_hip_frame_batch_device + vrt_hip_scatter_sparse_batch_device): the frame loop of main.cpp:257-335 with
 * the orbit's next n cameras handed over at once.  A member's shard of a sparse frame is a few microseconds of work behind
 * three dependent launches: frame by frame the group is launch-bound and slower than one GPU; in batches the launches are
 * paid once per n frames.  views: 16 n floats, origins: 3 n floats; every frame is rendered with in-kernel rays of its view
 * matrix at the image size of member 0's rays (vrt_hip_set_camera_view / set_plane must have been called once).  The group
 * keeps n - 1 mirror contexts per member (scene, options and table settings copied from the member's own context whenever
 * vrt_hip_state_generation() says they changed).  images_out: NULL, or n host pointers (each NULL or w*h u32); returns when
 * the frames are complete if wait != 0 or any image is wanted.  Each frame equals vrt_hip_group_frame's, bit for bit.
 * n <= 64. */
int vrt_hip_group_frame_batch(vrt_hip_group *g, int n, float tw, float th, const float *views, const float *origins,
                              int pack_flags, uint32_t *const *images_out, int wait);
/* Frame f of the last batch on member 0's device (valid until the next batch). */
const uint32_t *vrt_hip_group_batch_image_device(const vrt_hip_group *g, int f);

/* -------- point queries: replace transmittance / radiance (rt.h:32-54, 146-223) ----------- */
/* T_out[k] = transmittance<Exp,Erf>(o, n, s[k], all Gaussians of the scene), rt.h:32-54. */
int vrt_hip_transmittance(vrt_hip_ctx *ctx, const float o[3], const float n[3], const float *s, size_t ns,
                          float *T_out);
/* T_out[r] = broadcast_transmittance (rt.h:102-127): ray r has its own origin origins[3*r..], unit direction
 * dirs[3*r..] and sample point s[r]; all Gaussians of the scene. */
int vrt_hip_transmittance_rays(vrt_hip_ctx *ctx, size_t nrays, const float *origins, const float *dirs, const float *s,
                               float *T_out);
/* out[4*r..] = radiance / broadcast_radiance (rt.h:146-164, 205-223) for ray r with origin
 * origins[3*r..], unit direction dirs[3*r..], over all Gaussians of the scene (no tiling). */
int vrt_hip_radiance(vrt_hip_ctx *ctx, size_t nrays, const float *origins, const float *dirs, float *out);
/* Numeric cross-checks of rt.cpp:8-27 (transmittance_step uses fast_exp like the reference). */
int vrt_hip_transmittance_step(vrt_hip_ctx *ctx, const float o[3], const float n[3], const float *s, size_t ns,
                               float delta, float *T_out);
int vrt_hip_density(vrt_hip_ctx *ctx, size_t npts, const float *pts, float *D_out);

/* -------- elementwise approximations (approx.cpp) on device; for the accuracy study ------ */
int vrt_hip_eval_erf(vrt_hip_ctx *ctx, int erf_kind, const float *x, size_t n, float *y);
int vrt_hip_eval_exp(vrt_hip_ctx *ctx, int exp_kind, const float *x, size_t n, float *y);

/* Table mode for dense scenes (ON by default; NOT the reference's summation order, but inside its tolerance by a bound the
 * kernel checks per ray): blocks with many overlapping Gaussians evaluate the transmittance exponent of a ray at up to 384
 * equidistant nodes along it and interpolate the 5 n sample points (n * G erf terms per ray instead of 5 n^2).
 * `step` = requested node spacing in units of sqrt2 * sigma of the narrowest Gaussian of the block (default 0.05; 0 = off:
 * the exact kernels only, which reproduce the reference's per-term sums).  Every ray's worst-case radiance change --
 * sum over its samples of |term| * (0.0212 u^2 * K + 0.36 u^4 * S_all), see render_table_body in csrc/vrt_kernels.hip --
 * must stay below the budget (default 2.5e-5), or the block is redone at 0.6 of the spacing and then shaded exactly; so
 * are blocks with more than 2048 survivors.
 * Applies to the Exp / Erf pairs {vcl_exp, expf} x {A&S erf, erff}; other pairs are always exact. */
int vrt_hip_set_table_step(vrt_hip_ctx *ctx, float step);
int vrt_hip_set_table_budget(vrt_hip_ctx *ctx, float budget);

/* Budgeted ray-level cull of the one-wave ("block") kernel.  `cull_eps` (vrt_hip_set_options; replaces nothing in the reference,
 * which sums every Gaussian of a tile: rt.h:205-223) bounds what dropped Gaussians can change by worst-case counting; a ray's own
 * list is additionally cut by the SUM of what it drops: the smallest entries go while sum sigma*mag*exp(-x) <= kappa * 1365 * cull_eps
 * (default kappa 6: the ray's radiance changes by at most 3 * that = 2.46e-5 for albedos <= 1 -- the budget is divided by the scene's
 * largest albedo beyond that; 0 = off; cull_eps = 0 switches every cull off). */
int vrt_hip_set_cull_prune(vrt_hip_ctx *ctx, float kappa);

/* -------- statistics of the last render ----------------------------------------------------- */
typedef struct {
    double kernel_ms;        /* render kernel time of the last vrt_hip_render() (HIP events)       */
    double tiling_ms;        /* last vrt_hip_tile_gaussians() device time                           */
    uint64_t rays;           /* rays shaded                                                         */
    uint64_t blocks;         /* 8x8 pixel blocks (one wavefront each)                                */
    uint64_t list_entries;   /* sum over blocks of candidates kept by the block cull                 */
    uint64_t tile_entries;   /* sum over blocks of the (tile-culled) tile list length they scanned    */
    uint64_t overflow_blocks;/* dense-kernel blocks with more than 1024 survivors (streamed from scratch)      */
    uint64_t lane_entries;   /* sum over rays of the per-ray list length (fast path)                  */
    uint64_t lane_max_entries;/* sum over blocks of the longest per-ray list (the loop trip count)    */
    uint64_t shaded_blocks;  /* blocks that reached the shading loops (the rest were only cleared)    */
    uint64_t dense_blocks;   /* of those, blocks shaded by the 16-waves-per-block kernel              */
    double dense_busy_frac;  /* mean share of that kernel's duration its workgroups had blocks to work on */
    uint64_t table_blocks;   /* of the dense blocks, those shaded through the interpolation table (vrt_hip_set_table_step) */
    uint64_t lane_pairs;     /* sum over rays of (per-ray list length)^2: the (emitter, absorber) pairs the one-wave
                                kernel has to evaluate, 5 erf terms each -- the ALGORITHMIC work of its pair loops    */
    /* dense kernel: visits of an absorber by a chunk of 6 emitters (30 erf terms on every ray of the block when
     * evaluated), split by what the exact saturation tests decided */
    uint64_t dense_visits_full;   /* evaluated term by term                                              */
    uint64_t dense_visits_zero;   /* every term exactly 0 (absorber behind all samples): skipped         */
    uint64_t dense_visits_common; /* every term exactly -2 A_j (absorber in front of all samples): one fma */
    /* table kernel */
    uint64_t table_nodes;     /* sum over its blocks of the node count G                                     */
    uint64_t table_retries;   /* blocks that met the error budget only at the reduced spacing                */
    uint64_t table_skips;     /* (absorber, wave) visits settled by saturation (one add instead of NT terms) */
    uint64_t table_declined;  /* blocks handed to the exact kernel                                           */
    uint64_t table_coarser;   /* blocks done at a coarser spacing than requested (the bound left room)       */
    uint64_t table_empty;     /* of table_blocks, blocks no Gaussian reaches (the rim of a dense cell)       */
    uint64_t table_phase_ticks[8]; /* 10-ns ticks its workgroups spent, summed over blocks: set-up, cull, range, spacing
                                      estimate, kink weights, table, emission, reduction                      */
    uint64_t dense_launch_skips; /* frames of this context whose dense launch was left out (a frame of exactly the same state had
                                    reported that there is nothing for it); counted since the context was created */
} vrt_hip_stats;
int vrt_hip_get_stats(vrt_hip_ctx *ctx, vrt_hip_stats *out);
/* Enables per-block statistics collection (small atomics; off by default). */
int vrt_hip_enable_stats(vrt_hip_ctx *ctx, int on);
/* Kernel timing for roofline reports: while enabled, every render launch is bracketed with HIP events ON THE
 * STREAM IT RUNS ON (ring of the last 512 launches).  get() waits for them and returns the mean duration of
 * the one-wave-per-block render kernel, of the 16-waves-per-block (dense) kernel and of the list kernels.
 * on = 1: four events per frame (all three durations); on = 2: two events, around the one-wave render kernel only
 * (the other two durations read 0) -- an event costs the stream 2-3 us, which matters for 0.06 ms frames;
 * on = 3: like 2, on every 8th frame only. */
int vrt_hip_enable_kernel_timing(vrt_hip_ctx *ctx, int on);
int vrt_hip_get_kernel_timing(vrt_hip_ctx *ctx, double *render_ms, double *dense_ms, double *lists_ms, uint64_t *launches);

#ifdef __cplusplus
}
#endif
#endif
