// vrt_hip_group.cpp -- several GPUs driven from ONE host process through the C ABI (include/vrt_hip.h, "several GPUs"):
// what the reference does with a thread pool over image tiles inside one process (rt.h:355-399), one level up.
//
// One frame, tile-sharded:  member i renders the tiles of shard (i, n) into a SPARSE shard (only the 32x32-px cells a
// Gaussian reaches: 0.8 MB instead of 16 MB for `-g 64 -w 2048`) on its own device and stream and records an event;
// member 0's stream waits for the events (cross-device waits, no host round trip), then one kernel on member 0 fills
// the background and copies every stored cell to its place, reading the other members' shards where they lie:
// over xGMI when peer access exists (every pair of GPUs of an MI355X node), after a device-to-device copy of the
// shard's used prefix otherwise.  No collective library is involved: one process owns all devices, so the "gather" is
// peer loads of the assembling kernel -- the same wires RCCL's own send/recv kernels use (bench.py, one process per
// GPU, goes through RCCL instead).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/vrt_hip.h"

struct vrt_hip_group {
    struct Member {
        int device = 0;
        vrt_hip_ctx *ctx = nullptr;
        hipStream_t stream = nullptr;
        hipEvent_t done = nullptr;       // this member's shard of the current frame is complete
        uint32_t *shard = nullptr;       // sparse shard buffer on the member's device
        size_t shard_words = 0;
        uint32_t *staged = nullptr;      // copy on member 0's device when member 0 cannot read `shard` directly
        bool peer_ok = true;
        // frame batches: lanes[f] renders frame f of a batch (lanes[0] = ctx; the others mirror its scene and options)
        std::vector<vrt_hip_ctx *> lanes;
        uint64_t mirrored_gen = 0;       // vrt_hip_state_generation(ctx) the mirrors were copied at
        uint32_t *bshard = nullptr;      // the batch's sparse shards, frame f at bshard + f * bwords
        size_t bwords = 0;               // words per frame
        int bframes = 0;                 // frames the buffer holds
        uint32_t *bstaged = nullptr;
    };
    std::vector<Member> m;
    uint32_t *image = nullptr;           // assembled frame, member 0's device
    size_t image_px = 0;
    std::vector<uint32_t *> bimages;     // assembled frames of the last batch, member 0's device
    size_t bimage_px = 0;
    hipEvent_t assembled = nullptr;      // the previous frame has been assembled: shard buffers may be overwritten
    bool have_assembled = false;
    std::string err;
};

namespace {
std::string g_group_create_error;

int gfail(vrt_hip_group *g, int code, const std::string &msg)
{
    if (g) g->err = msg; else g_group_create_error = msg;
    return code;
}
#define GCHK(g, call)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (call);                                                                         \
        if (_e != hipSuccess) return gfail((g), VRT_HIP_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(_e)); \
    } while (0)
} // namespace

extern "C" {

int vrt_hip_group_create(const int *devices, int n, vrt_hip_group **out)
{
    if (!out) return gfail(nullptr, VRT_HIP_ERR_INVALID, "group_create: out is NULL");
    *out = nullptr;
    if (!devices || n < 1 || n > 64) return gfail(nullptr, VRT_HIP_ERR_INVALID, "group_create: 1..64 members");
    vrt_hip_group *g = new (std::nothrow) vrt_hip_group();
    if (!g) return gfail(nullptr, VRT_HIP_ERR_NOMEM, "group_create: out of host memory");
    g->m.resize(n);
    for (int i = 0; i < n; ++i) {
        auto &mb = g->m[i];
        mb.device = devices[i];
        int rc = vrt_hip_create(devices[i], &mb.ctx);
        if (rc != VRT_HIP_OK) {
            g_group_create_error = std::string("group_create: member ") + std::to_string(i) + ": " + vrt_hip_last_error(nullptr);
            vrt_hip_group_destroy(g);
            return rc;
        }
        if (hipSetDevice(mb.device) != hipSuccess || hipStreamCreateWithFlags(&mb.stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&mb.done, hipEventDisableTiming) != hipSuccess) {
            vrt_hip_group_destroy(g);
            return gfail(nullptr, VRT_HIP_ERR_HIP, "group_create: stream / event creation failed");
        }
        vrt_hip_set_shard(mb.ctx, i, n);
    }
    // member 0 reads the others' shards: enable peer access where the hardware offers it
    (void)hipSetDevice(g->m[0].device);
    for (int i = 1; i < n; ++i) {
        auto &mb = g->m[i];
        if (mb.device == g->m[0].device) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, g->m[0].device, mb.device) == hipSuccess && can) {
            const hipError_t e = hipDeviceEnablePeerAccess(mb.device, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) mb.peer_ok = false;
            (void)hipGetLastError();
        } else {
            mb.peer_ok = false;
        }
    }
    if (hipEventCreateWithFlags(&g->assembled, hipEventDisableTiming) != hipSuccess) {
        vrt_hip_group_destroy(g);
        return gfail(nullptr, VRT_HIP_ERR_HIP, "group_create: event creation failed");
    }
    *out = g;
    return VRT_HIP_OK;
}

void vrt_hip_group_destroy(vrt_hip_group *g)
{
    if (!g) return;
    for (auto &mb : g->m) {
        (void)hipSetDevice(mb.device);
        if (mb.stream) (void)hipStreamSynchronize(mb.stream);
    }
    if (!g->m.empty()) {
        (void)hipSetDevice(g->m[0].device);
        if (g->image) (void)hipFree(g->image);
        for (auto &mb : g->m) if (mb.staged) (void)hipFree(mb.staged);
        for (auto &mb : g->m) if (mb.bstaged) (void)hipFree(mb.bstaged);
        for (auto p : g->bimages) if (p) (void)hipFree(p);
        if (g->assembled) (void)hipEventDestroy(g->assembled);
    }
    for (auto &mb : g->m) {
        (void)hipSetDevice(mb.device);
        if (mb.shard) (void)hipFree(mb.shard);
        if (mb.bshard) (void)hipFree(mb.bshard);
        for (size_t k = 1; k < mb.lanes.size(); ++k) if (mb.lanes[k]) vrt_hip_destroy(mb.lanes[k]);
        if (mb.done) (void)hipEventDestroy(mb.done);
        if (mb.stream) (void)hipStreamDestroy(mb.stream);
        if (mb.ctx) vrt_hip_destroy(mb.ctx);
    }
    delete g;
}

int vrt_hip_group_size(const vrt_hip_group *g) { return g ? (int)g->m.size() : 0; }
vrt_hip_ctx *vrt_hip_group_ctx(vrt_hip_group *g, int member)
{
    return (g && member >= 0 && member < (int)g->m.size()) ? g->m[member].ctx : nullptr;
}
const char *vrt_hip_group_last_error(const vrt_hip_group *g) { return g ? g->err.c_str() : g_group_create_error.c_str(); }
const uint32_t *vrt_hip_group_image_device(const vrt_hip_group *g) { return g ? g->image : nullptr; }

int vrt_hip_group_frame(vrt_hip_group *g, float tw, float th, const float view[16], const float origin[3], int pack_flags,
                        uint32_t *image_out, int wait)
{
    if (!g || !view || !origin) return VRT_HIP_ERR_INVALID;
    const int n = (int)g->m.size();
    // ---- every member renders its shard (the loop only enqueues: the members run concurrently) ----
    for (int i = 0; i < n; ++i) {
        auto &mb = g->m[i];
        GCHK(g, hipSetDevice(mb.device));
        // the tile grid of this frame decides the shard capacity: set it before sizing the buffer
        int rc0 = vrt_hip_tile_gaussians_device(mb.ctx, tw, th, view, mb.stream);
        if (rc0 != VRT_HIP_OK) return gfail(g, rc0, std::string("group_frame: member ") + std::to_string(i) + ": " + vrt_hip_last_error(mb.ctx));
        const size_t words = vrt_hip_sparse_shard_words(mb.ctx);
        if (!words) return gfail(g, VRT_HIP_ERR_INVALID, "group_frame: set the rays of every member first (vrt_hip_set_camera_view / set_plane)");
        if (words > mb.shard_words) {
            GCHK(g, hipStreamSynchronize(mb.stream));
            // the previous frame's assembly on member 0's stream may still be reading this shard (peer access) or its
            // staged copy: wait for it on the host before either is freed (round-2 advisor finding)
            if (g->have_assembled) GCHK(g, hipEventSynchronize(g->assembled));
            if (mb.shard) (void)hipFree(mb.shard);
            mb.shard = nullptr; mb.shard_words = 0;
            GCHK(g, hipMalloc((void **)&mb.shard, words * sizeof(uint32_t)));
            mb.shard_words = words;
            if (mb.staged) { (void)hipSetDevice(g->m[0].device); (void)hipFree(mb.staged); mb.staged = nullptr; (void)hipSetDevice(mb.device); }
        }
        // the previous frame's assembly (on member 0) still reads this buffer
        if (g->have_assembled) GCHK(g, hipStreamWaitEvent(mb.stream, g->assembled, 0));
        const int rc = vrt_hip_frame_sparse_device(mb.ctx, tw, th, view, origin, pack_flags, mb.shard, mb.stream);
        if (rc != VRT_HIP_OK) return gfail(g, rc, std::string("group_frame: member ") + std::to_string(i) + ": " + vrt_hip_last_error(mb.ctx));
        GCHK(g, hipEventRecord(mb.done, mb.stream));
    }
    // ---- member 0 assembles ----
    auto &root = g->m[0];
    GCHK(g, hipSetDevice(root.device));
    std::vector<const uint32_t *> ptrs(n);
    for (int i = 0; i < n; ++i) {
        auto &mb = g->m[i];
        if (i) GCHK(g, hipStreamWaitEvent(root.stream, mb.done, 0));
        if (mb.device == root.device || mb.peer_ok) {
            ptrs[i] = mb.shard;
        } else {
            // no peer access: the used prefix of the shard is copied across.  Its length is data (header word 0), so
            // this path reads the header on the host first -- one round trip per member and frame, only without xGMI
            uint32_t hdr[4] = { 0, 0, 0, 0 };
            GCHK(g, hipSetDevice(mb.device));
            GCHK(g, hipEventSynchronize(mb.done));
            GCHK(g, hipMemcpy(hdr, mb.shard, sizeof hdr, hipMemcpyDeviceToHost));
            GCHK(g, hipSetDevice(root.device));
            if (!mb.staged) GCHK(g, hipMalloc((void **)&mb.staged, mb.shard_words * sizeof(uint32_t)));
            const size_t pix_off = (4 + (size_t)hdr[1] + 3) / 4 * 4;
            const size_t used = std::min(mb.shard_words, pix_off + (size_t)hdr[0] * 1024);
            GCHK(g, hipMemcpyPeerAsync(mb.staged, root.device, mb.shard, mb.device, used * sizeof(uint32_t), root.stream));
            ptrs[i] = mb.staged;
        }
    }
    const size_t npix = vrt_hip_image_pixels(root.ctx); // w * h of member 0's rays
    if (!npix) return gfail(g, VRT_HIP_ERR_INVALID, "group_frame: member 0 has no image size");
    if (npix > g->image_px) {
        GCHK(g, hipStreamSynchronize(root.stream));
        if (g->image) (void)hipFree(g->image);
        g->image = nullptr; g->image_px = 0;
        GCHK(g, hipMalloc((void **)&g->image, npix * sizeof(uint32_t)));
        g->image_px = npix;
    }
    // g->image is the group's own buffer and only this call writes it: the retained variant resets just the cells that went dark
    const int rc = vrt_hip_scatter_sparse_retained_device(root.ctx, ptrs.data(), n, pack_flags, g->image, root.stream);
    if (rc != VRT_HIP_OK) return gfail(g, rc, std::string("group_frame: assemble: ") + vrt_hip_last_error(root.ctx));
    GCHK(g, hipEventRecord(g->assembled, root.stream));
    g->have_assembled = true;
    if (image_out) GCHK(g, hipMemcpyAsync(image_out, g->image, npix * sizeof(uint32_t), hipMemcpyDeviceToHost, root.stream));
    if (image_out || wait) GCHK(g, hipStreamSynchronize(root.stream));
    return VRT_HIP_OK;
}

int vrt_hip_group_frame_batch(vrt_hip_group *g, int nf, float tw, float th, const float *views, const float *origins,
                              int pack_flags, uint32_t *const *images_out, int wait)
{
    if (!g || !views || !origins) return VRT_HIP_ERR_INVALID;
    if (nf < 1 || nf > 64) return gfail(g, VRT_HIP_ERR_INVALID, "group_frame_batch: 1..64 frames per batch");
    const int n = (int)g->m.size();
    uint32_t w = 0, h = 0;
    if (vrt_hip_get_image_size(g->m[0].ctx, &w, &h) != VRT_HIP_OK)
        return gfail(g, VRT_HIP_ERR_INVALID, "group_frame_batch: set the rays of member 0 once first (vrt_hip_set_camera_view / set_plane): the image size");
    // ---- every member: mirrors up to date, rays of every frame, shard buffer, one batch of launches ----
    for (int i = 0; i < n; ++i) {
        auto &mb = g->m[i];
        GCHK(g, hipSetDevice(mb.device));
        if (mb.lanes.empty()) mb.lanes.push_back(mb.ctx);
        const uint64_t gen = vrt_hip_state_generation(mb.ctx);
        while ((int)mb.lanes.size() < nf) {
            vrt_hip_ctx *c = nullptr;
            const int rc = vrt_hip_create(mb.device, &c);
            if (rc != VRT_HIP_OK) return gfail(g, rc, std::string("group_frame_batch: ") + vrt_hip_last_error(nullptr));
            mb.lanes.push_back(c);
            mb.mirrored_gen = 0; // the new lane has nothing yet
        }
        if (mb.mirrored_gen != gen) {
            for (size_t k = 1; k < mb.lanes.size(); ++k) {
                const int rc = vrt_hip_copy_state(mb.lanes[k], mb.ctx);
                if (rc != VRT_HIP_OK) return gfail(g, rc, std::string("group_frame_batch: member ") + std::to_string(i) + ": " + vrt_hip_last_error(mb.lanes[k]));
            }
            mb.mirrored_gen = gen;
        }
        for (int f = 0; f < nf; ++f) {
            int rc = vrt_hip_set_camera_view(mb.lanes[f], w, h, views + 16 * (size_t)f);
            // the tile grid decides the shard capacity: set it before sizing the buffer
            if (rc == VRT_HIP_OK && f == 0) rc = vrt_hip_tile_gaussians_device(mb.lanes[0], tw, th, views, mb.stream);
            if (rc != VRT_HIP_OK) return gfail(g, rc, std::string("group_frame_batch: member ") + std::to_string(i) + ": " + vrt_hip_last_error(mb.lanes[f]));
        }
        const size_t words = vrt_hip_sparse_shard_words(mb.lanes[0]);
        if (!words) return gfail(g, VRT_HIP_ERR_INVALID, "group_frame_batch: no shard geometry");
        if (words != mb.bwords || nf > mb.bframes) {
            GCHK(g, hipStreamSynchronize(mb.stream));
            if (g->have_assembled) GCHK(g, hipEventSynchronize(g->assembled)); // member 0 may still be reading the old buffer
            if (mb.bshard) (void)hipFree(mb.bshard);
            mb.bshard = nullptr; mb.bwords = 0; mb.bframes = 0;
            GCHK(g, hipMalloc((void **)&mb.bshard, words * (size_t)nf * sizeof(uint32_t)));
            mb.bwords = words; mb.bframes = nf;
            if (mb.bstaged) { (void)hipSetDevice(g->m[0].device); (void)hipFree(mb.bstaged); mb.bstaged = nullptr; (void)hipSetDevice(mb.device); }
        }
        if (g->have_assembled) GCHK(g, hipStreamWaitEvent(mb.stream, g->assembled, 0)); // the previous assembly still reads the shards
        std::vector<uint32_t *> outs(nf);
        for (int f = 0; f < nf; ++f) outs[f] = mb.bshard + (size_t)f * mb.bwords;
        const int rc = vrt_hip_frame_batch_device(mb.lanes.data(), nf, tw, th, views, origins, pack_flags, outs.data(), 2 /* sparse shards */, mb.stream);
        if (rc != VRT_HIP_OK) return gfail(g, rc, std::string("group_frame_batch: member ") + std::to_string(i) + ": " + vrt_hip_last_error(mb.lanes[0]));
        GCHK(g, hipEventRecord(mb.done, mb.stream));
    }
    // ---- member 0 assembles all frames with one launch ----
    auto &root = g->m[0];
    GCHK(g, hipSetDevice(root.device));
    std::vector<const uint32_t *> ptrs(n);
    for (int i = 0; i < n; ++i) {
        auto &mb = g->m[i];
        if (i) GCHK(g, hipStreamWaitEvent(root.stream, mb.done, 0));
        if (mb.device == root.device || mb.peer_ok) {
            ptrs[i] = mb.bshard;
        } else { // no peer access: the whole batch buffer is copied across
            if (!mb.bstaged) GCHK(g, hipMalloc((void **)&mb.bstaged, mb.bwords * (size_t)mb.bframes * sizeof(uint32_t)));
            GCHK(g, hipMemcpyPeerAsync(mb.bstaged, root.device, mb.bshard, mb.device, mb.bwords * (size_t)nf * sizeof(uint32_t), root.stream));
            ptrs[i] = mb.bstaged;
        }
    }
    const size_t npix = (size_t)w * h;
    if (npix != g->bimage_px || (int)g->bimages.size() < nf) {
        GCHK(g, hipStreamSynchronize(root.stream));
        if (npix != g->bimage_px) { for (auto p : g->bimages) if (p) (void)hipFree(p); g->bimages.clear(); }
        while ((int)g->bimages.size() < nf) {
            uint32_t *p = nullptr;
            GCHK(g, hipMalloc((void **)&p, npix * sizeof(uint32_t)));
            g->bimages.push_back(p);
        }
        g->bimage_px = npix;
    }
    // the group's own buffers, written by nothing but this call: retained assembly (only cells that went dark are reset)
    const int rc = vrt_hip_scatter_sparse_batch_device(root.ctx, ptrs.data(), n, g->m[0].bwords, nf, pack_flags, g->bimages.data(), 1, root.stream);
    if (rc != VRT_HIP_OK) return gfail(g, rc, std::string("group_frame_batch: assemble: ") + vrt_hip_last_error(root.ctx));
    GCHK(g, hipEventRecord(g->assembled, root.stream));
    g->have_assembled = true;
    bool any = false;
    if (images_out)
        for (int f = 0; f < nf; ++f)
            if (images_out[f]) { GCHK(g, hipMemcpyAsync(images_out[f], g->bimages[f], npix * sizeof(uint32_t), hipMemcpyDeviceToHost, root.stream)); any = true; }
    if (any || wait) GCHK(g, hipStreamSynchronize(root.stream));
    return VRT_HIP_OK;
}

const uint32_t *vrt_hip_group_batch_image_device(const vrt_hip_group *g, int f)
{
    return (g && f >= 0 && f < (int)g->bimages.size()) ? g->bimages[f] : nullptr;
}

int vrt_hip_group_sync(vrt_hip_group *g)
{
    if (!g) return VRT_HIP_ERR_INVALID;
    for (auto &mb : g->m) {
        GCHK(g, hipSetDevice(mb.device));
        GCHK(g, hipStreamSynchronize(mb.stream));
        const int rc = vrt_hip_sync(mb.ctx);
        if (rc != VRT_HIP_OK) return gfail(g, rc, vrt_hip_last_error(mb.ctx));
    }
    return VRT_HIP_OK;
}

} // extern "C"
