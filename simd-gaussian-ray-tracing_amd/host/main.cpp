// volumetric-ray-tracer -- MI355X drop-in for the reference CLI's quiet path
// (src/volumetric-ray-tracer/main.cpp, run with -q: no viewer).  Same flags, same scene producers,
// same per-frame loop (tile -> render -> PNG -> timing line -> orbit step), same output lines
// ("TIME: <ms> ms" / "AVG. TIME: <ms> ms (<n> frames)", main.cpp:308-315) so that runtimes.sh /
// gen-gif.sh keep working.  Rendering goes through the C ABI of libvrt_hip.so; the interactive
// Vulkan/ImGui viewer of the reference is out of scope, so -q is implied.
#include <getopt.h>
#include <time.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/vrt/vrt.hpp"
#include "png_writer.hpp"

// Own wording; the flag set (names, short forms, defaults) is the reference's (main.cpp:28-52, 75-91) because scripts
// written for it (runtimes.sh, gen-gif.sh) must keep working.
static const char *const HELP_MSG =
    "volumetric-ray-tracer (MI355X build) -- volumetric Gaussian renderer, quiet path of the reference CLI\n"
    "\n"
    "scene\n"
    "  -g, --grid [N]            N x N Gaussians on the plane z = 1 (N defaults to 4); the scene when no file is given\n"
    "  -f, --file PATH           one Gaussian per vertex of a Wavefront OBJ file; wins over --grid\n"
    "image\n"
    "  -w, --width PX            image width  (256); also the height unless --height is given\n"
    "  -h, --height PX           image height (256); also the width unless --width is given\n"
    "  -o, --output PATH         write the frame as PNG; with --frames N > 1: <stem>_<k>.<ext>, k = 1..N\n"
    "      --tiles N             N x N image tiles, each rendered with its own Gaussian subset (16)\n"
    "  -m, --mode 1..8           rendering mode of the reference: 1-4 untiled, 5-8 tiled; picks the pixel packing and\n"
    "                            the exp/erf pair of that mode (8 = tiled, SIMD-over-pixels semantics; the default)\n"
    "camera and animation\n"
    "  -c, --camera-offset Z     camera starts at (0, 0, Z) looking along +z (-4)\n"
    "      --focal-length F      distance from the camera to the projection plane (1)\n"
    "  -i, --initial-rotation D  orbit the camera D degrees about the y axis before the first frame (0)\n"
    "      --frames N            render N frames (1); prints the average frame time instead of one time\n"
    "  -r, --rotation D          total orbit angle over all frames, i.e. D/N degrees per frame (360)\n"
    "accepted for compatibility, no effect here\n"
    "  -t, --with-threads N      CPU worker threads of the reference; the GPU grid takes their place\n"
    "  -q, --quiet               the reference's 'no viewer' switch; this build never opens one\n"
    "this build only\n"
    "      --gpus N              use the first N GPUs (or the list in VRT_HIP_DEVICES): a frame that is written out or\n"
    "                            timed on its own is split by image tiles and assembled on the first GPU; the frames of\n"
    "                            an animation that only reports its average are dealt whole to the GPUs in turn (1)\n"
    "      --plane-arrays        upload per-pixel projection-plane points like the reference instead of generating\n"
    "                            rays in the kernel (same image, 12 more bytes per ray)\n"
    "      --cull-eps E          contributions below E are skipped (1e-9); 0 evaluates the reference's full sum\n"
    "      --table-step S        dense blocks tabulate the transmittance along each ray at nodes S*sqrt(2)*sigma apart\n"
    "                            (coarser where the kernel's own error bound leaves room) and interpolate; every ray stays\n"
    "                            within --table-budget of the exact sum or its block is shaded exactly (0.05; 0 = exact only)\n"
    "      --table-budget B      largest worst-case change of a ray's radiance the table may cause (2.5e-5)\n"
    "      --cull-prune K        budget factor of the ray-level prune: a ray drops its smallest entries while their sum stays\n"
    "                            below K * 1365 * cull-eps (6; 0 = off)\n"
    "      --help                this text\n";

struct cmd_args_t { // main.cpp:54-184
    u64 w = (u64)-1, h = (u64)-1;
    u64 grid_dim = 4;
    char *outfile = nullptr, *infile = nullptr;
    bool use_grid = false;
    u64 thread_count = 1, nr_frames = 1, tiles = 16, mode = 8;
    f32 rot = 360.f, inital_rot = 0.f, camera_offset = -4.f, focal_length = 1.f, cull_eps = 1e-9f, table_step = 0.05f, table_budget = 2.5e-5f, cull_prune = 6.f;
    bool plane_arrays = false;
    u64 gpus = 1;
    cmd_args_t(int argc, char **argv)
    {
        static struct option opts[] = {
            { "grid", optional_argument, NULL, 'g' }, { "file", required_argument, NULL, 'f' },
            { "output", required_argument, NULL, 'o' }, { "width", required_argument, NULL, 'w' },
            { "height", required_argument, NULL, 'h' }, { "with-threads", required_argument, NULL, 't' },
            { "quiet", no_argument, NULL, 'q' }, { "tiles", required_argument, NULL, 'l' },
            { "mode", required_argument, NULL, 'm' }, { "frames", required_argument, NULL, 's' },
            { "rotation", required_argument, NULL, 'r' }, { "initial-rotation", required_argument, NULL, 'i' },
            { "camera-offset", required_argument, NULL, 'c' }, { "focal-length", required_argument, NULL, 0xfe },
            { "help", no_argument, NULL, 0xff }, { "plane-arrays", no_argument, NULL, 0xfd },
            { "cull-eps", required_argument, NULL, 0xfc }, { "table-step", required_argument, NULL, 0xfb }, { "table-budget", required_argument, NULL, 0xf9 }, { "cull-prune", required_argument, NULL, 0xf8 },
            { "gpus", required_argument, NULL, 0xfa },
            { NULL, 0, NULL, 0 }
        };
        int lidx;
        for (;;) {
            const int c = getopt_long(argc, argv, "r:m:qw:o:f:g:h:t:c:i:", opts, &lidx);
            if (c == -1) break;
            switch (c) {
            case 'g': use_grid = true; if (optarg) grid_dim = strtoul(optarg, NULL, 10); break;
            case 'f': infile = optarg; break;
            case 'o': outfile = optarg; break;
            case 'w': w = strtoul(optarg, NULL, 10); if (h == (u64)-1) h = w; break;
            case 'h': h = strtoul(optarg, NULL, 10); if (w == (u64)-1) w = h; break;
            case 't': thread_count = strtoul(optarg, NULL, 10); break;
            case 'q': break;
            case 'l': tiles = strtoul(optarg, NULL, 10); break;
            case 's': nr_frames = strtoul(optarg, NULL, 10); break;
            case 'r': rot = strtof(optarg, NULL); break;
            case 'i': inital_rot = strtof(optarg, NULL); break;
            case 'c': camera_offset = strtof(optarg, NULL); break;
            case 0xff: fputs(HELP_MSG, stdout); exit(EXIT_SUCCESS);
            case 0xfe: focal_length = strtof(optarg, NULL); break;
            case 0xfd: plane_arrays = true; break;
            case 0xfc: cull_eps = strtof(optarg, NULL); break;
            case 0xfb: table_step = strtof(optarg, NULL); break;
            case 0xf9: table_budget = strtof(optarg, NULL); break;
            case 0xf8: cull_prune = strtof(optarg, NULL); break;
            case 0xfa: gpus = strtoul(optarg, NULL, 10); break;
            case 'm': mode = strtoul(optarg, NULL, 10); if (mode < 1 || mode > 8) mode = 8; break;
            default: break;
            }
        }
        if (w == (u64)-1) w = 256;
        if (h == (u64)-1) h = 256;
        if (use_grid && infile != nullptr) use_grid = false; // main.cpp:182: a file overrides the grid
        if (tiles == 0) tiles = 1;
        if (nr_frames == 0) nr_frames = 1;
        if (gpus == 0) gpus = 1;
    }
};

static double now_ms()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

// --gpus N: the frame loop of main.cpp:257-335 over a vrt_hip_group (csrc/vrt_hip_group.cpp).  Same output lines.
static int run_on_group(const cmd_args_t &cmd, const std::vector<vrt::gaussian_t> &gaussians, int pack, int ek, int rk)
{
    std::vector<int> devices;
    if (const char *list = getenv("VRT_HIP_DEVICES")) { // e.g. "0,0": two members on one GPU (how the path is tested on one GPU)
        for (const char *p = list; *p;) {
            char *end = nullptr;
            const long d = strtol(p, &end, 10);
            if (end == p) break;
            devices.push_back((int)d);
            p = (*end == ',') ? end + 1 : end;
        }
    }
    if (devices.empty()) for (u64 i = 0; i < cmd.gpus; ++i) devices.push_back((int)i);
    devices.resize(cmd.gpus, devices.back());
    // whole frames dealt to the members (no exchange) when no frame has to be in one place; four members per GPU then,
    // like the four contexts of the single-GPU loop
    // (VRT_CLI_GROUP_SHARD=1: tile-shard every frame even when none is written out -- to time the sharded path)
    const bool deal_frames = cmd.outfile == nullptr && cmd.nr_frames > 1 && getenv("VRT_CLI_GROUP_SHARD") == nullptr;
    if (deal_frames) { const size_t n = devices.size(); for (size_t k = 0; k < 3 * n; ++k) devices.push_back(devices[k % n]); }
    vrt_hip_group *grp = nullptr;
    if (vrt_hip_group_create(devices.data(), (int)devices.size(), &grp) != VRT_HIP_OK) {
        fprintf(stderr, "[ ERROR ]\t%s\n", vrt_hip_group_last_error(nullptr));
        return EXIT_FAILURE;
    }
    const int n = vrt_hip_group_size(grp);
    auto chk = [&](int rc, const char *what, vrt_hip_ctx *ctx) {
        if (rc != VRT_HIP_OK) {
            fprintf(stderr, "[ ERROR ]\t%s: %s\n", what, ctx ? vrt_hip_last_error(ctx) : vrt_hip_group_last_error(grp));
            exit(EXIT_FAILURE);
        }
    };
    for (int i = 0; i < n; ++i) {
        vrt_hip_ctx *ctx = vrt_hip_group_ctx(grp, i);
        chk(vrt_hip_set_gaussians_aos(ctx, gaussians.size(), gaussians.data()), "set_gaussians", ctx);
        chk(vrt_hip_set_options(ctx, ek, rk, cmd.cull_eps), "set_options", ctx);
        chk(vrt_hip_set_table_step(ctx, cmd.table_step), "set_table_step", ctx);
        chk(vrt_hip_set_table_budget(ctx, cmd.table_budget), "set_table_budget", ctx);
        chk(vrt_hip_set_cull_prune(ctx, cmd.cull_prune), "set_cull_prune", ctx);
        if (deal_frames) chk(vrt_hip_set_shard(ctx, 0, 1), "set_shard", ctx); // every member renders whole frames
    }
    const u64 width = cmd.w, height = cmd.h;
    std::vector<u32> image(width * height);
    vrt::camera_t cam({ 0.f, 0.f, cmd.camera_offset }, { 0.f, 1.f, 0.f }, { 0.f, 0.f, 1.f }, -90.f, 0.f,
                      cmd.plane_arrays ? width : 1, cmd.plane_arrays ? height : 1, cmd.focal_length);
    f32 angle = -90.f;
    cam.orbit(cmd.inital_rot);
    angle -= cmd.inital_rot;
    cam.turn(angle, 0.f);
    const f32 tw = 2.f / cmd.tiles, th = tw;
    if (getenv("VRT_CLI_NO_WARMUP") == nullptr) { // one untimed frame (see main): first-launch costs are not rendering time
        const f32 origin[3] = { cam.position[0], cam.position[1], cam.position[2] };
        for (int i = 0; i < n; ++i) {
            vrt_hip_ctx *ctx = vrt_hip_group_ctx(grp, i);
            if (cmd.plane_arrays)
                chk(vrt_hip_set_plane(ctx, (u32)width, (u32)height, cam.projection_plane.xs.data(), cam.projection_plane.ys.data(),
                                      cam.projection_plane.zs.data()), "set_plane", ctx);
            else
                chk(vrt_hip_set_camera_view(ctx, (u32)width, (u32)height, cam.view_matrix.data()), "set_camera_view", ctx);
            if (deal_frames) chk(vrt_hip_frame(ctx, tw, th, cam.view_matrix.data(), origin, pack, nullptr, 1), "frame", ctx);
        }
        if (!deal_frames) chk(vrt_hip_group_frame(grp, tw, th, cam.view_matrix.data(), origin, pack, nullptr, 1), "group_frame", nullptr);
    }
    f32 total_time = 0.f;
    double t_first = 0.0;
    if (!deal_frames && cmd.nr_frames > 1 && !cmd.plane_arrays) {
        // An animation whose frames are needed in one place (-o): the orbit's next cameras are known, so the frames go to the
        // group in batches -- one launch of each kernel per member and ONE assembly launch per batch instead of per frame
        // (a member's shard of a sparse frame is microseconds of work behind three dependent launches).
        const u64 per_batch = getenv("VRT_CLI_GROUP_BATCH") ? std::max(1, std::min(64, atoi(getenv("VRT_CLI_GROUP_BATCH")))) : 16;
        std::vector<std::vector<u32>> images(cmd.outfile ? per_batch : 0, std::vector<u32>(width * height));
        for (int i = 0; i < n; ++i) chk(vrt_hip_set_camera_view(vrt_hip_group_ctx(grp, i), (u32)width, (u32)height, cam.view_matrix.data()), "set_camera_view", vrt_hip_group_ctx(grp, i));
        if (getenv("VRT_CLI_NO_WARMUP") == nullptr) {
            // one untimed batch: the group creates its mirror contexts (per_batch - 1 per member) and their buffers on first use,
            // which is set-up like the first launch above, not rendering time
            const u64 nf = std::min<u64>(per_batch, cmd.nr_frames);
            std::vector<f32> views(16 * nf), origins(3 * nf);
            for (u64 f = 0; f < nf; ++f) {
                memcpy(&views[16 * f], cam.view_matrix.data(), 16 * sizeof(f32));
                for (int k = 0; k < 3; ++k) origins[3 * f + k] = cam.position[k];
            }
            chk(vrt_hip_group_frame_batch(grp, (int)nf, tw, th, views.data(), origins.data(), pack, nullptr, 1), "group_frame_batch", nullptr);
        }
        for (u64 done = 0; done < cmd.nr_frames;) {
            const u64 nf = std::min<u64>(per_batch, cmd.nr_frames - done);
            std::vector<f32> views(16 * nf), origins(3 * nf);
            std::vector<u32 *> outs(nf, nullptr);
            for (u64 f = 0; f < nf; ++f) { // main.cpp:329-334 between the frames
                memcpy(&views[16 * f], cam.view_matrix.data(), 16 * sizeof(f32));
                for (int k = 0; k < 3; ++k) origins[3 * f + k] = cam.position[k];
                if (cmd.outfile) outs[f] = images[f].data();
                const f32 angle_change = cmd.rot / cmd.nr_frames;
                cam.orbit(angle_change);
                angle -= angle_change;
                cam.turn(angle, 0.f);
            }
            const double t0 = now_ms();
            chk(vrt_hip_group_frame_batch(grp, (int)nf, tw, th, views.data(), origins.data(), pack, cmd.outfile ? outs.data() : nullptr, 1), "group_frame_batch", nullptr);
            total_time += (f32)(now_ms() - t0);
            if (cmd.outfile != nullptr)
                for (u64 f = 0; f < nf; ++f) {
                    const std::string of(cmd.outfile);
                    const size_t dot = of.find_last_of('.');
                    const std::string stem = of.substr(0, dot), ext = dot == std::string::npos ? "png" : of.substr(dot + 1);
                    const std::string path = stem + "_" + std::to_string(done + f + 1) + "." + ext;
                    if (!png::write_rgba(path.c_str(), (u32)width, (u32)height, images[f].data(), width * 4))
                        fprintf(stderr, "[ ERROR ]\tcould not write %s\n", path.c_str());
                }
            done += nf;
        }
        chk(vrt_hip_group_sync(grp), "sync", nullptr);
        printf("AVG. TIME: %g ms (%llu frames)\n", total_time / cmd.nr_frames, (unsigned long long)cmd.nr_frames);
        vrt_hip_group_destroy(grp);
        return EXIT_SUCCESS;
    }
    for (u64 frames = 1;; ++frames) {
        const f32 origin[3] = { cam.position[0], cam.position[1], cam.position[2] };
        const double t0 = now_ms();
        if (frames == 1) t_first = t0;
        auto set_rays = [&](vrt_hip_ctx *ctx) {
            if (cmd.plane_arrays)
                chk(vrt_hip_set_plane(ctx, (u32)width, (u32)height, cam.projection_plane.xs.data(), cam.projection_plane.ys.data(),
                                      cam.projection_plane.zs.data()), "set_plane", ctx);
            else
                chk(vrt_hip_set_camera_view(ctx, (u32)width, (u32)height, cam.view_matrix.data()), "set_camera_view", ctx);
        };
        if (deal_frames) {
            vrt_hip_ctx *ctx = vrt_hip_group_ctx(grp, (int)((frames - 1) % n));
            set_rays(ctx);
            chk(vrt_hip_frame(ctx, tw, th, cam.view_matrix.data(), origin, pack, nullptr, 0), "frame", ctx);
        } else {
            for (int i = 0; i < n; ++i) set_rays(vrt_hip_group_ctx(grp, i));
            chk(vrt_hip_group_frame(grp, tw, th, cam.view_matrix.data(), origin, pack, cmd.outfile ? image.data() : nullptr, 1), "group_frame", nullptr);
        }
        const f32 frame_time = (f32)(now_ms() - t0);
        if (cmd.outfile != nullptr) {
            const std::string of(cmd.outfile);
            const size_t dot = of.find_last_of('.');
            const std::string stem = of.substr(0, dot), ext = dot == std::string::npos ? "png" : of.substr(dot + 1);
            const std::string path = cmd.nr_frames > 1 ? stem + "_" + std::to_string(frames) + "." + ext : stem + "." + ext;
            if (!png::write_rgba(path.c_str(), (u32)width, (u32)height, image.data(), width * 4))
                fprintf(stderr, "[ ERROR ]\tcould not write %s\n", path.c_str());
        }
        if (cmd.nr_frames == 1) printf("TIME: %g ms\n", frame_time);
        total_time += frame_time;
        if (cmd.nr_frames == frames) {
            chk(vrt_hip_group_sync(grp), "sync", nullptr);
            if (deal_frames) total_time = (f32)(now_ms() - t_first);
            if (cmd.nr_frames > 1) printf("AVG. TIME: %g ms (%llu frames)\n", total_time / cmd.nr_frames, (unsigned long long)cmd.nr_frames);
            break;
        }
        const f32 angle_change = cmd.rot / cmd.nr_frames;
        cam.orbit(angle_change);
        angle -= angle_change;
        cam.turn(angle, 0.f);
    }
    vrt_hip_group_destroy(grp);
    return EXIT_SUCCESS;
}

int main(int argc, char **argv)
{
    cmd_args_t cmd(argc, argv);
    std::vector<vrt::gaussian_t> gaussians;
    if (cmd.infile != nullptr) {
        gaussians = read_from_obj(cmd.infile);
    } else { // main.cpp:194-205
        const u8 grid_dim = (u8)cmd.grid_dim;
        for (u8 i = 0; i < grid_dim; ++i)
            for (u8 j = 0; j < grid_dim; ++j)
                gaussians.push_back(vrt::gaussian_t{
                    { 1.f - (i * grid_dim + j) / (f32)(grid_dim * grid_dim), 0.f, 0.f + (i * grid_dim + j) / (f32)(grid_dim * grid_dim), 1.f },
                    { -1.f + 1.f / grid_dim + i * 1.f / (grid_dim / 2.f), -1.f + 1.f / grid_dim + j * 1.f / (grid_dim / 2.f), 1.f },
                    1.f / (2 * grid_dim), 1.f });
    }

    // mode table (main.cpp:150-177, 269-294): tiling, packing, Exp/Erf
    const bool use_tiling = cmd.mode >= 5;
    const u64 base_mode = (cmd.mode - 1) % 4 + 1;
    int pack = VRT_PACK_TRUNC | VRT_ALPHA_OPAQUE, ek = VRT_EXP_VCL, rk = VRT_ERF_AS;
    if (base_mode == 1) { ek = VRT_EXP_LIBM; rk = VRT_ERF_LIBM; }
    if (base_mode == 4) pack = VRT_PACK_ROUND | (use_tiling ? VRT_ALPHA_COMPUTED : VRT_ALPHA_OPAQUE);

    if (cmd.gpus > 1 && use_tiling) return run_on_group(cmd, gaussians, pack, ek, rk);
    if (cmd.gpus > 1) fprintf(stderr, "[ INFO ]\t--gpus applies to the tiled modes (5-8): an untiled frame has no tiles to shard; using one GPU\n");

    // An animation whose frames are not written out keeps four frames in flight: the frames alternate between four
    // contexts (each has its own HIP stream), so one frame's list kernel and the tail of its block kernel overlap the
    // other frames' work (like bench.py).  Frames that fill the chip for milliseconds gain nothing from that and lose to the
    // other frames' workgroups on their CUs (monkey 4096^2 orbit: 19.6 ms per frame on one context, 21.4 on two or four; a teapot
    // at 512^2, 1.4 ms of a half-empty chip: 1.40 / 0.71 / 0.51): a second untimed frame on the first context decides
    // (more than 3 ms: one context).  VRT_CLI_CONTEXTS=1..8 fixes the number.  Scheduling only: the images do not depend on it.
    int in_flight = 4; // one per hardware queue of the HIP runtime
    const bool fixed_contexts = getenv("VRT_CLI_CONTEXTS") != nullptr;
    if (fixed_contexts) in_flight = std::max(1, std::min(8, atoi(getenv("VRT_CLI_CONTEXTS"))));
    int nctx = (cmd.outfile == nullptr && cmd.nr_frames > 1 && getenv("VRT_CLI_TRACE") == nullptr) ? in_flight : 1;
    vrt_hip_ctx *ctxs[8] = {};
    const char *dev = getenv("VRT_HIP_DEVICE");
    vrt_hip_ctx *ctx = nullptr;
    auto chk = [&](int rc, const char *what) {
        if (rc != VRT_HIP_OK) { fprintf(stderr, "[ ERROR ]\t%s: %s\n", what, vrt_hip_last_error(ctx)); exit(EXIT_FAILURE); }
    };
    // (all contexts -- their streams -- are created before any work is enqueued: the runtime deals its hardware queues to streams as they
    // appear, and streams created between another context's copies and launches ended up two to a queue: 0.76 instead of 0.51 ms per
    // frame for the teapot at 512^2)
    const int n_created = nctx;
    for (int i = 0; i < n_created; ++i)
        if (vrt_hip_create(dev ? atoi(dev) : 0, &ctxs[i]) != VRT_HIP_OK) {
            fprintf(stderr, "[ ERROR ]\t%s\n", vrt_hip_last_error(nullptr));
            return EXIT_FAILURE;
        }
    auto configure = [&](int i) {
        ctx = ctxs[i];
        chk(vrt_hip_set_gaussians_aos(ctx, gaussians.size(), gaussians.data()), "set_gaussians");
        chk(vrt_hip_set_options(ctx, ek, rk, cmd.cull_eps), "set_options");
        chk(vrt_hip_set_table_step(ctx, cmd.table_step), "set_table_step");
        chk(vrt_hip_set_table_budget(ctx, cmd.table_budget), "set_table_budget");
        chk(vrt_hip_set_cull_prune(ctx, cmd.cull_prune), "set_cull_prune");
    };

    const u64 width = cmd.w, height = cmd.h;
    std::vector<u32> image(width * height);

    // main.cpp:247-255
    vrt::camera_t cam({ 0.f, 0.f, cmd.camera_offset }, { 0.f, 1.f, 0.f }, { 0.f, 0.f, 1.f }, -90.f, 0.f,
                      cmd.plane_arrays ? width : 1, cmd.plane_arrays ? height : 1, cmd.focal_length);
    f32 angle = -90.f;
    cam.orbit(cmd.inital_rot);
    angle -= cmd.inital_rot;
    cam.turn(angle, 0.f);

    // One untimed frame per context first: the reference's TIME line covers tiling + rendering (main.cpp:260-296), not
    // the one-off cost of a first GPU launch (code-object load, buffer allocation: ~9 ms).  VRT_CLI_NO_WARMUP=1 skips it.
    // The warm-up looks from ANOTHER pose (7 degrees further along the orbit): what the library keeps per camera -- tile cones,
    // per-origin tables, the dense-launch report -- is then not in place for the timed frame, which does a full frame's
    // work like the reference's (round-2 advisor: the warm-up used the timed frame's own pose).
    const bool warm_up = getenv("VRT_CLI_NO_WARMUP") == nullptr;
    vrt::camera_t wcam = cam;
    wcam.orbit(7.f);
    wcam.turn(angle - 7.f, 0.f);
    auto warm_frame = [&](int i) {
        ctx = ctxs[i];
        const f32 origin[3] = { wcam.position[0], wcam.position[1], wcam.position[2] };
        if (cmd.plane_arrays)
            chk(vrt_hip_set_plane(ctx, (u32)width, (u32)height, wcam.projection_plane.xs.data(), wcam.projection_plane.ys.data(),
                                  wcam.projection_plane.zs.data()), "set_plane");
        else
            chk(vrt_hip_set_camera_view(ctx, (u32)width, (u32)height, wcam.view_matrix.data()), "set_camera_view");
        if (use_tiling) {
            chk(vrt_hip_frame(ctx, 2.f / cmd.tiles, 2.f / cmd.tiles, wcam.view_matrix.data(), origin, pack, nullptr, 1), "frame");
        } else {
            chk(vrt_hip_clear_tiles(ctx), "clear_tiles");
            chk(vrt_hip_render(ctx, origin, pack, nullptr, nullptr), "render");
        }
    };
    for (int i = 0; i < n_created; ++i) configure(i);
    if (warm_up) {
        warm_frame(0);
        if (nctx > 1 && !fixed_contexts) { // how long does a frame of this scene take?
            const double p0 = now_ms();
            warm_frame(0);
            if (now_ms() - p0 > 3.0) nctx = 1;
        }
        for (int i = 1; i < nctx; ++i) warm_frame(i);
    }
    ctx = ctxs[0];

    f32 total_time = 0.f;
    double t_first = 0.0;
    for (u64 frames = 1;; ++frames) {
        ctx = ctxs[(frames - 1) % nctx];
        const f32 origin[3] = { cam.position[0], cam.position[1], cam.position[2] };
        // rays for this camera pose (the reference rebuilds the plane arrays in cam.turn(), outside its timers)
        if (cmd.plane_arrays)
            chk(vrt_hip_set_plane(ctx, (u32)width, (u32)height, cam.projection_plane.xs.data(), cam.projection_plane.ys.data(),
                                  cam.projection_plane.zs.data()), "set_plane");
        else
            chk(vrt_hip_set_camera_view(ctx, (u32)width, (u32)height, cam.view_matrix.data()), "set_camera_view");

        // Tiled modes: tile_gaussians + render in one call on the context's stream.  A frame whose image is needed on
        // the host (PNG) or whose time is printed is waited for; the frames of an animation that only reports its
        // average (main.cpp:310-315) are enqueued back to back and the clock stops after the last one.
        const bool need_image = cmd.outfile != nullptr;
        const bool trace = getenv("VRT_CLI_TRACE") != nullptr; // per-frame times on stderr (waits for every frame)
        const bool trace_async = trace && atoi(getenv("VRT_CLI_TRACE")) == 2;
        const bool wait = need_image || cmd.nr_frames == 1 || (trace && !trace_async);
        double t0 = now_ms();
        if (frames == 1) t_first = t0;
        if (use_tiling) {
            chk(vrt_hip_frame(ctx, 2.f / cmd.tiles, 2.f / cmd.tiles, cam.view_matrix.data(), origin, pack,
                              need_image ? image.data() : nullptr, wait ? 1 : 0), "frame");
        } else {
            chk(vrt_hip_clear_tiles(ctx), "clear_tiles");
            chk(vrt_hip_render(ctx, origin, pack, need_image ? image.data() : nullptr, nullptr), "render");
        }
        const f32 frame_time = (f32)(now_ms() - t0);
        if (trace) fprintf(stderr, "frame %llu angle %g: %g ms\n", (unsigned long long)frames, angle, frame_time);

        if (cmd.outfile != nullptr) { // main.cpp:299-307: <stem>_<frame>.<ext> when more than one frame
            const std::string of(cmd.outfile);
            const size_t dot = of.find_last_of('.');
            const std::string stem = of.substr(0, dot), ext = dot == std::string::npos ? "png" : of.substr(dot + 1);
            const std::string path = cmd.nr_frames > 1 ? stem + "_" + std::to_string(frames) + "." + ext : stem + "." + ext;
            if (!png::write_rgba(path.c_str(), (u32)width, (u32)height, image.data(), width * 4))
                fprintf(stderr, "[ ERROR ]\tcould not write %s\n", path.c_str());
        }
        if (cmd.nr_frames == 1) printf("TIME: %g ms\n", frame_time);
        total_time += frame_time;
        if (cmd.nr_frames == frames) {
            const double ts = now_ms();
            for (int i = 0; i < nctx; ++i) { ctx = ctxs[i]; chk(vrt_hip_sync(ctx), "sync"); }
            if (trace) fprintf(stderr, "final sync: %g ms\n", now_ms() - ts);
            if (!wait) total_time = (f32)(now_ms() - t_first); // frames were not waited for one by one
            if (cmd.nr_frames > 1) printf("AVG. TIME: %g ms (%llu frames)\n", total_time / cmd.nr_frames, (unsigned long long)cmd.nr_frames);
            break;
        }
        // main.cpp:329-334
        const f32 angle_change = cmd.rot / cmd.nr_frames;
        cam.orbit(angle_change);
        angle -= angle_change;
        cam.turn(angle, 0.f);
    }
    for (int i = 0; i < n_created; ++i) vrt_hip_destroy(ctxs[i]);
    return EXIT_SUCCESS;
}
