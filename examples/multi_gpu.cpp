// One frame from several GPUs: a C++ host program on the two C ABIs (include/vr_abi.h, include/vr_mg.h).
// One process per GPU, started by any launcher that gives every process its rank:
//
//   multi_gpu --rank R --world N --id-file /tmp/vr.id [--phantom 256] [--size 640x360] [--out frame.ppm]
//
// Rank 0 writes the RCCL unique id to the id file, the other ranks wait for it.  Every rank holds the
// whole phantom (sort-first frame) and, in a second context, its box of it (sort-last frames with both
// exchange schedules); rank 0 checks that all frames are the one a single GPU renders and writes it.
// The reference has no counterpart: it renders on one GPU (Source/VulkanApplication.cpp:316-321).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../include/vr_mg.h"

static void die(const char *what, const char *detail)
{
    std::fprintf(stderr, "%s: %s\n", what, detail);
    std::exit(1);
}
#define VR(ctx, call) do { if ((call) != VR_OK) die(#call, vr_last_error(ctx)); } while (0)
#define MG(mg, call) do { if ((call) != VR_OK) die(#call, vrmg_last_error(mg)); } while (0)

static void set_up(vr_ctx *c, int samples)
{
    const float xs[4] = {0.0f, 120.0f, 300.0f, 483.0f};
    const float ys[4][4] = {{0, .9f, 1, 1}, {0, .3f, .8f, 1}, {0, .1f, .5f, 1}, {0, 0, .05f, .6f}};
    vr_ctrl_pt pts[4][4];
    const vr_ctrl_pt *ptr[4];
    int n[4];
    for (int ch = 0; ch < 4; ch++) {
        for (int k = 0; k < 4; k++) pts[ch][k] = {xs[k], ys[ch][k]};
        ptr[ch] = pts[ch];
        n[ch] = 4;
    }
    VR(c, vr_set_transfer_function(c, ptr, n, 1));
    VR(c, vr_set_sampling(c, samples, nullptr));
    VR(c, vr_set_camera(c, 0.6f, 0.35f));
    VR(c, vr_set_mode(c, VR_MODE_FAST));
}

int main(int argc, char **argv)
{
    int rank = 0, world = 1, phantom = 128, width = 320, height = 200;
    std::string id_file = "/tmp/vr_mg.id", out;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto next = [&]() -> const char * { if (i + 1 >= argc) die(a.c_str(), "needs a value"); return argv[++i]; };
        if (a == "--rank") rank = std::atoi(next());
        else if (a == "--world") world = std::atoi(next());
        else if (a == "--id-file") id_file = next();
        else if (a == "--phantom") phantom = std::atoi(next());
        else if (a == "--size") { if (std::sscanf(next(), "%dx%d", &width, &height) != 2) return 2; }
        else if (a == "--out") out = next();
        else die("unknown option", a.c_str());
    }
    int devices = 0;
    if (hipGetDeviceCount(&devices) != hipSuccess || devices < 1) die("hipGetDeviceCount", "no GPU");
    const int device = rank % devices;
    const int samples = 2 * phantom;

    // the RCCL unique id: made on rank 0, read by the others
    unsigned char id[VRMG_UNIQUE_ID_BYTES];
    if (rank == 0) {
        if (vrmg_get_unique_id(id) != VR_OK) die("vrmg_get_unique_id", "failed");
        const std::string tmp = id_file + ".tmp";
        FILE *f = std::fopen(tmp.c_str(), "wb");
        if (!f || std::fwrite(id, 1, sizeof id, f) != sizeof id) die("id file", "cannot write");
        std::fclose(f);
        std::rename(tmp.c_str(), id_file.c_str());
    } else {
        for (int tries = 0;; tries++) {
            FILE *f = std::fopen(id_file.c_str(), "rb");
            if (f) {
                const size_t got = std::fread(id, 1, sizeof id, f);
                std::fclose(f);
                if (got == sizeof id) break;
            }
            if (tries > 600) die("id file", "never appeared");
            std::this_thread::sleep_for(std::chrono::milliseconds(100));
        }
    }

    // whole volume (sort-first) and this rank's box of it (sort-last), both from the phantom generator
    vr_ctx *whole = nullptr, *part = nullptr;
    if (vr_create(device, &whole) != VR_OK || vr_create(device, &part) != VR_OK) die("vr_create", "failed");
    const size_t voxels = (size_t)phantom * phantom * phantom;
    void *d_raw = nullptr;
    if (hipMalloc(&d_raw, voxels * 2) != hipSuccess) die("hipMalloc", "volume");
    VR(whole, vr_phantom_generate_device(whole, d_raw, VR_DTYPE_U16, phantom, 0x5EED1234u));
    VR(whole, vr_upload_volume_device(whole, d_raw, VR_DTYPE_U16, phantom, phantom, phantom, 0.0f));
    VR(whole, vr_synchronize(whole));
    // box grid: as cubic as powers of two allow
    int grid[3] = {1, 1, 1};
    for (int w = world, axis = 0; w > 1; w /= 2, axis = (axis + 1) % 3) {
        if (w % 2) die("--world", "a power of two, please");
        grid[axis] *= 2;
    }
    int bricks[3], lo[3], hi[3], vlo[3], vhi[3];
    vr_brick_grid(phantom, phantom, phantom, bricks);
    const int box[3] = {rank % grid[0], (rank / grid[0]) % grid[1], rank / (grid[0] * grid[1])};
    for (int k = 0; k < 3; k++) {
        lo[k] = box[k] * bricks[k] / grid[k];
        hi[k] = (box[k] + 1) * bricks[k] / grid[k];
    }
    if (vr_subvolume_voxels(phantom, phantom, phantom, lo, hi, vlo, vhi) != VR_OK) die("vr_subvolume_voxels", "failed");
    VR(part, vr_phantom_generate_box_device(part, d_raw, VR_DTYPE_U16, phantom, vlo, vhi, 0x5EED1234u));
    VR(part, vr_upload_subvolume_device(part, d_raw, VR_DTYPE_U16, phantom, phantom, phantom, lo, hi, 65535.0f));
    VR(part, vr_synchronize(part));
    (void)hipFree(d_raw);
    set_up(whole, samples);
    set_up(part, samples);

    vrmg *first = nullptr, *last = nullptr;
    if (vrmg_create(whole, id, rank, world, &first) != VR_OK) die("vrmg_create", "failed");
    const size_t frame_bytes = (size_t)width * height * 4;
    uint8_t *d_a = nullptr, *d_b = nullptr;
    if (hipMalloc(&d_a, frame_bytes) != hipSuccess || hipMalloc(&d_b, frame_bytes) != hipSuccess) die("hipMalloc", "frames");
    std::vector<uint8_t> single(frame_bytes), got(frame_bytes);
    VR(whole, vr_render(whole, width, height, single.data(), 0, nullptr));     // what one GPU renders

    MG(first, vrmg_render_sort_first(first, width, height, d_a));
    MG(first, vrmg_synchronize(first));       // the gather runs on the object's own stream
    int bad = 0;
    if (rank == 0) {
        if (hipMemcpy(got.data(), d_a, frame_bytes, hipMemcpyDeviceToHost) != hipSuccess) die("hipMemcpy", "frame");
        if (std::memcmp(got.data(), single.data(), frame_bytes) != 0) { std::fprintf(stderr, "sort-first frame differs\n"); bad = 1; }
    }
    MG(first, vrmg_destroy(first));

    // rank 0 has to publish a fresh id for the second communicator: reuse the file with a suffix
    const std::string id_file2 = id_file + ".2";
    if (rank == 0) {
        if (vrmg_get_unique_id(id) != VR_OK) die("vrmg_get_unique_id", "failed");
        const std::string tmp = id_file2 + ".tmp";
        FILE *f = std::fopen(tmp.c_str(), "wb");
        if (!f || std::fwrite(id, 1, sizeof id, f) != sizeof id) die("id file", "cannot write");
        std::fclose(f);
        std::rename(tmp.c_str(), id_file2.c_str());
    } else {
        for (int tries = 0;; tries++) {
            FILE *f = std::fopen(id_file2.c_str(), "rb");
            if (f) {
                const size_t n = std::fread(id, 1, sizeof id, f);
                std::fclose(f);
                if (n == sizeof id) break;
            }
            if (tries > 600) die("id file", "never appeared");
            std::this_thread::sleep_for(std::chrono::milliseconds(100));
        }
    }
    if (vrmg_create(part, id, rank, world, &last) != VR_OK) die("vrmg_create", "failed");
    MG(last, vrmg_set_box_grid(last, grid));
    for (int schedule : {VRMG_DIRECT_SEND, VRMG_BINARY_SWAP}) {
        uint64_t remarched = 0;
        MG(last, vrmg_render_sort_last(last, width, height, schedule, d_b, &remarched));
        MG(last, vrmg_synchronize(last));
        if (rank == 0) {
            if (hipMemcpy(got.data(), d_b, frame_bytes, hipMemcpyDeviceToHost) != hipSuccess) die("hipMemcpy", "frame");
            const bool same = std::memcmp(got.data(), single.data(), frame_bytes) == 0;
            std::printf("sort-last (%s): %s, %llu pixels re-marched\n", schedule == VRMG_DIRECT_SEND ? "direct send" : "binary swap",
                        same ? "identical to the single-GPU frame" : "DIFFERS", (unsigned long long)remarched);
            bad |= !same;
        }
    }
    MG(last, vrmg_destroy(last));
    if (rank == 0) {
        std::printf("sort-first: %s\n", bad ? "see above" : "identical to the single-GPU frame");
        if (!out.empty()) {
            FILE *f = std::fopen(out.c_str(), "wb");
            if (!f) die("cannot write", out.c_str());
            std::fprintf(f, "P6\n%d %d\n255\n", width, height);
            for (size_t p = 0; p < (size_t)width * height; p++) std::fwrite(&single[p * 4], 1, 3, f);
            std::fclose(f);
        }
        std::remove(id_file.c_str());
        std::remove(id_file2.c_str());
    }
    (void)hipFree(d_a);
    (void)hipFree(d_b);
    vr_destroy(whole);
    vr_destroy(part);
    return bad;
}
