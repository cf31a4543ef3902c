// Headless orbit driver: what the reference's render loop (Source/VulkanApplication.cpp:549-625)
// does with its window, done with files.  A C++ host program on the C ABI alone (include/vr_abi.h):
// loads a volume (a directory of big-endian u16 slices as Applications/VolumeRendering/Source/
// main.cpp:112-165 reads them, or the synthetic phantom), installs a transfer function
// (main.cpp:643-686), turns the camera (main.cpp:737-799) and writes every frame as a binary PPM.
//
//   orbit --phantom 256 --size 512x512 --frames 8 --out frames/      (or --slices DIR)
//   orbit --phantom 2048 --size 3840x2160 --frames 60                (no --out: timing only)
//
// Frames reach the host through the library's ring of page-locked images (vr_render_mapped /
// vr_map_frame): frame n is copied while frame n + 1 is rendered, and this loop writes frame n - 1 in the
// meantime -- the part of the reference's per-frame copyImage hand-off (main.cpp:903-909).  --pageable
// takes the older path instead, vr_render into a std::vector (a synchronous copy into pageable memory).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/vr_abi.h"

static vr_ctx *g_ctx = nullptr;

static void check(int rc, const char *what)
{
    if (rc == VR_OK) return;
    std::fprintf(stderr, "%s failed (%d): %s\n", what, rc, vr_last_error(g_ctx));
    std::exit(1);
}

int main(int argc, char **argv)
{
    std::string slices, out;
    int phantom = 0, width = 512, height = 512, frames = 8, samples = 200, mode = VR_MODE_FAST, dtype = VR_DTYPE_U16;
    float h0 = -180.0f, dh = 1.0f, v = 20.0f;
    bool pageable = false;
    int in_flight = 1;   // 2: vr_set_pipeline(ctx, 2); measured slower with the copy to the host than one frame at a time
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto next = [&]() -> const char * { if (i + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", a.c_str()); std::exit(2); } return argv[++i]; };
        if (a == "--slices") slices = next();
        else if (a == "--phantom") phantom = std::atoi(next());
        else if (a == "--u8") dtype = VR_DTYPE_U8;
        else if (a == "--size") { if (std::sscanf(next(), "%dx%d", &width, &height) != 2) return 2; }
        else if (a == "--frames") frames = std::atoi(next());
        else if (a == "--samples") samples = std::atoi(next());
        else if (a == "--mode") mode = std::atoi(next());
        else if (a == "--start") h0 = (float)std::atof(next());
        else if (a == "--step") dh = (float)std::atof(next());
        else if (a == "--elevation") v = (float)std::atof(next());
        else if (a == "--out") out = next();
        else if (a == "--pageable") pageable = true;
        else if (a == "--in-flight") in_flight = std::atoi(next());
        else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    if (slices.empty() && phantom <= 0) phantom = 128;

    check(vr_create(0, &g_ctx), "vr_create");
    if (!slices.empty()) {
        check(vr_load_volume_slices(g_ctx, slices.c_str()), "vr_load_volume_slices");
    } else {
        void *d_raw = nullptr;
        const size_t bytes = (size_t)phantom * phantom * phantom * (size_t)dtype;
        if (hipMalloc(&d_raw, bytes) != hipSuccess) { std::fprintf(stderr, "hipMalloc of %zu bytes failed\n", bytes); return 1; }
        check(vr_phantom_generate_device(g_ctx, d_raw, dtype, phantom, 0x5EED1234u), "vr_phantom_generate_device");
        check(vr_upload_volume_device(g_ctx, d_raw, dtype, phantom, phantom, phantom, 0.0f), "vr_upload_volume_device");
        (void)hipFree(d_raw);
    }
    int nx, ny, nz, dt;
    float mx;
    check(vr_volume_info(g_ctx, &nx, &ny, &nz, &dt, &mx), "vr_volume_info");

    // four control points per channel (x in editor pixels, the last x scales the axis: main.cpp:653)
    const float xs[4] = {0.0f, 120.0f, 300.0f, 483.0f};
    const float ys[4][4] = {{0, .9f, 1, 1}, {0, .3f, .8f, 1}, {0, .1f, .5f, 1}, {0, 0, .05f, .6f}};
    vr_ctrl_pt pts[4][4];
    const vr_ctrl_pt *ptr[4];
    int counts[4];
    for (int c = 0; c < 4; c++) {
        for (int k = 0; k < 4; k++) pts[c][k] = {xs[k], ys[c][k]};
        ptr[c] = pts[c];
        counts[c] = 4;
    }
    check(vr_set_transfer_function(g_ctx, ptr, counts, 1), "vr_set_transfer_function");
    check(vr_set_sampling(g_ctx, samples, nullptr), "vr_set_sampling");
    check(vr_set_mode(g_ctx, mode), "vr_set_mode");

    std::vector<uint8_t> frame(pageable ? (size_t)width * height * 4 : 0), rgb((size_t)width * height * 3);
    const float rad = 0.01745329251994329576923690768489f;
    auto write_frame = [&](const uint8_t *rgba, int f) {
        if (out.empty()) return;
        for (size_t p = 0; p < (size_t)width * height; p++) std::memcpy(&rgb[3 * p], &rgba[4 * p], 3);
        char name[64];
        std::snprintf(name, sizeof name, "/frame_%04d.ppm", f);
        const std::string path = out + name;
        FILE *fp = std::fopen(path.c_str(), "wb");
        if (!fp) { std::fprintf(stderr, "cannot write %s\n", path.c_str()); std::exit(1); }
        std::fprintf(fp, "P6\n%d %d\n255\n", width, height);
        std::fwrite(rgb.data(), 1, rgb.size(), fp);
        std::fclose(fp);
    };
    auto set_view = [&](int f) {
        float h = h0 + dh * (float)f;
        while (h > 180.0f) h -= 360.0f;
        check(vr_set_camera(g_ctx, h * rad, v * rad), "vr_set_camera");
    };
    if (!pageable) check(vr_set_pipeline(g_ctx, in_flight == 2 ? 2 : 1), "vr_set_pipeline");
    double enqueue_s = 0.0, wait_s = 0.0;   // host time inside vr_render_mapped and inside vr_map_frame
    unsigned long long checksum = 0;   // touches every frame's bytes on the host, so the timing includes their arrival
    const auto t_begin = std::chrono::steady_clock::now();
    if (pageable) {
        for (int f = 0; f < frames; f++) {
            set_view(f);
            check(vr_render(g_ctx, width, height, frame.data(), 0, nullptr), "vr_render");   // host pointer: includes the copy back
            checksum += frame[(size_t)f % frame.size()];
            write_frame(frame.data(), f);
        }
    } else {
        // the host takes frame f - lag while frames f - lag + 1 .. f are rendered and copied; with two frames in
        // flight it stays two behind, so that neither the march of the next frame nor the previous frame's
        // last kernels ever wait for it (the ring has three slots: f, f - 1 and the one being read)
        const int lag = in_flight == 2 ? 2 : 1;
        int ids[3] = {-1, -1, -1};   // slot of frame f % 3
        for (int f = 0; f < frames + lag; f++) {
            if (f < frames) {
                set_view(f);
                const auto t0 = std::chrono::steady_clock::now();
                check(vr_render_mapped(g_ctx, width, height, &ids[f % 3]), "vr_render_mapped");
                enqueue_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            }
            if (f >= lag) {
                const int g = f - lag;
                const uint8_t *pixels = nullptr;
                const auto t0 = std::chrono::steady_clock::now();
                check(vr_map_frame(g_ctx, ids[g % 3], &pixels), "vr_map_frame");
                wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                checksum += pixels[(size_t)g % ((size_t)width * height * 4)];
                write_frame(pixels, g);
                check(vr_unmap_frame(g_ctx, ids[g % 3]), "vr_unmap_frame");
            }
        }
    }
    const double render_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    std::printf("{\"volume\": [%d, %d, %d], \"image\": [%d, %d], \"samples_per_ray\": %d, \"frames\": %d, "
                "\"hand_off\": \"%s\", \"ms_per_frame_with_copy_to_host\": %.3f, \"host_ms_per_frame_enqueueing\": %.3f, "
                "\"host_ms_per_frame_waiting\": %.3f, \"checksum\": %llu}\n", nx, ny, nz, width, height, samples, frames,
                pageable ? "vr_render into pageable memory" : "vr_render_mapped ring of page-locked images", render_s / frames * 1e3,
                enqueue_s / frames * 1e3, wait_s / frames * 1e3, checksum);
    vr_destroy(g_ctx);
    return 0;
}
