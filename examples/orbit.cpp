// Headless orbit driver: what the reference's render loop (Source/VulkanApplication.cpp:549-625)
// does with its window, done with files.  A C++ host program on the C ABI alone (include/vr_abi.h):
// loads a volume (a directory of big-endian u16 slices as Applications/VolumeRendering/Source/
// main.cpp:112-165 reads them, or the synthetic phantom), installs a transfer function
// (main.cpp:643-686), turns the camera (main.cpp:737-799) and writes every frame as a binary PPM.
//
//   orbit --phantom 256 --size 512x512 --frames 8 --out frames/      (or --slices DIR)
//   orbit --phantom 2048 --size 3840x2160 --frames 60                (no --out: timing only)
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

    std::vector<uint8_t> frame((size_t)width * height * 4), rgb((size_t)width * height * 3);
    const float rad = 0.01745329251994329576923690768489f;
    double render_s = 0.0;
    for (int f = 0; f < frames; f++) {
        float h = h0 + dh * (float)f;
        while (h > 180.0f) h -= 360.0f;
        check(vr_set_camera(g_ctx, h * rad, v * rad), "vr_set_camera");
        const auto t0 = std::chrono::steady_clock::now();
        check(vr_render(g_ctx, width, height, frame.data(), 0, nullptr), "vr_render");   // host pointer: includes the copy back
        render_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (!out.empty()) {
            for (size_t p = 0; p < (size_t)width * height; p++) std::memcpy(&rgb[3 * p], &frame[4 * p], 3);
            char name[64];
            std::snprintf(name, sizeof name, "/frame_%04d.ppm", f);
            const std::string path = out + name;
            FILE *fp = std::fopen(path.c_str(), "wb");
            if (!fp) { std::fprintf(stderr, "cannot write %s\n", path.c_str()); return 1; }
            std::fprintf(fp, "P6\n%d %d\n255\n", width, height);
            std::fwrite(rgb.data(), 1, rgb.size(), fp);
            std::fclose(fp);
        }
    }
    std::printf("{\"volume\": [%d, %d, %d], \"image\": [%d, %d], \"samples_per_ray\": %d, \"frames\": %d, "
                "\"ms_per_frame_with_copy_to_host\": %.3f}\n", nx, ny, nz, width, height, samples, frames, render_s / frames * 1e3);
    vr_destroy(g_ctx);
    return 0;
}
