// Host side of libvr_hip.so: the context object and the extern "C" entry points declared in
// include/vr_abi.h.  Plays the part of the VolumeRendering application's initComputePipeline /
// preRenderLoop / renderCommands / postRenderLoop (Applications/VolumeRendering/Source/main.cpp:
// 203-573, 574-641, 800-935, 1246-1283) with HIP in place of the Vulkan objects.
#include "../../include/vr_abi.h"

#include <algorithm>
#include <chrono>
#include <cerrno>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <string>
#include <vector>

#include "vr_camera.hpp"
#include "vr_march.hpp"
#include "vr_sortlast.hpp"

struct vr_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // ring of event pairs around the main march kernel of the most recent frames
    static constexpr int kRing = 1024;
    std::vector<hipEvent_t> ring_a, ring_b;
    uint64_t frames = 0;
    std::string error;

    // volume
    void *bricks = nullptr;
    uint32_t *cells = nullptr;
    uint32_t *touched = nullptr;  // one bit per brick, stats only
    uint32_t *empty[vr::kLevels] = {nullptr, nullptr, nullptr, nullptr};  // occupancy pyramid for the current table
    int lx[vr::kLevels] = {0}, ly[vr::kLevels] = {0}, lz[vr::kLevels] = {0};
    size_t brick_count = 0;
    unsigned long long *micro = nullptr;  // per brick: bits of its 2^3-cell blocks
    int *occ = nullptr;                   // bounding box of the occupied bricks (6 ints)
    uint32_t *skip = nullptr;             // per brick: 2-bit code of the coarsest empty cube around it
    uint32_t *occ_list = nullptr;         // indices of the occupied bricks, for the per-tile depth ranges
    uint32_t *d_occ_count = nullptr;      // its length; stays on the device (k_splat_bricks reads it there)
    uint32_t *d_data_max = nullptr;       // where an upload's kernels leave the largest voxel value
    uint2 *tiles = nullptr;               // per 8x8 pixel tile: depth range of the occupied bricks (fast mode)
    size_t tiles_cap = 0;
    bool tile_clip = true;
    bool occupancy_dirty = true;
    bool occupancy_force = true;          // the next refresh rebuilds every occupancy table whatever the table says (new volume)
    hipEvent_t occ_ready = nullptr;       // recorded behind a refresh: the stream the depth ranges are built on waits for it
    // Fast mode pays off when rays become opaque or leave the object early.  With a table of low alpha
    // everywhere neither happens: every tap is marched, the rounding bound of the decision grows with the
    // number of taps blended, and the undecided rays are marched a second and third time by the replay.
    // Every fast frame that covers the whole image therefore reports, through a word of page-locked host
    // memory its last kernel writes, how many of its rays went through the replay.  The host looks at the
    // word when it next renders (it never waits for it): with more than one in a hundred of the pixels replayed, fast
    // frames are rendered by the skipping reference-order march (same bytes, every tap once) until table,
    // border, shading, sampling or volume change, with one fast frame in every kFallbackRetry to look again
    // (the view may have changed).
    bool fast_fallback = false;
    static constexpr int kFallbackRetry = 128;
    // Rays that end in the replay as a share of the frame's pixels: one in a thousand under a table that makes
    // rays opaque, several in a hundred under one that does not (and those frames hand half of their rays to the
    // wave march with most of their taps still to do: 7.9 ms against 3.5 for 512^3 at 1080p).
    static constexpr size_t kFallbackShare = 100;  // more than 1 / 100 of the pixels replayed: give way
    int fallback_frames = 0;
    unsigned long long *probe_word = nullptr;   // host, page-locked: frame sequence number << 32 | replayed rays
    unsigned long long *probe_dev = nullptr;    // the same word as the device addresses it
    uint32_t frame_seq = 0, probe_seen = 0, table_epoch = 0;
    struct ProbeMeta {
        uint32_t seq, epoch;
        size_t pixels;
        bool eligible;
    } probe_meta[8] = {};
    int dtype = 0, nx = 0, ny = 0, nz = 0, nbx = 0, nby = 0, nbz = 0;
    // sort-last: the bricks held are the box [brick_lo, brick_lo + (nbx, nby, nbz)) of the volume's grid
    bool sub = false;
    int brick_lo[3] = {0, 0, 0};
    float max_value = 0.0f, inv_hi = 0.0f, inv_lo = 0.0f;

    // transfer function
    float *lut = nullptr;  // 256 x RGBA on the device, persists like `transferFunction`: the current slot of lut_store
    // Tables in turn.  A call that changes the table but not the set of transparent entries (the host builds the
    // table too and knows) writes the next slot and lets the frames in flight finish on theirs, so the editor's
    // every-frame changes of colours and opacities do not serialise the frames; `lut` and `lut_info` point at the
    // current slot.  slot_frame: value of `frames` when a frame last took the slot.
    static constexpr int kLutSlots = 4;
    float *lut_store = nullptr;
    vr::LutInfo *info_store = nullptr;
    int lut_slot = 0;
    uint64_t slot_frame[kLutSlots] = {0, 0, 0, 0};
    float host_lut[1024] = {0};         // the table as the host builds it beside the device (bit for bit the same arithmetic)
    bool host_lut_valid = true;         // false once a graph addressed a pixel outside the table
    uint8_t built_bad[vr::kLutLds] = {0};   // per entry of the LDS table: not transparent, as of the last call that marked the occupancy dirty
    bool built_bad_valid = false;
    vr::CtrlPt *d_pts = nullptr;
    size_t d_pts_cap = 0;
    // The graphs of the last call and whether the table is what they make of a zeroed table.  The builder
    // writes values that depend on the graphs alone into entries that depend on the graphs alone, so the same
    // graphs applied again leave every bit of the table as it is: such a call (the reference makes one every
    // frame, main.cpp:800-804) returns without touching the GPU.
    std::vector<vr::CtrlPt> last_pts;
    int last_offs[5] = {0, 0, 0, 0, 0};
    bool table_from_zero = true;
    float border[4] = {0.0f, 0.0f, 0.0f, 0.0f};

    // camera
    bool raw_camera = false;
    float h_rad = 0.0f, v_rad = 0.0f;
    float raw_loc[4] = {0, 0, 0, 1};
    float raw_m[16] = {0};

    int waves_per_block = 0;   // waves per workgroup of the marches, one tile each; 0: two for launches of 65536 tiles and more (1 % faster there), else one (VR_WAVES_PER_BLOCK)
    // memory round trips (fetch rounds + passes of the skipping loop) a lane spends on a ray before a
    // wave takes the ray over
    // 0: by the size of the launch (lane_budget_for); set by vr_tune or VR_LANE_BUDGET otherwise
    int lane_budget = 0;
    // The same for the sort-last lane march.  Boxes cannot stop a ray as early as the whole-volume march
    // (they do not know what lies in front), so their rays need more taps.
    int partial_budget = 96;
    int samples = 200;  // VolumeRendering.comp:173
    int mode = VR_MODE_REFERENCE_ORDER;

    vr::Counters *counters = nullptr;  // device
    vr::LutInfo *lut_info = nullptr;   // device
    uint32_t *frame = nullptr;         // device staging for host-pointer renders
    size_t frame_cap = 0;
    float *partial = nullptr;          // device staging for host-pointer partial renders
    size_t partial_cap = 0;
    vr::ReplayRay *queue = nullptr;    // rays the fast march hands to the reference-order march
    size_t queue_cap = 0;
    vr::LongRay *long_queue = nullptr; // rays the lane march hands to the wave march
    // Two frames in flight (vr_set_pipeline): fast frames alternate between the buffers above and this
    // second set, and the kernels that finish a frame run on a stream of their own, beside the next
    // frame's march.
    // the shader's commented-out local illumination (vr_set_shading)
    bool shading = false;
    float scene[4] = {1.0f, 1.0f, 1.0f, 100.0f};   // main.cpp:90-98
    int pipeline_depth = 1;
    int long_group = 0;       // lanes per handed-over ray in k_march_long: 0 = by frame size, else 32 or 64 (VR_LONG_GROUP)
    int tail_blocks = 8192;   // grid of the wave-per-ray kernels: more, shorter queues per wave even out the rays' lengths (VR_TAIL_BLOCKS)
    hipStream_t tail_stream = nullptr;
    // Three sets of per-frame buffers in turn (set 0: the buffers above).  The kernels that finish frame n
    // get their execution slots as the march of frame n + 1 drains, so they end about when that march does;
    // with two sets the march of frame n + 2 had to wait for them (a tenth of the frame time), with three it
    // follows the march of frame n + 1 directly.
    static constexpr int kSets = 3;
    hipEvent_t main_done[kSets] = {nullptr, nullptr, nullptr}, tail_done[kSets] = {nullptr, nullptr, nullptr}, frame_done = nullptr;
    bool tail_busy[kSets] = {false, false, false};
    // a set's counters are cleared behind the frame's last kernel, on that kernel's stream, so that the
    // next frame on the set need not do it in front of its march (two small kernels between two marches)
    bool counters_clean[kSets] = {false, false, false};
    // output of the frame a set last rendered: a frame that writes into memory one of the frames still being
    // finished writes into waits for that frame first
    const unsigned char *set_out[kSets] = {nullptr, nullptr, nullptr};
    size_t set_out_bytes[kSets] = {0, 0, 0};
    int last_set = -1;
    uint64_t pipe_index = 0;
    vr::Counters *counters_alt[kSets - 1] = {nullptr, nullptr};
    vr::ReplayRay *queue_alt[kSets - 1] = {nullptr, nullptr};
    vr::LongRay *long_alt[kSets - 1] = {nullptr, nullptr};
    size_t alt_queue_cap = 0;
    // Two frames in flight: the per-tile depth ranges of a frame depend on the camera alone, so they are
    // built on a third stream while the previous frame is still marching; three range buffers in turn, as a
    // buffer is read until its frame's replay has finished (two frames later at most).
    hipStream_t prep_stream = nullptr;
    static constexpr int kRangeRing = 4;   // one more than the frames whose kernels can be running
    uint2 *tiles_ring[kRangeRing] = {nullptr, nullptr, nullptr, nullptr};
    size_t tiles_ring_cap = 0;
    hipEvent_t ranges_done[kRangeRing] = {nullptr, nullptr, nullptr, nullptr}, ranges_free[kRangeRing] = {nullptr, nullptr, nullptr, nullptr};
    bool ranges_busy[kRangeRing] = {false, false, false, false};
    uint64_t ring_index = 0;
    // Longest tiles first (vr::ordered_tile), for frames rendered one at a time: the lane march of a fast
    // frame notes how long each tile took, k_order_tiles sorts the tiles by that time behind the frame's last
    // kernel, and the next frame that uses the same set of per-frame buffers launches its tiles in that order
    // when it covers the same tiles.
    bool tile_order = true;             // VR_TILE_ORDER=0 switches it off
    bool tile_order_piped = false;      // VR_TILE_ORDER=2: also with two frames in flight (for measurements)
    uint32_t *order_lists[kSets] = {nullptr, nullptr, nullptr};   // per set of per-frame buffers
    uint32_t *tile_ticks[kSets] = {nullptr, nullptr, nullptr};    // time every tile took, per set
    size_t order_cap = 0;               // tiles per list
    bool order_valid[kSets] = {false, false, false};
    int order_key[kSets][7] = {{0}, {0}, {0}};   // the rectangle, split and tile count the lists of a set are for
    // Frames for a consumer on the host (vr_render_mapped / vr_map_frame): three slots in turn, each a device
    // image and a page-locked host image of the same size; the copy of frame n runs on a stream of its own
    // under the march of frame n + 1.
    struct MappedSlot {
        uint8_t *dev = nullptr, *host = nullptr;
        size_t cap = 0;               // bytes both images hold
        hipEvent_t copied = nullptr;  // recorded behind the slot's copy
        bool pending = false, mapped = false;
    };
    static constexpr int kMapped = 3;
    MappedSlot mapped[kMapped];
    hipStream_t copy_stream = nullptr;
    uint64_t mapped_index = 0;
    // test hook (VR_TEST_FAIL_ALLOC=n at vr_create): the n-th device allocation of a volume upload
    // fails, so that the error path of the upload can be exercised without exhausting the GPU
    int fail_alloc_in = 0;
};

// ---- scratch of the entry point that allocates nothing (vr_render_static) ---------------------------
// What a fast frame needs beyond its input and output: the frame's counters, the per-tile depth
// ranges, the queue of undecided pixels and the queue of handed-over rays.  vr_render takes them from
// buffers the context allocates on demand; vr_render_static takes them from this array, which the
// code object brings along: every GPU the library runs on holds one copy of it until the process ends.
// In host code the array's name is only a handle for hipGetSymbolAddress; the device address it yields
// (one per GPU) is what kernels and hipMemsetAsync get.
namespace static_scratch {
constexpr size_t kMaxPixels = (size_t)VR_STATIC_MAX_PIXELS;
constexpr size_t kMaxTiles = 2 * kMaxPixels / 64;           // 8x8-pixel tiles; ragged image edges need more than pixels / 64
constexpr size_t align_up(size_t n) { return (n + 255) & ~(size_t)255; }
constexpr size_t kCountersAt = 0;
constexpr size_t kTilesAt = kCountersAt + align_up(sizeof(vr::Counters));
constexpr size_t kQueueAt = kTilesAt + align_up((kMaxTiles + 1) * sizeof(uint2));
constexpr size_t kLongAt = kQueueAt + align_up(kMaxPixels * sizeof(vr::ReplayRay));
constexpr size_t kBytes = kLongAt + align_up(kMaxPixels * sizeof(vr::LongRay));
constexpr int kMaxDevices = 64;
}  // namespace static_scratch
static __device__ __align__(256) unsigned char scratch[static_scratch::kBytes];
// device address of `scratch` on each GPU, looked up once per GPU by vr_create with that GPU current
static unsigned char *g_scratch_ptr[static_scratch::kMaxDevices] = {nullptr};

namespace {

int fail(vr_ctx *ctx, int code, const std::string &msg)
{
    if (ctx) ctx->error = msg;
    return code;
}

#define VR_HIP(ctx, call)                                                                              \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return fail((ctx), VR_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));         \
    } while (0)

// Every device allocation and release of the library goes through these two.  Test hook: while the
// environment variable VR_TEST_FORBID_ALLOC is set to a non-zero number they refuse and count the attempt,
// so that a test can show that an entry point which promises not to allocate does not.
unsigned long long g_forbidden_calls = 0;
bool alloc_forbidden()
{
    const char *e = std::getenv("VR_TEST_FORBID_ALLOC");
    return e && std::atoi(e) != 0;
}
template <typename P>
hipError_t dev_alloc(P **ptr, size_t bytes)
{
    if (alloc_forbidden()) {
        g_forbidden_calls++;
        *ptr = nullptr;
        return hipErrorOutOfMemory;
    }
    return hipMalloc(ptr, bytes);
}
template <typename P>
hipError_t dev_free(P *ptr)
{
    if (alloc_forbidden()) {
        g_forbidden_calls++;
        return hipErrorInvalidValue;
    }
    return hipFree(ptr);
}

// Device allocation of a volume upload; honours the context's test hook.
template <typename P>
hipError_t volume_alloc(vr_ctx *ctx, P **ptr, size_t bytes)
{
    if (ctx->fail_alloc_in > 0 && --ctx->fail_alloc_in == 0) {
        *ptr = nullptr;
        return hipErrorOutOfMemory;
    }
    return dev_alloc(ptr, bytes);
}

int bind(vr_ctx *ctx)
{
    VR_HIP(ctx, hipSetDevice(ctx->device));
    return VR_OK;
}

// Orders the context's stream after the frame-finishing kernels still running on the tail stream:
// whatever is enqueued next may overwrite what they read.
void drain(vr_ctx *ctx)
{
    for (int k = 0; k < vr_ctx::kSets; k++)
        if (ctx->tail_busy[k]) {
            (void)hipStreamWaitEvent(ctx->stream, ctx->tail_done[k], 0);
            ctx->tail_busy[k] = false;
        }
}

void release_volume(vr_ctx *ctx)
{
    drain(ctx);
    if (ctx->bricks) (void)dev_free(ctx->bricks);
    if (ctx->cells) (void)dev_free(ctx->cells);
    if (ctx->touched) (void)dev_free(ctx->touched);
    for (int l = 0; l < vr::kLevels; l++) {
        if (ctx->empty[l]) (void)dev_free(ctx->empty[l]);
        ctx->empty[l] = nullptr;
    }
    if (ctx->micro) (void)dev_free(ctx->micro);
    if (ctx->skip) (void)dev_free(ctx->skip);
    if (ctx->occ_list) (void)dev_free(ctx->occ_list);
    ctx->micro = nullptr;
    ctx->skip = nullptr;
    ctx->occ_list = nullptr;
    ctx->bricks = nullptr;
    ctx->cells = nullptr;
    ctx->touched = nullptr;
    ctx->occupancy_dirty = true;
    ctx->occupancy_force = true;
    ctx->brick_count = 0;
    ctx->dtype = 0;
    ctx->sub = false;
    ctx->brick_lo[0] = ctx->brick_lo[1] = ctx->brick_lo[2] = 0;
}

vr::VolumeView view_of(const vr_ctx *ctx)
{
    vr::VolumeView v;
    v.bricks = ctx->bricks;
    v.cells = ctx->cells;
    v.nx = ctx->nx; v.ny = ctx->ny; v.nz = ctx->nz;
    v.nbx = ctx->nbx; v.nby = ctx->nby; v.nbz = ctx->nbz;
    v.fx = (float)ctx->nx; v.fy = (float)ctx->ny; v.fz = (float)ctx->nz;
    v.inv_hi = ctx->inv_hi; v.inv_lo = ctx->inv_lo;
    v.max_value = ctx->max_value;
    for (int l = 0; l < vr::kLevels; l++) {
        v.empty[l] = ctx->empty[l];
        v.lx[l] = ctx->lx[l]; v.ly[l] = ctx->ly[l]; v.lz[l] = ctx->lz[l];
    }
    v.micro = ctx->micro;
    v.skip = ctx->skip;
    v.occ = ctx->occ;
    v.ox = ctx->brick_lo[0] * vr::kBrick;
    v.oy = ctx->brick_lo[1] * vr::kBrick;
    v.oz = ctx->brick_lo[2] * vr::kBrick;
    return v;
}

int bricks_along(int voxels) { return (voxels + 2 * vr::kPad + vr::kBrick - 1) / vr::kBrick; }

// Voxels [lo, hi) of an axis of `voxels` voxels that the bricks [b_lo, b_hi) store (apron included).
void voxels_of_bricks(int voxels, int b_lo, int b_hi, int &lo, int &hi)
{
    lo = std::max(0, b_lo * vr::kBrick - vr::kPad);
    hi = std::min(voxels, b_hi * vr::kBrick - vr::kPad + 1);
    if (hi < lo) hi = lo;
}

// Re-tiles a linear device volume into bricks: one pass over d_raw, which is only read (big-endian
// values are swapped on the way).  Frees nothing of the caller's.  With a brick box (b_lo, b_hi not
// null) only those bricks are built and d_raw holds just the voxels they store.  The old volume is given
// up before the new one is built (the two need not fit side by side); the data maximum is known only
// after the pass, so an upload that fails -- an allocation, a max_value below the data maximum, a volume
// of zeros -- leaves the context without a volume (vr_render then answers VR_ERR_STATE), never with a
// half-built one.
template <typename T>
int ingest_device(vr_ctx *ctx, const T *d_raw, int nx, int ny, int nz, float max_value, bool big_endian, const int *b_lo = nullptr,
                  const int *b_hi = nullptr)
{
    const int dims[3] = {nx, ny, nz};
    int lo[3] = {0, 0, 0}, hi[3] = {bricks_along(nx), bricks_along(ny), bricks_along(nz)};
    const bool sub = b_lo != nullptr;
    if (sub)
        for (int k = 0; k < 3; k++) {
            if (b_lo[k] < 0 || b_hi[k] > hi[k] || b_lo[k] >= b_hi[k]) return fail(ctx, VR_ERR_ARG, "brick box outside the volume's brick grid");
            lo[k] = b_lo[k];
        }
    if (sub)
        for (int k = 0; k < 3; k++) hi[k] = b_hi[k];
    if (sub && !(max_value > 0.0f)) return fail(ctx, VR_ERR_ARG, "a sub-volume needs the whole volume's maximum as max_value");
    if (max_value > 65535.0f) return fail(ctx, VR_ERR_ARG, "max_value above 65535");
    vr::BrickSource src;
    int raw_ext[3];
    for (int k = 0; k < 3; k++) {
        int v_lo, v_hi;
        voxels_of_bricks(dims[k], lo[k], hi[k], v_lo, v_hi);
        src.b0[k] = lo[k];
        src.v0[k] = v_lo;
        raw_ext[k] = v_hi - v_lo;
    }
    src.rx = raw_ext[0];
    src.ry = raw_ext[1];
    src.rz = raw_ext[2];
    const size_t count = (size_t)raw_ext[0] * raw_ext[1] * raw_ext[2];
    if (count > 0 && !d_raw) return fail(ctx, VR_ERR_ARG, "null argument");
    if (sub && ctx->shading) return fail(ctx, VR_ERR_STATE, "shading is on: a sub-volume's apron does not hold the neighbours the gradient reads");
    const size_t brick_count = (size_t)(hi[0] - lo[0]) * (size_t)(hi[1] - lo[1]) * (size_t)(hi[2] - lo[2]);
    if (brick_count >= (1ull << 31)) return fail(ctx, VR_ERR_ARG, "volume too large");
    // From here on the old volume is gone.  Its buffers are kept when the new volume has the same brick grid and
    // voxel type (a time series, a re-upload): allocating and releasing 26 GB costs a hundred times what the pass
    // over the voxels does.
    const bool reuse = ctx->bricks && ctx->dtype == (int)sizeof(T) && ctx->brick_count == brick_count && ctx->nbx == hi[0] - lo[0] &&
                       ctx->nby == hi[1] - lo[1] && ctx->nbz == hi[2] - lo[2] && ctx->fail_alloc_in == 0;
    if (reuse) drain(ctx);   // frames still being finished read the old voxels
    else release_volume(ctx);
    ctx->nx = nx; ctx->ny = ny; ctx->nz = nz;
    ctx->nbx = hi[0] - lo[0];
    ctx->nby = hi[1] - lo[1];
    ctx->nbz = hi[2] - lo[2];
    ctx->sub = sub;
    for (int k = 0; k < 3; k++) ctx->brick_lo[k] = lo[k];
    ctx->brick_count = brick_count;
    ctx->dtype = (int)sizeof(T);
    hipError_t err = hipSuccess;
    if (!reuse) {
        err = volume_alloc(ctx, &ctx->bricks, brick_count * vr::kBrickElems * sizeof(T));
        if (err == hipSuccess) err = volume_alloc(ctx, &ctx->cells, brick_count * sizeof(uint32_t));
        if (err == hipSuccess) err = volume_alloc(ctx, &ctx->touched, ((brick_count + 31) / 32) * sizeof(uint32_t));
        for (int l = 0; l < vr::kLevels; l++) {
            ctx->lx[l] = l == 0 ? ctx->nbx : (ctx->lx[l - 1] + 3) / 4;
            ctx->ly[l] = l == 0 ? ctx->nby : (ctx->ly[l - 1] + 3) / 4;
            ctx->lz[l] = l == 0 ? ctx->nbz : (ctx->lz[l - 1] + 3) / 4;
            const size_t cubes = (size_t)ctx->lx[l] * ctx->ly[l] * ctx->lz[l];
            if (err == hipSuccess) err = volume_alloc(ctx, &ctx->empty[l], ((cubes + 31) / 32) * sizeof(uint32_t));
        }
        if (err == hipSuccess) err = volume_alloc(ctx, &ctx->micro, brick_count * sizeof(unsigned long long));
        if (err == hipSuccess) err = volume_alloc(ctx, &ctx->skip, ((brick_count + 15) / 16) * sizeof(uint32_t));
        if (err == hipSuccess) err = volume_alloc(ctx, &ctx->occ_list, brick_count * sizeof(uint32_t));
    }
    uint32_t data_max = 0;
    if (err == hipSuccess) err = hipMemsetAsync(ctx->d_data_max, 0, sizeof(uint32_t), ctx->stream);
    if (err == hipSuccess) {
        const int strips = (ctx->nbx + vr::kStrip - 1) / vr::kStrip;
        // a workgroup walks kIngestLayers brick layers of its strip, fewer when that leaves the chip short of workgroups
        const size_t columns = (size_t)strips * (size_t)ctx->nby;
        int z_steps = vr::kIngestLayers;
        // test hook: small volumes walk several layers too (by themselves they take one per workgroup)
        const char *forced = std::getenv("VR_TEST_INGEST_LAYERS");
        if (forced && std::atoi(forced) > 0) z_steps = std::atoi(forced);
        while (!forced && z_steps > 1 && columns * (size_t)((ctx->nbz + z_steps - 1) / z_steps) < 4096) z_steps /= 2;
        const size_t blocks_total = columns * (size_t)((ctx->nbz + z_steps - 1) / z_steps);
        const unsigned grid = (unsigned)(((blocks_total + 7) / 8) * 8);
        // rows of raw start on multiples of 8 voxels when its row length is one and the buffer itself is aligned
        const int vec_ok = (src.rx % 8 == 0) && (reinterpret_cast<uintptr_t>(d_raw) % (8 * sizeof(T)) == 0);
        // where the strips' first stored voxel sits in its group of 8 raw voxels: bricks start two voxels before a
        // multiple of 8, raw at voxel 0 of the volume or at a box's first stored voxel
        const int xo = (lo[0] * vr::kBrick - vr::kPad - src.v0[0]) & 7;
        auto launch = [&](auto kernel) {
            hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, ctx->stream, d_raw, (T *)ctx->bricks, ctx->cells, ctx->nbx, ctx->nby,
                               ctx->nbz, src, strips, z_steps, (uint32_t)blocks_total);
        };
        if (xo == 6) {
            if (big_endian) vec_ok ? launch(vr::k_ingest_bricks<T, true, 6, true>) : launch(vr::k_ingest_bricks<T, true, 6, false>);
            else vec_ok ? launch(vr::k_ingest_bricks<T, false, 6, true>) : launch(vr::k_ingest_bricks<T, false, 6, false>);
        } else if (xo == 0) {
            if (big_endian) vec_ok ? launch(vr::k_ingest_bricks<T, true, 0, true>) : launch(vr::k_ingest_bricks<T, true, 0, false>);
            else vec_ok ? launch(vr::k_ingest_bricks<T, false, 0, true>) : launch(vr::k_ingest_bricks<T, false, 0, false>);
        } else {
            err = hipErrorInvalidValue;   // cannot happen: voxels_of_bricks yields one of the two
        }
        hipLaunchKernelGGL(vr::k_cells_max, dim3((unsigned)std::min<size_t>((brick_count + 255) / 256, 1024)), dim3(256), 0, ctx->stream,
                           ctx->cells, (uint32_t)brick_count, ctx->d_data_max);
        if (err == hipSuccess) err = hipGetLastError();
    }
    if (err == hipSuccess) err = hipMemcpyAsync(&data_max, ctx->d_data_max, sizeof data_max, hipMemcpyDeviceToHost, ctx->stream);
    if (err == hipSuccess) err = hipStreamSynchronize(ctx->stream);
    if (err != hipSuccess) {
        (void)hipGetLastError();   // an allocation failure is not sticky; clear it for the calls that follow
        release_volume(ctx);
        return fail(ctx, VR_ERR_HIP, std::string("volume upload: ") + hipGetErrorString(err));
    }
    float mx = (float)data_max;  // main.cpp:144, exact for 16-bit values
    if (max_value > 0.0f) {
        if (max_value < mx) {
            release_volume(ctx);
            return fail(ctx, VR_ERR_ARG, "max_value is below the data maximum");
        }
        mx = max_value;
    }
    if (!(mx > 0.0f)) {
        release_volume(ctx);
        return fail(ctx, VR_ERR_ARG, "volume is all zero: the reference would divide by zero");
    }
    ctx->max_value = mx;
    // v / max == fma(v, inv_hi, v * inv_lo) bit for bit for all integers 0 <= v <= max <= 65535
    ctx->inv_hi = 1.0f / mx;
    ctx->inv_lo = (float)(1.0 / (double)mx - (double)ctx->inv_hi);
    ctx->occupancy_dirty = true;
    ctx->occupancy_force = true;
    ctx->table_epoch++;
    ctx->fast_fallback = false;
    return VR_OK;
}

int check_dims(vr_ctx *ctx, int dtype, int nx, int ny, int nz)
{
    if (dtype != VR_DTYPE_U8 && dtype != VR_DTYPE_U16) return fail(ctx, VR_ERR_ARG, "dtype must be VR_DTYPE_U8 or VR_DTYPE_U16");
    if (nx < 1 || ny < 1 || nz < 1 || nx > 8192 || ny > 8192 || nz > 8192) return fail(ctx, VR_ERR_ARG, "volume dimensions out of range");
    return VR_OK;
}

// Brings the occupancy tables up to date with the current table, border colour and volume.  Nothing here
// waits for the GPU: k_lut_derive compares the table with the one the occupancy tables were built for and
// leaves the verdict in device memory, where the kernels enqueued behind it read it; when the set of
// transparent entries is what it was (a colour changed, an alpha moved between non-zero values, lighting
// was switched) they return at once.
int refresh_occupancy(vr_ctx *ctx)
{
    if (!ctx->occupancy_dirty) return VR_OK;
    drain(ctx);   // frames still being finished read the tables rebuilt here
    const float4 border = make_float4(ctx->border[0], ctx->border[1], ctx->border[2], ctx->border[3]);
    hipLaunchKernelGGL(vr::k_lut_derive, dim3(1), dim3(256), 0, ctx->stream, reinterpret_cast<const float4 *>(ctx->lut), border,
                       ctx->shading ? vr::shade_bound(ctx->scene) : 0.0f, ctx->occupancy_force ? 1 : 0, ctx->lut_info, ctx->lut_info, ctx->occ,
                       ctx->d_occ_count);
    const unsigned bricks = (unsigned)ctx->brick_count;
    hipLaunchKernelGGL(vr::k_classify_bricks, dim3((bricks + 255) / 256), dim3(256), 0, ctx->stream, ctx->cells, bricks, ctx->inv_hi,
                       ctx->inv_lo, ctx->lut_info, ctx->empty[0]);
    for (int l = 1; l < vr::kLevels; l++) {
        const unsigned cubes = (unsigned)(ctx->lx[l] * ctx->ly[l] * ctx->lz[l]);
        hipLaunchKernelGGL(vr::k_classify_up, dim3((cubes + 255) / 256), dim3(256), 0, ctx->stream, ctx->lut_info, ctx->empty[l - 1],
                           ctx->lx[l - 1], ctx->ly[l - 1], ctx->lz[l - 1], ctx->lx[l], ctx->ly[l], ctx->lz[l], ctx->empty[l]);
    }
    hipLaunchKernelGGL(vr::k_occupied_bounds, dim3(1024), dim3(256), 0, ctx->stream, ctx->lut_info, ctx->empty[0], ctx->lx[0], ctx->ly[0],
                       ctx->lz[0], ctx->occ);
    const unsigned skip_words = (bricks + 15) / 16;
    hipLaunchKernelGGL(vr::k_pack_skip, dim3((skip_words + 255) / 256), dim3(256), 0, ctx->stream, ctx->lut_info, ctx->empty[0],
                       ctx->empty[1], ctx->empty[2], ctx->lx[0], ctx->ly[0], ctx->lz[0], ctx->lx[1], ctx->ly[1], ctx->lx[2], ctx->ly[2],
                       ctx->skip);
    const unsigned micro_blocks = (unsigned)std::min<size_t>((ctx->brick_count + 255) / 256, 1u << 14);   // a workgroup sifts 256 bricks at a time
    if (ctx->dtype == VR_DTYPE_U8)
        hipLaunchKernelGGL((vr::k_classify_micro<uint8_t>), dim3(micro_blocks), dim3(256), 0, ctx->stream, (const uint8_t *)ctx->bricks,
                           ctx->empty[0], ctx->cells, bricks, ctx->inv_hi, ctx->inv_lo, ctx->lut_info, ctx->micro);
    else
        hipLaunchKernelGGL((vr::k_classify_micro<uint16_t>), dim3(micro_blocks), dim3(256), 0, ctx->stream, (const uint16_t *)ctx->bricks,
                           ctx->empty[0], ctx->cells, bricks, ctx->inv_hi, ctx->inv_lo, ctx->lut_info, ctx->micro);
    const unsigned cubes4 = (unsigned)(((ctx->lx[0] + 3) / 4) * ((ctx->ly[0] + 3) / 4) * ((ctx->lz[0] + 3) / 4));
    hipLaunchKernelGGL(vr::k_list_occupied, dim3((cubes4 + vr::kListWaves - 1) / vr::kListWaves), dim3(64 * vr::kListWaves), 0, ctx->stream, ctx->lut_info, ctx->empty[0], ctx->empty[1],
                       ctx->lx[0], ctx->ly[0], ctx->lz[0], ctx->occ_list, ctx->d_occ_count);
    VR_HIP(ctx, hipGetLastError());
    VR_HIP(ctx, hipEventRecord(ctx->occ_ready, ctx->stream));
    ctx->occupancy_dirty = false;
    ctx->occupancy_force = false;
    return VR_OK;
}

// What the last finished fast frame reported (see vr_ctx::fast_fallback).  Reads host memory only.
void poll_probe(vr_ctx *ctx)
{
    const unsigned long long w = __atomic_load_n(ctx->probe_word, __ATOMIC_ACQUIRE);
    const uint32_t seq = (uint32_t)(w >> 32);
    if (seq == 0u || seq == ctx->probe_seen) return;
    ctx->probe_seen = seq;
    const vr_ctx::ProbeMeta &m = ctx->probe_meta[seq & 7u];
    if (m.seq != seq || m.epoch != ctx->table_epoch || !m.eligible) return;
    const bool translucent = (size_t)(w & 0xffffffffull) * vr_ctx::kFallbackShare > m.pixels;
    if (translucent && !ctx->fast_fallback) ctx->fallback_frames = 0;
    ctx->fast_fallback = translucent;
}

// Number of blocks k_splat_bricks is launched with: enough for the longest list of surface bricks the
// volume can have, capped (a block takes several runs of the list in turn).
unsigned splat_blocks(const vr_ctx *ctx) { return (unsigned)std::min<size_t>((ctx->brick_count + 255) / 256, 4096); }

// Sort-first split of `rows` image rows into 16-row bands dealt round-robin to `parts` parts.
int bands_of_part(int rows, int part, int parts)
{
    const int bands = (rows + 15) / 16;
    return bands > part ? (bands - part + parts - 1) / parts : 0;
}

int rows_of_part(int rows, int part, int parts)
{
    const int bands = (rows + 15) / 16, mine = bands_of_part(rows, part, parts);
    if (mine == 0) return 0;
    const int last_band = part + (mine - 1) * parts;                       // global index of my last band
    const int last_rows = last_band == bands - 1 ? rows - 16 * (bands - 1) : 16;  // the frame's last band may be short
    return (mine - 1) * 16 + last_rows;
}

struct FrameBuffers {
    vr::Counters *counters;
    vr::ReplayRay *queue;
    vr::LongRay *long_queue;
    uint2 *tiles;
    hipStream_t tail;        // stream of the kernels that finish the frame
    hipEvent_t main_done;    // recorded after the lane march when the tail runs on another stream, else null
    hipStream_t prep;        // stream the depth ranges are built on when it is not the main one, else null
    hipEvent_t ranges_done;  // recorded on `prep` after the ranges are built
    hipEvent_t ranges_free;  // to wait for on `prep` before the range buffer is overwritten, or null
    bool may_order;          // the frame may use (and allocate) the context's tile-order lists
    bool clear_after;        // clear the counters behind the frame's last kernel
    int set;                 // which of the two sets of per-frame buffers this is
    uint32_t seq;            // the frame's sequence number, for its report to the host (k_finish_frame)
};

// The lane budget when neither vr_tune nor the environment has set one.  The lane march is one wave per
// 8x8 tile.  A 4K frame gives each SIMD over a hundred waves in turn and long lane marches keep its
// vector pipes busy; a small frame (or a rank's share of a sort-first split) has a few waves per SIMD,
// the march is then a chain of dependent memory round trips, and handing rays over early lets the
// wave-per-ray kernel, which takes 64 to 128 taps per round trip, finish them side by side.  Measured
// on MI355X: 512x512 32 (0.19 ms per frame against 0.47 with 128), 1080p 64, 4K 128.
int lane_budget_for(const vr_ctx *ctx, int tile_waves)
{
    if (ctx->lane_budget > 0) return ctx->lane_budget;
    if (tile_waves < 8192) return 32;
    if (tile_waves < 65536) return 64;
    return vr::kLaneBudget;
}

template <typename T, bool STATS, bool SHADE>
void launch_frame_as(vr_ctx *ctx, const vr::FrameParams &fp, uint32_t *d_out, const FrameBuffers &fb)
{
    // one wave per 8x8 tile; a band of 16 rows is two tile rows
    const int tile_waves = ((fp.x1 - fp.x0 + 7) / 8) * 2 * bands_of_part(fp.y1 - fp.y0, fp.part, fp.parts);
    const int waves_per_block = ctx->waves_per_block > 0 ? ctx->waves_per_block : (tile_waves >= 65536 ? 2 : 1);
    const dim3 grid((unsigned)((tile_waves + waves_per_block - 1) / waves_per_block)), block(64 * waves_per_block);
    const vr::VolumeView vol = view_of(ctx);
    const float4 border = make_float4(ctx->border[0], ctx->border[1], ctx->border[2], ctx->border[3]);
    const float4 *lut = reinterpret_cast<const float4 *>(ctx->lut);
    hipStream_t s = ctx->stream;
    const int slot = (int)(ctx->frames % (uint64_t)vr_ctx::kRing);
    ctx->frames++;
    ctx->slot_frame[ctx->lut_slot] = ctx->frames;
    (void)hipEventRecord(ctx->ring_a[slot], s);
    const int mode = (ctx->mode == VR_MODE_FAST && ctx->fast_fallback) ? VR_MODE_REFERENCE_SKIP : ctx->mode;
    if (mode == VR_MODE_REFERENCE_ORDER) {
        hipLaunchKernelGGL((vr::k_march_exact<T, false, STATS, SHADE>), grid, block, 0, s, vol, fp, lut, border, d_out, fb.counters, ctx->touched);
        (void)hipEventRecord(ctx->ring_b[slot], s);
    } else if (mode == VR_MODE_REFERENCE_SKIP) {
        hipLaunchKernelGGL((vr::k_march_exact<T, true, STATS, SHADE>), grid, block, 0, s, vol, fp, lut, border, d_out, fb.counters, ctx->touched);
        (void)hipEventRecord(ctx->ring_b[slot], s);
    } else {
        // depth range of the occupied bricks per tile of the frame, then the front-to-back march, then the
        // rays it could not finish or decide, one wave per ray
        const int tile_px = 1 << vr::kRangeShift;
        const int tiles_x = (fp.width + tile_px - 1) / tile_px, tiles_y = (fp.height + tile_px - 1) / tile_px;
        const uint2 *tiles = nullptr;
        if (ctx->tile_clip && fb.tiles) {
            hipStream_t ps = fb.prep ? fb.prep : s;
            if (fb.prep && fb.ranges_free) (void)hipStreamWaitEvent(ps, fb.ranges_free, 0);
            if (fb.prep) (void)hipStreamWaitEvent(ps, ctx->occ_ready, 0);   // the list of surface bricks may just have been rebuilt
            (void)hipMemsetAsync(fb.tiles, 0, ((size_t)tiles_x * tiles_y + 1) * sizeof(uint2), ps);   // + the word that switches the ranges off
            hipLaunchKernelGGL(vr::k_splat_bricks, dim3(splat_blocks(ctx)), dim3(256), 0, ps, vol, fp, ctx->occ_list, ctx->d_occ_count,
                               fb.tiles, tiles_x, tiles_y, fb.counters);
            if (fb.prep) {
                (void)hipEventRecord(fb.ranges_done, ps);
                (void)hipStreamWaitEvent(s, fb.ranges_done, 0);
            }
            tiles = fb.tiles;
            (void)hipEventRecord(ctx->ring_a[slot], s);   // the ring times the march kernel alone
        }
        const dim3 replay_grid((unsigned)ctx->tail_blocks);  // 4 waves each, one ray per wave and pass
        // launch order of the tiles: by the time they took in the previous frame, when that frame was over the
        // same tiles.  Worth it once a launch is several times the waves the chip holds at once.
        const uint32_t *order_in = nullptr;
        uint32_t *order_out = nullptr, *ticks = nullptr;
        if (!STATS && fb.may_order && ctx->tile_order && tile_waves >= 8192) {
            if ((size_t)tile_waves > ctx->order_cap) {
                for (int k = 0; k < vr_ctx::kSets; k++) {
                    if (ctx->order_lists[k]) (void)dev_free(ctx->order_lists[k]);
                    if (ctx->tile_ticks[k]) (void)dev_free(ctx->tile_ticks[k]);
                    ctx->order_lists[k] = ctx->tile_ticks[k] = nullptr;
                    ctx->order_valid[k] = false;
                }
                ctx->order_cap = 0;
                const size_t words = (size_t)vr::kOrderLists + (size_t)vr::kOrderBuckets * (size_t)tile_waves;
                bool ok = true;
                for (int k = 0; k < vr_ctx::kSets && ok; k++)
                    ok = dev_alloc(&ctx->order_lists[k], words * sizeof(uint32_t)) == hipSuccess &&
                         dev_alloc(&ctx->tile_ticks[k], (size_t)tile_waves * sizeof(uint32_t)) == hipSuccess;
                if (ok) ctx->order_cap = (size_t)tile_waves;
                else (void)hipGetLastError();   // no lists: launch order
            }
            if (ctx->order_cap > 0) {
                const int key[7] = {fp.x0, fp.y0, fp.x1, fp.y1, fp.part, fp.parts, tile_waves};
                if (ctx->order_valid[fb.set] && std::memcmp(key, ctx->order_key[fb.set], sizeof key) == 0) order_in = ctx->order_lists[fb.set];
                order_out = ctx->order_lists[fb.set];
                ticks = ctx->tile_ticks[fb.set];
                std::memcpy(ctx->order_key[fb.set], key, sizeof key);
                ctx->order_valid[fb.set] = true;
            }
        }
        hipLaunchKernelGGL((vr::k_march_fast<T, STATS, SHADE>), grid, block, 0, s, vol, fp, lut, border, ctx->lut_info, d_out, fb.counters, ctx->touched, fb.queue, fb.long_queue, lane_budget_for(ctx, tile_waves), tiles, tiles_x,
                           order_in, ticks, (uint32_t)ctx->order_cap, (uint32_t)tile_waves);
        (void)hipEventRecord(ctx->ring_b[slot], s);
        if (STATS)
            hipLaunchKernelGGL(vr::k_count_bits, dim3(1024), dim3(256), 0, s, ctx->touched, (ctx->brick_count + 31) / 32, &fb.counters->main_bricks);
        hipStream_t ts = fb.tail;
        if (fb.main_done) {
            (void)hipEventRecord(fb.main_done, s);
            (void)hipStreamWaitEvent(ts, fb.main_done, 0);
        }
        // a whole wave per handed-over ray for large frames (long rays of large volumes), two rays per wave for
        // small ones, where most rays are handed over with few taps left (VR_LONG_GROUP=32|64 overrides)
        const bool two_per_wave = ctx->long_group ? ctx->long_group == 32 : tile_waves < 65536;
        if (two_per_wave)
            hipLaunchKernelGGL((vr::k_march_long<T, 32, STATS, SHADE>), replay_grid, dim3(256), 0, ts, vol, fp, lut, border, ctx->lut_info, d_out, fb.counters, ctx->touched, fb.queue, fb.long_queue);
        else
            hipLaunchKernelGGL((vr::k_march_long<T, 64, STATS, SHADE>), replay_grid, dim3(256), 0, ts, vol, fp, lut, border, ctx->lut_info, d_out, fb.counters, ctx->touched, fb.queue, fb.long_queue);
        hipLaunchKernelGGL((vr::k_replay_exact<T, STATS, SHADE>), replay_grid, dim3(256), 0, ts, vol, fp, lut, border, ctx->lut_info, d_out, fb.counters, ctx->touched, fb.queue, tiles, tiles_x);
        hipLaunchKernelGGL(vr::k_finish_frame, dim3(1), dim3(64), 0, ts, fb.counters, ctx->probe_dev, fb.seq, fb.clear_after ? 1 : 0);
        if (order_out) {
            // behind the frame's last kernel: the march before it has read these lists, and the next reader is
            // the march of the next frame on this set of buffers, which waits for this set's last kernels
            (void)hipMemsetAsync(order_out, 0, vr::kOrderBuckets * sizeof(uint32_t), ts);
            hipLaunchKernelGGL(vr::k_order_tiles, dim3((tile_waves + 1023) / 1024), dim3(1024), 0, ts, ticks, (uint32_t)tile_waves, order_out,
                               (uint32_t)ctx->order_cap);
        }
    }
}

template <typename T>
void launch_frame(vr_ctx *ctx, const vr::FrameParams &fp, uint32_t *d_out, bool stats, const FrameBuffers &fb)
{
    if (ctx->shading) {
        if (stats) launch_frame_as<T, true, true>(ctx, fp, d_out, fb);
        else launch_frame_as<T, false, true>(ctx, fp, d_out, fb);
    } else {
        if (stats) launch_frame_as<T, true, false>(ctx, fp, d_out, fb);
        else launch_frame_as<T, false, false>(ctx, fp, d_out, fb);
    }
}

}  // namespace

extern "C" {

int vr_create(int device, vr_ctx **out)
{
    if (!out) return VR_ERR_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return VR_ERR_HIP;
    vr_ctx *ctx = new vr_ctx();
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess ||
        dev_alloc(&ctx->lut_store, (size_t)vr_ctx::kLutSlots * 1024 * sizeof(float)) != hipSuccess ||
        dev_alloc(&ctx->counters, sizeof(vr::Counters)) != hipSuccess ||
        dev_alloc(&ctx->info_store, (size_t)vr_ctx::kLutSlots * sizeof(vr::LutInfo)) != hipSuccess ||
        dev_alloc(&ctx->occ, 6 * sizeof(int)) != hipSuccess ||
        dev_alloc(&ctx->d_occ_count, sizeof(uint32_t)) != hipSuccess ||
        dev_alloc(&ctx->d_data_max, sizeof(uint32_t)) != hipSuccess ||
        hipMemset(ctx->d_occ_count, 0, sizeof(uint32_t)) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->occ_ready, hipEventDisableTiming) != hipSuccess ||
        hipEventRecord(ctx->occ_ready, ctx->stream) != hipSuccess ||
        hipHostMalloc((void **)&ctx->probe_word, 64, hipHostMallocDefault) != hipSuccess ||
        hipHostGetDevicePointer((void **)&ctx->probe_dev, ctx->probe_word, 0) != hipSuccess ||
        hipMemset(ctx->lut_store, 0, (size_t)vr_ctx::kLutSlots * 1024 * sizeof(float)) != hipSuccess ||  // main.cpp:110: zero-initialised table
        hipMemset(ctx->info_store, 0, (size_t)vr_ctx::kLutSlots * sizeof(vr::LutInfo)) != hipSuccess ||
        hipMemset(ctx->counters, 0, sizeof(vr::Counters)) != hipSuccess ||
        hipMemcpyToSymbol(HIP_SYMBOL(vr::d_phantom), vr::PHANTOM, sizeof(vr::PHANTOM)) != hipSuccess) {
        vr_destroy(ctx);
        return VR_ERR_HIP;
    }
    *ctx->probe_word = 0ull;
    ctx->lut = ctx->lut_store;
    ctx->lut_info = ctx->info_store;
    if (const char *e = std::getenv("VR_TEST_FAIL_ALLOC")) ctx->fail_alloc_in = std::max(0, std::atoi(e));
    if (const char *e = std::getenv("VR_TILE_CLIP")) ctx->tile_clip = std::atoi(e) != 0;
    if (const char *e = std::getenv("VR_TILE_ORDER")) {
        ctx->tile_order = std::atoi(e) != 0;
        ctx->tile_order_piped = std::atoi(e) == 2;
    }
    if (const char *e = std::getenv("VR_LANE_BUDGET")) ctx->lane_budget = std::max(1, std::atoi(e));
    if (const char *e = std::getenv("VR_PARTIAL_BUDGET")) ctx->partial_budget = std::max(1, std::atoi(e));
    if (const char *e = std::getenv("VR_LONG_GROUP")) ctx->long_group = std::atoi(e) == 32 ? 32 : (std::atoi(e) == 64 ? 64 : 0);
    if (const char *e = std::getenv("VR_TAIL_BLOCKS")) ctx->tail_blocks = std::max(64, std::min(65536, std::atoi(e))) & ~31;   // a multiple of 32: for_queue_entries
    if (const char *e = std::getenv("VR_WAVES_PER_BLOCK")) ctx->waves_per_block = std::max(1, std::min(4, std::atoi(e)));
    if (device < static_scratch::kMaxDevices && !g_scratch_ptr[device]) {
        // this GPU's copy of the static scratch array (hipSetDevice(device) above made the GPU current)
        unsigned char *ptr = nullptr;
        if (hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch)) != hipSuccess || !ptr) {
            vr_destroy(ctx);
            return VR_ERR_HIP;
        }
        g_scratch_ptr[device] = ptr;
    }
    ctx->ring_a.resize(vr_ctx::kRing, nullptr);
    ctx->ring_b.resize(vr_ctx::kRing, nullptr);
    for (int i = 0; i < vr_ctx::kRing; i++)
        if (hipEventCreate(&ctx->ring_a[i]) != hipSuccess || hipEventCreate(&ctx->ring_b[i]) != hipSuccess) {
            vr_destroy(ctx);
            return VR_ERR_HIP;
        }
    *out = ctx;
    return VR_OK;
}

int vr_destroy(vr_ctx *ctx)
{
    if (!ctx) return VR_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->prep_stream) (void)hipStreamSynchronize(ctx->prep_stream);
    if (ctx->tail_stream) (void)hipStreamSynchronize(ctx->tail_stream);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (int k = 0; k < vr_ctx::kSets; k++) ctx->tail_busy[k] = false;
    release_volume(ctx);
    for (int k = 0; k < vr_ctx::kSets - 1; k++) {
        if (ctx->counters_alt[k]) (void)dev_free(ctx->counters_alt[k]);
        if (ctx->queue_alt[k]) (void)dev_free(ctx->queue_alt[k]);
        if (ctx->long_alt[k]) (void)dev_free(ctx->long_alt[k]);
    }
    for (int k = 0; k < vr_ctx::kRangeRing; k++) {
        if (ctx->tiles_ring[k]) (void)dev_free(ctx->tiles_ring[k]);
        if (ctx->ranges_done[k]) (void)hipEventDestroy(ctx->ranges_done[k]);
        if (ctx->ranges_free[k]) (void)hipEventDestroy(ctx->ranges_free[k]);
    }
    if (ctx->prep_stream) (void)hipStreamDestroy(ctx->prep_stream);
    if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
    for (vr_ctx::MappedSlot &m : ctx->mapped) {
        if (m.dev) (void)dev_free(m.dev);
        if (m.host) (void)hipHostFree(m.host);
        if (m.copied) (void)hipEventDestroy(m.copied);
    }
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    for (int k = 0; k < vr_ctx::kSets; k++) {
        if (ctx->main_done[k]) (void)hipEventDestroy(ctx->main_done[k]);
        if (ctx->tail_done[k]) (void)hipEventDestroy(ctx->tail_done[k]);
    }
    if (ctx->frame_done) (void)hipEventDestroy(ctx->frame_done);
    if (ctx->tail_stream) (void)hipStreamDestroy(ctx->tail_stream);
    if (ctx->lut_store) (void)dev_free(ctx->lut_store);
    if (ctx->d_pts) (void)dev_free(ctx->d_pts);
    if (ctx->counters) (void)dev_free(ctx->counters);
    if (ctx->frame) (void)dev_free(ctx->frame);
    if (ctx->partial) (void)dev_free(ctx->partial);
    if (ctx->queue) (void)dev_free(ctx->queue);
    if (ctx->long_queue) (void)dev_free(ctx->long_queue);
    if (ctx->info_store) (void)dev_free(ctx->info_store);
    if (ctx->occ) (void)dev_free(ctx->occ);
    if (ctx->d_occ_count) (void)dev_free(ctx->d_occ_count);
    if (ctx->d_data_max) (void)dev_free(ctx->d_data_max);
    if (ctx->occ_ready) (void)hipEventDestroy(ctx->occ_ready);
    if (ctx->probe_word) (void)hipHostFree(ctx->probe_word);
    if (ctx->tiles) (void)dev_free(ctx->tiles);
    for (int k = 0; k < vr_ctx::kSets; k++) {
        if (ctx->order_lists[k]) (void)dev_free(ctx->order_lists[k]);
        if (ctx->tile_ticks[k]) (void)dev_free(ctx->tile_ticks[k]);
    }
    for (hipEvent_t e : ctx->ring_a)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->ring_b)
        if (e) (void)hipEventDestroy(e);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return VR_OK;
}

const char *vr_last_error(const vr_ctx *ctx) { return ctx ? ctx->error.c_str() : "null context"; }

void *vr_stream(vr_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int vr_synchronize(vr_ctx *ctx)
{
    if (!ctx) return VR_ERR_ARG;
    if (int rc = bind(ctx)) return rc;
    drain(ctx);
    VR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VR_OK;
}

int vr_set_pipeline(vr_ctx *ctx, int frames_in_flight)
{
    if (!ctx) return VR_ERR_ARG;
    if (frames_in_flight != 1 && frames_in_flight != 2) return fail(ctx, VR_ERR_ARG, "frames in flight must be 1 or 2");
    if (int rc = bind(ctx)) return rc;
    drain(ctx);
    if (frames_in_flight == 2 && !ctx->tail_stream) {
        // the second stream, its events and the second set of counters exist only in contexts that ask for them
        for (int k = 0; k < vr_ctx::kSets; k++) {
            VR_HIP(ctx, hipEventCreateWithFlags(&ctx->main_done[k], hipEventDisableTiming));
            VR_HIP(ctx, hipEventCreateWithFlags(&ctx->tail_done[k], hipEventDisableTiming));
        }
        for (int k = 0; k < vr_ctx::kRangeRing; k++) {
            VR_HIP(ctx, hipEventCreateWithFlags(&ctx->ranges_done[k], hipEventDisableTiming));
            VR_HIP(ctx, hipEventCreateWithFlags(&ctx->ranges_free[k], hipEventDisableTiming));
        }
        VR_HIP(ctx, hipStreamCreateWithFlags(&ctx->prep_stream, hipStreamNonBlocking));
        for (int k = 0; k < vr_ctx::kSets - 1; k++) {
            VR_HIP(ctx, dev_alloc(&ctx->counters_alt[k], sizeof(vr::Counters)));
            VR_HIP(ctx, hipMemsetAsync(ctx->counters_alt[k], 0, sizeof(vr::Counters), ctx->stream));
        }
        // The finishing kernels are short chains of dependent steps and may be given their few waves first
        // (VR_TAIL_PRIORITY=1).  Opt-in: with several processes sharing one GPU a high-priority queue was
        // seen to stall the other processes' queues for good, so it is only meant for a GPU of one's own.
        const char *prio = std::getenv("VR_TAIL_PRIORITY");
        if (prio && std::atoi(prio) != 0) {
            int least = 0, greatest = 0;
            VR_HIP(ctx, hipDeviceGetStreamPriorityRange(&least, &greatest));
            VR_HIP(ctx, hipStreamCreateWithPriority(&ctx->tail_stream, hipStreamNonBlocking, greatest));
        } else {
            VR_HIP(ctx, hipStreamCreateWithFlags(&ctx->tail_stream, hipStreamNonBlocking));
        }
    }
    ctx->pipeline_depth = frames_in_flight;
    return VR_OK;
}

int vr_wait_frame(vr_ctx *ctx, void *stream)
{
    if (!ctx) return VR_ERR_ARG;
    if (int rc = bind(ctx)) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (!ctx->frame_done) VR_HIP(ctx, hipEventCreateWithFlags(&ctx->frame_done, hipEventDisableTiming));
    VR_HIP(ctx, hipEventRecord(ctx->frame_done, ctx->stream));
    VR_HIP(ctx, hipStreamWaitEvent(s, ctx->frame_done, 0));
    if (ctx->last_set >= 0 && ctx->tail_busy[ctx->last_set]) VR_HIP(ctx, hipStreamWaitEvent(s, ctx->tail_done[ctx->last_set], 0));
    return VR_OK;
}

int vr_upload_volume_device(vr_ctx *ctx, const void *d_voxels, int dtype, int nx, int ny, int nz, float max_value)
{
    if (!ctx || !d_voxels) return fail(ctx, VR_ERR_ARG, "null argument");
    if (int rc = check_dims(ctx, dtype, nx, ny, nz)) return rc;
    if (int rc = bind(ctx)) return rc;
    if (dtype == VR_DTYPE_U8) return ingest_device<uint8_t>(ctx, (const uint8_t *)d_voxels, nx, ny, nz, max_value, false);
    return ingest_device<uint16_t>(ctx, (const uint16_t *)d_voxels, nx, ny, nz, max_value, false);
}

int vr_upload_volume(vr_ctx *ctx, const void *voxels, int dtype, int nx, int ny, int nz, float max_value)
{
    if (!ctx || !voxels) return fail(ctx, VR_ERR_ARG, "null argument");
    if (int rc = check_dims(ctx, dtype, nx, ny, nz)) return rc;
    if (int rc = bind(ctx)) return rc;
    const size_t bytes = (size_t)nx * ny * nz * (size_t)dtype;
    void *d_raw = nullptr;
    VR_HIP(ctx, dev_alloc(&d_raw, bytes));
    hipError_t e = hipMemcpyAsync(d_raw, voxels, bytes, hipMemcpyHostToDevice, ctx->stream);
    int rc = VR_OK;
    if (e != hipSuccess) rc = fail(ctx, VR_ERR_HIP, std::string("hipMemcpyAsync: ") + hipGetErrorString(e));
    if (rc == VR_OK) rc = vr_upload_volume_device(ctx, d_raw, dtype, nx, ny, nz, max_value);
    (void)dev_free(d_raw);
    return rc;
}

int vr_load_volume_slices(vr_ctx *ctx, const char *directory)
{
    if (!ctx || !directory) return fail(ctx, VR_ERR_ARG, "null argument");
    if (int rc = bind(ctx)) return rc;
    namespace fs = std::filesystem;
    std::vector<std::pair<int, fs::path>> files;
    std::error_code ec;
    for (fs::directory_iterator it(directory, ec), end; !ec && it != end; it.increment(ec)) {
        // main.cpp:123-128: the integer after the extension dot, or the whole file name
        const fs::path &p = it->path();
        std::string key = p.has_extension() ? p.extension().string().substr(1) : p.filename().string();
        // std::stoi there: leading blanks, a sign and digits, anything after them ignored; a key without
        // digits (a hidden file, a name ending in a dot) or beyond int throws and ends the application
        errno = 0;
        char *rest = nullptr;
        const long number = std::strtol(key.c_str(), &rest, 10);
        if (rest == key.c_str() || errno == ERANGE || number < INT_MIN || number > INT_MAX)
            return fail(ctx, VR_ERR_IO, "slice file name without a slice number: " + p.string());
        files.emplace_back((int)number, p);
    }
    if (ec) return fail(ctx, VR_ERR_IO, std::string("cannot list ") + directory + ": " + ec.message());
    if (files.empty()) return fail(ctx, VR_ERR_IO, std::string("Unavailable volume data files for loading at ") + directory);  // main.cpp:158
    std::stable_sort(files.begin(), files.end(), [](const auto &a, const auto &b) { return a.first < b.first; });

    std::vector<uint16_t> raw;
    size_t first_count = 0;
    for (size_t f = 0; f < files.size(); f++) {
        std::ifstream in(files[f].second, std::ios::binary | std::ios::ate);
        if (!in) return fail(ctx, VR_ERR_IO, "Can't open file at " + files[f].second.string());  // main.cpp:134
        const size_t values = (size_t)in.tellg() / 2;  // a trailing odd byte is dropped, as main.cpp:138 does
        in.seekg(0);
        const size_t at = raw.size();
        raw.resize(at + values);
        in.read(reinterpret_cast<char *>(raw.data() + at), (std::streamsize)(values * 2));
        if (f == 0) first_count = values;
    }
    const int side = (int)std::sqrt((double)first_count);  // main.cpp:153
    const int slices = (int)files.size();
    if (side < 1 || (size_t)side * side * slices != raw.size())
        return fail(ctx, VR_ERR_IO, "slice files are not equally sized squares");
    uint16_t *d_raw = nullptr;
    VR_HIP(ctx, dev_alloc(&d_raw, raw.size() * 2));
    hipError_t e = hipMemcpyAsync(d_raw, raw.data(), raw.size() * 2, hipMemcpyHostToDevice, ctx->stream);
    int rc = VR_OK;
    if (e != hipSuccess) rc = fail(ctx, VR_ERR_HIP, std::string("hipMemcpyAsync: ") + hipGetErrorString(e));
    if (rc == VR_OK) rc = ingest_device<uint16_t>(ctx, d_raw, side, side, slices, 0.0f, true);
    (void)dev_free(d_raw);
    return rc;
}

int vr_volume_info(const vr_ctx *ctx, int *nx, int *ny, int *nz, int *dtype, float *max_value)
{
    if (!ctx) return VR_ERR_ARG;
    if (!ctx->bricks) return VR_ERR_STATE;
    if (nx) *nx = ctx->nx;
    if (ny) *ny = ctx->ny;
    if (nz) *nz = ctx->nz;
    if (dtype) *dtype = ctx->dtype;
    if (max_value) *max_value = ctx->max_value;
    return VR_OK;
}

int vr_read_volume_normalized(vr_ctx *ctx, float *out)
{
    if (!ctx || !out) return fail(ctx, VR_ERR_ARG, "null argument");
    if (!ctx->bricks) return fail(ctx, VR_ERR_STATE, "no volume uploaded");
    if (ctx->sub) return fail(ctx, VR_ERR_STATE, "the context holds a sub-volume");
    if (int rc = bind(ctx)) return rc;
    const size_t count = (size_t)ctx->nx * ctx->ny * ctx->nz;
    float *d = nullptr;
    VR_HIP(ctx, dev_alloc(&d, count * sizeof(float)));
    const int blocks = (int)std::min<size_t>((count + 255) / 256, 256 * 16);
    if (ctx->dtype == VR_DTYPE_U8)
        hipLaunchKernelGGL((vr::k_read_normalized<uint8_t>), dim3(blocks), dim3(256), 0, ctx->stream, view_of(ctx), d);
    else
        hipLaunchKernelGGL((vr::k_read_normalized<uint16_t>), dim3(blocks), dim3(256), 0, ctx->stream, view_of(ctx), d);
    hipError_t e = hipMemcpyAsync(out, d, count * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)dev_free(d);
    if (e != hipSuccess) return fail(ctx, VR_ERR_HIP, hipGetErrorString(e));
    return VR_OK;
}

int vr_histogram(vr_ctx *ctx, float out[100])
{
    if (!ctx || !out) return fail(ctx, VR_ERR_ARG, "null argument");
    if (!ctx->bricks) return fail(ctx, VR_ERR_STATE, "no volume uploaded");
    if (ctx->sub) return fail(ctx, VR_ERR_STATE, "the context holds a sub-volume");
    if (int rc = bind(ctx)) return rc;
    // main.cpp:166-181 on the sorted normalised values, which are runs of float(v)/float(max), v ascending:
    // lower_bound(x) = number of values < x, upper_bound(x) = number of values <= x, so bin i counts the raw
    // values from the first v with !(float(v)/max < lo) up to the first v with hi < float(v)/max
    const float mx = ctx->max_value;
    const int top = (int)mx;
    const size_t values = ctx->dtype == VR_DTYPE_U8 ? 256 : 65536;
    std::vector<uint8_t> bin_of(values, (uint8_t)vr::kHistBins);
    const float step = 1.0f / 100;                        // main.cpp:170
    for (int i = 0; i < vr::kHistBins; i++) {
        const float lo = step * (float)i;                  // main.cpp:174, float
        const double hi = (double)step * ((double)i + 0.9);   // main.cpp:175, double
        int a = 0, b = top + 1;
        while (a < b) { const int m = (a + b) / 2; if ((float)m / mx < lo) a = m + 1; else b = m; }
        const int first = a;
        a = 0; b = top + 1;
        while (a < b) { const int m = (a + b) / 2; if (!(hi < (double)((float)m / mx))) a = m + 1; else b = m; }
        for (int v = first; v < a && v < (int)values; v++) bin_of[(size_t)v] = (uint8_t)i;
    }
    uint8_t *d_bin = nullptr;
    unsigned long long *d_hist = nullptr;
    VR_HIP(ctx, dev_alloc(&d_bin, values));
    if (dev_alloc(&d_hist, vr::kHistBins * sizeof(unsigned long long)) != hipSuccess) {
        (void)dev_free(d_bin);
        return fail(ctx, VR_ERR_HIP, "out of device memory for the histogram");
    }
    hipError_t e = hipMemsetAsync(d_hist, 0, vr::kHistBins * sizeof(unsigned long long), ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_bin, bin_of.data(), values, hipMemcpyHostToDevice, ctx->stream);
    // one workgroup walks rows of bricks along x; two of them fit a compute unit's LDS
    const unsigned blocks = (unsigned)std::min<size_t>((size_t)ctx->nby * ctx->nbz, 2 * 256);
    if (e == hipSuccess) {
        if (ctx->dtype == VR_DTYPE_U8)
            hipLaunchKernelGGL((vr::k_bin_histogram<uint8_t>), dim3(blocks), dim3(vr::kHistThreads), 0, ctx->stream, view_of(ctx), d_bin, d_hist);
        else
            hipLaunchKernelGGL((vr::k_bin_histogram<uint16_t>), dim3(blocks), dim3(vr::kHistThreads), 0, ctx->stream, view_of(ctx), d_bin, d_hist);
        e = hipGetLastError();
    }
    unsigned long long counts[vr::kHistBins] = {};
    if (e == hipSuccess) e = hipMemcpyAsync(counts, d_hist, sizeof counts, hipMemcpyDeviceToHost, ctx->stream);
    const hipError_t waited = hipStreamSynchronize(ctx->stream);   // also before bin_of and counts go out of scope
    if (e == hipSuccess) e = waited;
    (void)dev_free(d_hist);
    (void)dev_free(d_bin);
    if (e != hipSuccess) return fail(ctx, VR_ERR_HIP, hipGetErrorString(e));

    size_t count_max = 0;
    for (int i = 0; i < vr::kHistBins; i++) {
        const float c = (float)(long long)counts[i];       // main.cpp:176
        out[i] = c;
        if (c > (float)count_max) count_max = (size_t)c;   // main.cpp:178
    }
    for (int i = 0; i < 100; i++) out[i] /= (float)count_max;   // main.cpp:180
    return VR_OK;
}

namespace {
// k_lut_update on the host, statement for statement (updateTransferFunction, main.cpp:643-686): the same IEEE
// operations in the same order, so the table is the device's bit for bit.  Returns false when a graph
// addresses a pixel outside the table (the device then flags the table and so must the caller).
bool host_lut_update(float *lut, const std::vector<vr::CtrlPt> &pts, const int offs[5])
{
    bool ok = true;
    for (int ch = 0; ch < 4; ch++) {
        const int begin = offs[ch], cnt = offs[ch + 1] - offs[ch];
        if (cnt < 1) continue;
        const float x_last = pts[(size_t)begin + cnt - 1].x;
        for (int j = 1; j < cnt; j++) {
            const vr::CtrlPt ga = pts[(size_t)begin + j - 1], gb = pts[(size_t)begin + j];
            const float xa = (ga.x / x_last) * 255.0f;
            const float xb = (gb.x / x_last) * 255.0f;
            const float dir = gb.intensity - ga.intensity;
            const float total = xb - xa + 1.0f;
            uint32_t count = 0;
            if (total >= 1.0f) count = total >= 4294967296.0f ? 0xFFFFFFFFu : (uint32_t)total;
            if (count > 1024u) {
                ok = false;
                count = 0;
            }
            for (uint32_t k = 0; k < count; k++) {
                const float val = ga.intensity + dir * ((float)k / (total - 1.0f));
                const float pixel = xa + (float)k;
                if (pixel >= 0.0f && pixel < 256.0f) lut[(int)pixel * 4 + ch] = val;
                else ok = false;
            }
        }
    }
    return ok;
}

// Per entry of the marches' LDS table (two border entries, the 256 of the table, two border entries): the entry
// is not transparent, by k_lut_derive's test.
void bad_entries(const float *lut, const float border[4], uint8_t *bad)
{
    for (int i = 0; i < vr::kLutLds; i++) {
        const float *v = (i >= 2 && i < 258) ? lut + (size_t)(i - 2) * 4 : border;
        const bool finite = std::isfinite(v[0]) && std::isfinite(v[1]) && std::isfinite(v[2]);
        bad[i] = (v[3] != 0.0f || !finite) ? 1 : 0;
    }
}
}  // namespace

int vr_set_transfer_function(vr_ctx *ctx, const vr_ctrl_pt *const points[4], const int counts[4], int reset_state)
{
    if (!ctx || !points || !counts) return fail(ctx, VR_ERR_ARG, "null argument");
    if (int rc = bind(ctx)) return rc;
    std::vector<vr::CtrlPt> packed;
    int offs[5] = {0, 0, 0, 0, 0};
    for (int c = 0; c < 4; c++) {
        if (counts[c] < 0 || counts[c] > 4096) return fail(ctx, VR_ERR_ARG, "control point count out of range");
        if (counts[c] > 0 && !points[c]) return fail(ctx, VR_ERR_ARG, "null control graph");
        for (int k = 0; k < counts[c]; k++) {
            const vr_ctrl_pt &p = points[c][k];
            // The reference keeps every graph sorted by x with x >= 0 (main.cpp:1009-1012, 1024);
            // anything else indexes outside the table there.
            if (!(p.x >= 0.0f) || !std::isfinite(p.x) || (k > 0 && p.x < points[c][k - 1].x))
                return fail(ctx, VR_ERR_ARG, "control points must have finite x >= 0 in ascending order");
            packed.push_back({p.x, p.intensity});
        }
        if (counts[c] > 0 && !(points[c][counts[c] - 1].x > 0.0f))
            return fail(ctx, VR_ERR_ARG, "the last control point needs x > 0 (x is divided by it, main.cpp:653)");
        offs[c + 1] = (int)packed.size();
    }
    const bool same = packed.size() == ctx->last_pts.size() && std::memcmp(offs, ctx->last_offs, sizeof offs) == 0 &&
                      (packed.empty() || std::memcmp(packed.data(), ctx->last_pts.data(), packed.size() * sizeof(vr::CtrlPt)) == 0);
    if (same && (!reset_state || ctx->table_from_zero)) return VR_OK;   // every bit of the table stays as it is
    ctx->table_from_zero = reset_state != 0;
    ctx->last_pts = packed;
    std::memcpy(ctx->last_offs, offs, sizeof offs);
    ctx->table_epoch++;
    ctx->fast_fallback = false;

    // The table on the host, to see whether the set of transparent entries moves: the occupancy tables depend
    // on the table through that set alone.
    if (reset_state) {
        std::memset(ctx->host_lut, 0, sizeof ctx->host_lut);
        ctx->host_lut_valid = true;
    }
    if (!host_lut_update(ctx->host_lut, packed, offs)) ctx->host_lut_valid = false;
    uint8_t bad[vr::kLutLds];
    bad_entries(ctx->host_lut, ctx->border, bad);
    const bool set_kept = ctx->host_lut_valid && ctx->built_bad_valid && !ctx->occupancy_dirty &&
                          std::memcmp(bad, ctx->built_bad, sizeof bad) == 0 && packed.size() <= (size_t)vr::kLutArgPoints;
    if (set_kept) {
        // Same set: the occupancy tables stay, and so do the frames in flight -- the table goes into the next
        // slot (a copy of this one, or zeros, updated there) together with what the marches derive from it.
        const int next = (ctx->lut_slot + 1) % vr_ctx::kLutSlots;
        if (ctx->slot_frame[next] != 0 && ctx->frames - ctx->slot_frame[next] < (uint64_t)vr_ctx::kSets)
            drain(ctx);   // a frame that read the slot may still be finished (several table calls between two frames)
        float *lut = ctx->lut_store + (size_t)next * 1024;
        vr::LutInfo *info = ctx->info_store + next;
        if (reset_state) VR_HIP(ctx, hipMemsetAsync(lut, 0, 1024 * sizeof(float), ctx->stream));
        else VR_HIP(ctx, hipMemcpyAsync(lut, ctx->lut, 1024 * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
        if (!packed.empty()) {
            vr::LutPointsArg arg;
            std::memcpy(arg.p, packed.data(), packed.size() * sizeof(vr::CtrlPt));
            hipLaunchKernelGGL((vr::k_lut_update<true>), dim3(1), dim3(256), 0, ctx->stream, lut, (const vr::CtrlPt *)nullptr, arg, offs[0],
                               offs[1], offs[2], offs[3], offs[4], ctx->counters);
        }
        const float4 border = make_float4(ctx->border[0], ctx->border[1], ctx->border[2], ctx->border[3]);
        hipLaunchKernelGGL(vr::k_lut_derive, dim3(1), dim3(256), 0, ctx->stream, reinterpret_cast<const float4 *>(lut), border,
                           ctx->shading ? vr::shade_bound(ctx->scene) : 0.0f, 0, ctx->lut_info, info, ctx->occ, ctx->d_occ_count);
        VR_HIP(ctx, hipGetLastError());
        ctx->lut = lut;
        ctx->lut_info = info;
        ctx->lut_slot = next;
        return VR_OK;
    }
    std::memcpy(ctx->built_bad, bad, sizeof bad);
    ctx->built_bad_valid = true;
    ctx->occupancy_dirty = true;
    drain(ctx);   // frames still being finished read the table and the occupancy tables
    if (reset_state) VR_HIP(ctx, hipMemsetAsync(ctx->lut, 0, 1024 * sizeof(float), ctx->stream));
    if (packed.empty()) return VR_OK;
    if (packed.size() <= (size_t)vr::kLutArgPoints) {
        // the points travel with the launch: no staging buffer, no copy, nothing to wait for
        vr::LutPointsArg arg;
        std::memcpy(arg.p, packed.data(), packed.size() * sizeof(vr::CtrlPt));
        hipLaunchKernelGGL((vr::k_lut_update<true>), dim3(1), dim3(256), 0, ctx->stream, ctx->lut, (const vr::CtrlPt *)nullptr, arg, offs[0],
                           offs[1], offs[2], offs[3], offs[4], ctx->counters);
        VR_HIP(ctx, hipGetLastError());
        return VR_OK;
    }
    if (packed.size() > ctx->d_pts_cap) {
        if (ctx->d_pts) (void)dev_free(ctx->d_pts);
        ctx->d_pts = nullptr;
        ctx->d_pts_cap = 0;
        const size_t cap = std::max<size_t>(1024, packed.size() * 2);
        VR_HIP(ctx, dev_alloc(&ctx->d_pts, cap * sizeof(vr::CtrlPt)));
        ctx->d_pts_cap = cap;
    }
    // synchronous on purpose: `packed` lives on this stack frame
    VR_HIP(ctx, hipMemcpyAsync(ctx->d_pts, packed.data(), packed.size() * sizeof(vr::CtrlPt), hipMemcpyHostToDevice, ctx->stream));
    VR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    hipLaunchKernelGGL((vr::k_lut_update<false>), dim3(1), dim3(256), 0, ctx->stream, ctx->lut, ctx->d_pts, vr::LutPointsArg{}, offs[0], offs[1],
                       offs[2], offs[3], offs[4], ctx->counters);
    VR_HIP(ctx, hipGetLastError());
    return VR_OK;
}

int vr_get_lut(vr_ctx *ctx, float out[1024])
{
    if (!ctx || !out) return fail(ctx, VR_ERR_ARG, "null argument");
    if (int rc = bind(ctx)) return rc;
    vr::Counters c;
    VR_HIP(ctx, hipMemcpyAsync(out, ctx->lut, 1024 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    VR_HIP(ctx, hipMemcpyAsync(&c, ctx->counters, sizeof c, hipMemcpyDeviceToHost, ctx->stream));
    VR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (c.lut_error) return fail(ctx, VR_ERR_ARG, "a control graph addressed a pixel outside the 256-entry table");
    // the table the host builds beside the device's (it decides whether a change concerns the occupancy tables)
    // has to be the same table; entries that are not numbers on both sides count as equal
    if (ctx->host_lut_valid)
        for (int i = 0; i < 1024; i++)
            if (std::memcmp(&out[i], &ctx->host_lut[i], sizeof(float)) != 0 && !(std::isnan(out[i]) && std::isnan(ctx->host_lut[i])))
                return fail(ctx, VR_ERR_STATE, "the host's copy of the table differs from the device's at entry " + std::to_string(i / 4));
    return VR_OK;
}

float vr_control_point_intensity(float y, float child_height) { return 1.0f - (y / (child_height - 1)); }  // main.cpp:952

int vr_set_camera(vr_ctx *ctx, float horizontal_rad, float vertical_rad)
{
    if (!ctx) return VR_ERR_ARG;
    if (!std::isfinite(horizontal_rad) || !std::isfinite(vertical_rad)) return fail(ctx, VR_ERR_ARG, "camera angles must be finite");
    ctx->raw_camera = false;
    ctx->h_rad = horizontal_rad;
    ctx->v_rad = vertical_rad;
    return VR_OK;
}

int vr_set_camera_raw(vr_ctx *ctx, const float location[4], const float to_world[16])
{
    if (!ctx || !location || !to_world) return fail(ctx, VR_ERR_ARG, "null argument");
    ctx->raw_camera = true;
    std::memcpy(ctx->raw_loc, location, sizeof ctx->raw_loc);
    std::memcpy(ctx->raw_m, to_world, sizeof ctx->raw_m);
    return VR_OK;
}

int vr_get_camera(vr_ctx *ctx, int width, int height, float view_pyramid[20], float to_world[16])
{
    if (!ctx || !view_pyramid || !to_world) return fail(ctx, VR_ERR_ARG, "null argument");
    if (width < 2 || height < 2) return fail(ctx, VR_ERR_ARG, "image must be at least 2x2 (1/(extent-1), main.cpp:772-773)");
    if (ctx->raw_camera) {
        std::memset(view_pyramid, 0, 20 * sizeof(float));
        view_pyramid[0] = vr::cam::kPyramidHeight;
        std::memcpy(view_pyramid + 4, ctx->raw_loc, 4 * sizeof(float));
        std::memcpy(to_world, ctx->raw_m, 16 * sizeof(float));
        return VR_OK;
    }
    vr::cam::pixel_transform(ctx->h_rad, ctx->v_rad, (uint32_t)width, (uint32_t)height, view_pyramid, to_world);
    return VR_OK;
}

int vr_camera_uniforms(float horizontal_rad, float vertical_rad, int width, int height, float view_pyramid[20],
                       float to_world[16])
{
    if (!view_pyramid || !to_world || width < 2 || height < 2) return VR_ERR_ARG;
    vr::cam::pixel_transform(horizontal_rad, vertical_rad, (uint32_t)width, (uint32_t)height, view_pyramid, to_world);
    return VR_OK;
}

int vr_set_sampling(vr_ctx *ctx, int samples_per_ray, const float tf_border_rgba[4])
{
    if (!ctx) return VR_ERR_ARG;
    if (samples_per_ray < 0 || samples_per_ray > (1 << 20)) return fail(ctx, VR_ERR_ARG, "samples_per_ray out of range");
    if (ctx->samples != samples_per_ray) {
        ctx->table_epoch++;
        ctx->fast_fallback = false;
    }
    ctx->samples = samples_per_ray;
    if (tf_border_rgba && std::memcmp(ctx->border, tf_border_rgba, sizeof ctx->border) != 0) {
        std::memcpy(ctx->border, tf_border_rgba, sizeof ctx->border);
        ctx->occupancy_dirty = true;
        ctx->table_epoch++;
        ctx->fast_fallback = false;
    }
    return VR_OK;
}

int vr_set_shading(vr_ctx *ctx, int enabled, const float scene[4])
{
    if (!ctx) return VR_ERR_ARG;
    float next[4] = {1.0f, 1.0f, 1.0f, 100.0f};   // main.cpp:90-98
    if (scene) {
        std::memcpy(next, scene, sizeof next);
        // the window's sliders keep the first three in [0,1] and the exponent in [1,128] (main.cpp:1212-1215)
        for (int k = 0; k < 3; k++)
            if (!(next[k] >= 0.0f && next[k] <= 1.0f)) return fail(ctx, VR_ERR_ARG, "ambient, diffuse and specular must lie in [0,1]");
        if (!(next[3] >= 1.0f && next[3] <= 128.0f)) return fail(ctx, VR_ERR_ARG, "shininess must lie in [1,128]");
    }
    if (enabled && ctx->sub) return fail(ctx, VR_ERR_STATE, "the context holds a sub-volume: its apron does not hold the neighbours the gradient reads");
    const bool on = enabled != 0;
    if (on != ctx->shading || std::memcmp(next, ctx->scene, sizeof next) != 0) {
        if (int rc = bind(ctx)) return rc;
        drain(ctx);
        ctx->shading = on;
        std::memcpy(ctx->scene, next, sizeof next);
        ctx->occupancy_dirty = true;   // the colour range the early decisions rely on includes the illumination
        ctx->table_epoch++;
        ctx->fast_fallback = false;
    }
    return VR_OK;
}

int vr_set_mode(vr_ctx *ctx, int mode)
{
    if (!ctx) return VR_ERR_ARG;
    if (mode != VR_MODE_REFERENCE_ORDER && mode != VR_MODE_FAST && mode != VR_MODE_REFERENCE_SKIP) return fail(ctx, VR_ERR_ARG, "unknown mode");
    ctx->mode = mode;
    return VR_OK;
}

// use_static: the frame's scratch comes from the static array instead of the context's buffers, and
// nothing is allocated or freed (fast mode, device output, no counters).
static int render_impl(vr_ctx *ctx, int width, int height, int x0, int y0, int x1, int y1, int part, int parts, uint8_t *out,
                       int out_is_device, vr_stats *stats, bool use_static = false)
{
    if (!ctx || !out) return fail(ctx, VR_ERR_ARG, "null argument");
    if (parts < 1 || part < 0 || part >= parts) return fail(ctx, VR_ERR_ARG, "bad sort-first part");
    if (!ctx->bricks) return fail(ctx, VR_ERR_STATE, "no volume uploaded");
    if (ctx->sub) return fail(ctx, VR_ERR_STATE, "the context holds a sub-volume: use vr_render_partial");
    if (width < 2 || height < 2 || width > 65536 || height > 65536) return fail(ctx, VR_ERR_ARG, "image size out of range");
    if (x0 < 0 || y0 < 0 || x1 > width || y1 > height || x0 >= x1 || y0 >= y1) return fail(ctx, VR_ERR_ARG, "bad pixel rectangle");
    if (int rc = bind(ctx)) return rc;

    vr::FrameParams fp;
    float pyramid[20];
    if (int rc = vr_get_camera(ctx, width, height, pyramid, fp.m)) return rc;
    std::memcpy(fp.loc, pyramid + 4, sizeof fp.loc);
    fp.width = width; fp.height = height;
    fp.x0 = x0; fp.y0 = y0; fp.x1 = x1; fp.y1 = y1;
    fp.samples = ctx->samples;
    fp.part = part;
    fp.parts = parts;
    std::memcpy(fp.scene, ctx->scene, sizeof fp.scene);

    const size_t pixels = (size_t)(x1 - x0) * (size_t)rows_of_part(y1 - y0, part, parts);
    if (pixels == 0) {
        if (stats) std::memset(stats, 0, sizeof *stats);
        return VR_OK;
    }
    uint32_t *d_out = reinterpret_cast<uint32_t *>(out);
    if (!out_is_device) {
        if (pixels > ctx->frame_cap) {
            if (ctx->frame) (void)dev_free(ctx->frame);
            ctx->frame = nullptr;
            ctx->frame_cap = 0;
            VR_HIP(ctx, dev_alloc(&ctx->frame, pixels * 4));
            ctx->frame_cap = pixels;
        }
        d_out = ctx->frame;
    }
    const bool want_stats = stats != nullptr;
    if (ctx->mode != VR_MODE_REFERENCE_ORDER && ctx->occupancy_dirty) {
        drain(ctx);
        if (int rc = refresh_occupancy(ctx)) return rc;
    }
    poll_probe(ctx);
    if (ctx->mode == VR_MODE_FAST && ctx->fast_fallback && ++ctx->fallback_frames >= vr_ctx::kFallbackRetry) {
        ctx->fast_fallback = false;   // one fast frame to look again; its report decides what follows
        ctx->fallback_frames = 0;
    }
    const bool fast = ctx->mode == VR_MODE_FAST && !ctx->fast_fallback;
    unsigned char *sp = nullptr;
    if (use_static) {
        const int tile_px = 1 << vr::kRangeShift;
        const size_t tile_count = (size_t)((width + tile_px - 1) / tile_px) * (size_t)((height + tile_px - 1) / tile_px);
        if (pixels > static_scratch::kMaxPixels || tile_count > static_scratch::kMaxTiles)
            return fail(ctx, VR_ERR_ARG, "frame too large for the static scratch: at most " + std::to_string(static_scratch::kMaxPixels) +
                                             " pixels and " + std::to_string(static_scratch::kMaxTiles) + " 8x8 tiles");
        sp = ctx->device < static_scratch::kMaxDevices ? g_scratch_ptr[ctx->device] : nullptr;
        if (!sp) return fail(ctx, VR_ERR_STATE, "no static scratch on this device");
    }
    if (!use_static && fast && pixels > ctx->queue_cap) {
        if (ctx->queue) (void)dev_free(ctx->queue);
        if (ctx->long_queue) (void)dev_free(ctx->long_queue);
        ctx->queue = nullptr;
        ctx->long_queue = nullptr;
        ctx->queue_cap = 0;
        VR_HIP(ctx, dev_alloc(&ctx->queue, pixels * sizeof(vr::ReplayRay)));
        VR_HIP(ctx, dev_alloc(&ctx->long_queue, pixels * sizeof(vr::LongRay)));
        ctx->queue_cap = pixels;
    }
    if (fast && !use_static) {
        const int tile_px = 1 << vr::kRangeShift;
        const size_t tile_count = (size_t)((width + tile_px - 1) / tile_px) * (size_t)((height + tile_px - 1) / tile_px);
        if (tile_count > ctx->tiles_cap) {
            if (ctx->tiles) (void)dev_free(ctx->tiles);
            ctx->tiles = nullptr;
            ctx->tiles_cap = 0;
            VR_HIP(ctx, dev_alloc(&ctx->tiles, (tile_count + 1) * sizeof(uint2)));
            ctx->tiles_cap = tile_count;
        }
    }
    // Two frames in flight: only for fast frames into device memory without counters.  Every other
    // frame first lets the frames in flight finish and then runs on the one stream.
    const bool piped = fast && out_is_device && !want_stats && ctx->pipeline_depth == 2 && !use_static;
    if (!piped) drain(ctx);
    FrameBuffers fb{ctx->counters, ctx->queue, ctx->long_queue, ctx->tiles, ctx->stream, nullptr, nullptr, nullptr, nullptr};
    if (use_static)
        fb = FrameBuffers{reinterpret_cast<vr::Counters *>(sp + static_scratch::kCountersAt),
                          reinterpret_cast<vr::ReplayRay *>(sp + static_scratch::kQueueAt),
                          reinterpret_cast<vr::LongRay *>(sp + static_scratch::kLongAt),
                          reinterpret_cast<uint2 *>(sp + static_scratch::kTilesAt), ctx->stream, nullptr, nullptr, nullptr, nullptr};
    int set = 0, ring = -1;
    hipStream_t main_s = ctx->stream;
    if (piped) {
        set = (int)(ctx->pipe_index++ % (uint64_t)vr_ctx::kSets);
        const int tile_px = 1 << vr::kRangeShift;
        const size_t tile_count = (size_t)((width + tile_px - 1) / tile_px) * (size_t)((height + tile_px - 1) / tile_px);
        if (pixels > ctx->alt_queue_cap || tile_count > ctx->tiles_ring_cap) {
            drain(ctx);
            VR_HIP(ctx, hipStreamSynchronize(ctx->stream));
            for (int k = 0; k < vr_ctx::kSets - 1; k++) {
                if (ctx->queue_alt[k]) (void)dev_free(ctx->queue_alt[k]);
                if (ctx->long_alt[k]) (void)dev_free(ctx->long_alt[k]);
                ctx->queue_alt[k] = nullptr;
                ctx->long_alt[k] = nullptr;
            }
            ctx->alt_queue_cap = 0;
            for (int k = 0; k < vr_ctx::kRangeRing; k++) {
                if (ctx->tiles_ring[k]) (void)dev_free(ctx->tiles_ring[k]);
                ctx->tiles_ring[k] = nullptr;
                ctx->ranges_busy[k] = false;
            }
            ctx->tiles_ring_cap = 0;
            for (int k = 0; k < vr_ctx::kSets - 1; k++) {
                VR_HIP(ctx, dev_alloc(&ctx->queue_alt[k], pixels * sizeof(vr::ReplayRay)));
                VR_HIP(ctx, dev_alloc(&ctx->long_alt[k], pixels * sizeof(vr::LongRay)));
            }
            for (int k = 0; k < vr_ctx::kRangeRing; k++) VR_HIP(ctx, dev_alloc(&ctx->tiles_ring[k], (tile_count + 1) * sizeof(uint2)));
            ctx->alt_queue_cap = pixels;
            ctx->tiles_ring_cap = tile_count;
        }
        if (set > 0)
            fb = FrameBuffers{ctx->counters_alt[set - 1], ctx->queue_alt[set - 1], ctx->long_alt[set - 1], nullptr, ctx->stream, nullptr, nullptr, nullptr, nullptr};
        ring = (int)(ctx->ring_index++ % (uint64_t)vr_ctx::kRangeRing);
        fb.tiles = ctx->tiles_ring[ring];
        fb.prep = ctx->prep_stream;
        fb.ranges_done = ctx->ranges_done[ring];
        fb.ranges_free = ctx->ranges_busy[ring] ? ctx->ranges_free[ring] : nullptr;
        // this set's buffers were last used two frames ago; and whatever frame still being finished writes
        // into the memory this frame is going to write into comes first
        const unsigned char *o0 = reinterpret_cast<const unsigned char *>(d_out), *o1 = o0 + pixels * 4;
        for (int k = 0; k < vr_ctx::kSets; k++) {
            if (!ctx->tail_busy[k]) continue;
            const bool overlaps = ctx->set_out[k] < o1 && o0 < ctx->set_out[k] + ctx->set_out_bytes[k];
            if (k == set || overlaps) {
                VR_HIP(ctx, hipStreamWaitEvent(main_s, ctx->tail_done[k], 0));
                ctx->tail_busy[k] = false;
            }
        }
        ctx->set_out[set] = o0;
        ctx->set_out_bytes[set] = pixels * 4;
        fb.tail = ctx->tail_stream;
        fb.main_done = ctx->main_done[set];
    }
    // with two frames in flight the kernels that finish a frame fill the end of the next frame's march
    // already; the longest tiles first then only competes with them (measured: 1 to 3 % slower)
    fb.may_order = !use_static && (!piped || ctx->tile_order_piped);
    fb.set = set;
    // the counters also carry the fast march's queue fill, so they are cleared every frame
    if (!(piped && ctx->counters_clean[set])) VR_HIP(ctx, hipMemsetAsync(fb.counters, 0, offsetof(vr::Counters, lut_error), main_s));
    for (int k = 0; k < vr_ctx::kSets; k++)
        if (k == set || !piped) ctx->counters_clean[k] = false;   // frames outside the pipeline use set 0's and tell nobody
    fb.clear_after = piped;
    if (fb.clear_after) ctx->counters_clean[set] = true;
    if (fast) {
        fb.seq = ++ctx->frame_seq;
        if (fb.seq == 0u) fb.seq = ++ctx->frame_seq;   // 0 means "no report yet"
        // a rectangle cut out of the frame says little about the table (one on the outline is all replay)
        ctx->probe_meta[fb.seq & 7u] = vr_ctx::ProbeMeta{fb.seq, ctx->table_epoch, pixels,
                                                         pixels >= 4096 && x0 == 0 && y0 == 0 && x1 == width && y1 == height};
    }
    if (want_stats)
        VR_HIP(ctx, hipMemsetAsync(ctx->touched, 0, ((ctx->brick_count + 31) / 32) * sizeof(uint32_t), main_s));

    if (!piped) VR_HIP(ctx, hipEventRecord(ctx->ev0, main_s));
    if (ctx->dtype == VR_DTYPE_U8)
        launch_frame<uint8_t>(ctx, fp, d_out, want_stats, fb);
    else
        launch_frame<uint16_t>(ctx, fp, d_out, want_stats, fb);
    VR_HIP(ctx, hipGetLastError());
    if (piped) {
        VR_HIP(ctx, hipEventRecord(ctx->tail_done[set], ctx->tail_stream));
        ctx->tail_busy[set] = true;
        ctx->last_set = set;
        VR_HIP(ctx, hipEventRecord(ctx->ranges_free[ring], ctx->tail_stream));   // the replay was the last reader of the ranges
        ctx->ranges_busy[ring] = true;
    } else {
        ctx->last_set = -1;
    }
    if (!piped) VR_HIP(ctx, hipEventRecord(ctx->ev1, main_s));

    if (!out_is_device) {
        VR_HIP(ctx, hipMemcpyAsync(out, d_out, pixels * 4, hipMemcpyDeviceToHost, ctx->stream));
        VR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (want_stats) {
        vr::Counters c;
        VR_HIP(ctx, hipMemcpyAsync(&c, ctx->counters, sizeof c, hipMemcpyDeviceToHost, ctx->stream));
        std::vector<uint32_t> bits((ctx->brick_count + 31) / 32);
        VR_HIP(ctx, hipMemcpyAsync(bits.data(), ctx->touched, bits.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        VR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        uint64_t touched = 0;
        for (uint32_t w : bits) touched += (uint64_t)__builtin_popcount(w);
        std::memset(stats, 0, sizeof *stats);
        stats->rays = pixels;
        stats->rays_hit = c.rays_hit;
        stats->samples_nominal = c.rays_hit * (uint64_t)(ctx->samples + 2);
        stats->samples_evaluated = c.samples_evaluated;
        stats->cells_read = c.cells_read;
        stats->bricks_touched = touched;
        stats->replay_rays = c.replay_rays;
        stats->long_rays = c.long_count;
        stats->long_samples = c.long_samples;
        stats->replay_samples = c.replay_samples;
        stats->main_bricks = c.main_bricks;
        for (int k = 0; k < 24; k++) {
            stats->rays_by_lane_taps[k] = c.hist_lane[k];
            stats->rays_by_wave_taps[k] = c.hist_long[k];
        }
        const uint64_t brick_bytes = (uint64_t)vr::kBrick * vr::kBrick * vr::kBrick * (uint64_t)ctx->dtype;
        // bricks fetched + (fast/skip modes) the occupancy pyramid's bits + LUT + frame
        uint64_t occupancy = 0;
        if (ctx->mode != VR_MODE_REFERENCE_ORDER)
            for (int l = 0; l < vr::kLevels; l++) occupancy += ((uint64_t)ctx->lx[l] * ctx->ly[l] * ctx->lz[l] + 7) / 8;
        stats->algorithmic_bytes = brick_bytes * touched + occupancy + (uint64_t)vr::kLutLds * 16ull + 4ull * pixels;
        float ms = 0.0f;
        VR_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
        stats->kernel_ms = ms;
    }
    return VR_OK;
}

int vr_render_rect(vr_ctx *ctx, int width, int height, int x0, int y0, int x1, int y1, uint8_t *out, int out_is_device,
                   vr_stats *stats)
{
    return render_impl(ctx, width, height, x0, y0, x1, y1, 0, 1, out, out_is_device, stats);
}

int vr_render(vr_ctx *ctx, int width, int height, uint8_t *out, int out_is_device, vr_stats *stats)
{
    return render_impl(ctx, width, height, 0, 0, width, height, 0, 1, out, out_is_device, stats);
}

int vr_render_static(vr_ctx *ctx, int width, int height, uint8_t *d_out)
{
    if (!ctx) return VR_ERR_ARG;
    const int mode = ctx->mode;
    const unsigned long long refused = g_forbidden_calls;
    ctx->mode = VR_MODE_FAST;
    const int rc = render_impl(ctx, width, height, 0, 0, width, height, 0, 1, d_out, 1, nullptr, true);
    ctx->mode = mode;
    if (rc == VR_OK && g_forbidden_calls != refused) return fail(ctx, VR_ERR_STATE, "vr_render_static tried to allocate or free device memory");
    return rc;
}

int vr_static_max_pixels(void) { return (int)static_scratch::kMaxPixels; }

int vr_render_mapped(vr_ctx *ctx, int width, int height, int *frame_id)
{
    if (!ctx || !frame_id) return fail(ctx, VR_ERR_ARG, "null argument");
    if (width < 2 || height < 2 || width > 65536 || height > 65536) return fail(ctx, VR_ERR_ARG, "image size out of range");
    if (int rc = bind(ctx)) return rc;
    const int id = (int)(ctx->mapped_index % (uint64_t)vr_ctx::kMapped);
    vr_ctx::MappedSlot &m = ctx->mapped[id];
    if (m.mapped) return fail(ctx, VR_ERR_STATE, "the oldest of the three mapped frames has not been given back (vr_unmap_frame)");
    if (!ctx->copy_stream) VR_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    if (!m.copied) VR_HIP(ctx, hipEventCreateWithFlags(&m.copied, hipEventDisableTiming));
    const size_t bytes = (size_t)width * (size_t)height * 4;
    if (bytes > m.cap) {
        if (m.pending) VR_HIP(ctx, hipEventSynchronize(m.copied));
        m.pending = false;
        if (m.dev) (void)dev_free(m.dev);
        if (m.host) (void)hipHostFree(m.host);
        m.dev = m.host = nullptr;
        m.cap = 0;
        VR_HIP(ctx, dev_alloc(&m.dev, bytes));
        hipError_t e = hipHostMalloc((void **)&m.host, bytes, hipHostMallocDefault);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            (void)dev_free(m.dev);
            m.dev = nullptr;
            return fail(ctx, VR_ERR_HIP, std::string("hipHostMalloc of the frame's page-locked image: ") + hipGetErrorString(e));
        }
        m.cap = bytes;
    }
    // the copy that last read this slot's device image comes before the frame that overwrites it
    if (m.pending) VR_HIP(ctx, hipStreamWaitEvent(ctx->stream, m.copied, 0));
    if (int rc = render_impl(ctx, width, height, 0, 0, width, height, 0, 1, m.dev, 1, nullptr)) return rc;
    // the copy waits for the frame on a stream of its own, so that the next frame's march need not wait for it
    if (int rc = vr_wait_frame(ctx, ctx->copy_stream)) return rc;
    VR_HIP(ctx, hipMemcpyAsync(m.host, m.dev, bytes, hipMemcpyDeviceToHost, ctx->copy_stream));
    VR_HIP(ctx, hipEventRecord(m.copied, ctx->copy_stream));
    m.pending = true;
    ctx->mapped_index++;
    *frame_id = id;
    return VR_OK;
}

int vr_map_frame(vr_ctx *ctx, int frame_id, const uint8_t **pixels)
{
    if (!ctx || !pixels) return fail(ctx, VR_ERR_ARG, "null argument");
    if (frame_id < 0 || frame_id >= vr_ctx::kMapped) return fail(ctx, VR_ERR_ARG, "no such frame");
    vr_ctx::MappedSlot &m = ctx->mapped[frame_id];
    if (!m.pending) return fail(ctx, VR_ERR_STATE, "no frame has been rendered into this slot");
    if (int rc = bind(ctx)) return rc;
    VR_HIP(ctx, hipEventSynchronize(m.copied));
    m.mapped = true;
    *pixels = m.host;
    return VR_OK;
}

int vr_unmap_frame(vr_ctx *ctx, int frame_id)
{
    if (!ctx) return VR_ERR_ARG;
    if (frame_id < 0 || frame_id >= vr_ctx::kMapped) return fail(ctx, VR_ERR_ARG, "no such frame");
    ctx->mapped[frame_id].mapped = false;
    return VR_OK;
}

int vr_render_bands(vr_ctx *ctx, int width, int height, int part, int parts, uint8_t *out, int out_is_device, vr_stats *stats)
{
    return render_impl(ctx, width, height, 0, 0, width, height, part, parts, out, out_is_device, stats);
}

int vr_march_kernel_times(vr_ctx *ctx, float *ms, int capacity, int *count)
{
    if (!ctx || !ms || !count || capacity < 0) return fail(ctx, VR_ERR_ARG, "null argument");
    if (int rc = bind(ctx)) return rc;
    VR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const uint64_t have = std::min<uint64_t>(ctx->frames, (uint64_t)vr_ctx::kRing);
    const int n = (int)std::min<uint64_t>(have, (uint64_t)capacity);
    for (int i = 0; i < n; i++) {  // oldest of the n most recent frames first
        const int slot = (int)((ctx->frames - (uint64_t)n + (uint64_t)i) % (uint64_t)vr_ctx::kRing);
        VR_HIP(ctx, hipEventElapsedTime(&ms[i], ctx->ring_a[slot], ctx->ring_b[slot]));
    }
    *count = n;
    return VR_OK;
}

int vr_tune(vr_ctx *ctx, int width, int height, int part, int parts, int *chosen_budget)
{
    if (!ctx) return VR_ERR_ARG;
    if (!ctx->bricks) return fail(ctx, VR_ERR_STATE, "no volume uploaded");
    if (parts < 1 || part < 0 || part >= parts) return fail(ctx, VR_ERR_ARG, "bad part");
    if (width < 2 || height < 2 || width > 65536 || height > 65536) return fail(ctx, VR_ERR_ARG, "image size out of range");
    if (rows_of_part(height, part, parts) == 0) {   // more parts than bands: this part renders nothing, nothing to measure
        if (chosen_budget) *chosen_budget = lane_budget_for(ctx, 0);
        return VR_OK;
    }
    if (int rc = bind(ctx)) return rc;
    if (ctx->sub) {
        // a box of a split volume: the lane march of its partial images has a budget of its own (a box
        // cannot stop a ray as early as the whole-volume march, and how long its rays are depends on the box)
        if (part != 0 || parts != 1) return fail(ctx, VR_ERR_ARG, "a box of a split volume renders whole partial images");
        if (ctx->shading) return fail(ctx, VR_ERR_STATE, "shading is on: the partial march has no lighting");
        float *d_partial = nullptr;
        VR_HIP(ctx, dev_alloc(&d_partial, (size_t)width * (size_t)height * vr::kPartial * sizeof(float)));
        const int before = ctx->partial_budget;
        int best = before, rc = VR_OK;
        double best_s = 1.0e300;
        for (int budget : {48, 64, 80, 96, 112, 128, 160}) {
            ctx->partial_budget = budget;
            double fastest = 1.0e300;
            for (int rep = 0; rep < 3 && rc == VR_OK; rep++) {
                const auto t0 = std::chrono::steady_clock::now();
                for (int k = 0; k < 3 && rc == VR_OK; k++) rc = vr_render_partial(ctx, width, height, d_partial, 1);
                if (rc == VR_OK) rc = vr_synchronize(ctx);
                const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (rep > 0 && s < fastest) fastest = s;
            }
            if (rc != VR_OK) break;
            if (fastest < best_s) {
                best_s = fastest;
                best = budget;
            }
        }
        ctx->partial_budget = rc == VR_OK ? best : before;
        (void)hipStreamSynchronize(ctx->stream);
        (void)dev_free(d_partial);
        if (rc == VR_OK && chosen_budget) *chosen_budget = best;
        return rc;
    }
    const size_t pixels = (size_t)width * (size_t)rows_of_part(height, part, parts);
    // one allocation, three frames in turn (frames in flight must not share their output)
    const size_t frame_bytes = (pixels * 4 + 255) & ~(size_t)255;
    uint8_t *scratch = nullptr;
    VR_HIP(ctx, dev_alloc(&scratch, 3 * frame_bytes));
    const int mode = ctx->mode, before = ctx->lane_budget;
    ctx->mode = VR_MODE_FAST;
    // How long a lane stays with its ray before it hands it to a wave of its own: the best value
    // depends on how the rays' lengths are spread and on how many waves the launch has, i.e. on volume,
    // table, image size and share.  Each candidate renders a run of frames the way the caller will (one
    // or two frames in flight, as set) and is judged by the wall time of the run.
    const int candidates[] = {8, 16, 32, 48, 64, 96, 128, 160};
    constexpr int kRun = 6;
    int best = before > 0 ? before : lane_budget_for(ctx, 0), rc = VR_OK;
    double best_s = 1.0e300;
    for (int budget : candidates) {
        ctx->lane_budget = budget;
        double fastest = 1.0e300;
        for (int rep = 0; rep < 3 && rc == VR_OK; rep++) {      // the first run also warms the caches (and is the fast mode's probe)
            const auto t0 = std::chrono::steady_clock::now();
            for (int k = 0; k < kRun && rc == VR_OK; k++)
                rc = render_impl(ctx, width, height, 0, 0, width, height, part, parts, scratch + (size_t)(k % 3) * frame_bytes, 1, nullptr);
            if (rc == VR_OK) rc = vr_synchronize(ctx);
            const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (rep > 0 && s < fastest) fastest = s;
        }
        if (rc != VR_OK) break;
        if (fastest < best_s) {
            best_s = fastest;
            best = budget;
        }
    }
    ctx->lane_budget = rc == VR_OK ? best : before;
    ctx->mode = mode;
    drain(ctx);
    (void)hipStreamSynchronize(ctx->stream);
    (void)dev_free(scratch);
    if (rc == VR_OK && chosen_budget) *chosen_budget = best;
    return rc;
}

#ifdef VR_DIAG
// diagnostic build only (make diag): device buffer of 4 x uint64 per wave of the next fast march
int vr_diag_set_timeline(vr_ctx *ctx, void *d_buffer)
{
    if (!ctx) return VR_ERR_ARG;
    if (int rc = bind(ctx)) return rc;
    VR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    VR_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(vr::g_vr_timeline), &d_buffer, sizeof d_buffer));
    return VR_OK;
}
#endif

int vr_band_rows(int height, int part, int parts)
{
    if (height < 1 || parts < 1 || part < 0 || part >= parts) return VR_ERR_ARG;
    return rows_of_part(height, part, parts);
}

int vr_assemble_bands(vr_ctx *ctx, int width, int height, int parts, const uint8_t *d_gathered, uint64_t part_stride_bytes,
                      uint8_t *d_frame)
{
    if (!ctx || !d_gathered || !d_frame) return fail(ctx, VR_ERR_ARG, "null argument");
    if (width < 1 || height < 1 || width > 65536 || height > 65536 || parts < 1) return fail(ctx, VR_ERR_ARG, "bad image size or part count");
    if (part_stride_bytes % 4 != 0 || part_stride_bytes < (uint64_t)rows_of_part(height, 0, parts) * (uint64_t)width * 4ull)
        return fail(ctx, VR_ERR_ARG, "part stride smaller than the largest part or not a multiple of 4");
    if (int rc = bind(ctx)) return rc;
    const size_t pixels = (size_t)width * (size_t)height;
    hipLaunchKernelGGL(vr::k_assemble_bands, dim3((unsigned)((pixels + 255) / 256)), dim3(256), 0, ctx->stream,
                       reinterpret_cast<const uint32_t *>(d_gathered), (size_t)(part_stride_bytes / 4), parts, width, height,
                       reinterpret_cast<uint32_t *>(d_frame));
    VR_HIP(ctx, hipGetLastError());
    return VR_OK;
}

int vr_count_hit_rays(vr_ctx *ctx, int width, int height, int x0, int y0, int x1, int y1, uint64_t *hits)
{
    if (!ctx || !hits) return fail(ctx, VR_ERR_ARG, "null argument");
    if (width < 2 || height < 2 || x0 < 0 || y0 < 0 || x1 > width || y1 > height || x0 >= x1 || y0 >= y1)
        return fail(ctx, VR_ERR_ARG, "bad image size or pixel rectangle");
    if (int rc = bind(ctx)) return rc;
    vr::FrameParams fp;
    float pyramid[20];
    if (int rc = vr_get_camera(ctx, width, height, pyramid, fp.m)) return rc;
    std::memcpy(fp.loc, pyramid + 4, sizeof fp.loc);
    fp.width = width; fp.height = height;
    fp.x0 = x0; fp.y0 = y0; fp.x1 = x1; fp.y1 = y1;
    fp.samples = ctx->samples;
    fp.part = 0;
    fp.parts = 1;
    std::memcpy(fp.scene, ctx->scene, sizeof fp.scene);
    drain(ctx);
    VR_HIP(ctx, hipMemsetAsync(ctx->counters, 0, offsetof(vr::Counters, lut_error), ctx->stream));
    ctx->counters_clean[0] = false;
    const int tiles = ((x1 - x0 + 7) / 8) * 2 * ((y1 - y0 + 15) / 16);
    hipLaunchKernelGGL(vr::k_count_hits, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, ctx->stream, fp, &ctx->counters->rays_hit);
    unsigned long long h = 0;
    VR_HIP(ctx, hipMemcpyAsync(&h, &ctx->counters->rays_hit, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    VR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *hits = h;
    return VR_OK;
}

int vr_phantom_generate_device(vr_ctx *ctx, void *d_out, int dtype, int n, uint32_t seed)
{
    const int lo[3] = {0, 0, 0}, hi[3] = {n, n, n};
    return vr_phantom_generate_box_device(ctx, d_out, dtype, n, lo, hi, seed);
}

int vr_phantom_generate_box_device(vr_ctx *ctx, void *d_out, int dtype, int n, const int lo[3], const int hi[3], uint32_t seed)
{
    if (!ctx || !d_out || !lo || !hi) return fail(ctx, VR_ERR_ARG, "null argument");
    if (int rc = check_dims(ctx, dtype, n, n, n)) return rc;
    vr::PhantomBox box;
    for (int k = 0; k < 3; k++) {
        if (lo[k] < 0 || hi[k] > n || lo[k] >= hi[k]) return fail(ctx, VR_ERR_ARG, "bad voxel box");
        box.lo[k] = (uint32_t)lo[k];
        box.ext[k] = (uint32_t)(hi[k] - lo[k]);
    }
    if (int rc = bind(ctx)) return rc;
    const size_t count = (size_t)box.ext[0] * box.ext[1] * box.ext[2];
    const int blocks = (int)std::min<size_t>((count + 255) / 256, 256 * 32);
    if (dtype == VR_DTYPE_U8)
        hipLaunchKernelGGL((vr::k_phantom<uint8_t>), dim3(blocks), dim3(256), 0, ctx->stream, (uint8_t *)d_out, (uint32_t)n, 255u, seed, box);
    else
        hipLaunchKernelGGL((vr::k_phantom<uint16_t>), dim3(blocks), dim3(256), 0, ctx->stream, (uint16_t *)d_out, (uint32_t)n, 65535u, seed, box);
    VR_HIP(ctx, hipGetLastError());
    return VR_OK;
}

// ---- sort-last ---------------------------------------------------------------------------------------

int vr_brick_grid(int nx, int ny, int nz, int bricks[3])
{
    if (!bricks || nx < 1 || ny < 1 || nz < 1) return VR_ERR_ARG;
    bricks[0] = bricks_along(nx);
    bricks[1] = bricks_along(ny);
    bricks[2] = bricks_along(nz);
    return VR_OK;
}

int vr_subvolume_voxels(int nx, int ny, int nz, const int brick_lo[3], const int brick_hi[3], int voxel_lo[3], int voxel_hi[3])
{
    if (!brick_lo || !brick_hi || !voxel_lo || !voxel_hi || nx < 1 || ny < 1 || nz < 1) return VR_ERR_ARG;
    const int dims[3] = {nx, ny, nz};
    for (int k = 0; k < 3; k++) {
        if (brick_lo[k] < 0 || brick_hi[k] > bricks_along(dims[k]) || brick_lo[k] >= brick_hi[k]) return VR_ERR_ARG;
        voxels_of_bricks(dims[k], brick_lo[k], brick_hi[k], voxel_lo[k], voxel_hi[k]);
    }
    return VR_OK;
}

int vr_occupied_brick_counts(vr_ctx *ctx, int cell_bricks, uint32_t *d_counts)
{
    if (!ctx || !d_counts) return fail(ctx, VR_ERR_ARG, "null argument");
    if (cell_bricks < 1) return fail(ctx, VR_ERR_ARG, "cell size must be at least one brick");
    if (!ctx->bricks) return fail(ctx, VR_ERR_STATE, "no volume uploaded");
    if (int rc = bind(ctx)) return rc;
    if (int rc = refresh_occupancy(ctx)) return rc;
    const int cnx = (bricks_along(ctx->nx) + cell_bricks - 1) / cell_bricks, cny = (bricks_along(ctx->ny) + cell_bricks - 1) / cell_bricks;
    hipLaunchKernelGGL(vr::k_count_occupied, dim3(2048), dim3(256), 0, ctx->stream, ctx->empty[0], ctx->nbx, ctx->nby, ctx->nbz,
                       ctx->brick_lo[0], ctx->brick_lo[1], ctx->brick_lo[2], cell_bricks, cnx, cny, d_counts);
    VR_HIP(ctx, hipGetLastError());
    return VR_OK;
}

int vr_upload_subvolume_device(vr_ctx *ctx, const void *d_voxels, int dtype, int nx, int ny, int nz, const int brick_lo[3],
                               const int brick_hi[3], float max_value)
{
    if (!ctx || !brick_lo || !brick_hi) return fail(ctx, VR_ERR_ARG, "null argument");
    if (int rc = check_dims(ctx, dtype, nx, ny, nz)) return rc;
    if (int rc = bind(ctx)) return rc;
    if (dtype == VR_DTYPE_U8) return ingest_device<uint8_t>(ctx, (const uint8_t *)d_voxels, nx, ny, nz, max_value, false, brick_lo, brick_hi);
    return ingest_device<uint16_t>(ctx, (const uint16_t *)d_voxels, nx, ny, nz, max_value, false, brick_lo, brick_hi);
}

static int frame_params(vr_ctx *ctx, int width, int height, vr::FrameParams &fp)
{
    if (width < 2 || height < 2 || width > 65536 || height > 65536) return fail(ctx, VR_ERR_ARG, "image size out of range");
    float pyramid[20];
    if (int rc = vr_get_camera(ctx, width, height, pyramid, fp.m)) return rc;
    std::memcpy(fp.loc, pyramid + 4, sizeof fp.loc);
    fp.width = width; fp.height = height;
    fp.x0 = 0; fp.y0 = 0; fp.x1 = width; fp.y1 = height;
    fp.samples = ctx->samples;
    fp.part = 0;
    fp.parts = 1;
    std::memcpy(fp.scene, ctx->scene, sizeof fp.scene);
    return VR_OK;
}

int vr_render_partial(vr_ctx *ctx, int width, int height, float *out, int out_is_device)
{
    if (!ctx || !out) return fail(ctx, VR_ERR_ARG, "null argument");
    if (!ctx->bricks) return fail(ctx, VR_ERR_STATE, "no volume uploaded");
    if (ctx->shading) return fail(ctx, VR_ERR_STATE, "shading is on: the partial march has no lighting");
    if (int rc = bind(ctx)) return rc;
    vr::FrameParams fp;
    if (int rc = frame_params(ctx, width, height, fp)) return rc;
    const size_t pixels = (size_t)width * height;
    float *d_out = out;
    drain(ctx);
    if (!out_is_device) {
        if (pixels > ctx->partial_cap) {
            if (ctx->partial) (void)dev_free(ctx->partial);
            ctx->partial = nullptr;
            ctx->partial_cap = 0;
            VR_HIP(ctx, dev_alloc(&ctx->partial, pixels * vr::kPartial * sizeof(float)));
            ctx->partial_cap = pixels;
        }
        d_out = ctx->partial;
    }
    if (pixels > ctx->queue_cap) {
        if (ctx->queue) (void)dev_free(ctx->queue);
        if (ctx->long_queue) (void)dev_free(ctx->long_queue);
        ctx->queue = nullptr;
        ctx->long_queue = nullptr;
        ctx->queue_cap = 0;
        VR_HIP(ctx, dev_alloc(&ctx->queue, pixels * sizeof(vr::ReplayRay)));
        VR_HIP(ctx, dev_alloc(&ctx->long_queue, pixels * sizeof(vr::LongRay)));
        ctx->queue_cap = pixels;
    }
    if (int rc = refresh_occupancy(ctx)) return rc;
    VR_HIP(ctx, hipMemsetAsync(ctx->counters, 0, offsetof(vr::Counters, lut_error), ctx->stream));
    ctx->counters_clean[0] = false;
    const int tiles = ((width + 7) / 8) * 2 * ((height + 15) / 16);
    const vr::VolumeView vol = view_of(ctx);
    const float4 border = make_float4(ctx->border[0], ctx->border[1], ctx->border[2], ctx->border[3]);
    const float4 *lut = reinterpret_cast<const float4 *>(ctx->lut);
    const dim3 long_grid(2048);
    // depth range of this box's surface bricks per tile, as in the whole-volume fast mode
    const int tile_px = 1 << vr::kRangeShift;
    const int tiles_x = (width + tile_px - 1) / tile_px, tiles_y = (height + tile_px - 1) / tile_px;
    const size_t tile_count = (size_t)tiles_x * tiles_y;
    if (tile_count > ctx->tiles_cap) {
        if (ctx->tiles) (void)dev_free(ctx->tiles);
        ctx->tiles = nullptr;
        ctx->tiles_cap = 0;
        VR_HIP(ctx, dev_alloc(&ctx->tiles, (tile_count + 1) * sizeof(uint2)));
        ctx->tiles_cap = tile_count;
    }
    const uint2 *range = nullptr;
    if (ctx->tile_clip) {
        VR_HIP(ctx, hipMemsetAsync(ctx->tiles, 0, (tile_count + 1) * sizeof(uint2), ctx->stream));
        hipLaunchKernelGGL(vr::k_splat_bricks, dim3(splat_blocks(ctx)), dim3(256), 0, ctx->stream, vol, fp, ctx->occ_list,
                           ctx->d_occ_count, ctx->tiles, tiles_x, tiles_y, ctx->counters);
        range = ctx->tiles;
    }
    if (ctx->dtype == VR_DTYPE_U8) {
        hipLaunchKernelGGL((vr::k_march_partial<uint8_t>), dim3((unsigned)tiles), dim3(64), 0, ctx->stream, vol, fp, lut, border, ctx->lut_info, d_out, ctx->counters, ctx->long_queue, ctx->partial_budget, range, tiles_x);
        hipLaunchKernelGGL((vr::k_partial_long<uint8_t>), long_grid, dim3(256), 0, ctx->stream, vol, fp, lut, border, ctx->lut_info, d_out, ctx->counters, ctx->long_queue);
    } else {
        hipLaunchKernelGGL((vr::k_march_partial<uint16_t>), dim3((unsigned)tiles), dim3(64), 0, ctx->stream, vol, fp, lut, border, ctx->lut_info, d_out, ctx->counters, ctx->long_queue, ctx->partial_budget, range, tiles_x);
        hipLaunchKernelGGL((vr::k_partial_long<uint16_t>), long_grid, dim3(256), 0, ctx->stream, vol, fp, lut, border, ctx->lut_info, d_out, ctx->counters, ctx->long_queue);
    }
    VR_HIP(ctx, hipGetLastError());
    if (!out_is_device) {
        VR_HIP(ctx, hipMemcpyAsync(out, d_out, pixels * vr::kPartial * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        VR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return VR_OK;
}

static int pixel_span(vr_ctx *ctx, uint64_t first_pixel, uint64_t count)
{
    if (count == 0 || count > 0x7fffffffull || first_pixel + count > 0xffffffffull) return fail(ctx, VR_ERR_ARG, "bad pixel span");
    return VR_OK;
}

int vr_composite_pair(vr_ctx *ctx, int width, int height, uint64_t first_pixel, uint64_t count, const float *d_mine,
                      const float *d_theirs, int axis, int mine_is_low, float *d_out)
{
    if (!ctx || !d_mine || !d_theirs || !d_out) return fail(ctx, VR_ERR_ARG, "null argument");
    if (axis < 0 || axis > 2) return fail(ctx, VR_ERR_ARG, "axis must be 0, 1 or 2");
    if (!ctx->bricks) return fail(ctx, VR_ERR_STATE, "no volume uploaded");
    if (int rc = pixel_span(ctx, first_pixel, count)) return rc;
    if (int rc = bind(ctx)) return rc;
    vr::FrameParams fp;
    if (int rc = frame_params(ctx, width, height, fp)) return rc;
    hipLaunchKernelGGL(vr::k_composite_pair, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, ctx->stream, fp, (float)ctx->nx,
                       (float)ctx->ny, (float)ctx->nz, (uint32_t)first_pixel, (uint32_t)count, d_mine, d_theirs, axis, mine_is_low, d_out);
    VR_HIP(ctx, hipGetLastError());
    return VR_OK;
}

int vr_composite_ordered(vr_ctx *ctx, int width, int height, uint64_t first_pixel, uint64_t count, const float *d_parts,
                         const int *d_slot_of_box, const int grid[3], float *d_out)
{
    if (!ctx || !d_parts || !d_slot_of_box || !grid || !d_out) return fail(ctx, VR_ERR_ARG, "null argument");
    if (grid[0] < 1 || grid[1] < 1 || grid[2] < 1 || (int64_t)grid[0] * grid[1] * grid[2] > 4096) return fail(ctx, VR_ERR_ARG, "bad box grid");
    if (!ctx->bricks) return fail(ctx, VR_ERR_STATE, "no volume uploaded");
    if (int rc = pixel_span(ctx, first_pixel, count)) return rc;
    if (int rc = bind(ctx)) return rc;
    vr::FrameParams fp;
    if (int rc = frame_params(ctx, width, height, fp)) return rc;
    const vr::BoxGrid g{grid[0], grid[1], grid[2]};
    hipLaunchKernelGGL(vr::k_composite_ordered, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, ctx->stream, fp, (float)ctx->nx,
                       (float)ctx->ny, (float)ctx->nz, (uint32_t)first_pixel, (uint32_t)count, d_parts, d_slot_of_box, g, d_out);
    VR_HIP(ctx, hipGetLastError());
    return VR_OK;
}

int vr_decide_partials(vr_ctx *ctx, int width, int height, uint64_t first_pixel, uint64_t count, const float *d_partials,
                       uint8_t *d_out_rgba8, uint32_t *d_undecided, uint32_t *d_undecided_count)
{
    if (!ctx || !d_partials || !d_out_rgba8 || !d_undecided || !d_undecided_count) return fail(ctx, VR_ERR_ARG, "null argument");
    if (!ctx->bricks) return fail(ctx, VR_ERR_STATE, "no volume uploaded");
    if (int rc = pixel_span(ctx, first_pixel, count)) return rc;
    if (int rc = bind(ctx)) return rc;
    vr::FrameParams fp;
    if (int rc = frame_params(ctx, width, height, fp)) return rc;
    if (int rc = refresh_occupancy(ctx)) return rc;   // brings the table's colour range and rounding bound up to date
    VR_HIP(ctx, hipMemsetAsync(d_undecided_count, 0, sizeof(uint32_t), ctx->stream));
    hipLaunchKernelGGL(vr::k_decide_partials, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, ctx->stream, fp, ctx->lut_info,
                       (uint32_t)first_pixel, (uint32_t)count, d_partials, reinterpret_cast<uint32_t *>(d_out_rgba8), d_undecided,
                       d_undecided_count);
    VR_HIP(ctx, hipGetLastError());
    return VR_OK;
}

int vr_chain_stage(vr_ctx *ctx, int width, int height, const uint32_t *d_pixels, uint64_t count, const float *d_colour_in,
                   float *d_colour_out, int stage, const int grid[3], const int box[3])
{
    if (!ctx || !d_pixels || !d_colour_in || !d_colour_out || !grid || !box) return fail(ctx, VR_ERR_ARG, "null argument");
    if (grid[0] < 1 || grid[1] < 1 || grid[2] < 1 || (int64_t)grid[0] * grid[1] * grid[2] > 4096) return fail(ctx, VR_ERR_ARG, "bad box grid");
    for (int k = 0; k < 3; k++)
        if (box[k] < 0 || box[k] >= grid[k]) return fail(ctx, VR_ERR_ARG, "box outside the grid");
    if (stage < 0 || stage >= grid[0] * grid[1] * grid[2]) return fail(ctx, VR_ERR_ARG, "stage outside the grid");
    if (!ctx->bricks) return fail(ctx, VR_ERR_STATE, "no volume uploaded");
    if (int rc = pixel_span(ctx, 0, count)) return rc;
    if (int rc = bind(ctx)) return rc;
    vr::FrameParams fp;
    if (int rc = frame_params(ctx, width, height, fp)) return rc;
    if (int rc = refresh_occupancy(ctx)) return rc;
    const vr::VolumeView vol = view_of(ctx);
    const float4 border = make_float4(ctx->border[0], ctx->border[1], ctx->border[2], ctx->border[3]);
    const float4 *lut = reinterpret_cast<const float4 *>(ctx->lut);
    const vr::BoxGrid g{grid[0], grid[1], grid[2]};
    const unsigned blocks = (unsigned)std::min<uint64_t>((count + 3) / 4, 2048);
    if (ctx->dtype == VR_DTYPE_U8)
        hipLaunchKernelGGL((vr::k_chain_exact<uint8_t>), dim3(blocks), dim3(256), 0, ctx->stream, vol, fp, lut, border, ctx->lut_info, d_pixels, (uint32_t)count,
                           reinterpret_cast<const float4 *>(d_colour_in), reinterpret_cast<float4 *>(d_colour_out), stage, g, box[0], box[1], box[2]);
    else
        hipLaunchKernelGGL((vr::k_chain_exact<uint16_t>), dim3(blocks), dim3(256), 0, ctx->stream, vol, fp, lut, border, ctx->lut_info, d_pixels, (uint32_t)count,
                           reinterpret_cast<const float4 *>(d_colour_in), reinterpret_cast<float4 *>(d_colour_out), stage, g, box[0], box[1], box[2]);
    VR_HIP(ctx, hipGetLastError());
    return VR_OK;
}

int vr_store_chain(vr_ctx *ctx, const uint32_t *d_pixels, const float *d_colour, uint64_t count, uint64_t first_pixel,
                   uint64_t pixel_count, uint8_t *d_out_rgba8)
{
    if (!ctx || !d_pixels || !d_colour || !d_out_rgba8) return fail(ctx, VR_ERR_ARG, "null argument");
    if (int rc = pixel_span(ctx, 0, count)) return rc;
    if (int rc = pixel_span(ctx, first_pixel, pixel_count)) return rc;
    if (int rc = bind(ctx)) return rc;
    hipLaunchKernelGGL(vr::k_store_chain, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, ctx->stream, d_pixels,
                       reinterpret_cast<const float4 *>(d_colour), (uint32_t)count, (uint32_t)first_pixel, (uint32_t)pixel_count,
                       reinterpret_cast<uint32_t *>(d_out_rgba8));
    VR_HIP(ctx, hipGetLastError());
    return VR_OK;
}

}  // extern "C"
