// libvr_mg.so: one frame from several GPUs over RCCL (include/vr_mg.h).  Host code only: every kernel
// is reached through the public C ABI of libvr_hip.so (include/vr_abi.h); the collectives are RCCL
// point-to-point groups, all-gathers and all-reduces.
// The reference renders on one GPU from one queue (Source/VulkanApplication.cpp:316-321): no counterpart.
#include "../../include/vr_mg.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static_assert(VRMG_UNIQUE_ID_BYTES == sizeof(ncclUniqueId), "the unique id travels as VRMG_UNIQUE_ID_BYTES bytes");

namespace {
constexpr int kPartialFloats = 5;   // F0, F1, F2, Tr, S per pixel (vr_render_partial)
constexpr int kBandRows = 16;       // rows of a sort-first band (vr_render_bands)
constexpr int kTurns = 3;           // sort-first: band buffers and exchanges of this many frames in turn

// Test hook: while VRMG_TEST_FAIL_ALLOC=n is in the environment, the n-th allocation of every frame call is refused.
int g_allocations_this_call = 0;
bool refused_by_test()
{
    const char *e = std::getenv("VRMG_TEST_FAIL_ALLOC");
    return e && ++g_allocations_this_call == std::atoi(e);
}

struct Buffer {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t fit(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        if (refused_by_test()) return hipErrorOutOfMemory;
        const hipError_t e = hipMalloc(&p, bytes);
        if (e == hipSuccess) cap = bytes;
        else (void)hipGetLastError();
        return e;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <typename T>
    T *as() const { return static_cast<T *>(p); }
};
}  // namespace

struct vrmg {
    vr_ctx *ctx = nullptr;
    ncclComm_t comm = nullptr;
    bool own_comm = false;
    int rank = 0, world = 1, device = 0;
    hipStream_t stream = nullptr;        // the context's stream: sort-last frames are one ordered queue on it
    hipStream_t comm_stream = nullptr;   // sort-first: the gather of frame n beside the march of frame n + 1
    hipEvent_t last_done = nullptr;      // behind the last operation of the most recent frame
    bool have_frame = false;
    std::string error;
    int grid[3] = {1, 1, 1};
    bool grid_set = false;
    Buffer status;                       // two words: what the ranks agree on after a frame's buffers were sized
    // sort-first
    Buffer bands[kTurns], gather_in;     // rank 0: where the peers' bands arrive, one part after the other
    hipEvent_t gathered[kTurns] = {nullptr, nullptr, nullptr};
    bool turn_busy[kTurns] = {false, false, false};
    uint64_t turn = 0;
    size_t agreed_bands = 0;             // band buffer size all ranks are known to hold
    // sort-last
    Buffer partial, parts, work[2], merged, bytes, undecided, count, counts_all, lists, pixels, colour[2], frame, slots;
    size_t agreed_padded[2] = {0, 0};    // per schedule: frame size all ranks are known to hold buffers for
    size_t agreed_chain = 0, agreed_longest = 0;   // re-march: capacities (pixels in all, per rank) all ranks are known to hold
};

namespace {

int fail(vrmg *mg, int code, const std::string &msg)
{
    if (mg) mg->error = msg;
    return code;
}

// The first failure of a frame; later steps still run (see "Errors and peers" in vr_mg.h).
struct Status {
    vrmg *mg;
    int code = VR_OK;
    void hip(hipError_t e, const char *what)
    {
        if (e != hipSuccess && code == VR_OK) code = fail(mg, VR_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
    }
    void nccl(ncclResult_t r, const char *what)
    {
        if (r != ncclSuccess && code == VR_OK) code = fail(mg, VR_ERR_HIP, std::string(what) + ": " + ncclGetErrorString(r));
    }
    void vr(int rc, const char *what)
    {
        if (rc != VR_OK && code == VR_OK) code = fail(mg, rc, std::string(what) + ": " + vr_last_error(mg->ctx));
    }
};
#define ST_HIP(st, call) (st).hip((call), #call)
#define ST_NCCL(st, call) (st).nccl((call), #call)
#define ST_VR(st, call) (st).vr((call), #call)

// All ranks learn whether any of them failed so far (collective; waits on the host).  Called only in frames
// whose buffer sizes grew, which are the same frames on every rank.
int agree(vrmg *mg, hipStream_t s, int local_code, const char *what)
{
    const int flag = local_code != VR_OK ? 1 : 0;
    int any = flag;
    bool comm_ok = true;
    if (hipMemcpyAsync(mg->status.p, &flag, sizeof flag, hipMemcpyHostToDevice, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess)   // `flag` lives on this stack frame
        comm_ok = false;
    // the all-reduce is entered whatever happened here: the peers are in it
    if (ncclAllReduce(mg->status.p, mg->status.as<int>() + 1, 1, ncclInt, ncclMax, mg->comm, s) != ncclSuccess) comm_ok = false;
    if (hipMemcpyAsync(&any, mg->status.as<int>() + 1, sizeof any, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess)
        comm_ok = false;
    if (local_code != VR_OK) return local_code;
    if (!comm_ok) return fail(mg, VR_ERR_HIP, std::string(what) + ": the ranks could not agree on the outcome");
    if (any) return fail(mg, VR_ERR_HIP, std::string(what) + ": failed on another rank");
    return VR_OK;
}

int create_common(vr_ctx *ctx, int rank, int world, vrmg **out, vrmg *&mg)
{
    if (!ctx || !out) return VR_ERR_ARG;
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return VR_ERR_ARG;
    mg = new vrmg();
    mg->ctx = ctx;
    mg->rank = rank;
    mg->world = world;
    mg->stream = static_cast<hipStream_t>(vr_stream(ctx));
    bool ok = vr_synchronize(ctx) == VR_OK && hipGetDevice(&mg->device) == hipSuccess;   // vr_synchronize makes the context's GPU current
    ok = ok && hipStreamCreateWithFlags(&mg->comm_stream, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&mg->last_done, hipEventDisableTiming) == hipSuccess;
    for (int k = 0; k < kTurns && ok; k++) ok = hipEventCreateWithFlags(&mg->gathered[k], hipEventDisableTiming) == hipSuccess;
    ok = ok && mg->status.fit(2 * sizeof(int)) == hipSuccess;
    ok = ok && mg->count.fit(2 * sizeof(uint32_t)) == hipSuccess;
    if (!ok) {
        vrmg_destroy(mg);
        mg = nullptr;
        return VR_ERR_HIP;
    }
    return VR_OK;
}

// Pixel span a rank holds after all binary-swap rounds (round r: the rank whose bit r is set keeps the upper half).
void swap_span(int rank, int world, uint64_t padded, uint64_t &first, uint64_t &count)
{
    first = 0;
    count = padded;
    for (int r = 0; (1 << r) < world; r++) {
        count /= 2;
        if ((rank >> r) & 1) first += count;
    }
}

uint64_t span_first(int rank, int world, int schedule, uint64_t padded)
{
    uint64_t first = (uint64_t)rank * (padded / (uint64_t)world), count = 0;
    if (schedule == VRMG_BINARY_SWAP) swap_span(rank, world, padded, first, count);
    return first;
}

}  // namespace

extern "C" {

int vrmg_plan_sort_last(int rank, int world, const int grid[3], int schedule, int width, int height, vrmg_plan *plan)
{
    if (!grid || !plan || world < 1 || rank < 0 || rank >= world) return VR_ERR_ARG;
    if (grid[0] < 1 || grid[1] < 1 || grid[2] < 1 || (long long)grid[0] * grid[1] * grid[2] != world) return VR_ERR_ARG;
    if (schedule != VRMG_DIRECT_SEND && schedule != VRMG_BINARY_SWAP) return VR_ERR_ARG;
    if (width < 2 || height < 2 || (uint64_t)width * (uint64_t)height > 0x7fffffffull) return VR_ERR_ARG;
    std::memset(plan, 0, sizeof *plan);
    plan->pixels = (uint64_t)width * (uint64_t)height;
    const uint64_t unit = (uint64_t)world * 64u;
    plan->padded = (plan->pixels + unit - 1) / unit * unit;
    plan->span = plan->padded / (uint64_t)world;
    plan->first = (uint64_t)rank * plan->span;
    plan->box[0] = rank % grid[0];
    plan->box[1] = (rank / grid[0]) % grid[1];
    plan->box[2] = rank / (grid[0] * grid[1]);
    if (schedule == VRMG_DIRECT_SEND) return VR_OK;
    // binary swap: round r pairs the ranks that differ in bit r; with the box index x fastest those are first the
    // x neighbours (log2 gx rounds), then y, then z, and every merged pair is separated by a plane normal to that axis
    int axes[VRMG_MAX_ROUNDS], n = 0;
    for (int axis = 0; axis < 3; axis++) {
        if (grid[axis] & (grid[axis] - 1)) return VR_ERR_ARG;   // needs a power-of-two number of boxes per axis
        for (int g = grid[axis]; g > 1; g >>= 1) {
            if (n == VRMG_MAX_ROUNDS) return VR_ERR_ARG;
            axes[n++] = axis;
        }
    }
    plan->rounds = n;
    uint64_t at = 0, count = plan->padded;
    for (int r = 0; r < n; r++) {
        const bool upper = (rank >> r) & 1;
        count /= 2;
        vrmg_round &rd = plan->round[r];
        rd.partner = rank ^ (1 << r);
        rd.axis = axes[r];
        rd.mine_is_low = upper ? 0 : 1;
        rd.keep_offset = upper ? count : 0;
        rd.send_offset = upper ? 0 : count;
        rd.count = count;
        rd.first_pixel = at + rd.keep_offset;
        rd.out_buffer = count == plan->span ? 2 : (r & 1);
        at += rd.keep_offset;
    }
    plan->first = at;
    return VR_OK;
}

int vrmg_get_unique_id(void *out_bytes)
{
    if (!out_bytes) return VR_ERR_ARG;
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return VR_ERR_HIP;
    std::memcpy(out_bytes, &id, sizeof id);
    return VR_OK;
}

int vrmg_create(vr_ctx *ctx, const void *unique_id_bytes, int rank, int world, vrmg **out)
{
    if (!unique_id_bytes) return VR_ERR_ARG;
    vrmg *mg = nullptr;
    if (int rc = create_common(ctx, rank, world, out, mg)) return rc;
    ncclUniqueId id;
    std::memcpy(&id, unique_id_bytes, sizeof id);
    if (ncclCommInitRank(&mg->comm, world, id, rank) != ncclSuccess) {
        mg->comm = nullptr;
        vrmg_destroy(mg);
        return VR_ERR_HIP;
    }
    mg->own_comm = true;
    *out = mg;
    return VR_OK;
}

int vrmg_create_from_comm(vr_ctx *ctx, void *nccl_comm, int rank, int world, vrmg **out)
{
    if (!nccl_comm) return VR_ERR_ARG;
    vrmg *mg = nullptr;
    if (int rc = create_common(ctx, rank, world, out, mg)) return rc;
    mg->comm = static_cast<ncclComm_t>(nccl_comm);
    *out = mg;
    return VR_OK;
}

int vrmg_destroy(vrmg *mg)
{
    if (!mg) return VR_OK;
    (void)vr_synchronize(mg->ctx);
    if (mg->comm_stream) (void)hipStreamSynchronize(mg->comm_stream);
    for (Buffer *b : {&mg->status, &mg->bands[0], &mg->bands[1], &mg->bands[2], &mg->gather_in, &mg->partial, &mg->parts, &mg->work[0], &mg->work[1],
                      &mg->merged, &mg->bytes, &mg->undecided, &mg->count, &mg->counts_all, &mg->lists, &mg->pixels, &mg->colour[0],
                      &mg->colour[1], &mg->frame, &mg->slots})
        b->release();
    if (mg->own_comm && mg->comm) (void)ncclCommDestroy(mg->comm);
    for (hipEvent_t e : mg->gathered)
        if (e) (void)hipEventDestroy(e);
    if (mg->last_done) (void)hipEventDestroy(mg->last_done);
    if (mg->comm_stream) (void)hipStreamDestroy(mg->comm_stream);
    delete mg;
    return VR_OK;
}

const char *vrmg_last_error(const vrmg *mg) { return mg ? mg->error.c_str() : "null object"; }

int vrmg_wait_frame(vrmg *mg, void *stream)
{
    if (!mg) return VR_ERR_ARG;
    if (!mg->have_frame) return VR_OK;
    if (hipSetDevice(mg->device) != hipSuccess) return fail(mg, VR_ERR_HIP, "hipSetDevice failed");
    const hipError_t e = hipStreamWaitEvent(static_cast<hipStream_t>(stream), mg->last_done, 0);
    return e == hipSuccess ? VR_OK : fail(mg, VR_ERR_HIP, std::string("hipStreamWaitEvent: ") + hipGetErrorString(e));
}

int vrmg_synchronize(vrmg *mg)
{
    if (!mg) return VR_ERR_ARG;
    if (vr_synchronize(mg->ctx) != VR_OK) return fail(mg, VR_ERR_HIP, vr_last_error(mg->ctx));
    const hipError_t e = hipStreamSynchronize(mg->comm_stream);
    return e == hipSuccess ? VR_OK : fail(mg, VR_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
}

int vrmg_render_sort_first(vrmg *mg, int width, int height, uint8_t *d_frame)
{
    if (!mg) return VR_ERR_ARG;
    if (mg->rank == 0 && !d_frame) return fail(mg, VR_ERR_ARG, "rank 0 needs a frame buffer");
    const int rows0 = vr_band_rows(height, 0, mg->world);      // the largest part
    if (rows0 <= 0 || width < 2) return fail(mg, VR_ERR_ARG, "bad image size");
    if (hipSetDevice(mg->device) != hipSuccess) return fail(mg, VR_ERR_HIP, "hipSetDevice failed");
    const size_t row_bytes = (size_t)width * 4u, stride = (size_t)rows0 * row_bytes;
    const int world = mg->world, rank = mg->rank;
    Status st{mg};
    g_allocations_this_call = 0;

    // sizing first; when the band buffers have to grow, the ranks agree that all of them could
    if (stride > mg->agreed_bands) {
        if (hipStreamSynchronize(mg->comm_stream) != hipSuccess || vr_synchronize(mg->ctx) != VR_OK)   // exchanges in flight read the old buffers
            st.code = fail(mg, VR_ERR_HIP, "cannot wait for the frames in flight");
        for (int k = 0; k < kTurns; k++) {
            ST_HIP(st, mg->bands[k].fit(stride));
            mg->turn_busy[k] = false;
        }
        if (rank == 0 && world > 1) ST_HIP(st, mg->gather_in.fit(stride * (size_t)world));
        if (int rc = agree(mg, mg->comm_stream, st.code, "sort-first band buffers")) return rc;
        mg->agreed_bands = stride;
    }

    // this rank's bands: buffers in turn, so that the march of this frame need not wait for the gather of the last
    const int turn = (int)(mg->turn++ % (uint64_t)kTurns);
    uint8_t *bands = mg->bands[turn].as<uint8_t>();
    if (mg->turn_busy[turn]) ST_HIP(st, hipStreamWaitEvent(mg->stream, mg->gathered[turn], 0));   // the gather three frames ago read this buffer
    const int my_rows = vr_band_rows(height, rank, world);
    if (my_rows > 0) ST_VR(st, vr_render_bands(mg->ctx, width, height, rank, world, bands, 1, nullptr));
    ST_VR(st, vr_wait_frame(mg->ctx, mg->comm_stream));   // the exchange follows the frame on its own stream

    // gather on rank 0: one point-to-point transfer per peer, all in one group (rank 0's own bands stay where they are)
    ST_NCCL(st, ncclGroupStart());
    if (rank != 0 && my_rows > 0) ST_NCCL(st, ncclSend(bands, (size_t)my_rows * row_bytes, ncclUint8, 0, mg->comm, mg->comm_stream));
    if (rank == 0)
        for (int p = 1; p < world; p++) {
            const int rows = vr_band_rows(height, p, world);
            if (rows > 0) ST_NCCL(st, ncclRecv(mg->gather_in.as<uint8_t>() + (size_t)p * stride, (size_t)rows * row_bytes, ncclUint8, p, mg->comm, mg->comm_stream));
        }
    ST_NCCL(st, ncclGroupEnd());
    // ... and every part's bands into their rows of the frame: band j of part p is band j * world + p of the frame,
    // so a part is one strided copy (plus one for a short last band)
    if (rank == 0) {
        const int frame_bands = (height + kBandRows - 1) / kBandRows;
        const size_t band_bytes = (size_t)kBandRows * row_bytes;
        for (int p = 0; p < world; p++) {
            const uint8_t *src = p == 0 ? bands : mg->gather_in.as<uint8_t>() + (size_t)p * stride;
            const int count = frame_bands > p ? (frame_bands - p + world - 1) / world : 0;      // bands of part p
            if (count == 0) continue;
            const int last = p + (count - 1) * world;                                           // its last band in the frame
            const int last_rows = std::min(kBandRows, height - last * kBandRows);
            const int whole = last_rows == kBandRows ? count : count - 1;
            if (whole > 0)
                ST_HIP(st, hipMemcpy2DAsync(d_frame + (size_t)p * band_bytes, (size_t)world * band_bytes, src, band_bytes, band_bytes, (size_t)whole,
                                            hipMemcpyDeviceToDevice, mg->comm_stream));
            if (whole < count)
                ST_HIP(st, hipMemcpyAsync(d_frame + (size_t)last * band_bytes, src + (size_t)whole * band_bytes, (size_t)last_rows * row_bytes,
                                          hipMemcpyDeviceToDevice, mg->comm_stream));
        }
    }
    ST_HIP(st, hipEventRecord(mg->gathered[turn], mg->comm_stream));
    mg->turn_busy[turn] = true;
    ST_HIP(st, hipEventRecord(mg->last_done, mg->comm_stream));
    mg->have_frame = true;
    return st.code;
}

int vrmg_set_box_grid(vrmg *mg, const int grid[3])
{
    if (!mg || !grid) return VR_ERR_ARG;
    if (grid[0] < 1 || grid[1] < 1 || grid[2] < 1 || (long long)grid[0] * grid[1] * grid[2] != mg->world)
        return fail(mg, VR_ERR_ARG, "the box grid must have one box per rank");
    if (hipSetDevice(mg->device) != hipSuccess) return fail(mg, VR_ERR_HIP, "hipSetDevice failed");
    // direct send: slot s of the exchange holds rank s's image, and a rank's box index (x fastest) is its rank
    std::vector<int> slots(mg->world);
    for (int r = 0; r < mg->world; r++) slots[r] = r;
    if (mg->slots.fit(slots.size() * sizeof(int)) != hipSuccess ||
        hipMemcpyAsync(mg->slots.p, slots.data(), slots.size() * sizeof(int), hipMemcpyHostToDevice, mg->stream) != hipSuccess ||
        hipStreamSynchronize(mg->stream) != hipSuccess)   // `slots` lives on this stack frame
        return fail(mg, VR_ERR_HIP, "cannot upload the slot table");
    std::memcpy(mg->grid, grid, sizeof mg->grid);
    mg->grid_set = true;
    return VR_OK;
}

int vrmg_render_sort_last(vrmg *mg, int width, int height, int schedule, uint8_t *d_frame, uint64_t *remarched)
{
    if (!mg) return VR_ERR_ARG;
    if (mg->rank == 0 && !d_frame) return fail(mg, VR_ERR_ARG, "rank 0 needs a frame buffer");
    if (!mg->grid_set) return fail(mg, VR_ERR_STATE, "vrmg_set_box_grid first");
    if (schedule != VRMG_DIRECT_SEND && schedule != VRMG_BINARY_SWAP) return fail(mg, VR_ERR_ARG, "unknown schedule");
    if (width < 2 || height < 2 || (uint64_t)width * (uint64_t)height > 0x7fffffffull) return fail(mg, VR_ERR_ARG, "bad image size");
    const int world = mg->world, rank = mg->rank;
    vrmg_plan plan;
    if (vrmg_plan_sort_last(rank, world, mg->grid, schedule, width, height, &plan) != VR_OK)
        return fail(mg, VR_ERR_ARG, "binary swap needs a power-of-two number of boxes per axis");
    const uint64_t pixels = plan.pixels, padded = plan.padded, span = plan.span, first = plan.first;
    const size_t fl = sizeof(float);
    if (hipSetDevice(mg->device) != hipSuccess) return fail(mg, VR_ERR_HIP, "hipSetDevice failed");
    hipStream_t s = mg->stream;
    Status st{mg};
    g_allocations_this_call = 0;

    // every buffer of steps 1 to 3 and of the final gather, before the first collective
    if (padded > mg->agreed_padded[schedule]) {
        if (vr_synchronize(mg->ctx) != VR_OK) st.code = fail(mg, VR_ERR_HIP, vr_last_error(mg->ctx));   // frames in flight read the old buffers
        ST_HIP(st, mg->partial.fit(padded * kPartialFloats * fl));
        ST_HIP(st, mg->merged.fit(span * kPartialFloats * fl));
        ST_HIP(st, mg->bytes.fit(span * 4));
        ST_HIP(st, mg->undecided.fit(span * 4));
        ST_HIP(st, mg->counts_all.fit((size_t)world * 2 * sizeof(uint32_t)));
        if (rank == 0) ST_HIP(st, mg->frame.fit(padded * 4));
        if (schedule == VRMG_DIRECT_SEND) {
            ST_HIP(st, mg->parts.fit((size_t)world * span * kPartialFloats * fl));
        } else {
            ST_HIP(st, mg->work[0].fit(padded / 2 * kPartialFloats * fl + 16));
            ST_HIP(st, mg->work[1].fit(padded / 2 * kPartialFloats * fl + 16));
            ST_HIP(st, mg->parts.fit(padded / 2 * kPartialFloats * fl + 16));   // what the partner sends
        }
        if (int rc = agree(mg, s, st.code, "sort-last frame buffers")) return rc;
        mg->agreed_padded[schedule] = padded;
    }
    // rows past the last pixel are padding: vr_render_partial leaves them alone, the merges read them as misses
    ST_HIP(st, hipMemsetAsync(mg->partial.as<float>() + pixels * kPartialFloats, 0, (padded - pixels) * kPartialFloats * fl, s));

    // 1. this box's partial image
    ST_VR(st, vr_render_partial(mg->ctx, width, height, mg->partial.as<float>(), 1));

    // 2. exchange + merge in visibility order
    if (schedule == VRMG_DIRECT_SEND) {
        const size_t chunk = span * kPartialFloats;
        ST_NCCL(st, ncclGroupStart());
        for (int p = 0; p < world; p++) {
            ST_NCCL(st, ncclSend(mg->partial.as<float>() + (size_t)p * chunk, chunk, ncclFloat, p, mg->comm, s));
            ST_NCCL(st, ncclRecv(mg->parts.as<float>() + (size_t)p * chunk, chunk, ncclFloat, p, mg->comm, s));
        }
        ST_NCCL(st, ncclGroupEnd());
        ST_VR(st, vr_composite_ordered(mg->ctx, width, height, first, span, mg->parts.as<float>(), mg->slots.as<int>(), mg->grid, mg->merged.as<float>()));
    } else {
        float *held = mg->partial.as<float>();
        for (int r = 0; r < plan.rounds; r++) {
            const vrmg_round &rd = plan.round[r];
            ST_NCCL(st, ncclGroupStart());
            ST_NCCL(st, ncclSend(held + rd.send_offset * kPartialFloats, rd.count * kPartialFloats, ncclFloat, rd.partner, mg->comm, s));
            ST_NCCL(st, ncclRecv(mg->parts.p, rd.count * kPartialFloats, ncclFloat, rd.partner, mg->comm, s));
            ST_NCCL(st, ncclGroupEnd());
            float *out = rd.out_buffer == 2 ? mg->merged.as<float>() : mg->work[rd.out_buffer].as<float>();
            ST_VR(st, vr_composite_pair(mg->ctx, width, height, rd.first_pixel, rd.count, held + rd.keep_offset * kPartialFloats,
                                        mg->parts.as<float>(), rd.axis, rd.mine_is_low, out));
            held = out;
        }
        if (plan.rounds == 0)
            ST_HIP(st, hipMemcpyAsync(mg->merged.p, mg->partial.p, span * kPartialFloats * fl, hipMemcpyDeviceToDevice, s));
    }

    // 3. bytes of the span; pixels the merged numbers cannot decide are listed
    ST_VR(st, vr_decide_partials(mg->ctx, width, height, first, span, mg->merged.as<float>(), mg->bytes.as<uint8_t>(), mg->undecided.as<uint32_t>(),
                                 mg->count.as<uint32_t>()));

    // 4. the listed pixels of all spans, re-marched in the shader's order, the colour handed from box to box.  The
    // ranks exchange how many pixels each has listed -- the one host wait of a frame, the re-march is sized from it
    // -- and, in the same message, whether steps 1 to 3 went through: a frame that failed anywhere ends here, on
    // every rank.
    ST_HIP(st, hipMemsetAsync(mg->count.as<uint32_t>() + 1, st.code != VR_OK ? 0xff : 0, sizeof(uint32_t), s));
    std::vector<uint32_t> counts((size_t)world * 2, 0);
    ST_NCCL(st, ncclAllGather(mg->count.p, mg->counts_all.p, 2, ncclUint32, mg->comm, s));
    ST_HIP(st, hipMemcpyAsync(counts.data(), mg->counts_all.p, counts.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    ST_HIP(st, hipStreamSynchronize(s));
    if (st.code != VR_OK) return st.code;
    uint64_t total = 0;
    uint32_t longest = 0;
    for (int r = 0; r < world; r++) {
        if (counts[2 * r + 1] != 0) return fail(mg, VR_ERR_HIP, "the frame failed on rank " + std::to_string(r));
        if (counts[2 * r] > span) return fail(mg, VR_ERR_STATE, "a rank listed more pixels than its span holds");
        total += counts[2 * r];
        longest = std::max(longest, counts[2 * r]);
    }
    if (remarched) *remarched = total;
    if (total > 0) {
        if (total > mg->agreed_chain || longest > mg->agreed_longest) {
            // grown with room to spare, so that a slowly growing list does not mean an agreement every frame
            const size_t chain = std::max<size_t>(total + total / 2, 4096), per_rank = std::min<size_t>(std::max<size_t>(longest + longest / 2, 1024), span);
            ST_HIP(st, mg->lists.fit((size_t)world * per_rank * 4));
            ST_HIP(st, mg->pixels.fit(chain * 4));
            ST_HIP(st, mg->colour[0].fit(chain * 16));
            ST_HIP(st, mg->colour[1].fit(chain * 16));
            if (int rc = agree(mg, s, st.code, "re-march buffers")) return rc;
            mg->agreed_chain = chain;
            mg->agreed_longest = per_rank;
        }
        ST_NCCL(st, ncclAllGather(mg->undecided.p, mg->lists.p, longest, ncclUint32, mg->comm, s));
        size_t at = 0;
        for (int r = 0; r < world; r++) {
            if (counts[2 * r])
                ST_HIP(st, hipMemcpyAsync(mg->pixels.as<uint32_t>() + at, mg->lists.as<uint32_t>() + (size_t)r * longest, (size_t)counts[2 * r] * 4,
                                          hipMemcpyDeviceToDevice, s));
            at += counts[2 * r];
        }
        ST_HIP(st, hipMemsetAsync(mg->colour[0].p, 0, total * 16, s));
        int cur = 0;
        for (int stage = 0; stage < world; stage++) {
            ST_VR(st, vr_chain_stage(mg->ctx, width, height, mg->pixels.as<uint32_t>(), total, mg->colour[cur].as<float>(),
                                     mg->colour[cur ^ 1].as<float>(), stage, mg->grid, plan.box));
            // exactly one rank wrote a non-zero colour for each pixel: the sum hands it on
            ST_NCCL(st, ncclAllReduce(mg->colour[cur ^ 1].p, mg->colour[cur ^ 1].p, total * 4, ncclFloat, ncclSum, mg->comm, s));
            cur ^= 1;
        }
        ST_VR(st, vr_store_chain(mg->ctx, mg->pixels.as<uint32_t>(), mg->colour[cur].as<float>(), total, first, span, mg->bytes.as<uint8_t>()));
    }

    // gather the spans on rank 0
    ST_NCCL(st, ncclGroupStart());
    ST_NCCL(st, ncclSend(mg->bytes.p, span * 4, ncclUint8, 0, mg->comm, s));
    if (rank == 0)
        for (int r = 0; r < world; r++)
            ST_NCCL(st, ncclRecv(mg->frame.as<uint8_t>() + span_first(r, world, schedule, padded) * 4, span * 4, ncclUint8, r, mg->comm, s));
    ST_NCCL(st, ncclGroupEnd());
    if (rank == 0) ST_HIP(st, hipMemcpyAsync(d_frame, mg->frame.p, pixels * 4, hipMemcpyDeviceToDevice, s));
    ST_HIP(st, hipEventRecord(mg->last_done, s));
    mg->have_frame = true;
    return st.code;
}

}  // extern "C"
