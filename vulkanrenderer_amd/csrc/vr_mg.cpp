// libvr_mg.so: one frame from several GPUs over RCCL (include/vr_mg.h).  Host code only: every kernel
// is reached through the public C ABI of libvr_hip.so (include/vr_abi.h); the collectives are RCCL
// point-to-point groups, all-gathers and all-reduces issued on the context's own stream.
// The reference renders on one GPU from one queue (Source/VulkanApplication.cpp:316-321): no counterpart.
#include "../../include/vr_mg.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

static_assert(VRMG_UNIQUE_ID_BYTES == sizeof(ncclUniqueId), "the unique id travels as VRMG_UNIQUE_ID_BYTES bytes");

namespace {
constexpr int kPartialFloats = 5;   // F0, F1, F2, Tr, S per pixel (vr_render_partial)

struct Buffer {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t fit(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        const hipError_t e = hipMalloc(&p, bytes);
        if (e == hipSuccess) cap = bytes;
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
    hipStream_t stream = nullptr;
    std::string error;
    int grid[3] = {1, 1, 1};
    bool grid_set = false;
    // sort-first
    Buffer bands, gathered;
    // sort-last
    Buffer partial, parts, work[2], merged, bytes, undecided, count, counts_all, lists, pixels, colour[2], frame, slots;
};

namespace {

int fail(vrmg *mg, int code, const std::string &msg)
{
    if (mg) mg->error = msg;
    return code;
}

#define MG_HIP(mg, call)                                                                           \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) return fail((mg), VR_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define MG_NCCL(mg, call)                                                                          \
    do {                                                                                           \
        ncclResult_t r_ = (call);                                                                  \
        if (r_ != ncclSuccess) return fail((mg), VR_ERR_HIP, std::string(#call) + ": " + ncclGetErrorString(r_)); \
    } while (0)
#define MG_VR(mg, call)                                                                            \
    do {                                                                                           \
        int rc_ = (call);                                                                          \
        if (rc_ != VR_OK) return fail((mg), rc_, std::string(#call) + ": " + vr_last_error((mg)->ctx)); \
    } while (0)

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
    if (vr_synchronize(ctx) != VR_OK || hipGetDevice(&mg->device) != hipSuccess) {   // vr_synchronize makes the context's GPU current
        delete mg;
        return VR_ERR_HIP;
    }
    return VR_OK;
}

}  // namespace

extern "C" {

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
        delete mg;
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
    for (Buffer *b : {&mg->bands, &mg->gathered, &mg->partial, &mg->parts, &mg->work[0], &mg->work[1], &mg->merged, &mg->bytes,
                      &mg->undecided, &mg->count, &mg->counts_all, &mg->lists, &mg->pixels, &mg->colour[0], &mg->colour[1], &mg->frame,
                      &mg->slots})
        b->release();
    if (mg->own_comm && mg->comm) (void)ncclCommDestroy(mg->comm);
    delete mg;
    return VR_OK;
}

const char *vrmg_last_error(const vrmg *mg) { return mg ? mg->error.c_str() : "null object"; }

int vrmg_render_sort_first(vrmg *mg, int width, int height, uint8_t *d_frame)
{
    if (!mg) return VR_ERR_ARG;
    if (mg->rank == 0 && !d_frame) return fail(mg, VR_ERR_ARG, "rank 0 needs a frame buffer");
    const int rows0 = vr_band_rows(height, 0, mg->world);      // the largest part
    if (rows0 <= 0 || width < 2) return fail(mg, VR_ERR_ARG, "bad image size");
    const size_t stride = (size_t)rows0 * (size_t)width * 4u;
    if (vr_synchronize(mg->ctx) != VR_OK) return fail(mg, VR_ERR_HIP, vr_last_error(mg->ctx));   // binds the GPU; buffers below may be reallocated
    MG_HIP(mg, mg->bands.fit(stride));
    if (mg->rank == 0) MG_HIP(mg, mg->gathered.fit(stride * (size_t)mg->world));
    if (vr_band_rows(height, mg->rank, mg->world) > 0)
        MG_VR(mg, vr_render_bands(mg->ctx, width, height, mg->rank, mg->world, mg->bands.as<uint8_t>(), 1, nullptr));
    MG_VR(mg, vr_wait_frame(mg->ctx, mg->stream));     // two frames in flight: the bands are complete on the stream
    // gather on rank 0: one point-to-point transfer per peer, all in one group
    MG_NCCL(mg, ncclGroupStart());
    MG_NCCL(mg, ncclSend(mg->bands.p, stride, ncclUint8, 0, mg->comm, mg->stream));
    if (mg->rank == 0)
        for (int p = 0; p < mg->world; p++)
            MG_NCCL(mg, ncclRecv(mg->gathered.as<uint8_t>() + (size_t)p * stride, stride, ncclUint8, p, mg->comm, mg->stream));
    MG_NCCL(mg, ncclGroupEnd());
    if (mg->rank == 0)
        MG_VR(mg, vr_assemble_bands(mg->ctx, width, height, mg->world, mg->gathered.as<uint8_t>(), stride, d_frame));
    return VR_OK;
}

int vrmg_set_box_grid(vrmg *mg, const int grid[3])
{
    if (!mg || !grid) return VR_ERR_ARG;
    if (grid[0] < 1 || grid[1] < 1 || grid[2] < 1 || (long long)grid[0] * grid[1] * grid[2] != mg->world)
        return fail(mg, VR_ERR_ARG, "the box grid must have one box per rank");
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
    std::vector<int> rounds;   // binary swap: the axis merged in each round (x pairs first, then y, then z)
    if (schedule == VRMG_BINARY_SWAP) {
        for (int axis = 0; axis < 3; axis++) {
            if (mg->grid[axis] & (mg->grid[axis] - 1)) return fail(mg, VR_ERR_ARG, "binary swap needs a power-of-two number of boxes per axis");
            for (int g = mg->grid[axis]; g > 1; g >>= 1) rounds.push_back(axis);
        }
    }
    const uint64_t pixels = (uint64_t)width * (uint64_t)height;
    const uint64_t unit = (uint64_t)world * 64u;
    const uint64_t padded = (pixels + unit - 1) / unit * unit;
    const uint64_t span = padded / (uint64_t)world;
    uint64_t first = (uint64_t)rank * span, span_check = span;
    if (schedule == VRMG_BINARY_SWAP) swap_span(rank, world, padded, first, span_check);
    const size_t fl = sizeof(float);
    const int box[3] = {rank % mg->grid[0], (rank / mg->grid[0]) % mg->grid[1], rank / (mg->grid[0] * mg->grid[1])};

    if (vr_synchronize(mg->ctx) != VR_OK) return fail(mg, VR_ERR_HIP, vr_last_error(mg->ctx));   // binds the GPU; buffers may be reallocated
    MG_HIP(mg, mg->partial.fit(padded * kPartialFloats * fl));
    MG_HIP(mg, mg->merged.fit(span * kPartialFloats * fl));
    MG_HIP(mg, mg->bytes.fit(span * 4));
    MG_HIP(mg, mg->undecided.fit(span * 4));
    MG_HIP(mg, mg->count.fit(4));
    MG_HIP(mg, mg->counts_all.fit((size_t)world * 4));
    if (rank == 0) MG_HIP(mg, mg->frame.fit(padded * 4));
    // rows past the last pixel are padding: vr_render_partial leaves them alone, the merges read them as misses
    MG_HIP(mg, hipMemsetAsync(mg->partial.as<float>() + pixels * kPartialFloats, 0, (padded - pixels) * kPartialFloats * fl, mg->stream));

    // 1. this box's partial image
    MG_VR(mg, vr_render_partial(mg->ctx, width, height, mg->partial.as<float>(), 1));

    // 2. exchange + merge in visibility order
    if (schedule == VRMG_DIRECT_SEND) {
        MG_HIP(mg, mg->parts.fit((size_t)world * span * kPartialFloats * fl));
        MG_HIP(mg, mg->slots.fit((size_t)world * sizeof(int)));
        std::vector<int> slots(world);
        for (int r = 0; r < world; r++) slots[r] = r;   // slot s of the exchange holds rank s's image; a rank's box index is its rank
        MG_HIP(mg, hipMemcpyAsync(mg->slots.p, slots.data(), slots.size() * sizeof(int), hipMemcpyHostToDevice, mg->stream));
        MG_HIP(mg, hipStreamSynchronize(mg->stream));   // `slots` lives on this stack frame
        const size_t chunk = span * kPartialFloats;
        MG_NCCL(mg, ncclGroupStart());
        for (int p = 0; p < world; p++) {
            MG_NCCL(mg, ncclSend(mg->partial.as<float>() + (size_t)p * chunk, chunk, ncclFloat, p, mg->comm, mg->stream));
            MG_NCCL(mg, ncclRecv(mg->parts.as<float>() + (size_t)p * chunk, chunk, ncclFloat, p, mg->comm, mg->stream));
        }
        MG_NCCL(mg, ncclGroupEnd());
        MG_VR(mg, vr_composite_ordered(mg->ctx, width, height, first, span, mg->parts.as<float>(), mg->slots.as<int>(), mg->grid, mg->merged.as<float>()));
    } else {
        MG_HIP(mg, mg->work[0].fit(padded / 2 * kPartialFloats * fl + 16));
        MG_HIP(mg, mg->work[1].fit(padded / 2 * kPartialFloats * fl + 16));
        MG_HIP(mg, mg->parts.fit(padded / 2 * kPartialFloats * fl + 16));   // what the partner sends
        float *held = mg->partial.as<float>();
        uint64_t at = 0, count = padded;
        for (size_t r = 0; r < rounds.size(); r++) {
            const int partner = rank ^ (1 << r);
            const bool upper = (rank >> r) & 1;
            count /= 2;
            const uint64_t keep = upper ? count : 0, send = upper ? 0 : count;   // offsets inside `held`
            MG_NCCL(mg, ncclGroupStart());
            MG_NCCL(mg, ncclSend(held + send * kPartialFloats, count * kPartialFloats, ncclFloat, partner, mg->comm, mg->stream));
            MG_NCCL(mg, ncclRecv(mg->parts.p, count * kPartialFloats, ncclFloat, partner, mg->comm, mg->stream));
            MG_NCCL(mg, ncclGroupEnd());
            float *out = count == span ? mg->merged.as<float>() : mg->work[r & 1].as<float>();
            MG_VR(mg, vr_composite_pair(mg->ctx, width, height, at + keep, count, held + keep * kPartialFloats, mg->parts.as<float>(),
                                        rounds[r], upper ? 0 : 1, out));
            held = out;
            at += keep;
        }
        if (rounds.empty())
            MG_HIP(mg, hipMemcpyAsync(mg->merged.p, mg->partial.p, span * kPartialFloats * fl, hipMemcpyDeviceToDevice, mg->stream));
        if (at != first || count != span) return fail(mg, VR_ERR_STATE, "binary swap left an unexpected span");
    }

    // 3. bytes of the span; pixels the merged numbers cannot decide are listed
    MG_VR(mg, vr_decide_partials(mg->ctx, width, height, first, span, mg->merged.as<float>(), mg->bytes.as<uint8_t>(), mg->undecided.as<uint32_t>(),
                                 mg->count.as<uint32_t>()));

    // 4. the listed pixels of all spans, re-marched in the shader's order, the colour handed from box to box
    std::vector<uint32_t> counts(world, 0);
    MG_NCCL(mg, ncclAllGather(mg->count.p, mg->counts_all.p, 1, ncclUint32, mg->comm, mg->stream));
    MG_HIP(mg, hipMemcpyAsync(counts.data(), mg->counts_all.p, (size_t)world * 4, hipMemcpyDeviceToHost, mg->stream));
    MG_HIP(mg, hipStreamSynchronize(mg->stream));
    uint64_t total = 0;
    uint32_t longest = 0;
    for (uint32_t c : counts) {
        total += c;
        longest = std::max(longest, c);
    }
    if (remarched) *remarched = total;
    if (total > 0) {
        MG_HIP(mg, mg->lists.fit((size_t)world * longest * 4));
        MG_HIP(mg, mg->pixels.fit(total * 4));
        MG_HIP(mg, mg->colour[0].fit(total * 16));
        MG_HIP(mg, mg->colour[1].fit(total * 16));
        MG_NCCL(mg, ncclAllGather(mg->undecided.p, mg->lists.p, longest, ncclUint32, mg->comm, mg->stream));
        size_t at = 0;
        for (int r = 0; r < world; r++) {
            if (counts[r])
                MG_HIP(mg, hipMemcpyAsync(mg->pixels.as<uint32_t>() + at, mg->lists.as<uint32_t>() + (size_t)r * longest, (size_t)counts[r] * 4,
                                          hipMemcpyDeviceToDevice, mg->stream));
            at += counts[r];
        }
        MG_HIP(mg, hipMemsetAsync(mg->colour[0].p, 0, total * 16, mg->stream));
        int cur = 0;
        for (int stage = 0; stage < world; stage++) {
            MG_VR(mg, vr_chain_stage(mg->ctx, width, height, mg->pixels.as<uint32_t>(), total, mg->colour[cur].as<float>(),
                                     mg->colour[cur ^ 1].as<float>(), stage, mg->grid, box));
            // exactly one rank wrote a non-zero colour for each pixel: the sum hands it on
            MG_NCCL(mg, ncclAllReduce(mg->colour[cur ^ 1].p, mg->colour[cur ^ 1].p, total * 4, ncclFloat, ncclSum, mg->comm, mg->stream));
            cur ^= 1;
        }
        MG_VR(mg, vr_store_chain(mg->ctx, mg->pixels.as<uint32_t>(), mg->colour[cur].as<float>(), total, first, span, mg->bytes.as<uint8_t>()));
    }

    // gather the spans on rank 0
    MG_NCCL(mg, ncclGroupStart());
    MG_NCCL(mg, ncclSend(mg->bytes.p, span * 4, ncclUint8, 0, mg->comm, mg->stream));
    if (rank == 0)
        for (int r = 0; r < world; r++) {
            uint64_t f = (uint64_t)r * span, c = span;
            if (schedule == VRMG_BINARY_SWAP) swap_span(r, world, padded, f, c);
            MG_NCCL(mg, ncclRecv(mg->frame.as<uint8_t>() + f * 4, span * 4, ncclUint8, r, mg->comm, mg->stream));
        }
    MG_NCCL(mg, ncclGroupEnd());
    if (rank == 0) MG_HIP(mg, hipMemcpyAsync(d_frame, mg->frame.p, pixels * 4, hipMemcpyDeviceToDevice, mg->stream));
    return VR_OK;
}

}  // extern "C"
