/*
 * C ABI of the multi-GPU layer (libvr_mg.so): one frame from several MI355X, one process and one
 * vr_ctx per GPU, exchanges over RCCL (xGMI inside a node).
 *
 * The reference drives one GPU and one queue (Source/VulkanApplication.cpp:316-321) and has no
 * counterpart for any of this; the frame produced is still the one its shader defines
 * (Applications/VolumeRendering/Shader/VolumeRendering.comp:154-206), byte for byte the frame
 * vr_render gives on one GPU.  Every entry point drives the kernels of include/vr_abi.h through that
 * header's public functions and issues its collectives on the context's own stream (vr_stream), so a
 * frame is one ordered queue of kernels and transfers per GPU.  Conventions as in vr_abi.h: 0 or a
 * negative VR_ERR_* code, message from vrmg_last_error, no exceptions, not thread-safe per object.
 *
 * Two ways to split a frame:
 *   sort-first  every rank holds the whole volume (vr_upload_volume*) and renders the 16-row bands b with
 *               b % world == rank; the bands are gathered on rank 0 (point-to-point sends) and put together;
 *   sort-last   every rank holds one box of the volume's brick grid (vr_upload_subvolume_device) and
 *               renders a partial image of it; the partial images are exchanged and merged in
 *               visibility order (all at once, or pairwise in log2(world) rounds), turned into bytes,
 *               the few pixels the merged numbers cannot decide are re-marched from box to box in the
 *               shader's own order, and the spans are gathered on rank 0.
 * Transfer function, camera and sampling are set on every rank's vr_ctx by the caller, identically.
 */
#ifndef VR_MG_H
#define VR_MG_H

#include "vr_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vrmg vrmg;

#define VRMG_UNIQUE_ID_BYTES 128
#define VRMG_DIRECT_SEND 0 /* one exchange: every rank gets all boxes' partials of its span of pixels (all links busy) */
#define VRMG_BINARY_SWAP 1 /* log2(world) pairwise rounds, partner rank ^ (1 << round); power-of-two box grids */

/* ncclGetUniqueId: called on one rank, the bytes are handed to the other ranks by whatever launched them. */
int vrmg_get_unique_id(void *out_bytes);
/* Joins the communicator of `world` ranks (ncclCommInitRank on the context's GPU; collective: every rank
 * calls it).  The object borrows ctx, which must outlive it. */
int vrmg_create(vr_ctx *ctx, const void *unique_id_bytes, int rank, int world, vrmg **out);
/* The same on a communicator the caller already has (an ncclComm_t of the RCCL this library is linked to);
 * it is not destroyed by vrmg_destroy. */
int vrmg_create_from_comm(vr_ctx *ctx, void *nccl_comm, int rank, int world, vrmg **out);
int vrmg_destroy(vrmg *mg);
const char *vrmg_last_error(const vrmg *mg);

/* Errors and peers.  A frame is a fixed sequence of collectives, and a rank that left it half-way would leave
 * its peers waiting in the next one.  Therefore: everything a frame allocates is sized before its first
 * collective, and whenever those sizes grow (first frame, larger image, other schedule, a longer list of
 * re-marched pixels) the ranks agree on the outcome with one small all-reduce before going on -- an allocation
 * that fails on one rank makes every rank return VR_ERR_HIP for that frame.  A kernel launch or RCCL call that
 * fails on one rank afterwards is remembered, the rank still issues the rest of the frame's collectives (its
 * contribution is then undefined) and returns the error at the end; in a sort-last frame the first three steps'
 * status travels with the count exchange, so all ranks return an error for such a frame.  A group of
 * point-to-point calls is always closed. */

/* Sort-first frame.  d_frame: width*height*4 bytes of device memory on rank 0 (ignored elsewhere).  Every
 * peer's bands arrive in one message and are copied into their rows of d_frame, on a stream of the object's own,
 * beside the next frame's march (band buffers of three frames in turn): the frame is complete for `stream`
 * after vrmg_wait_frame(mg, stream) and for the host after vrmg_synchronize(mg), not when the call returns.
 * The host waits for nothing. */
int vrmg_render_sort_first(vrmg *mg, int width, int height, uint8_t *d_frame);
/* Orders `stream` (a hipStream_t, NULL = the default stream) after the most recent frame of either kind. */
int vrmg_wait_frame(vrmg *mg, void *stream);
/* Waits on the host for everything the object and its context have enqueued. */
int vrmg_synchronize(vrmg *mg);

/* Sort-last: the box grid (grid[0]*grid[1]*grid[2] == world); rank r holds box (r % gx, (r / gx) % gy,
 * r / (gx*gy)).  The boxes may be of unequal size as long as they are cut z planes first, then y planes per
 * z layer, then x planes per (z, y) cell; the context must already hold its box. */
int vrmg_set_box_grid(vrmg *mg, const int grid[3]);
/* Sort-last frame.  d_frame as above, complete after vrmg_wait_frame / vrmg_synchronize; remarched (may be
 * NULL) receives the number of pixels of the frame that went through the box-to-box re-march.  Waits on the
 * host once per frame, for that number (the ranks size the re-march from it). */
int vrmg_render_sort_last(vrmg *mg, int width, int height, int schedule, uint8_t *d_frame, uint64_t *remarched);

/* The schedule arithmetic of a sort-last frame as pure host code (no GPU, no communicator): which pixels a
 * rank sends, keeps and merges in every binary-swap round, and the span of pixels it ends up with under
 * either schedule.  vrmg_render_sort_last follows exactly this plan; it is exported so that the plan can be
 * checked for every rank of a world without the GPUs. */
#define VRMG_MAX_ROUNDS 16
typedef struct vrmg_round {
    int partner;          /* rank ^ (1 << round) */
    int axis;             /* the plane between the two halves merged is normal to this axis (x pairs first, then y, then z) */
    int mine_is_low;      /* this rank's boxes lie on the side of the smaller indices */
    int out_buffer;       /* where the merged half goes: 0 / 1 = the two scratch buffers in turn, 2 = the final span */
    uint64_t keep_offset; /* pixels, inside what the rank holds before the round: the half it keeps ... */
    uint64_t send_offset; /* ... and the half it sends */
    uint64_t count;       /* pixels in each half */
    uint64_t first_pixel; /* frame pixel index of the kept half's first pixel */
} vrmg_round;
typedef struct vrmg_plan {
    uint64_t pixels;      /* width * height */
    uint64_t padded;      /* rounded up to a multiple of 64 * world: what is exchanged */
    uint64_t span;        /* padded / world: pixels a rank decides and sends to rank 0 */
    uint64_t first;       /* frame pixel index of this rank's span */
    int box[3];           /* this rank's box in the grid */
    int rounds;           /* binary swap: log2(world); direct send: 0 */
    vrmg_round round[VRMG_MAX_ROUNDS];
} vrmg_plan;
int vrmg_plan_sort_last(int rank, int world, const int grid[3], int schedule, int width, int height, vrmg_plan *plan);

#ifdef __cplusplus
}
#endif
#endif
