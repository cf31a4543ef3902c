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

/* Sort-first frame.  d_frame: width*height*4 bytes of device memory on rank 0 (ignored elsewhere), complete
 * on the context's stream when the call returns there (vr_synchronize to wait on the host). */
int vrmg_render_sort_first(vrmg *mg, int width, int height, uint8_t *d_frame);

/* Sort-last: the box grid (grid[0]*grid[1]*grid[2] == world); rank r holds box (r % gx, (r / gx) % gy,
 * r / (gx*gy)).  The boxes may be of unequal size as long as they are cut z planes first, then y planes per
 * z layer, then x planes per (z, y) cell; the context must already hold its box. */
int vrmg_set_box_grid(vrmg *mg, const int grid[3]);
/* Sort-last frame.  d_frame as above; remarched (may be NULL) receives the number of pixels of the frame
 * that went through the box-to-box re-march.  Waits on the host once per frame (for that number). */
int vrmg_render_sort_last(vrmg *mg, int width, int height, int schedule, uint8_t *d_frame, uint64_t *remarched);

#ifdef __cplusplus
}
#endif
#endif
