/*
 * C ABI of the MI355X direct-volume-rendering core (libvr_hip.so).
 *
 * The library replaces the VolumeRendering application's GPU path of the reference — the
 * Vulkan compute pipeline that the application records every frame — by HIP kernels for
 * gfx950.  Each entry point names the reference code it stands in for (paths relative to the
 * reference checkout).  Conventions: every function returns 0 on success or a negative
 * VR_ERR_* code; the message of the last failure of a context is available from
 * vr_last_error(); no exception crosses this boundary (the reference throws
 * std::runtime_error and catches once in VulkanApplication::run, Source/VulkanApplication.cpp:
 * 188-217).  A context belongs to one GPU and is not thread-safe, like the reference's single
 * render thread; different contexts may be used from different threads.  Host pointers are
 * borrowed for the duration of the call only.  All work is issued on the context's stream.
 */
#ifndef VR_ABI_H
#define VR_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VR_OK 0
#define VR_ERR_ARG (-1)      /* invalid argument */
#define VR_ERR_HIP (-2)      /* a HIP runtime call failed */
#define VR_ERR_STATE (-3)    /* call order violated (e.g. render before upload) */
#define VR_ERR_IO (-4)       /* file system error while reading slices */

#define VR_DTYPE_U8 1
#define VR_DTYPE_U16 2

/* Render strategies.  Both produce the frame the reference shader defines. */
#define VR_MODE_REFERENCE_ORDER 0 /* back-to-front, every tap (VolumeRendering.comp:183-202) */
#define VR_MODE_FAST 1            /* front-to-back, macrocell skipping, early termination; for a table under which
                                     rays do not become opaque (more than one in a hundred of the pixels of a whole
                                     frame undecided) the following frames take VR_MODE_REFERENCE_SKIP's path */
#define VR_MODE_REFERENCE_SKIP 2  /* back-to-front, taps in exactly transparent bricks not fetched */

typedef struct vr_ctx vr_ctx;

/* One control point of a transfer-function control graph: ControlPoint{position.x, intensity},
 * Applications/VolumeRendering/Source/main.cpp:101-105 (position.y only feeds intensity). */
typedef struct vr_ctrl_pt {
    float x;
    float intensity;
} vr_ctrl_pt;

/* Work and traffic counters of one frame (filled only when a stats pointer is passed; the
 * counting variant of the kernels is slower than the plain one). */
typedef struct vr_stats {
    uint64_t rays;              /* pixels rendered by this call */
    uint64_t rays_hit;          /* rays that hit the volume box */
    uint64_t samples_nominal;   /* rays_hit * (samples_per_ray + 2): taps the reference executes */
    uint64_t samples_evaluated; /* trilinear taps actually fetched */
    uint64_t cells_read;        /* macrocell (min,max) entries read */
    uint64_t bricks_touched;    /* distinct 8^3 bricks holding a voxel of an evaluated tap */
    uint64_t replay_rays;       /* rays re-marched in reference order by the fast mode */
    uint64_t algorithmic_bytes; /* 1024 B (u16; 512 B u8) * bricks_touched + 4 B * distinct cells
                                   + LUT + 4 B * rays */
    float kernel_ms;            /* device time of the frame's kernels (hipEvent) */
    uint32_t pad_;
    uint64_t long_rays;         /* rays the lane-per-ray march handed to the wave-per-ray march */
    /* fast mode, where the frame's taps were fetched: */
    uint64_t long_samples;      /* taps fetched by the wave-per-ray continuation of handed-over rays */
    uint64_t replay_samples;    /* taps fetched by the reference-order replay */
    uint64_t main_bricks;       /* distinct bricks the lane-per-ray march alone touched */
    uint32_t rays_by_lane_taps[24]; /* hit rays by floor(log2(taps the lane-per-ray march fetched for them + 1)) */
    uint32_t rays_by_wave_taps[24]; /* handed-over rays by floor(log2(taps the wave-per-ray march fetched + 1)) */
} vr_stats;

/* Context life cycle.  Stands in for initComputePipeline / postRenderLoop,
 * main.cpp:203-573 and 1246-1283. */
int vr_create(int device, vr_ctx **out);
int vr_destroy(vr_ctx *ctx);
const char *vr_last_error(const vr_ctx *ctx);
/* The HIP stream all work of the context is issued on (a hipStream_t). */
void *vr_stream(vr_ctx *ctx);
int vr_synchronize(vr_ctx *ctx);

/* Volume upload: loadVolumeData's normalisation (main.cpp:144,161-164) + the staging copy into
 * the 3-D image (main.cpp:260-280, 601-613).  voxels: nx*ny*nz values, x fastest, then y, then
 * slice, little-endian.  max_value <= 0: use the data maximum, as the reference does; a
 * positive value must not be below the data maximum.  The volume is re-tiled into 8^3 bricks
 * and a per-brick (min,max) macrocell grid is built.  vr_upload_volume takes a host pointer,
 * vr_upload_volume_device a device pointer on the context's GPU. */
int vr_upload_volume(vr_ctx *ctx, const void *voxels, int dtype, int nx, int ny, int nz, float max_value);
int vr_upload_volume_device(vr_ctx *ctx, const void *d_voxels, int dtype, int nx, int ny, int nz, float max_value);

/* loadVolumeData, main.cpp:112-165: every file of the directory, ordered by the integer after
 * the last dot (or the whole name), headerless big-endian u16, width = height =
 * floor(sqrt(values in the first slice)).  Byte swap, maximum and re-tiling run on the GPU. */
int vr_load_volume_slices(vr_ctx *ctx, const char *directory);
int vr_volume_info(const vr_ctx *ctx, int *nx, int *ny, int *nz, int *dtype, float *max_value);
/* Normalised float copy of the resident volume (what the reference keeps in `intensities`,
 * main.cpp:39), written to a host buffer of nx*ny*nz floats.  For checks. */
int vr_read_volume_normalized(vr_ctx *ctx, float *out);

/* prepareHistogramVolumeData, main.cpp:166-181: the 100-bin histogram of the normalised voxels the
 * transfer-function editor draws behind its control graphs (bin i counts the values in
 * [i/100, (i + 0.9)/100], all bins divided by the largest count).  The voxel values are counted on
 * the GPU; out receives 100 floats. */
int vr_histogram(vr_ctx *ctx, float out[100]);

/* updateTransferFunction, main.cpp:643-686, plus the upload of the table, main.cpp:687-736.
 * points[c] / counts[c]: control graph of channel c (R,G,B,A), sorted by x.  The table lives on
 * the GPU and persists between calls exactly like `transferFunction` (main.cpp:110); entries no
 * segment covers keep their previous value.  reset_state != 0 zeroes it first.  counts[c] == 0
 * leaves channel c untouched.  Meant to be called every frame, as renderCommands does (main.cpp:800-804):
 * graphs equal to the last call's, which leave every bit of the table as it is, return without touching
 * the GPU; changed graphs of up to 256 points in all travel with the kernel launch.  A change that leaves
 * the set of transparent table entries as it is (colours, opacities between non-zero values) does not
 * disturb frames in flight: the new table is written beside the one they read.  A change of that set orders
 * the next frame behind them, and the tables the fast modes skip by are brought up to date on the GPU, for
 * the bricks the change concerns, when that frame is rendered.  The host waits for nothing (more than 256
 * points: one copy it waits for). */
int vr_set_transfer_function(vr_ctx *ctx, const vr_ctrl_pt *const points[4], const int counts[4], int reset_state);
int vr_get_lut(vr_ctx *ctx, float out[1024]);
/* getIntensity, main.cpp:948-953 */
float vr_control_point_intensity(float y, float child_height);

/* Camera: setupViewPyramid + updatePixelTransformationUniforms, main.cpp:182-202, 737-799.
 * horizontal in [-pi, pi], vertical in [-pi/2, pi/2] (the ImGui sliders, main.cpp:1210-1211). */
int vr_set_camera(vr_ctx *ctx, float horizontal_rad, float vertical_rad);
/* Directly sets what the shader reads: ViewPyramid.location and PixelTransform.toWorld
 * (VolumeRendering.comp:12-22).  to_world is column-major and is used for every image size. */
int vr_set_camera_raw(vr_ctx *ctx, const float location[4], const float to_world[16]);
/* The two uniform blocks for a width x height image: 80-byte ViewPyramid
 * {height, location, side, up, forward} and the 64-byte toWorld matrix. */
int vr_get_camera(vr_ctx *ctx, int width, int height, float view_pyramid[20], float to_world[16]);
/* The same uniforms without a context (pure host arithmetic, usable without a GPU). */
int vr_camera_uniforms(float horizontal_rad, float vertical_rad, int width, int height,
                       float view_pyramid[20], float to_world[16]);

/* samples_per_ray: the shader constant `samples` (VolumeRendering.comp:173, 200 there); a ray
 * takes samples_per_ray + 2 taps.  tf_border_rgba: border colour of the sampler as seen by the
 * transfer-function fetch (main.cpp:380), NULL keeps the current one (default 0,0,0,0). */
int vr_set_sampling(vr_ctx *ctx, int samples_per_ray, const float tf_border_rgba[4]);
int vr_set_mode(vr_ctx *ctx, int mode);

/* Local illumination: the gradient normal (computeNormal, VolumeRendering.comp:109-119) and the
 * Blinn-Phong term (shading, :123-152) that the shader adds to a tap's colour at :195-198 -- lines its
 * author has commented out, so the reference itself renders with enabled = 0, the default here.
 * scene = the push constants `Scene` {ambient, diffuse, specular, shininess} (main.cpp:80-98, pushed
 * at :808; sliders :1212-1215 keep them in [0,1]^3 x [1,128], enforced here); NULL = the reference's
 * values {1, 1, 1, 100}.  Light at (0,0,-1), light colour 1, ambient light 0.3 as in the shader.
 * normalize() of a zero vector, undefined in GLSL, yields the zero vector (a tap in a flat
 * neighbourhood gets the ambient term only).  Works in every mode; frames stay identical between
 * the modes.  Contexts holding a sub-volume refuse it (the gradient reads beyond the box's apron). */
int vr_set_shading(vr_ctx *ctx, int enabled, const float scene[4]);

/* One frame: the dispatch of VolumeRendering.comp (main.cpp:823-828) and its imageStore
 * (VolumeRendering.comp:205).  out: width*height*4 bytes RGBA8, row-major, pixel (0,0) first;
 * a host pointer (out_is_device == 0, copied back and synchronised) or a device pointer
 * (out_is_device != 0, asynchronous on the context's stream).  stats may be NULL. */
int vr_render(vr_ctx *ctx, int width, int height, uint8_t *out, int out_is_device, vr_stats *stats);
/* The same for the pixel rectangle [x0,x1) x [y0,y1) of the width x height frame: out holds
 * (x1-x0)*(y1-y0)*4 bytes, rectangle rows contiguous.  For sort-first tiling. */
int vr_render_rect(vr_ctx *ctx, int width, int height, int x0, int y0, int x1, int y1,
                   uint8_t *out, int out_is_device, vr_stats *stats);

/* One frame without any allocation.  No counterpart in the reference (whose frame loop owns all its
 * Vulkan objects up front, main.cpp:203-573, and never runs under a capture); it is for callers that
 * cannot allocate device memory: code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc.  Same frame as vr_render(ctx, width, height, d_out, 1, NULL) with VR_MODE_FAST, whatever mode
 * is set, never pipelined; d_out is a device pointer, the call is asynchronous on the context's stream.
 * The scratch a fast frame needs (counters, per-tile depth ranges, the queues of handed-over and of
 * undecided rays) comes from a static device array inside the library (about 70 MB, held by every GPU the
 * library runs on until the process ends) instead of buffers allocated on demand, so the frame is limited
 * to VR_STATIC_MAX_PIXELS pixels and 2 * VR_STATIC_MAX_PIXELS / 64 tiles of 8x8 pixels (VR_ERR_ARG
 * beyond).  The call allocates and frees nothing, on the device or through any library; the volume and
 * the transfer function must have been set before (their calls allocate).  The call enqueues kernels and
 * memsets only: no copy, no wait (the first frame after a change of table, border colour, shading or volume
 * rebuilds the occupancy tables on the stream, in front of the march).  All contexts on one GPU share that
 * GPU's array: calls that overlap on one GPU, from two contexts or two streams, race. */
#define VR_STATIC_MAX_PIXELS (1280 * 720) /* the reference's window, main.cpp:30 */
int vr_render_static(vr_ctx *ctx, int width, int height, uint8_t *d_out);
int vr_static_max_pixels(void);

/* Frames for a consumer on the host.  The reference hands every frame on with a device-side image copy into
 * the image its window presents (copyImage raycastedImage -> presentedImgui, main.cpp:903-909); a host
 * program that presents, encodes or stores frames needs them in host memory instead, and vr_render with a
 * host pointer pays a synchronous copy into pageable memory for that (several times the render time at 4K).
 * vr_render_mapped renders a frame (current mode; with vr_set_pipeline(ctx, 2) in flight like any other)
 * into one of three slots the context owns -- a device image and a page-locked host image each -- and
 * enqueues the copy to the host on a stream of its own, so the copy of frame n runs under the march of
 * frame n + 1; the call returns at once with the slot in *frame_id.  vr_map_frame waits until that frame has
 * arrived and yields its width*height*4 bytes (RGBA8, row-major) in page-locked memory, valid until
 * vr_unmap_frame; a slot is used again by the third vr_render_mapped after it, which answers VR_ERR_STATE
 * while the slot is still mapped.  Typical loop: render_mapped(n); map(n - 1); consume; unmap(n - 1).
 * Measured at 3840x2160 on the 2048^3 phantom (examples/orbit): 1.58 ms per frame on the host with one frame
 * at a time, against 6.4 ms through vr_render's host pointer; with two frames in flight the copy ends up in a
 * chain with the previous frame's finishing kernels and a frame takes 2.5 ms, so leave the pipeline at 1 here. */
int vr_render_mapped(vr_ctx *ctx, int width, int height, int *frame_id);
int vr_map_frame(vr_ctx *ctx, int frame_id, const uint8_t **pixels);
int vr_unmap_frame(vr_ctx *ctx, int frame_id);

/* Frames in flight (no counterpart in the reference, which waits on a fence every frame,
 * Source/VulkanApplication.cpp:560-562).  With frames_in_flight == 2, VR_MODE_FAST frames rendered
 * into device memory without a stats pointer are pipelined: the kernels that finish a frame (the few
 * rays the main march hands over) run on a second stream beside the main marches of the next two frames.
 * Rules for the caller: a frame is complete on stream `s` after vr_wait_frame(ctx, s), or on the host
 * after vr_synchronize(); work enqueued directly on vr_stream() is NOT ordered after the last frame.
 * A frame whose output overlaps the output of a frame that is still being finished waits for that frame by
 * itself, so any number of output buffers is correct; three used in turn keep the pipeline full (with two
 * every march waits for the frame before last, about 3 % of the frame time).  Every other entry point
 * first lets the frames in flight finish.  Default: 1 (a frame is complete on the context's stream
 * when vr_render returns; such frames launch their tiles longest first, by the times the previous frame
 * over the same tiles took, which shortens a frame rendered alone by about 8 %). */
int vr_set_pipeline(vr_ctx *ctx, int frames_in_flight);
/* Makes `stream` (a hipStream_t, NULL = the default stream) wait for the most recent frame. */
int vr_wait_frame(vr_ctx *ctx, void *stream);

/* Sort-first split for multi-GPU rendering (no counterpart in the reference, which drives one
 * GPU): the frame's rows are cut into bands of 16 rows, dealt round-robin to `parts` renderers;
 * this call renders the bands b with b % parts == part in one launch and writes them densely,
 * band after band: out holds vr_band_rows(height, part, parts) * width * 4 bytes.  Pixels are
 * independent (VolumeRendering.comp:159 uses only its own invocation id), so the union of the
 * parts is the frame vr_render produces. */
int vr_render_bands(vr_ctx *ctx, int width, int height, int part, int parts,
                    uint8_t *out, int out_is_device, vr_stats *stats);
/* Rows a part owns (>= 0), or VR_ERR_ARG. */
int vr_band_rows(int height, int part, int parts);
/* Puts a frame together from the parts' band buffers after they have been gathered on one GPU (the
 * gather itself is the caller's: one grouped send/recv or gather on its RCCL communicator):
 * d_gathered holds part p's vr_render_bands output at byte offset p * part_stride_bytes
 * (part_stride_bytes >= the largest part, part 0's).  Runs on the context's stream; d_frame gets
 * width*height*4 bytes, the frame vr_render produces.  No counterpart in the reference (one GPU). */
int vr_assemble_bands(vr_ctx *ctx, int width, int height, int parts, const uint8_t *d_gathered,
                      uint64_t part_stride_bytes, uint8_t *d_frame);

/* Device time (ms, HIP events on the context's stream) of the main march kernel of the most
 * recent frames, oldest first: at most `capacity` and at most the last 1024 frames.  Waits for
 * the stream.  Measurement aid (the reference only shows whole-loop FPS in the window title,
 * Source/VulkanApplication.cpp:617-620). */
int vr_march_kernel_times(vr_ctx *ctx, float *ms, int capacity, int *count);
/* Picks the fast mode's one tuning parameter by measurement: renders the current camera's frame (or
 * the sort-first share part/parts of it) in short runs, with the frames in flight the context is set to,
 * for each candidate of the per-lane work budget
 * -- how long a lane marches its ray before a whole wave takes it over -- and keeps the fastest.
 * Frames are identical for every value; only the time changes with volume, table and image size.
 * Without this call the budget follows the size of the launch (small frames hand rays over early).
 * On a context that holds a box of a split volume (vr_upload_subvolume_device) the same is done for the
 * march of vr_render_partial, whose budget is a separate one (part 0 of 1 only).
 * chosen_budget may be NULL.  No counterpart in the reference. */
int vr_tune(vr_ctx *ctx, int width, int height, int part, int parts, int *chosen_budget);

/* Number of rays of the rectangle that hit the volume box; the reference marches
 * samples_per_ray + 2 taps on each of them (VolumeRendering.comp:180-184). */
int vr_count_hit_rays(vr_ctx *ctx, int width, int height, int x0, int y0, int x1, int y1, uint64_t *hits);

/* Synthetic CT phantom used as measurement input (no counterpart in the reference): n^3 voxels
 * written to device memory, then usable with vr_upload_volume_device.  d_out: n^3 * dtype bytes. */
int vr_phantom_generate_device(vr_ctx *ctx, void *d_out, int dtype, int n, uint32_t seed);
/* The voxels [lo, hi) of the same n^3 phantom, x fastest: what one GPU of a sort-last split holds. */
int vr_phantom_generate_box_device(vr_ctx *ctx, void *d_out, int dtype, int n, const int lo[3], const int hi[3],
                                   uint32_t seed);

/* ---- Sort-last rendering: the volume split over several GPUs -------------------------------------
 * No counterpart in the reference, which renders on one GPU from one 3-D image (main.cpp:244-258);
 * the frame produced is still the one VolumeRendering.comp defines.  The volume's grid of 8^3
 * bricks (vr_brick_grid) is cut into boxes, one per context.  Tap t of a ray belongs to the box
 * holding the low corner of its filter footprint, so every tap of the shader loop
 * (VolumeRendering.comp:184-201) is evaluated by exactly one context, with the arithmetic of the
 * single-GPU path.  A frame is made in four steps:
 *   1. vr_render_partial on every context: per pixel 5 floats (F0, F1, F2, Tr, S) - colour sum,
 *      transmittance and rounding-bound weight of the box's taps taken front to back;
 *   2. the partial images are exchanged and merged in visibility order, either pairwise
 *      (vr_composite_pair, binary swap) or all at once (vr_composite_ordered, direct send);
 *   3. vr_decide_partials turns merged partials into RGBA8 and lists the pixels whose bytes the
 *      merged numbers cannot decide (a rounding boundary inside the error interval);
 *   4. those pixels are re-marched in the shader's order, the float colour handed from box to box
 *      back to front: for stage = 0 .. boxes-1 every context calls vr_chain_stage and the outputs
 *      are summed over the contexts (all but one are zero) to give the next stage's input;
 *      vr_store_chain then writes their bytes.
 * Pixels are addressed by their row-major index in the width x height frame; a span may run past
 * the last pixel (padding, rendered as misses).  All pointers named d_* are device pointers on the
 * context's GPU, all calls are asynchronous on the context's stream.  Camera, transfer function and
 * sampling must be set identically on all contexts. */

/* Bricks per axis of an nx x ny x nz volume. */
int vr_brick_grid(int nx, int ny, int nz, int bricks[3]);
/* Voxel box [voxel_lo, voxel_hi) the bricks [brick_lo, brick_hi) store (their 8^3 cells plus a
 * one-voxel apron, clipped to the volume): what vr_upload_subvolume_device expects. */
int vr_subvolume_voxels(int nx, int ny, int nz, const int brick_lo[3], const int brick_hi[3], int voxel_lo[3],
                        int voxel_hi[3]);
/* Like vr_upload_volume_device for the box [brick_lo, brick_hi) of the nx x ny x nz volume's brick
 * grid: d_voxels holds only the voxels vr_subvolume_voxels names, x fastest.  max_value must be the
 * maximum of the whole volume (> 0): main.cpp:161-164 divides every voxel by it.  Afterwards the
 * context renders with vr_render_partial only. */
int vr_upload_subvolume_device(vr_ctx *ctx, const void *d_voxels, int dtype, int nx, int ny, int nz,
                               const int brick_lo[3], const int brick_hi[3], float max_value);
/* For cutting boxes of equal work: the bricks of this context (whole volume or box) that the current
 * transfer function does not make transparent, counted per cube of cell_bricks^3 bricks of the WHOLE
 * volume's brick grid and ADDED to d_counts (device, uint32, ceil(bricks[k] / cell_bricks) cubes per axis,
 * x fastest; the caller clears it and sums the contexts' grids, e.g. with one all-reduce).  Asynchronous on
 * the context's stream. */
int vr_occupied_brick_counts(vr_ctx *ctx, int cell_bricks, uint32_t *d_counts);
/* Step 1.  out: width*height*5 floats, host (synchronous) or device pointer.  Also valid on a
 * context holding the whole volume (one box). */
int vr_render_partial(vr_ctx *ctx, int width, int height, float *out, int out_is_device);
/* Step 2, pairwise: the boxes (or unions of boxes) behind d_mine and d_theirs are separated by a
 * plane normal to `axis` (0 x, 1 y, 2 z in voxel index order); mine_is_low != 0 when d_mine's lies
 * on the side of the smaller indices.  count partials each, for pixels first_pixel ... */
int vr_composite_pair(vr_ctx *ctx, int width, int height, uint64_t first_pixel, uint64_t count, const float *d_mine,
                      const float *d_theirs, int axis, int mine_is_low, float *d_out);
/* Step 2, all at once: d_parts holds one image of count partials per slot; d_slot_of_box[(bz *
 * grid[1] + by) * grid[0] + bx] is the slot with the image of box (bx, by, bz) of the box grid. */
int vr_composite_ordered(vr_ctx *ctx, int width, int height, uint64_t first_pixel, uint64_t count,
                         const float *d_parts, const int *d_slot_of_box, const int grid[3], float *d_out);
/* Step 3.  d_out_rgba8: count*4 bytes; d_undecided: room for count pixel indices;
 * d_undecided_count: one uint32 (cleared by the call). */
int vr_decide_partials(vr_ctx *ctx, int width, int height, uint64_t first_pixel, uint64_t count,
                       const float *d_partials, uint8_t *d_out_rgba8, uint32_t *d_undecided,
                       uint32_t *d_undecided_count);
/* Step 4.  d_pixels: count frame pixel indices (the same list on every context); d_colour_in /
 * d_colour_out: count float4 (rgb + unused); box: this context's position in the box grid. */
int vr_chain_stage(vr_ctx *ctx, int width, int height, const uint32_t *d_pixels, uint64_t count,
                   const float *d_colour_in, float *d_colour_out, int stage, const int grid[3], const int box[3]);
/* Writes the RGBA8 value (imageStore, VolumeRendering.comp:205) of those of the count listed
 * pixels that fall into [first_pixel, first_pixel + pixel_count) to d_out_rgba8[pixel - first_pixel]. */
int vr_store_chain(vr_ctx *ctx, const uint32_t *d_pixels, const float *d_colour, uint64_t count,
                   uint64_t first_pixel, uint64_t pixel_count, uint8_t *d_out_rgba8);

#ifdef __cplusplus
}
#endif
#endif
