// Sort-last rendering for gfx950: the volume is cut into boxes of bricks, one per GPU; every GPU
// marches the taps that fall into its box and the per-pixel results are combined afterwards.
// The reference drives one GPU (one queue, Source/VulkanApplication.cpp:316-321), so nothing here
// has a counterpart there beyond the shader loop itself (Applications/VolumeRendering/Shader/
// VolumeRendering.comp:183-202), whose frame all of this has to reproduce.
//
// Ownership.  Tap t of a ray belongs to the box that holds the low corner of its filter footprint
// (the padded cell tap_cell() returns); bricks carry a one-voxel apron, so the owner has all eight
// corners.  Boxes tile the padded brick grid, so every tap has exactly one owner, the back-most tap
// (t == 0, whose colour enters unweighted) included.
//
// Partial result of a box for a pixel, five floats (F0, F1, F2, Tr, S): with the box's taps taken
// front to back, F = sum Tf(t) a_t c_t, Tr = prod (1 - a_t), S = sum of Tf(t) over its taps with
// a_t != 0 (Tf(t): product of 1 - a over the box's taps in front of t).  The owner of tap 0 adds
// Tr * c_0 and sets Tr = 0.  A ray that misses the volume has all five zero.  Two partials whose
// boxes a plane separates combine, front f and back b, as
//     F = F_f + Tr_f F_b,   Tr = Tr_f Tr_b,   S = S_f + Tr_f S_b,
// which is associative, so any order of pairwise merges that respects front/back gives the ray's
// totals.  The totals decide the RGBA8 value exactly as in k_march_fast; the few pixels they cannot
// decide are re-marched in reference order by passing the float colour from box to box, back to
// front (k_chain_exact), so the bytes equal the single-GPU frame.
#pragma once

#include "vr_march.hpp"

namespace vr {

// Whether the texel coordinate of an axis grows with the tap index, i.e. towards the eye
// (sign of RayModel::b).  Every rank evaluates these expressions on the same inputs, so all agree.
struct Towards {
    bool x, y, z;
};
__device__ __forceinline__ Towards march_direction(const Ray &r, float step, float fx, float fy, float fz)
{
    Towards u;
    u.x = -(step * r.d[0]) * fx >= 0.0f;
    u.y = -(step * r.d[1]) * fy >= 0.0f;
    u.z = (step * r.d[2]) * fz >= 0.0f;
    return u;
}

// Boxes form a gx x gy x gz grid.  Along a ray each texel coordinate is monotone in the tap index,
// so the boxes a ray visits have, axis by axis, non-decreasing grid coordinates once an axis the ray
// descends along is mirrored; the lexicographic rank of the mirrored coordinates is therefore a
// back-to-front order valid for every ray with these directions.
struct BoxGrid {
    int gx, gy, gz;
};
__device__ __forceinline__ int visit_rank(const Towards &u, const BoxGrid &g, int ci, int cj, int ck)
{
    const int i = u.x ? ci : g.gx - 1 - ci, j = u.y ? cj : g.gy - 1 - cj, k = u.z ? ck : g.gz - 1 - ck;
    return (k * g.gy + j) * g.gx + i;
}
// Linear index (ck * gy + cj) * gx + ci of the box with the given back-to-front rank.
__device__ __forceinline__ int box_of_visit_rank(const Towards &u, const BoxGrid &g, int rank)
{
    const int i = rank % g.gx, j = (rank / g.gx) % g.gy, k = rank / (g.gx * g.gy);
    const int ci = u.x ? i : g.gx - 1 - i, cj = u.y ? j : g.gy - 1 - j, ck = u.z ? k : g.gz - 1 - k;
    return (ck * g.gy + cj) * g.gx + ci;
}

__device__ __forceinline__ float ray_step(const Ray &r, int samples)
{
    return (r.tmax - r.tmin) / (float)(uint32_t)(samples + 1);
}

// ---- partial march ------------------------------------------------------------------------------
// One lane per ray, front to back over the taps this context owns: the lane march of the whole-volume
// path (march_lanes) with the ownership test, without the decision.  out: kPartial floats per pixel.
// A lane that has spent `budget` memory round trips with more to go hands its ray to k_partial_long.
#ifndef VR_PARTIAL_MIN_WAVES
#define VR_PARTIAL_MIN_WAVES 7
#endif
template <typename T>
__global__ __launch_bounds__(256, VR_PARTIAL_MIN_WAVES) void k_march_partial(VolumeView vol, FrameParams fp, const float4 *__restrict__ lut,
                                                      float4 border, const LutInfo *__restrict__ info,
                                                      float *__restrict__ out, Counters *ctr,
                                                      LongRay *__restrict__ long_queue, int budget,
                                                      const uint2 *__restrict__ tiles, int tiles_x)
{
    __shared__ float4 s_lut[kLutLds];
    stage_lut(s_lut, lut, border);
    const Decider dec(info);
    int px, py;
    uint32_t pixel_id;
    const bool in_rect = lane_pixel(fp, px, py, pixel_id);
    march_lanes<T, false, true>(vol, fp, s_lut, dec, nullptr, ctr, nullptr, nullptr, long_queue, &ctr->long_count, budget, in_rect,
                                px, py, pixel_id, out, tiles, tiles_x);
}

// Continuation of the rays k_march_partial ran out of budget on, one wave per ray (continue_ray).
template <typename T>
__global__ __launch_bounds__(256) void k_partial_long(VolumeView vol, FrameParams fp, const float4 *__restrict__ lut,
                                                     float4 border, const LutInfo *__restrict__ info,
                                                     float *__restrict__ out, Counters *ctr,
                                                     const LongRay *__restrict__ long_queue)
{
    __shared__ float4 s_lut[kLutLds];
    stage_lut(s_lut, lut, border);
    const Decider dec(info);
    for_queue_entries<1>(ctr->long_count, [&](uint32_t entry, bool valid) {
        const LongRay e = long_queue[entry];
        uint32_t rgba;
        int dark_from;
        continue_ray<T, 64, false, true>(vol, fp, s_lut, dec, ctr, nullptr, e, valid, rgba, dark_from, out);
    });
}

// ---- combining partial results --------------------------------------------------------------------
struct PartialD {
    double f0, f1, f2, tr, s;
};
__device__ __forceinline__ PartialD load_partial(const float *p)
{
    return PartialD{(double)p[0], (double)p[1], (double)p[2], (double)p[3], (double)p[4]};
}
__device__ __forceinline__ void under(PartialD &front, const PartialD &back)
{
    front.f0 = __builtin_fma(front.tr, back.f0, front.f0);
    front.f1 = __builtin_fma(front.tr, back.f1, front.f1);
    front.f2 = __builtin_fma(front.tr, back.f2, front.f2);
    front.s = __builtin_fma(front.tr, back.s, front.s);
    front.tr = front.tr * back.tr;
}
__device__ __forceinline__ void store_partial(float *p, const PartialD &v)
{
    p[0] = (float)v.f0;
    p[1] = (float)v.f1;
    p[2] = (float)v.f2;
    p[3] = (float)v.tr;
    p[4] = (float)v.s * 1.0000002f;   // S is an upper bound: never round it down
}

// Pixel `index` of the frame, row-major; false for the padding behind the last pixel.
__device__ __forceinline__ bool ray_of_pixel(const FrameParams &fp, uint32_t index, Ray &r)
{
    if (index >= (uint32_t)fp.width * (uint32_t)fp.height) return false;
    r = make_ray(fp, (int)(index % (uint32_t)fp.width), (int)(index / (uint32_t)fp.width));
    return true;
}

// Two partial images of the pixels [first, first + count) whose boxes lie on the two sides of a plane
// normal to `axis`; `mine_low`: the box of `mine` is on the side of the smaller coordinates.
__global__ __launch_bounds__(256) void k_composite_pair(FrameParams fp, float fx, float fy, float fz, uint32_t first,
                                                       uint32_t count, const float *__restrict__ mine,
                                                       const float *__restrict__ theirs, int axis, int mine_low,
                                                       float *__restrict__ out)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= count) return;
    Ray r;
    PartialD res{0.0, 0.0, 0.0, 0.0, 0.0};
    if (ray_of_pixel(fp, first + i, r) && r.hit) {
        const Towards u = march_direction(r, ray_step(r, fp.samples), fx, fy, fz);
        const bool up = axis == 0 ? u.x : (axis == 1 ? u.y : u.z);
        // coordinate growing towards the eye: the high side is in front
        const bool mine_front = (mine_low != 0) != up;
        const PartialD a = load_partial(mine + (size_t)i * kPartial), b = load_partial(theirs + (size_t)i * kPartial);
        res = mine_front ? a : b;
        under(res, mine_front ? b : a);
    }
    store_partial(out + (size_t)i * kPartial, res);
}

// All boxes' partial images of the pixels [first, first + count) at once.  parts: one image of count
// pixels per slot; slot_of_box[(ck * gy + cj) * gx + ci]: the slot holding the image of box (ci, cj, ck).
__global__ __launch_bounds__(256) void k_composite_ordered(FrameParams fp, float fx, float fy, float fz, uint32_t first,
                                                          uint32_t count, const float *__restrict__ parts,
                                                          const int *__restrict__ slot_of_box, BoxGrid grid,
                                                          float *__restrict__ out)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= count) return;
    Ray r;
    PartialD res{0.0, 0.0, 0.0, 0.0, 0.0};
    if (ray_of_pixel(fp, first + i, r) && r.hit) {
        const Towards u = march_direction(r, ray_step(r, fp.samples), fx, fy, fz);
        const int boxes = grid.gx * grid.gy * grid.gz;
        res.tr = 1.0;
        for (int rank = boxes - 1; rank >= 0; rank--) {   // front to back
            const int slot = slot_of_box[box_of_visit_rank(u, grid, rank)];
            under(res, load_partial(parts + ((size_t)slot * count + i) * kPartial));
        }
    }
    store_partial(out + (size_t)i * kPartial, res);
}

// Final bytes of the pixels [first, first + count) from their combined partials.  The interval
// argument of k_march_fast with two more terms: the floats the partials travelled in (at most 16
// roundings of values bounded by the largest colour) and the light the boxes dropped below kTrCut
// (the first box of a ray to stop early leaves out at most 2 * kTrCut * colour; every box behind it
// is attenuated by that box's transmittance again).
// Pixels the interval cannot decide are appended to `list` (frame pixel indices).
__global__ __launch_bounds__(256) void k_decide_partials(FrameParams fp, const LutInfo *__restrict__ info, uint32_t first,
                                                        uint32_t count, const float *__restrict__ partials,
                                                        uint32_t *__restrict__ out, uint32_t *__restrict__ list,
                                                        uint32_t *list_count)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= count) return;
    const float *p = partials + (size_t)i * kPartial;
    const float trf = p[3];
    const float taps = (float)(fp.samples + 2);
    const float spread = (p[4] * 1.001f + (trf + 2.0f * (float)kTrCut) * taps) * 1.000001f;
    uint32_t bytes = 0xff000000u;
    bool same = info->fast_ok != 0;
    // the interval ends in float, pushed outwards by more than forming them can lose (5 u of their
    // magnitude, as in Decider::decide)
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float cmin = info->cmin[c], cmax = info->cmax[c];
        const float mag = fmaxf(fabsf(cmin), fabsf(cmax));
        const float e = (info->kappa[c] * spread + (16.0f * kUlpHalf + 2.5f * (float)kTrCut) * mag) * 1.000001f;
        const float m = fabsf(p[c]) + trf * mag + e;
        const float slack = __builtin_fmaf(m, 5.0f * kUlpHalf, 1e-38f);
        const float lo = (__builtin_fmaf(trf, cmin, p[c]) - e) - slack;
        const float hi = (__builtin_fmaf(trf, cmax, p[c]) + e) + slack;
        const uint32_t bl = to_unorm8(lo), bh = to_unorm8(hi);
        same = same && (bl == bh);
        bytes |= bl << (8 * c);
    }
    bool listed = false;
    if (!same) {
        // with a table the interval argument does not hold for, every ray that hits is re-marched
        Ray r;
        listed = ray_of_pixel(fp, first + i, r) && r.hit;
        if (!listed) bytes = 0xff000000u;
    }
    // one addition per wave reserves the list entries of its pixels (additions on one word take their turn)
    const unsigned long long votes = __ballot(listed);
    if (votes) {
        const int lane = threadIdx.x & 63;
        uint32_t base = 0;
        if (lane == __builtin_ctzll(votes)) base = atomicAdd(list_count, (uint32_t)__popcll(votes));
        base = (uint32_t)__builtin_amdgcn_readlane((int)base, __builtin_ctzll(votes));
        if (listed) list[base + (uint32_t)__popcll(votes & ((1ull << lane) - 1ull))] = first + i;
    }
    out[i] = bytes;
}

// ---- reference-order march across boxes --------------------------------------------------------------
// One step of the back-to-front pass over the pixels in `list`: the box whose back-to-front rank for
// a pixel equals `stage` takes the float colour accumulated by the boxes behind it (c_in), blends its
// own taps over it in the shader's order with the shader's expression, and writes the result; every
// other box writes zeros, so that the sum over all boxes of c_out is the colour to carry into the
// next stage.  After the last stage it is the shader's colour.  One wave per ray, as k_replay_exact.
template <typename T>
__global__ __launch_bounds__(256) void k_chain_exact(VolumeView vol, FrameParams fp, const float4 *__restrict__ lut,
                                                    float4 border, const LutInfo *__restrict__ info,
                                                    const uint32_t *__restrict__ list, uint32_t entries,
                                                    const float4 *__restrict__ c_in, float4 *__restrict__ c_out, int stage,
                                                    BoxGrid grid, int ci, int cj, int ck)
{
    __shared__ float4 s_lut[kLutLds];
    stage_lut(s_lut, lut, border);
    const Decider dec(info);
    const int lane = threadIdx.x & 63;
    const uint32_t waves = gridDim.x * 4u;
    for (uint32_t entry = blockIdx.x * 4u + (threadIdx.x >> 6); entry < entries; entry += waves) {
        Ray r;
        const uint32_t pixel = list[entry];
        const bool valid = ray_of_pixel(fp, pixel, r);   // identical in every lane
        const float step = valid ? ray_step(r, fp.samples) : 0.0f;
        const Towards u = march_direction(r, step, vol.fx, vol.fy, vol.fz);
        if (!valid || visit_rank(u, grid, ci, cj, ck) != stage) {
            if (lane == 0) c_out[entry] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            continue;
        }
        const float4 carried = c_in[entry];
        const float in[3] = {carried.x, carried.y, carried.z};
        float res[3];
        replay_ray<T, false, true>(vol, fp, s_lut, dec, nullptr, nullptr, nullptr, pixel, nullptr, 0, in, res);
        if (lane == 0) c_out[entry] = make_float4(res[0], res[1], res[2], 0.0f);
    }
}

// Bytes of the re-marched pixels that fall into [first, first + count).
__global__ __launch_bounds__(256) void k_store_chain(const uint32_t *__restrict__ list, const float4 *__restrict__ colour,
                                                    uint32_t entries, uint32_t first, uint32_t count, uint32_t *__restrict__ out)
{
    const uint32_t e = blockIdx.x * 256u + threadIdx.x;
    if (e >= entries) return;
    const uint32_t pixel = list[e];
    if (pixel < first || pixel - first >= count) return;
    const float4 c = colour[e];
    out[pixel - first] = pack_rgba8(c.x, c.y, c.z);
}

}  // namespace vr
