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

constexpr int kPartial = 5;             // floats per pixel of a partial result
// A box stops marching a ray once the transmittance of its own taps is below this: what it leaves
// out changes the ray's colour by at most 2 * kTrCut * (largest colour), which k_decide_partials
// adds to its error interval.
constexpr double kTrCut = 0x1p-20;

// Cell of a tap in the coordinates of this context's brick box; false when another box owns the tap.
__device__ __forceinline__ bool tap_cell_local(const Tap &p, const VolumeView &v, Cell &c)
{
    c = tap_cell(p, v);
    c.sx -= v.ox;
    c.sy -= v.oy;
    c.sz -= v.oz;
    return (unsigned)c.sx < (unsigned)(v.nbx * kBrick) && (unsigned)c.sy < (unsigned)(v.nby * kBrick) &&
           (unsigned)c.sz < (unsigned)(v.nbz * kBrick);
}

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
// Writes a ray's partial result; adds the back-most tap first when the box owns it.
template <typename T>
__device__ __forceinline__ void finish_partial(const VolumeView &vol, const Ray &r, float step, const float4 *s_lut,
                                               double F0, double F1, double F2, double Tr, float S, float *o)
{
    if (!(Tr < kTrCut)) {
        // back-most tap, if it is this box's: its colour enters unweighted (VolumeRendering.comp:189-191)
        const Tap p = tap_position(r, vol, step, 0);
        Cell c;
        if (tap_cell_local(p, vol, c)) {
            const float4 tf = sample_tf(s_lut, sample_volume<T>(c, brick_of(c, vol), vol));
            F0 = __builtin_fma(Tr, (double)tf.x, F0);
            F1 = __builtin_fma(Tr, (double)tf.y, F1);
            F2 = __builtin_fma(Tr, (double)tf.z, F2);
            Tr = 0.0;
        }
    }
    o[0] = (float)F0;
    o[1] = (float)F1;
    o[2] = (float)F2;
    o[3] = (float)Tr;
    o[4] = S;
}

// One lane per ray, front to back over the taps this context owns, stepping over empty space like
// k_march_fast.  out: kPartial floats per pixel of the rectangle.  A lane that has fetched `budget`
// taps with more to go hands its ray to k_partial_long.
template <typename T>
__global__ __launch_bounds__(256) void k_march_partial(VolumeView vol, FrameParams fp, const float4 *__restrict__ lut,
                                                      float4 border, float *__restrict__ out, Counters *ctr,
                                                      LongRay *__restrict__ long_queue, int budget)
{
    __shared__ float4 s_lut[kLutLds];
    stage_lut(s_lut, lut, border);
    int px, py;
    uint32_t pixel_id;
    const bool in_rect = lane_pixel(fp, px, py, pixel_id);
    const Ray r = make_ray(fp, in_rect ? px : fp.x0, in_rect ? py : fp.y0);
    const int last = fp.samples + 1;
    const float step = ray_step(r, fp.samples);
    const RayModel rm = ray_model(r, vol, step, -1.0f);

    double F0 = 0.0, F1 = 0.0, F2 = 0.0, Tr = 1.0;
    float S = 0.0f;
    int t = last < rm.t_first ? last : rm.t_first;
    bool marching = in_rect && r.hit;
    uint32_t solid = 0xffffffffu;
    unsigned long long solid_mask = 0;
    bool in_air = true;
    for (int round = 0; round < budget; round++) {
        bool have = false;
        Cell c;
        uint32_t brick = 0;
        while (marching && !have) {
            if (t < rm.t_stop) { marching = false; break; }
            if (in_air) {
                const int n = empty_span(vol, rm, t);
                if (n > 0) {
                    t -= n;
                    continue;
                }
            }
            const Tap p = tap_position(r, vol, step, t);
            if (!tap_cell_local(p, vol, c)) {   // another box's tap (only near the faces of this one)
                t -= 1;
                in_air = true;
                continue;
            }
            brick = brick_of(c, vol);
            if (brick != solid) {
                const int level = empty_level(vol, c.sx >> 3, c.sy >> 3, c.sz >> 3);
                if (level >= 0) {
                    const int n = empty_span_at(vol, rm, t, level, c.sx >> 3, c.sy >> 3, c.sz >> 3);
                    t -= n > 1 ? n : 1;
                    in_air = true;
                    continue;
                }
                solid = brick;
                solid_mask = vol.micro[brick];
            }
            in_air = false;
            if (micro_empty(solid_mask, c)) {
                const int n = block_span(rm, t, c);
                t -= n > 1 ? n : 1;
            } else {
                have = true;
            }
        }
        if (__ballot(have) == 0ull) break;
        if (have) {
            const float4 tf = sample_tf(s_lut, sample_volume<T>(c, brick, vol));
            if (tf.w != 0.0f) {
                const double a = (double)tf.w;
                const double w = Tr * a;
                F0 = __builtin_fma(w, (double)tf.x, F0);
                F1 = __builtin_fma(w, (double)tf.y, F1);
                F2 = __builtin_fma(w, (double)tf.z, F2);
                S += (float)Tr;
                Tr = Tr * (1.0 - a);
                if (Tr < kTrCut) marching = false;
            }
            t -= 1;
        }
    }
    if (!in_rect) return;
    float *o = out + (size_t)pixel_id * kPartial;
    if (!r.hit) {
        o[0] = o[1] = o[2] = o[3] = o[4] = 0.0f;
        return;
    }
    if (marching) {
        // out of budget with taps left: a wave continues this ray (k_partial_long)
        LongRay e;
        e.pixel_id = pixel_id; e.t = t; e.S = S; e.pad_ = 0.0f;
        e.F0 = F0; e.F1 = F1; e.F2 = F2; e.Tr = Tr;
        long_queue[atomicAdd(&ctr->long_count, 1u)] = e;
        return;
    }
    finish_partial<T>(vol, r, step, s_lut, F0, F1, F2, Tr, S, o);
}

// Continuation of the rays k_march_partial ran out of budget on, one wave per ray, 64 consecutive
// taps per round, folded front to back in tap order (the accumulation of k_march_partial).
template <typename T>
__global__ __launch_bounds__(256) void k_partial_long(VolumeView vol, FrameParams fp, const float4 *__restrict__ lut,
                                                     float4 border, float *__restrict__ out, const Counters *ctr,
                                                     const LongRay *__restrict__ long_queue)
{
    __shared__ float4 s_lut[kLutLds];
    stage_lut(s_lut, lut, border);
    const int lane = threadIdx.x & 63;
    const uint32_t waves = gridDim.x * 4u;
    const uint32_t entries = ctr->long_count;
    for (uint32_t entry = blockIdx.x * 4u + (threadIdx.x >> 6); entry < entries; entry += waves) {
        const LongRay e = long_queue[entry];
        int px, py;
        pixel_of_index(fp, e.pixel_id, px, py);
        const Ray r = make_ray(fp, px, py);          // identical in every lane
        const float step = ray_step(r, fp.samples);
        const RayModel rm = ray_model(r, vol, step, -1.0f);
        double F0 = e.F0, F1 = e.F1, F2 = e.F2, Tr = e.Tr;
        float S = e.S;
        int t = e.t;
        if (t > rm.t_first) t = rm.t_first;
        while (t >= rm.t_stop && !(Tr < kTrCut)) {
            const int n = empty_span(vol, rm, t);   // wave-uniform
            if (n > 0) {
                t -= n;
                continue;
            }
            const int tt = t - lane;
            bool fetch = tt >= rm.t_stop;
            Cell c;
            uint32_t brick = 0;
            if (fetch) {
                const Tap p = tap_position(r, vol, step, tt);
                fetch = tap_cell_local(p, vol, c);
                if (fetch) {
                    brick = brick_of(c, vol);
                    if (brick_empty(brick, vol) || micro_empty(vol.micro[brick], c)) fetch = false;
                }
            }
            float4 tf = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (fetch) tf = sample_tf(s_lut, sample_volume<T>(c, brick, vol));
            unsigned long long todo = __ballot(fetch && tf.w != 0.0f);
            while (todo && !(Tr < kTrCut)) {
                const int src = __builtin_ctzll(todo);   // lowest lane = front-most tap of the chunk
                todo &= todo - 1;
                const double a = (double)__shfl(tf.w, src);
                const double w = Tr * a;
                F0 = __builtin_fma(w, (double)__shfl(tf.x, src), F0);
                F1 = __builtin_fma(w, (double)__shfl(tf.y, src), F1);
                F2 = __builtin_fma(w, (double)__shfl(tf.z, src), F2);
                S += (float)Tr;
                Tr = Tr * (1.0 - a);
            }
            t -= 64;
        }
        float res[kPartial];
        finish_partial<T>(vol, r, step, s_lut, F0, F1, F2, Tr, S, res);
        if (lane == 0) {
            float *o = out + (size_t)e.pixel_id * kPartial;
            for (int k = 0; k < kPartial; k++) o[k] = res[k];
        }
    }
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
    const double tr = (double)p[3];
    const float taps = (float)(fp.samples + 2);
    const float spread = p[4] * 1.001f + ((float)tr + 2.0f * (float)kTrCut) * taps;
    uint32_t bytes = 0xff000000u;
    bool same = info->fast_ok != 0;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const double mag = fmax(fabs((double)info->cmin[c]), fabs((double)info->cmax[c]));
        const double e = (double)info->kappa[c] * (double)spread + (16.0 * (double)kUlpHalf + 2.5 * kTrCut) * mag;
        double lo = (double)p[c] + tr * (double)info->cmin[c] - e;
        double hi = (double)p[c] + tr * (double)info->cmax[c] + e;
        lo -= fabs(lo) * 1e-7 + 1e-38;
        hi += fabs(hi) * 1e-7 + 1e-38;
        const uint32_t bl = to_unorm8((float)lo), bh = to_unorm8((float)hi);
        same = same && (bl == bh);
        bytes |= bl << (8 * c);
    }
    if (!same) {
        // with a table the interval argument does not hold for, every ray that hits is re-marched
        Ray r;
        const bool hit = ray_of_pixel(fp, first + i, r) && r.hit;
        if (hit) list[atomicAdd(list_count, 1u)] = first + i;
        else bytes = 0xff000000u;
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
                                                    float4 border, const uint32_t *__restrict__ list, uint32_t entries,
                                                    const float4 *__restrict__ c_in, float4 *__restrict__ c_out, int stage,
                                                    BoxGrid grid, int ci, int cj, int ck)
{
    __shared__ float4 s_lut[kLutLds];
    stage_lut(s_lut, lut, border);
    const int lane = threadIdx.x & 63;
    const uint32_t waves = gridDim.x * 4u;
    for (uint32_t entry = blockIdx.x * 4u + (threadIdx.x >> 6); entry < entries; entry += waves) {
        Ray r;
        const bool valid = ray_of_pixel(fp, list[entry], r);   // identical in every lane
        const float step = valid ? ray_step(r, fp.samples) : 0.0f;
        const Towards u = march_direction(r, step, vol.fx, vol.fy, vol.fz);
        if (!valid || visit_rank(u, grid, ci, cj, ck) != stage) {
            if (lane == 0) c_out[entry] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            continue;
        }
        const float4 carried = c_in[entry];
        float cr = carried.x, cg = carried.y, cb = carried.z;
        if (r.hit) {
            const int last = fp.samples + 1;
            const RayModel rm = ray_model(r, vol, step, 1.0f);
            int base = 0;
            while (base <= last) {
                if (base > 0) {
                    if (base < rm.t_stop) base = rm.t_stop;
                    if (base > rm.t_first || base > last) break;
                    const int n = empty_span(vol, rm, base);
                    if (n > 0) {
                        base += n;
                        continue;
                    }
                }
                const int t = base + lane;
                bool fetch = t <= last;
                Cell c;
                uint32_t brick = 0;
                if (fetch) {
                    const Tap p = tap_position(r, vol, step, t);
                    fetch = tap_cell_local(p, vol, c);
                    if (fetch) {
                        brick = brick_of(c, vol);
                        if (t > 0 && (brick_empty(brick, vol) || micro_empty(vol.micro[brick], c))) fetch = false;
                    }
                }
                float4 tf = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (fetch) tf = sample_tf(s_lut, sample_volume<T>(c, brick, vol));
                const bool identity = tf.w == 0.0f && isfinite(tf.x) && isfinite(tf.y) && isfinite(tf.z);
                unsigned long long todo = __ballot(fetch && (t == 0 || !identity));
                while (todo) {
                    const int src = __builtin_ctzll(todo);
                    todo &= todo - 1;
                    const float a = __shfl(tf.w, src), x = __shfl(tf.x, src), y = __shfl(tf.y, src), z = __shfl(tf.z, src);
                    if (base + src == 0) {
                        cr = x; cg = y; cb = z;
                    } else {
                        const float ia = 1.0f - a;
                        cr = a * x + ia * cr;
                        cg = a * y + ia * cg;
                        cb = a * z + ia * cb;
                    }
                }
                base += 64;
            }
        }
        if (lane == 0) c_out[entry] = make_float4(cr, cg, cb, 0.0f);
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
