// Ray-march kernels for gfx950: occupancy classification, the reference-order march and the
// front-to-back march with empty-space skipping and early termination.
//
// Both marches evaluate a tap exactly as Applications/VolumeRendering/Shader/VolumeRendering.comp
// does (same position, same filter arithmetic, same transfer-function fetch), so the alpha and
// colour of a tap are bit-identical in both.  They differ in which taps they visit and in how
// they combine them.
#pragma once

#include "vr_kernels.hpp"

namespace vr {

// ---- what the marches need to know about the current transfer function ----------------------
struct LutInfo {
    // bad_prefix[i]: number of entries j < i of the 260-entry LDS table (kLutLds) that are "bad"
    // for skipping: alpha != 0 or a non-finite colour.  A run of good entries means every
    // intensity filtered from them has alpha exactly 0 and blends as the identity.
    uint32_t bad_prefix[kLutLds + 1];
    float cmin[3], cmax[3];  // range of the colour channels over table and border
    float kappa[3];          // per channel: bound of the rounding error one blend step adds
    uint32_t fast_ok;        // all alphas in [0,1] and all colours finite
    uint32_t pad_;
};

constexpr float kUlpHalf = 5.9604644775390625e-08f;  // 2^-24

// One block.  Derives LutInfo from the table and the border colour.
__global__ __launch_bounds__(256) void k_lut_derive(const float4 *__restrict__ lut, float4 border, LutInfo *info)
{
    __shared__ uint32_t s_bad[kLutLds];
    __shared__ float s_min[3][256], s_max[3][256];
    __shared__ uint32_t s_ok[256];
    const int tid = threadIdx.x;
    float mn[3] = {border.x, border.y, border.z}, mx[3] = {border.x, border.y, border.z};
    uint32_t ok = 1;
    for (int i = tid; i < kLutLds; i += 256) {
        float4 v = border;
        if (i >= 2 && i < 258) v = lut[i - 2];
        const bool finite = isfinite(v.x) && isfinite(v.y) && isfinite(v.z);
        s_bad[i] = (v.w != 0.0f || !finite) ? 1u : 0u;
        if (!finite || !(v.w >= 0.0f && v.w <= 1.0f)) ok = 0;
        const float ch[3] = {v.x, v.y, v.z};
        for (int c = 0; c < 3; c++) {
            if (ch[c] < mn[c]) mn[c] = ch[c];
            if (ch[c] > mx[c]) mx[c] = ch[c];
        }
    }
    for (int c = 0; c < 3; c++) {
        s_min[c][tid] = mn[c];
        s_max[c][tid] = mx[c];
    }
    s_ok[tid] = ok;
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0;
        for (int i = 0; i < kLutLds; i++) {
            info->bad_prefix[i] = run;
            run += s_bad[i];
        }
        info->bad_prefix[kLutLds] = run;
        uint32_t all_ok = 1;
        float lo[3] = {s_min[0][0], s_min[1][0], s_min[2][0]}, hi[3] = {s_max[0][0], s_max[1][0], s_max[2][0]};
        for (int k = 0; k < 256; k++) {
            all_ok &= s_ok[k];
            for (int c = 0; c < 3; c++) {
                if (s_min[c][k] < lo[c]) lo[c] = s_min[c][k];
                if (s_max[c][k] > hi[c]) hi[c] = s_max[c][k];
            }
        }
        for (int c = 0; c < 3; c++) {
            // a filtered colour stays inside [lo, hi] up to one rounding; widen a little
            const float mag = fmaxf(fabsf(lo[c]), fabsf(hi[c]));
            const float slack = mag * 1e-6f + 1e-30f;
            info->cmin[c] = lo[c] - slack;
            info->cmax[c] = hi[c] + slack;
            // One blend step C' = fl(fl(a*c) + fl(fl(1-a)*C)) with |c|,|C| <= mag adds at most
            // 2^-24 * mag * (a + 2(1-a) + 1) <= 3 * 2^-24 * mag; 3.25 covers second-order terms.
            info->kappa[c] = 3.25f * kUlpHalf * (mag + slack);
        }
        info->fast_ok = all_ok;
        info->pad_ = 0;
    }
}

// Range of LDS-table entries a filtered intensity between lo and hi (normalised) can read,
// widened by one entry on each side.
__device__ __forceinline__ void tf_entry_range(float lo, float hi, int &first, int &last)
{
    const float xl = floorf(lo * 256.0f - 0.5f), xh = floorf(hi * 256.0f - 0.5f);
    first = safe_index(xl, 256) + 2 - 1;
    last = safe_index(xh, 256) + 3 + 1;
    first = first < 0 ? 0 : first;
    last = last > kLutLds - 1 ? kLutLds - 1 : last;
}

// One bit per brick: set when the transfer function is exactly transparent (and finite) over the
// whole value range of the brick's 9^3 voxels, i.e. for every tap whose filter footprint comes
// from this brick.
__global__ __launch_bounds__(256) void k_classify_bricks(const uint32_t *__restrict__ cells, uint32_t brick_count,
                                                        float inv_hi, float inv_lo, const LutInfo *__restrict__ info,
                                                        uint32_t *__restrict__ empty_fine)
{
    const uint32_t b = blockIdx.x * 256u + threadIdx.x;
    bool empty = false;
    if (b < brick_count) {
        const uint32_t mm = cells[b];
        const float lo = (float)(mm & 0xffffu), hi = (float)(mm >> 16);
        int first, last;
        tf_entry_range(__builtin_fmaf(lo, inv_hi, lo * inv_lo), __builtin_fmaf(hi, inv_hi, hi * inv_lo), first, last);
        empty = info->bad_prefix[last + 1] == info->bad_prefix[first];
    }
    const unsigned long long mask = __ballot(empty);
    const int lane = threadIdx.x & 63;
    const uint32_t word = (blockIdx.x * 256u + (threadIdx.x & ~63u)) >> 5;
    if (lane == 0 && word * 32u < brick_count) empty_fine[word] = (uint32_t)mask;
    if (lane == 32 && (word + 1) * 32u < brick_count) empty_fine[word + 1] = (uint32_t)(mask >> 32);
}

// One bit per supercell of 4^3 bricks: set when all of its bricks inside the grid are empty.
__global__ __launch_bounds__(256) void k_classify_coarse(const uint32_t *__restrict__ empty_fine, int nbx, int nby, int nbz,
                                                        int ncx, int ncy, int ncz, uint32_t *__restrict__ empty_coarse)
{
    const uint32_t sc = blockIdx.x * 256u + threadIdx.x;
    const uint32_t count = (uint32_t)(ncx * ncy * ncz);
    bool empty = false;
    if (sc < count) {
        const int cx = sc % ncx, cy = (sc / ncx) % ncy, cz = sc / (ncx * ncy);
        empty = true;
        for (int dz = 0; dz < 4 && empty; dz++)
            for (int dy = 0; dy < 4 && empty; dy++) {
                const int bz = cz * 4 + dz, by = cy * 4 + dy;
                if (bz >= nbz || by >= nby) continue;
                for (int dx = 0; dx < 4; dx++) {
                    const int bx = cx * 4 + dx;
                    if (bx >= nbx) continue;
                    const uint32_t b = (uint32_t)((bz * nby + by) * nbx + bx);
                    if (!((empty_fine[b >> 5] >> (b & 31)) & 1u)) empty = false;
                }
            }
    }
    const unsigned long long mask = __ballot(empty);
    const int lane = threadIdx.x & 63;
    const uint32_t word = (blockIdx.x * 256u + (threadIdx.x & ~63u)) >> 5;
    if (lane == 0 && word * 32u < count) empty_coarse[word] = (uint32_t)mask;
    if (lane == 32 && (word + 1) * 32u < count) empty_coarse[word + 1] = (uint32_t)(mask >> 32);
}

// ---- stepping over empty space ----------------------------------------------------------------
// Along a ray the padded texel coordinate g = u*size - 0.5 + kPad is linear in the tap index.
// dir holds its change per march step; inv the reciprocals.  steps_inside() returns how many
// consecutive taps, starting with the current one, are certain to have their low corner in the
// axis-aligned region [lo, lo + extent)^3: the region is shrunk by `margin` texels on every
// side, far more than the rounding difference between this linear model and the positions the
// taps really get, so the count never includes a tap outside the region.
struct MarchDir {
    float dx, dy, dz;
    float ix, iy, iz;
    float margin;
};

__device__ __forceinline__ MarchDir march_dir(const Ray &r, const VolumeView &v, float step, float sign)
{
    MarchDir m;
    // one step towards the eye (tap index - 1) moves the position by +step*d; u = x + .5,
    // v = y + .5, w = -z inside the box (VolumeRendering.comp:168,187)
    m.dx = sign * (step * r.d[0] * v.fx);
    m.dy = sign * (step * r.d[1] * v.fy);
    m.dz = sign * -(step * r.d[2] * v.fz);
    m.ix = m.dx != 0.0f ? 1.0f / m.dx : 0.0f;
    m.iy = m.dy != 0.0f ? 1.0f / m.dy : 0.0f;
    m.iz = m.dz != 0.0f ? 1.0f / m.dz : 0.0f;
    const float big = fmaxf(v.fx, fmaxf(v.fy, v.fz));
    m.margin = fmaxf(0.02f, big * 1e-5f);
    return m;
}

__device__ __forceinline__ float axis_steps(float g, float lo, float extent, float d, float inv, float margin)
{
    if (d > 0.0f) return (lo + extent - margin - g) * inv;
    if (d < 0.0f) return (lo + margin - g) * inv;
    return 1e9f;
}

__device__ __forceinline__ int steps_inside(const Tap &p, int lox, int loy, int loz, float extent, const MarchDir &m)
{
    float n = axis_steps(p.x + (float)kPad, (float)lox, extent, m.dx, m.ix, m.margin);
    n = fminf(n, axis_steps(p.y + (float)kPad, (float)loy, extent, m.dy, m.iy, m.margin));
    n = fminf(n, axis_steps(p.z + (float)kPad, (float)loz, extent, m.dz, m.iz, m.margin));
    n = fminf(n, 1e9f);
    return 1 + ((n > 0.0f) ? (int)n : 0);
}

__device__ __forceinline__ bool coarse_empty(const Cell &c, const VolumeView &v)
{
    const uint32_t sc = (uint32_t)(((c.sz >> 5) * v.ncy + (c.sy >> 5)) * v.ncx + (c.sx >> 5));
    return (v.empty_coarse[sc >> 5] >> (sc & 31)) & 1u;
}

__device__ __forceinline__ bool fine_empty(uint32_t brick, const VolumeView &v)
{
    return (v.empty_fine[brick >> 5] >> (brick & 31)) & 1u;
}

// ---- reference-order march ---------------------------------------------------------------------
// Back to front over the samples + 2 taps exactly as the shader loops (VolumeRendering.comp:
// 183-202), same blend expression, so the colour is bit-identical to it.  With SKIP, a tap whose
// brick (or supercell) is marked empty is not fetched: its alpha is exactly 0 and the blend
// a*c + (1-a)*C leaves C unchanged bit for bit.  The back-most tap (t == 0) is always fetched
// because the shader assigns its colour instead of blending.
// With QUEUE the kernel marches the pixels listed in `queue` (count in ctr->replay_count) instead
// of a rectangle; the fast march uses this to re-do rays whose byte it could not decide.
template <typename T, bool SKIP, bool STATS, bool QUEUE>
__global__ __launch_bounds__(256) void k_march_exact(VolumeView vol, FrameParams fp, const float4 *__restrict__ lut,
                                                    float4 border, uint32_t *__restrict__ out, Counters *ctr,
                                                    uint32_t *touched, const uint32_t *__restrict__ queue)
{
    __shared__ float4 s_lut[kLutLds];
    stage_lut(s_lut, lut, border);
    uint32_t entry = blockIdx.x * 256u + threadIdx.x;
    const uint32_t stride = gridDim.x * 256u;
    const uint32_t entries = QUEUE ? ctr->replay_count : 1u;
    for (; QUEUE ? entry < entries : true; entry += stride) {
        int px, py;
        uint32_t out_index;
        if (QUEUE) {
            out_index = queue[entry];
            pixel_of_index(fp, out_index, px, py);
        } else {
            if (!lane_pixel(fp, px, py, out_index)) return;
        }
        const Ray r = make_ray(fp, px, py);
        float cr = 0.0f, cg = 0.0f, cb = 0.0f;
        unsigned long long evaluated = 0, lookups = 0;
        if (r.hit) {
            const int last = fp.samples + 1;
            const float step = (r.tmax - r.tmin) / (float)(uint32_t)last;
            const MarchDir md = march_dir(r, vol, step, -1.0f);
            int t = 0;
            while (t <= last) {
                const Tap p = tap_position(r, vol, step, t);
                const Cell c = tap_cell(p, vol);
                const uint32_t brick = brick_of(c, vol);
                if (SKIP && t > 0) {
                    if (STATS) lookups++;
                    if (coarse_empty(c, vol)) {
                        t += steps_inside(p, c.sx & ~31, c.sy & ~31, c.sz & ~31, 32.0f, md);
                        continue;
                    }
                    if (fine_empty(brick, vol)) {
                        t += steps_inside(p, c.sx & ~7, c.sy & ~7, c.sz & ~7, 8.0f, md);
                        continue;
                    }
                }
                const float inten = sample_volume<T>(c, brick, vol);
                const float4 tf = sample_tf(s_lut, inten);
                if (t == 0) {
                    cr = tf.x; cg = tf.y; cb = tf.z;
                } else {
                    const float a = tf.w, ia = 1.0f - tf.w;
                    cr = a * tf.x + ia * cr;
                    cg = a * tf.y + ia * cg;
                    cb = a * tf.z + ia * cb;
                }
                if (STATS) {
                    evaluated++;
                    atomicOr(&touched[brick >> 5], 1u << (brick & 31));
                }
                t++;
            }
        }
        out[out_index] = pack_rgba8(cr, cg, cb);
        if (STATS) {
            if (!QUEUE) atomicAdd(&ctr->rays_hit, r.hit ? 1ull : 0ull);
            if (QUEUE) atomicAdd(&ctr->replay_rays, 1ull);
            atomicAdd(&ctr->samples_evaluated, evaluated);
            atomicAdd(&ctr->cells_read, lookups);
        }
        if (!QUEUE) break;
    }
}

// ---- front-to-back march ------------------------------------------------------------------------
// Visits the taps from the eye towards the back, steps over empty bricks and supercells, and
// stops a ray as soon as its RGBA8 value is decided.
//
// The shader's result for a ray is C = sum_t Tf(t) a_t c_t + Tf(0) c_0 with Tf(t) the product of
// (1 - a_s) over the taps s in front of t, evaluated back to front in float.  This kernel
// accumulates F = sum over the visited taps and the transmittance Tr in double (exact for this
// purpose) and keeps, per channel, an interval that must contain the shader's float result:
//     [F + Tr*cmin - E,  F + Tr*cmax + E],   E = kappa * (S + Tr * taps_left),
// where the taps not yet visited contribute Tr times a weighted mean of colours in [cmin,cmax]
// (alphas lie in [0,1]), S is the sum of Tr over the visited taps with a != 0 and kappa bounds
// the rounding error one float blend step adds (attenuated by everything in front of it; steps
// with a == 0 are exact).  float -> unorm8 is monotone, so when both interval ends convert to the
// same byte in all three channels the pixel is final.  A ray whose bytes are still undecided at
// the back tap is appended to a queue and re-marched in reference order by k_march_exact.
// The loop leaves a wave when no lane has a tap left: lanes first step over empty space
// (cheap, divergent), then all lanes that found a tap fetch it together.
template <typename T, bool STATS>
__global__ __launch_bounds__(256) void k_march_fast(VolumeView vol, FrameParams fp, const float4 *__restrict__ lut,
                                                   float4 border, const LutInfo *__restrict__ info,
                                                   uint32_t *__restrict__ out, Counters *ctr, uint32_t *touched,
                                                   uint32_t *__restrict__ queue)
{
    __shared__ float4 s_lut[kLutLds];
    stage_lut(s_lut, lut, border);
    int px, py;
    uint32_t pixel_id;
    const bool in_rect = lane_pixel(fp, px, py, pixel_id);

    const Ray r = make_ray(fp, in_rect ? px : fp.x0, in_rect ? py : fp.y0);
    const bool fast_ok = info->fast_ok != 0;
    const float cmin[3] = {info->cmin[0], info->cmin[1], info->cmin[2]};
    const float cmax[3] = {info->cmax[0], info->cmax[1], info->cmax[2]};
    const float kappa[3] = {info->kappa[0], info->kappa[1], info->kappa[2]};

    const int last = fp.samples + 1;
    const float step = (r.tmax - r.tmin) / (float)(uint32_t)last;
    const MarchDir md = march_dir(r, vol, step, 1.0f);

    double F0 = 0.0, F1 = 0.0, F2 = 0.0, Tr = 1.0;
    float S = 0.0f;
    int t = last;                       // front-most tap
    bool marching = in_rect && r.hit;   // still has taps with index >= 1 to visit
    bool decided = false;
    uint32_t rgba = 0xff000000u;        // miss: clear colour (VolumeRendering.comp:174,180)
    unsigned long long evaluated = 0, lookups = 0;

    // is the byte of every channel fixed, given the taps still to come?
    auto try_decide = [&](int taps_left) -> bool {
        const double tr = Tr;
        const float spread = S * 1.001f + (float)tr * (float)taps_left;
        uint32_t bytes = 0xff000000u;
        bool same = true;
        const double Fc[3] = {F0, F1, F2};
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const double e = (double)kappa[c] * (double)spread;
            double lo = Fc[c] + tr * (double)cmin[c] - e;
            double hi = Fc[c] + tr * (double)cmax[c] + e;
            lo -= fabs(lo) * 1e-7 + 1e-38;      // survive the rounding of the conversion to float
            hi += fabs(hi) * 1e-7 + 1e-38;
            const uint32_t bl = to_unorm8((float)lo), bh = to_unorm8((float)hi);
            same = same && (bl == bh);
            bytes |= bl << (8 * c);
        }
        if (same) rgba = bytes;
        return same;
    };

    while (true) {
        // 1. step over empty space until this lane stands on a tap that must be fetched
        bool have = false;
        Cell c;
        uint32_t brick = 0;
        while (marching && !have) {
            if (t < 1) { marching = false; break; }
            const Tap p = tap_position(r, vol, step, t);
            c = tap_cell(p, vol);
            brick = brick_of(c, vol);
            if (STATS) lookups++;
            if (coarse_empty(c, vol)) {
                t -= steps_inside(p, c.sx & ~31, c.sy & ~31, c.sz & ~31, 32.0f, md);
            } else if (fine_empty(brick, vol)) {
                t -= steps_inside(p, c.sx & ~7, c.sy & ~7, c.sz & ~7, 8.0f, md);
            } else {
                have = true;
            }
        }
        // 2. the wave leaves the loop when no lane has a tap to fetch
        if (__ballot(have) == 0ull) break;
        if (have) {
            const float inten = sample_volume<T>(c, brick, vol);
            const float4 tf = sample_tf(s_lut, inten);
            if (STATS) {
                evaluated++;
                atomicOr(&touched[brick >> 5], 1u << (brick & 31));
            }
            if (tf.w != 0.0f) {
                const double a = (double)tf.w;
                const double w = Tr * a;
                F0 = __builtin_fma(w, (double)tf.x, F0);
                F1 = __builtin_fma(w, (double)tf.y, F1);
                F2 = __builtin_fma(w, (double)tf.z, F2);
                S += (float)Tr;
                Tr = Tr * (1.0 - a);
                // early termination: worth testing once little light can still come through
                if (fast_ok && Tr < 0.03125 && try_decide(t)) {
                    decided = true;
                    marching = false;
                }
            }
            t -= 1;
        }
    }

    if (in_rect) {
        if (r.hit && !decided) {
            // back-most tap: its colour enters unweighted by its alpha (VolumeRendering.comp:189-191)
            const Tap p = tap_position(r, vol, step, 0);
            const Cell c = tap_cell(p, vol);
            const uint32_t brick = brick_of(c, vol);
            const float4 tf = sample_tf(s_lut, sample_volume<T>(c, brick, vol));
            if (STATS) {
                evaluated++;
                atomicOr(&touched[brick >> 5], 1u << (brick & 31));
            }
            F0 = __builtin_fma(Tr, (double)tf.x, F0);
            F1 = __builtin_fma(Tr, (double)tf.y, F1);
            F2 = __builtin_fma(Tr, (double)tf.z, F2);
            Tr = 0.0;
            decided = fast_ok && try_decide(0);
            if (!decided) queue[atomicAdd(&ctr->replay_count, 1u)] = pixel_id;
        }
        if (!r.hit || decided) out[pixel_id] = rgba;
        if (STATS) {
            atomicAdd(&ctr->rays_hit, r.hit ? 1ull : 0ull);
            atomicAdd(&ctr->samples_evaluated, evaluated);
            atomicAdd(&ctr->cells_read, lookups);
        }
    }
}

// Rays of the rectangle that hit the box (the reference marches samples + 2 taps on each).
__global__ __launch_bounds__(256) void k_count_hits(FrameParams fp, unsigned long long *hits)
{
    int px, py;
    uint32_t out_index;
    const bool hit = lane_pixel(fp, px, py, out_index) && make_ray(fp, px, py).hit;
    const unsigned long long mask = __ballot(hit);
    if ((threadIdx.x & 63) == 0 && mask) atomicAdd(hits, (unsigned long long)__popcll(mask));
}

}  // namespace vr
