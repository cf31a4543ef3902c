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

// Waves per SIMD the plain fast march is compiled for (register budget 512 / this): 7 measured best
// (5 and 6: 4 % slower, 8 spills: 19 % slower); the variants with counters or lighting need more registers.
#ifndef VR_FAST_MIN_WAVES
#define VR_FAST_MIN_WAVES 7
#endif

#ifdef VR_DIAG
__device__ unsigned long long *g_vr_timeline = nullptr;   // kTimelineWords per wave of the lane march (tools/timeline.py)
constexpr int kTimelineWords = 10;
#endif

// ---- what the marches need to know about the current transfer function ----------------------
struct LutInfo {
    // bad_prefix[i]: number of entries j < i of the 260-entry LDS table (kLutLds) that are "bad"
    // for skipping: alpha != 0 or a non-finite colour.  A run of good entries means every
    // intensity filtered from them has alpha exactly 0 and blends as the identity.
    uint32_t bad_prefix[kLutLds + 1];
    float cmin[3], cmax[3];  // range of the colour channels over table and border
    float kappa[3];          // per channel: bound of the rounding error one blend step adds
    uint32_t fast_ok;        // all alphas in [0,1] and all colours finite
    // What the last k_lut_derive found when it compared the table with the one the occupancy tables were
    // built for.  The occupancy of a brick or block depends on the table only through the set of bad entries,
    // so the kernels that rebuild the occupancy tables are always enqueued after a change of table and return
    // at once when `changed` is 0: the host never has to ask.
    uint32_t changed;        // the set of bad entries differs, or the volume is new: rebuild
    int chg_first, chg_last; // entries of the LDS table whose badness changed lie in [chg_first, chg_last]
};

constexpr float kUlpHalf = 5.9604644775390625e-08f;  // 2^-24

// One block.  Derives LutInfo from the table and the border colour.
// shade_hi: largest value the local illumination adds to a tap's colour (0 when it is off).
// prev: what was derived from the table the occupancy tables were built for (may be `info` itself: it is read
// before anything is written).
// force: the occupancy tables have to be rebuilt whatever the table says (new volume; `prev` may be
// uninitialised then).  When a rebuild is due the bounding box of the occupied bricks and the length of the
// list of surface bricks are reset here, in front of the kernels that rebuild them.
__global__ __launch_bounds__(256) void k_lut_derive(const float4 *__restrict__ lut, float4 border, float shade_hi, int force,
                                                   const LutInfo *prev, LutInfo *info, int *occ, uint32_t *occ_count)
{
    __shared__ uint32_t s_bad[kLutLds], s_old[kLutLds];
    __shared__ float s_min[3][256], s_max[3][256];
    __shared__ uint32_t s_ok[256];
    __shared__ int s_first, s_last;
    const int tid = threadIdx.x;
    float mn[3] = {border.x, border.y, border.z}, mx[3] = {border.x, border.y, border.z};
    uint32_t ok = 1;
    for (int i = tid; i < kLutLds; i += 256) {
        float4 v = border;
        if (i >= 2 && i < 258) v = lut[i - 2];
        const bool finite = isfinite(v.x) && isfinite(v.y) && isfinite(v.z);
        s_bad[i] = (v.w != 0.0f || !finite) ? 1u : 0u;
        s_old[i] = force ? 2u : prev->bad_prefix[i + 1] - prev->bad_prefix[i];   // 2: differs from any new value
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
    if (tid == 0) {
        s_first = kLutLds;
        s_last = -1;
    }
    __syncthreads();
    // prefix counts and the range of entries whose badness changed, every lane its own entries
    for (int i = tid; i <= kLutLds; i += 256) {
        uint32_t run = 0;
        for (int j = 0; j < i; j++) run += s_bad[j];
        info->bad_prefix[i] = run;
        if (i < kLutLds && s_bad[i] != s_old[i]) {
            atomicMin(&s_first, i);
            atomicMax(&s_last, i);
        }
    }
    // colour ranges and the all-is-well flag: folded pairwise
    for (int stride = 128; stride > 0; stride >>= 1) {
        __syncthreads();
        if (tid < stride) {
            for (int c = 0; c < 3; c++) {
                if (s_min[c][tid + stride] < s_min[c][tid]) s_min[c][tid] = s_min[c][tid + stride];
                if (s_max[c][tid + stride] > s_max[c][tid]) s_max[c][tid] = s_max[c][tid + stride];
            }
            s_ok[tid] &= s_ok[tid + stride];
        }
    }
    __syncthreads();
    if (tid == 0) {
        const int first = s_first, last = s_last;
        info->changed = last >= 0 ? 1u : 0u;
        info->chg_first = first;
        info->chg_last = last;
        if (last >= 0) {
            occ[0] = occ[1] = occ[2] = 0x7fffffff;   // empty box: lo > hi
            occ[3] = occ[4] = occ[5] = -1;
            *occ_count = 0u;
        }
        const uint32_t all_ok = s_ok[0];
        float lo[3] = {s_min[0][0], s_min[1][0], s_min[2][0]}, hi[3] = {s_max[0][0], s_max[1][0], s_max[2][0]};
        for (int c = 0; c < 3; c++) {
            hi[c] += shade_hi;   // taps in front of the back-most one carry colour + illumination
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
    }
}

// Range of LDS-table entries a tap can read whose eight corners lie between lo and hi (normalised).
// The filter's three levels of fma(w, b - a, a) keep a tap's intensity inside [lo, hi] up to a rounding of
// the difference and one of the result per level, under four units in the last place of hi in all, so the
// range is taken from [lo - hi * 1e-6, hi + hi * 1e-6] with the look-up's own arithmetic (sample_tf), which is
// monotone.  A constant neighbourhood filters to exactly its value and gets exactly the two entries that value
// reads: air at 0 is then the border entry and entry 0, whatever the table does from entry 1 on.
__device__ __forceinline__ void tf_entry_range(float lo, float hi, int &first, int &last)
{
    const float slack = hi * 1e-6f;
    const float xl = floorf((lo - slack) * 256.0f - 0.5f), xh = floorf((hi + slack) * 256.0f - 0.5f);
    first = safe_index(xl, 256) + 2;
    last = safe_index(xh, 256) + 3;
    first = first < 0 ? 0 : first;
    last = last > kLutLds - 1 ? kLutLds - 1 : last;
}

__device__ __forceinline__ void store_ballot(bool flag, uint32_t first_item, uint32_t item_count, uint32_t *bits)
{
    const unsigned long long mask = __ballot(flag);
    const int lane = threadIdx.x & 63;
    const uint32_t word = first_item >> 5;  // first_item is a multiple of 64
    if (lane == 0 && word * 32u < item_count) bits[word] = (uint32_t)mask;
    if (lane == 32 && (word + 1) * 32u < item_count) bits[word + 1] = (uint32_t)(mask >> 32);
}

// Level 0 of the occupancy pyramid: a brick is empty when the transfer function is exactly
// transparent (and finite) over the whole value range of its 9^3 voxels, i.e. for every tap whose
// filter footprint comes from this brick.
__global__ __launch_bounds__(256) void k_classify_bricks(const uint32_t *__restrict__ cells, uint32_t brick_count,
                                                        float inv_hi, float inv_lo, const LutInfo *__restrict__ info,
                                                        uint32_t *__restrict__ empty0)
{
    if (!info->changed) return;
    const uint32_t b = blockIdx.x * 256u + threadIdx.x;
    bool empty = false;
    if (b < brick_count) {
        const uint32_t mm = cells[b];
        const float lo = (float)(mm & 0xffffu), hi = (float)(mm >> 16);
        int first, last;
        tf_entry_range(__builtin_fmaf(lo, inv_hi, lo * inv_lo), __builtin_fmaf(hi, inv_hi, hi * inv_lo), first, last);
        empty = info->bad_prefix[last + 1] == info->bad_prefix[first];
    }
    store_ballot(empty, blockIdx.x * 256u + (threadIdx.x & ~63u), brick_count, empty0);
}

// Level l+1 from level l: a cube is empty when its (up to) 4^3 children inside the grid are.
__global__ __launch_bounds__(256) void k_classify_up(const LutInfo *__restrict__ info, const uint32_t *__restrict__ child, int cnx,
                                                    int cny, int cnz, int px, int py, int pz, uint32_t *__restrict__ parent)
{
    if (!info->changed) return;
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    const uint32_t count = (uint32_t)(px * py * pz);
    bool empty = false;
    if (p < count) {
        const int x = p % px, y = (p / px) % py, z = p / (px * py);
        empty = true;
        for (int dz = 0; dz < 4 && empty; dz++)
            for (int dy = 0; dy < 4 && empty; dy++) {
                const int cz = z * 4 + dz, cy = y * 4 + dy;
                if (cz >= cnz || cy >= cny) continue;
                for (int dx = 0; dx < 4; dx++) {
                    const int cx = x * 4 + dx;
                    if (cx >= cnx) continue;
                    const uint32_t c = (uint32_t)((cz * cny + cy) * cnx + cx);
                    if (!((child[c >> 5] >> (c & 31)) & 1u)) empty = false;
                }
            }
    }
    store_ballot(empty, blockIdx.x * 256u + (threadIdx.x & ~63u), count, parent);
}

// Below level 0: one 64-bit word per brick, one bit per block of 2^3 cells.  A tap whose low
// corner lies in block (bx, by, bz) of the brick filters the voxels [2b, 2b+2]^3 (local, apron
// included); the block is empty when the transfer function is exactly transparent over their range.
// One wave per brick, one lane per block.  Empty bricks get all ones without touching their voxels.
// After a change of table only the bricks whose value range reads an entry whose badness changed are
// looked at again (the range of a block lies inside its brick's): dragging a control point re-reads the
// voxels of the bricks around the threshold it moves, not the volume.
template <typename T>
__global__ __launch_bounds__(256) void k_classify_micro(const T *__restrict__ bricks, const uint32_t *__restrict__ empty0,
                                                       const uint32_t *__restrict__ cells, uint32_t brick_count, float inv_hi,
                                                       float inv_lo, const LutInfo *__restrict__ info,
                                                       unsigned long long *__restrict__ micro)
{
    if (!info->changed) return;
    const int chg_first = info->chg_first, chg_last = info->chg_last;
    const int lane = threadIdx.x & 63;
    const uint32_t waves = gridDim.x * 4u;
    const int bx = (lane & 3) * 2, by = ((lane >> 2) & 3) * 2, bz = (lane >> 4) * 2;
    // 64 bricks at a time, one per lane, to find those the change concerns (few: the bricks whose values straddle
    // the entries that changed); the wave then takes these one after the other
    for (uint32_t base = (blockIdx.x * 4u + (threadIdx.x >> 6)) * 64u; base < brick_count; base += waves * 64u) {
        bool todo = false;
        const uint32_t mine = base + (uint32_t)lane;
        if (mine < brick_count) {
            const uint32_t mm = cells[mine];
            const float lo = (float)(mm & 0xffffu), hi = (float)(mm >> 16);
            int first, last;
            tf_entry_range(__builtin_fmaf(lo, inv_hi, lo * inv_lo), __builtin_fmaf(hi, inv_hi, hi * inv_lo), first, last);
            todo = !(last < chg_first || first > chg_last);   // else nothing this brick reads has changed: its word stands
            if (todo && ((empty0[mine >> 5] >> (mine & 31)) & 1u)) {
                micro[mine] = ~0ull;
                todo = false;
            }
        }
        for (unsigned long long left = __ballot(todo); left != 0ull; left &= left - 1ull) {
            const uint32_t brick = base + (uint32_t)__builtin_ctzll(left);
            const T *src = bricks + (size_t)brick * kBrickElems + (bz * kRow + by) * kRow + bx;
            uint32_t mn = 0xffffffffu, mx = 0;
#pragma unroll
            for (int z = 0; z < 3; z++)
#pragma unroll
                for (int y = 0; y < 3; y++)
#pragma unroll
                    for (int x = 0; x < 3; x++) {
                        const uint32_t v = src[(z * kRow + y) * kRow + x];
                        mn = v < mn ? v : mn;
                        mx = v > mx ? v : mx;
                    }
            const float lo = (float)mn, hi = (float)mx;
            int first, last;
            tf_entry_range(__builtin_fmaf(lo, inv_hi, lo * inv_lo), __builtin_fmaf(hi, inv_hi, hi * inv_lo), first, last);
            const bool empty = info->bad_prefix[last + 1] == info->bad_prefix[first];
            unsigned long long mask = __ballot(empty);
#ifdef VR_COARSE_MICRO
            {   // experiment: blocks of 4^3 cells (a group of 8 blocks is empty only if all of them are)
                unsigned long long gm = 0;
                for (int k = 0; k < 8; k++) {
                    const int x = ((lane & 3) & ~1) | (k & 1), y = (((lane >> 2) & 3) & ~1) | ((k >> 1) & 1), z = ((lane >> 4) & ~1) | (k >> 2);
                    gm |= 1ull << ((z << 4) | (y << 2) | x);
                }
                mask = __ballot((mask & gm) == gm);
            }
#endif
            if (lane == 0) micro[brick] = mask;
        }
    }
}

__device__ __forceinline__ bool micro_empty(unsigned long long mask, const Cell &c)
{
    const int bit = (((c.sz & 7) >> 1) << 4) | (((c.sy & 7) >> 1) << 2) | ((c.sx & 7) >> 1);
    return (mask >> bit) & 1ull;
}

// ---- stepping over empty space ----------------------------------------------------------------
// Along a ray the padded texel coordinate G = u*size - 0.5 + kPad is, in real arithmetic, linear
// in the tap index t: G(t) = a + t*b.  The marches walk this line model to find empty regions and
// to count how many taps lie in them; taps that are fetched always get their exact position.
// The model and the exact float position of a tap differ only by rounding (a few 1e-3 texels for
// a 2048^3 volume), so every test against a region is made against the region shrunk by `margin`
// texels on each side: a tap whose model position is inside the shrunk region has its true low
// corner inside the region.
struct RayModel {
    float ax, ay, az;   // G(t) = a + t * b
    float bx, by, bz;
    // for counting taps while walking in the march direction (tap index changing by `dir` per step):
    // 1 / (dir * b), 0 where the ray does not move along the axis (it never leaves through a face of that
    // axis); positive where the walk increases the coordinate (it leaves through the upper face)
    float ix, iy, iz;
    float margin;
    // Taps outside [t_stop, t_first] lie outside the bounding box of all occupied bricks (widened
    // by a voxel and a tap): they are in empty bricks and need no look at all.
    int t_first, t_stop;
};

// The line itself, without the clip against the occupied bricks (t_first, t_stop left open).
__device__ __forceinline__ RayModel ray_line(const Ray &r, const VolumeView &v, float step, float dir)
{
    RayModel m;
    m.ax = (r.o[0] + r.tmax * r.d[0] + 0.5f) * v.fx - 0.5f + (float)(kPad - v.ox);
    m.ay = (r.o[1] + r.tmax * r.d[1] + 0.5f) * v.fy - 0.5f + (float)(kPad - v.oy);
    m.az = -(r.o[2] + r.tmax * r.d[2]) * v.fz - 0.5f + (float)(kPad - v.oz);
    m.bx = -(step * r.d[0]) * v.fx;
    m.by = -(step * r.d[1]) * v.fy;
    m.bz = (step * r.d[2]) * v.fz;
    const float dx = dir * m.bx, dy = dir * m.by, dz = dir * m.bz;
    m.ix = dx != 0.0f ? __builtin_amdgcn_rcpf(dx) : 0.0f;
    m.iy = dy != 0.0f ? __builtin_amdgcn_rcpf(dy) : 0.0f;
    m.iz = dz != 0.0f ? __builtin_amdgcn_rcpf(dz) : 0.0f;
    const float big = fmaxf(v.fx, fmaxf(v.fy, v.fz));
    m.margin = fmaxf(0.02f, big * 1e-5f);
    m.t_first = 0x7fffffff;
    m.t_stop = 1;
    return m;
}

__device__ __forceinline__ RayModel ray_model(const Ray &r, const VolumeView &v, float step, float dir)
{
    RayModel m;
    // world(t) = o + (tmax - t*step)*d; u = x + .5, v = y + .5, w = -z inside the box
    // (VolumeRendering.comp:168,187)
    // (for a sub-volume: measured from the first cell of its brick box)
    m.ax = (r.o[0] + r.tmax * r.d[0] + 0.5f) * v.fx - 0.5f + (float)(kPad - v.ox);
    m.ay = (r.o[1] + r.tmax * r.d[1] + 0.5f) * v.fy - 0.5f + (float)(kPad - v.oy);
    m.az = -(r.o[2] + r.tmax * r.d[2]) * v.fz - 0.5f + (float)(kPad - v.oz);
    m.bx = -(step * r.d[0]) * v.fx;
    m.by = -(step * r.d[1]) * v.fy;
    m.bz = (step * r.d[2]) * v.fz;
    // the reciprocals only serve the tap counts of the line model, which the margin protects
    const float dx = dir * m.bx, dy = dir * m.by, dz = dir * m.bz;
    m.ix = dx != 0.0f ? __builtin_amdgcn_rcpf(dx) : 0.0f;
    m.iy = dy != 0.0f ? __builtin_amdgcn_rcpf(dy) : 0.0f;
    m.iz = dz != 0.0f ? __builtin_amdgcn_rcpf(dz) : 0.0f;
    const float big = fmaxf(v.fx, fmaxf(v.fy, v.fz));
    m.margin = fmaxf(0.02f, big * 1e-5f);
    if (v.occ[0] > v.occ[3]) {
        // no brick is occupied (a fully transparent table): no tap in front of the back-most one matters
        m.t_first = 0;
        m.t_stop = 0x7fffffff;
        return m;
    }
    // clip the tap range against the box of occupied bricks: G(t) = a + t*b inside [L, U] per axis
    float t_lo = 0.0f, t_hi = 3e9f;
    const float a[3] = {m.ax, m.ay, m.az}, b[3] = {m.bx, m.by, m.bz}, inv[3] = {dir * m.ix, dir * m.iy, dir * m.iz};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float L = (float)(v.occ[k] * kBrick) - 1.0f - m.margin, U = (float)((v.occ[3 + k] + 1) * kBrick) + 1.0f + m.margin;
        if (b[k] != 0.0f) {
            const float t1 = (L - a[k]) * inv[k], t2 = (U - a[k]) * inv[k];
            t_lo = fmaxf(t_lo, fminf(t1, t2));
            t_hi = fminf(t_hi, fmaxf(t1, t2));
        } else if (!(a[k] >= L && a[k] <= U)) {
            t_hi = -1.0f;
        }
    }
    // one tap of slack on each side for the rounding of the model
    m.t_first = t_hi >= t_lo ? (int)fminf(t_hi + 2.0f, 2e9f) : 0;
    m.t_stop = t_hi >= t_lo ? (int)fmaxf(t_lo - 1.0f, 1.0f) : 0x7fffffff;
    if (m.t_stop < 1) m.t_stop = 1;
    return m;
}

// Bounding box (brick coordinates, inclusive) of the occupied bricks; lo > hi when there is none.
__global__ __launch_bounds__(256) void k_occupied_bounds(const LutInfo *__restrict__ info, const uint32_t *__restrict__ empty0, int nx,
                                                        int ny, int nz, int *occ)
{
    if (!info->changed) return;
    __shared__ int s_lo[3][4], s_hi[3][4];
    const uint32_t bricks = (uint32_t)(nx * ny * nz);
    int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {-1, -1, -1};
    for (uint32_t b = blockIdx.x * 256u + threadIdx.x; b < bricks; b += gridDim.x * 256u) {
        if ((empty0[b >> 5] >> (b & 31)) & 1u) continue;
        const int c[3] = {(int)(b % (uint32_t)nx), (int)((b / (uint32_t)nx) % (uint32_t)ny), (int)(b / (uint32_t)(nx * ny))};
        for (int k = 0; k < 3; k++) {
            lo[k] = min(lo[k], c[k]);
            hi[k] = max(hi[k], c[k]);
        }
    }
    for (int k = 0; k < 3; k++) {
        for (int o = 32; o > 0; o >>= 1) {
            lo[k] = min(lo[k], __shfl_xor(lo[k], o));
            hi[k] = max(hi[k], __shfl_xor(hi[k], o));
        }
        if ((threadIdx.x & 63) == 0) {
            s_lo[k][threadIdx.x >> 6] = lo[k];
            s_hi[k][threadIdx.x >> 6] = hi[k];
        }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = threadIdx.x;
        atomicMin(&occ[k], min(min(s_lo[k][0], s_lo[k][1]), min(s_lo[k][2], s_lo[k][3])));
        atomicMax(&occ[3 + k], max(max(s_hi[k][0], s_hi[k][1]), max(s_hi[k][2], s_hi[k][3])));
    }
}

// Occupied bricks of this context's box, counted per cube of `cell`^3 bricks of the WHOLE volume's brick
// grid (x fastest; cnx x cny cubes per slice of the count grid) and added to `counts`: what a caller that
// splits a volume over several GPUs needs to cut boxes of equal work.  (bx0, by0, bz0): first brick of the box.
__global__ __launch_bounds__(256) void k_count_occupied(const uint32_t *__restrict__ empty0, int nbx, int nby, int nbz, int bx0,
                                                       int by0, int bz0, int cell, int cnx, int cny, uint32_t *counts)
{
    const uint32_t bricks = (uint32_t)(nbx * nby * nbz);
    for (uint32_t b = blockIdx.x * 256u + threadIdx.x; b < bricks; b += gridDim.x * 256u) {
        if ((empty0[b >> 5] >> (b & 31)) & 1u) continue;
        const int x = (int)(b % (uint32_t)nbx) + bx0, y = (int)((b / (uint32_t)nbx) % (uint32_t)nby) + by0,
                  z = (int)(b / (uint32_t)(nbx * nby)) + bz0;
        atomicAdd(&counts[((size_t)(z / cell) * cny + (size_t)(y / cell)) * cnx + (size_t)(x / cell)], 1u);
    }
}

// ---- depth range of the occupied bricks per pixel tile (8x8 pixels, 1 << kRangeShift) ---------------
// Before a frame every occupied brick is projected into the image and widens, for each tile its
// projection touches, the tile's range [near, far] of distances from the eye.  A ray then only looks
// at the taps inside its tile's range: rays of tiles no brick touches are finished at once, the others
// start a few voxels in front of the first occupied brick instead of at the volume's face.
// Stored per tile as two numbers that only grow, so that a cleared buffer means "nothing":
// x = 0x7f800000 - bits(near), y = bits(far).  A tile is the 8x8 pixels one wave of the lane march owns, so
// the range is as tight as the wave can use; a brick of a 2048^3 volume projects to about 3x3 pixels of a
// 4K frame and touches one to four tiles (the cost of this pass is its atomics, merged per block in LDS).
#ifndef VR_RANGE_SHIFT
#define VR_RANGE_SHIFT 3
#endif
constexpr int kRangeShift = VR_RANGE_SHIFT;
// Lists the occupied bricks that have an empty (or no) neighbour across at least one face: only those
// can be the first or last occupied brick of a ray, whatever the camera.  One wave looks at one cube of
// 4^3 bricks and appends its finds in one piece, so bricks that are neighbours in space are neighbours
// in the list (k_splat_bricks merges the ranges of a block's bricks per tile before it goes to memory).
constexpr int kListWaves = 16;   // cubes a workgroup of k_list_occupied looks at
__global__ __launch_bounds__(64 * kListWaves) void k_list_occupied(const LutInfo *__restrict__ info, const uint32_t *__restrict__ empty0,
                                                                  const uint32_t *__restrict__ empty1, int nbx, int nby, int nbz,
                                                                  uint32_t *__restrict__ list, uint32_t *count)
{
    if (!info->changed) return;
    __shared__ uint32_t s_finds[kListWaves], s_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gx_n = (nbx + 3) >> 2, gy_n = (nby + 3) >> 2, gz_n = (nbz + 3) >> 2;
    const uint32_t group = blockIdx.x * (uint32_t)kListWaves + (uint32_t)wave;
    bool keep = false;
    int bx = 0, by = 0, bz = 0;
    // the cube is a cell of the pyramid's next level (same grid, same numbering): most of a volume is empty cubes
    if (group < (uint32_t)(gx_n * gy_n * gz_n) && !((empty1[group >> 5] >> (group & 31)) & 1u)) {
        const int gx = (int)(group % (uint32_t)gx_n), gy = (int)((group / (uint32_t)gx_n) % (uint32_t)gy_n),
                  gz = (int)(group / (uint32_t)(gx_n * gy_n));
        bx = gx * 4 + (lane & 3);
        by = gy * 4 + ((lane >> 2) & 3);
        bz = gz * 4 + (lane >> 4);
        auto empty_at = [&](int x, int y, int z) -> bool {
            if ((unsigned)x >= (unsigned)nbx || (unsigned)y >= (unsigned)nby || (unsigned)z >= (unsigned)nbz) return true;
            const uint32_t b = (uint32_t)((z * nby + y) * nbx + x);
            return (empty0[b >> 5] >> (b & 31)) & 1u;
        };
        if (bx < nbx && by < nby && bz < nbz && !empty_at(bx, by, bz))
            keep = empty_at(bx - 1, by, bz) || empty_at(bx + 1, by, bz) || empty_at(bx, by - 1, bz) || empty_at(bx, by + 1, bz) ||
                   empty_at(bx, by, bz - 1) || empty_at(bx, by, bz + 1);
    }
    // one addition to the list's length per workgroup: additions on one word take their turn
    const unsigned long long mask = __ballot(keep);
    if (lane == 0) s_finds[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (int k = 0; k < kListWaves; k++) total += s_finds[k];
        s_base = total ? atomicAdd(count, total) : 0u;
    }
    __syncthreads();
    if (keep) {
        uint32_t base = s_base;
        for (int k = 0; k < wave; k++) base += s_finds[k];
        list[base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = (uint32_t)((bz * nby + by) * nbx + bx);
    }
}

// Only bricks a ray can reach first or leave last have something to say.  A ray enters a brick through
// a face turned towards the eye, out of the brick on the other side of that face; if that neighbour is
// occupied it is nearer and speaks for the ray.  So a brick widens `near` only if a neighbour across
// one of its eye-facing faces is empty (or outside the grid), and `far` only if one across a face
// turned away from the eye is.  That leaves the surface layers: a fraction of the atomics.
__device__ __forceinline__ bool brick_is_empty(const VolumeView &vol, int bx, int by, int bz)
{
    if ((unsigned)bx >= (unsigned)vol.nbx || (unsigned)by >= (unsigned)vol.nby || (unsigned)bz >= (unsigned)vol.nbz) return true;
    const uint32_t b = (uint32_t)((bz * vol.nby + by) * vol.nbx + bx);
    return (vol.empty[0][b >> 5] >> (b & 31)) & 1u;
}

constexpr uint32_t kSplatSlots = 1024, kNoTile = 0xffffffffu;

// One word behind the tiles' ranges says that the ranges could not be built for this camera (a brick
// behind or around the eye): the marches then take every ray's full tap range.  It travels with the
// ranges because the pass that writes both may run ahead of the frame's other set-up, on a stream of its own.
__device__ __forceinline__ size_t clip_flag_index(int tiles_x, int height)
{
    return (size_t)tiles_x * (size_t)((height + (1 << kRangeShift) - 1) >> kRangeShift);
}
__device__ __forceinline__ bool clip_disabled(const uint2 *__restrict__ tiles, int tiles_x, const FrameParams &fp)
{
    return tiles[clip_flag_index(tiles_x, fp.height)].x != 0u;
}

// Raises the range of one tile in global memory.  Both numbers only grow, so a range that would not
// raise what the tile already holds has nothing to write; atomics on one address are served one
// after the other.
__device__ __forceinline__ void raise_tile(uint2 *t, uint32_t nb, uint32_t fb)
{
    if (nb != 0u && __atomic_load_n(&t->x, __ATOMIC_RELAXED) < nb) atomicMax(&t->x, nb);
    if (fb != 0u && __atomic_load_n(&t->y, __ATOMIC_RELAXED) < fb) atomicMax(&t->y, fb);
}

// One brick of the list: the ranges of the tiles it covers go into the block's LDS table, or, where
// the table is crowded, straight to global memory.
__device__ __forceinline__ void splat_one(const VolumeView &vol, const FrameParams &fp, uint32_t brick, uint2 *__restrict__ tiles,
                                          int tiles_x, int tiles_y, Counters *ctr, uint32_t *s_tile, uint32_t *s_near,
                                          uint32_t *s_far)
{
    const int bx = (int)(brick % (uint32_t)vol.nbx), by = (int)((brick / (uint32_t)vol.nbx) % (uint32_t)vol.nby),
              bz = (int)(brick / (uint32_t)(vol.nbx * vol.nby));
    // the eye in texel coordinates (inverse of the mapping below)
    const float eye[3] = {(fp.loc[0] + 0.5f) * vol.fx + 1.5f, (fp.loc[1] + 0.5f) * vol.fy + 1.5f, -fp.loc[2] * vol.fz + 1.5f};
    const int b3[3] = {bx, by, bz}, org[3] = {vol.ox, vol.oy, vol.oz};
    bool front = false, back = false;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        int lo[3] = {bx, by, bz}, hi[3] = {bx, by, bz};
        lo[k] -= 1;
        hi[k] += 1;
        const bool lo_empty = brick_is_empty(vol, lo[0], lo[1], lo[2]), hi_empty = brick_is_empty(vol, hi[0], hi[1], hi[2]);
        const float f0 = (float)(b3[k] * kBrick + org[k]), f1 = f0 + (float)kBrick;
        if (eye[k] < f0 + 1.0f) {          // the low face is (or may be) turned towards the eye
            front = front || lo_empty;
            back = back || hi_empty;
        }
        if (eye[k] > f1 - 1.0f) {
            front = front || hi_empty;
            back = back || lo_empty;
        }
        if (eye[k] >= f0 - 1.0f && eye[k] <= f1 + 1.0f) back = back || lo_empty || hi_empty;   // eye beside the brick: both can be exits
    }
    if (!front && !back) return;
    // Taps whose low corner lies in the brick have texel coordinates G in [8b, 8b + 8) per axis; one
    // texel more on each side covers the rounding of a tap's position.  G -> world: u = (G - 1.5) / size,
    // x = u - 0.5, y = v - 0.5, z = -w (VolumeRendering.comp:168,187).
    const float gx0 = (float)(bx * kBrick + vol.ox - 1), gy0 = (float)(by * kBrick + vol.oy - 1), gz0 = (float)(bz * kBrick + vol.oz - 1);
    const float ext = (float)(kBrick + 2);
    const float x0 = (gx0 - 1.5f) / vol.fx - 0.5f, x1 = (gx0 + ext - 1.5f) / vol.fx - 0.5f;
    const float y0 = (gy0 - 1.5f) / vol.fy - 0.5f, y1 = (gy0 + ext - 1.5f) / vol.fy - 0.5f;
    const float z0 = -(gz0 - 1.5f) / vol.fz, z1 = -(gz0 + ext - 1.5f) / vol.fz;
    // pixel of a world point P: px m0 + py m1 + m3 - loc = s (P - loc), solved with Cramer's rule
    const float m0[3] = {fp.m[0], fp.m[1], fp.m[2]}, m1[3] = {fp.m[4], fp.m[5], fp.m[6]};
    const float r[3] = {fp.loc[0] - fp.m[12], fp.loc[1] - fp.m[13], fp.loc[2] - fp.m[14]};
    float pxmin = 3e38f, pxmax = -3e38f, pymin = 3e38f, pymax = -3e38f;
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const float q[3] = {-((k & 1 ? x1 : x0) - fp.loc[0]), -((k & 2 ? y1 : y0) - fp.loc[1]), -((k & 4 ? z1 : z0) - fp.loc[2])};
        const float c1[3] = {m1[1] * q[2] - m1[2] * q[1], m1[2] * q[0] - m1[0] * q[2], m1[0] * q[1] - m1[1] * q[0]};   // m1 x q
        const float c2[3] = {r[1] * q[2] - r[2] * q[1], r[2] * q[0] - r[0] * q[2], r[0] * q[1] - r[1] * q[0]};        // r x q
        const float c3[3] = {m1[1] * r[2] - m1[2] * r[1], m1[2] * r[0] - m1[0] * r[2], m1[0] * r[1] - m1[1] * r[0]};   // m1 x r
        const float det = m0[0] * c1[0] + m0[1] * c1[1] + m0[2] * c1[2];
        const float inv = 1.0f / det;
        const float px = (r[0] * c1[0] + r[1] * c1[1] + r[2] * c1[2]) * inv;
        const float py = (m0[0] * c2[0] + m0[1] * c2[1] + m0[2] * c2[2]) * inv;
        const float sc = (m0[0] * c3[0] + m0[1] * c3[1] + m0[2] * c3[2]) * inv;
        // the corner must lie in front of the eye, well away from it, and project to a finite pixel
        if (!(sc > 1e-4f) || !(fabsf(px) < 1e7f) || !(fabsf(py) < 1e7f)) ok = false;
        pxmin = fminf(pxmin, px); pxmax = fmaxf(pxmax, px);
        pymin = fminf(pymin, py); pymax = fmaxf(pymax, py);
    }
    if (!ok) {
        atomicOr(&tiles[(size_t)tiles_x * tiles_y].x, 1u);
        return;
    }
    // Distance from the eye (what a tap's index measures along its ray) of the nearest and the farthest point
    // of the box: per axis the gap to the box, or zero inside its slab, and the farther of the two faces.
    const float cx = 0.5f * (x0 + x1) - fp.loc[0], cy = 0.5f * (y0 + y1) - fp.loc[1], cz = 0.5f * (z0 + z1) - fp.loc[2];
    const float hx = 0.5f * fabsf(x1 - x0), hy = 0.5f * fabsf(y1 - y0), hz = 0.5f * fabsf(z1 - z0);
    const float nx = fmaxf(fabsf(cx) - hx, 0.0f), ny = fmaxf(fabsf(cy) - hy, 0.0f), nz = fmaxf(fabsf(cz) - hz, 0.0f);
    const float fx = fabsf(cx) + hx, fy = fabsf(cy) + hy, fz = fabsf(cz) + hz;
    const float near = fmaxf(sqrtf(nx * nx + ny * ny + nz * nz) * 0.9999f - 1e-6f, 0.0f);
    const float far = sqrtf(fx * fx + fy * fy + fz * fz) * 1.0001f + 1e-6f;
    const int tx0 = max(0, (int)floorf(pxmin - 1.5f) >> kRangeShift), tx1 = min(tiles_x - 1, (int)floorf(pxmax + 1.5f) >> kRangeShift);
    const int ty0 = max(0, (int)floorf(pymin - 1.5f) >> kRangeShift), ty1 = min(tiles_y - 1, (int)floorf(pymax + 1.5f) >> kRangeShift);
    if (tx0 > tx1 || ty0 > ty1) return;   // off screen
    if ((tx1 - tx0 + 1) * (ty1 - ty0 + 1) > 4096) {   // eye very close to a brick: not worth the writes
        atomicOr(&tiles[(size_t)tiles_x * tiles_y].x, 1u);
        return;
    }
    const uint32_t nb = front ? 0x7f800000u - __float_as_uint(near) : 0u, fb = back ? __float_as_uint(far) : 0u;
    for (int ty = ty0; ty <= ty1; ty++) {
        // only tiles this launch renders: inside the rectangle and, with a sort-first split, in one of
        // this part's 16-row bands
        const int r0 = (ty << kRangeShift) - fp.y0, r1 = r0 + (1 << kRangeShift) - 1;   // rectangle rows of the tile
        if (r1 < 0 || r0 >= fp.y1 - fp.y0) continue;
        const int band0 = (r0 < 0 ? 0 : r0) >> 4, band1 = r1 >> 4;
        bool mine = false;
        for (int b = band0; b <= band1; b++) mine = mine || (b % fp.parts == fp.part);
        if (!mine) continue;
        for (int tx = tx0; tx <= tx1; tx++) {
            if (((tx + 1) << kRangeShift) <= fp.x0 || (tx << kRangeShift) >= fp.x1) continue;
            const uint32_t tile = (uint32_t)(ty * tiles_x + tx);
            uint32_t slot = (tile * 2654435761u) >> 22;     // top 10 bits: kSplatSlots entries
            bool placed = false;
            for (int probe = 0; probe < 8 && !placed; probe++) {
                const uint32_t held = atomicCAS(&s_tile[slot], kNoTile, tile);
                if (held == kNoTile || held == tile) {
                    if (nb != 0u) atomicMax(&s_near[slot], nb);
                    if (fb != 0u) atomicMax(&s_far[slot], fb);
                    placed = true;
                }
                slot = (slot + 1u) & (kSplatSlots - 1u);
            }
            if (!placed) raise_tile(tiles + tile, nb, fb);
        }
    }
}

// Bricks next to each other in the list project onto the same few tiles, so a block first merges its
// bricks' ranges per tile in LDS (a small open-addressing table keyed by the tile) and then raises
// each tile of the table once in global memory: far fewer global atomics, which are the cost here.
// The list's length is read from device memory (the kernel that rebuilds the list after a change of table
// leaves it there; the host never learns it): the grid is sized for the longest list a volume can have and a
// block takes every gridDim.x-th run of 256 entries, flushing its table after each.
__global__ __launch_bounds__(256) void k_splat_bricks(VolumeView vol, FrameParams fp, const uint32_t *__restrict__ list,
                                                     const uint32_t *__restrict__ count_ptr, uint2 *__restrict__ tiles, int tiles_x,
                                                     int tiles_y, Counters *ctr)
{
    static_assert(kSplatSlots == 1024, "the hash keeps the top 10 bits");
    __shared__ uint32_t s_tile[kSplatSlots], s_near[kSplatSlots], s_far[kSplatSlots];
    const uint32_t count = *count_ptr;
    for (uint32_t base = blockIdx.x * 256u; base < count; base += gridDim.x * 256u) {
        for (uint32_t k = threadIdx.x; k < kSplatSlots; k += 256u) {
            s_tile[k] = kNoTile;
            s_near[k] = 0u;
            s_far[k] = 0u;
        }
        __syncthreads();
        const uint32_t i = base + threadIdx.x;
        if (i < count) splat_one(vol, fp, list[i], tiles, tiles_x, tiles_y, ctr, s_tile, s_near, s_far);
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < kSplatSlots; k += 256u) {
            const uint32_t tile = s_tile[k];
            if (tile != kNoTile) raise_tile(tiles + tile, s_near[k], s_far[k]);
        }
        __syncthreads();
    }
}

// Narrows a ray's tap range to its tile's depth range.
__device__ __forceinline__ void clip_to_tile(RayModel &m, const Ray &r, float step, const uint2 *__restrict__ tiles,
                                             int tiles_x, int px, int py, const FrameParams &fp)
{
    if (!tiles || clip_disabled(tiles, tiles_x, fp) || !(step > 0.0f)) return;
    const uint2 range = tiles[(size_t)(py >> kRangeShift) * tiles_x + (px >> kRangeShift)];
    if (range.y == 0u) {   // no occupied brick projects into this tile
        m.t_first = 0;
        m.t_stop = 0x7fffffff;
        return;
    }
    const float near = __uint_as_float(0x7f800000u - range.x), far = __uint_as_float(range.y);
    const float inv = 1.0f / step;
    // tap t lies at distance tmax - t * step from the eye (VolumeRendering.comp:185-186); two taps of slack
    const float first = (r.tmax - near) * inv + 2.0f, stop = (r.tmax - far) * inv - 2.0f;
    if (first < (float)m.t_first) m.t_first = first < 0.0f ? 0 : (int)first;
    if (stop > (float)m.t_stop) m.t_stop = (int)fminf(stop, 2e9f);
}

// Steps (possibly fractional) until the model position leaves [lo, lo + extent) shrunk by the
// margin along one axis; negative when it is not inside now.  Branch-free on purpose.
__device__ __forceinline__ float axis_span(float g, float lo, float extent, float inv, float margin)
{
    const float a = lo + margin, b = lo + extent - margin;
    const float target = inv > 0.0f ? b : a;
    float n = (target - g) * inv;
    if (inv == 0.0f) n = 1e9f;                // not moving along this axis
    return (g >= a && g <= b) ? n : -1.0f;
}

// Taps t, t + dir, t + 2*dir, ... whose model position stays inside the cube [lo, lo + extent)^3
// shrunk by the margin; 0 when tap t itself is not inside.
__device__ __forceinline__ int span_inside(float gx, float gy, float gz, int lox, int loy, int loz, float extent,
                                           const RayModel &m)
{
    float n = axis_span(gx, (float)lox, extent, m.ix, m.margin);
    n = fminf(n, axis_span(gy, (float)loy, extent, m.iy, m.margin));
    n = fminf(n, axis_span(gz, (float)loz, extent, m.iz, m.margin));
    n = fminf(n, 1e9f) * 0.9999f;             // the reciprocals are good to 1 ulp: round the count down
    return n >= 0.0f ? 1 + (int)n : 0;
}

// The pyramid folded into one table the marches read: two bits per brick, 0 = occupied, otherwise
// 1 + the coarsest of the levels 0..2 whose cube around the brick is empty.  One load answers
// "how far can I step from here".
__global__ __launch_bounds__(256) void k_pack_skip(const LutInfo *__restrict__ info, const uint32_t *__restrict__ e0,
                                                  const uint32_t *__restrict__ e1, const uint32_t *__restrict__ e2, int nx0, int ny0,
                                                  int nz0, int nx1, int ny1, int nx2, int ny2, uint32_t *__restrict__ skip)
{
    if (!info->changed) return;
    const uint32_t word = blockIdx.x * 256u + threadIdx.x;
    const uint32_t bricks = (uint32_t)(nx0 * ny0 * nz0);
    if (word * 16u >= bricks) return;
    uint32_t packed = 0;
    for (uint32_t k = 0; k < 16u; k++) {
        const uint32_t b = word * 16u + k;
        if (b >= bricks) break;
        const int x = b % nx0, y = (b / nx0) % ny0, z = b / (nx0 * ny0);
        const uint32_t i1 = (uint32_t)(((z >> 2) * ny1 + (y >> 2)) * nx1 + (x >> 2));
        const uint32_t i2 = (uint32_t)(((z >> 4) * ny2 + (y >> 4)) * nx2 + (x >> 4));
        uint32_t code = (e0[b >> 5] >> (b & 31)) & 1u;
        if ((e1[i1 >> 5] >> (i1 & 31)) & 1u) code = 2;
        if ((e2[i2 >> 5] >> (i2 & 31)) & 1u) code = 3;
        packed |= code << (2 * k);
    }
    skip[word] = packed;
}

__device__ __forceinline__ uint32_t skip_code(uint32_t brick, const VolumeView &v)
{
    return (v.skip[brick >> 4] >> ((brick & 15u) * 2u)) & 3u;
}

// Coarsest empty pyramid level (0..2) that contains brick (cx, cy, cz), or -1.
__device__ __forceinline__ int empty_level(const VolumeView &v, int cx, int cy, int cz)
{
    return (int)skip_code((uint32_t)((cz * v.ly[0] + cy) * v.lx[0] + cx), v) - 1;
}

__device__ __forceinline__ bool brick_empty(uint32_t brick, const VolumeView &v)
{
    return skip_code(brick, v) != 0u;
}

// How many taps, starting at t and walking in direction dir, can be stepped over because the line
// model puts them inside an empty cube of the pyramid.  0: tap t needs an exact look.
__device__ __forceinline__ int empty_span(const VolumeView &v, const RayModel &m, int t)
{
    const float ft = (float)t;
    const float gx = __builtin_fmaf(ft, m.bx, m.ax), gy = __builtin_fmaf(ft, m.by, m.ay), gz = __builtin_fmaf(ft, m.bz, m.az);
    int cx = (int)fminf(fmaxf(gx, 0.0f), 1e7f) >> 3;
    int cy = (int)fminf(fmaxf(gy, 0.0f), 1e7f) >> 3;
    int cz = (int)fminf(fmaxf(gz, 0.0f), 1e7f) >> 3;
    cx = cx < v.lx[0] ? cx : v.lx[0] - 1;
    cy = cy < v.ly[0] ? cy : v.ly[0] - 1;
    cz = cz < v.lz[0] ? cz : v.lz[0] - 1;
    const int level = empty_level(v, cx, cy, cz);
    if (level < 0) return 0;
    const int shift = 3 + 2 * level;
    return span_inside(gx, gy, gz, (cx >> (2 * level)) << shift, (cy >> (2 * level)) << shift, (cz >> (2 * level)) << shift,
                       (float)(1 << shift), m);
}

// The same count when the tap's exact brick (cx, cy, cz) is already known to lie in an empty cube of
// pyramid level `level`: taps t, t + dir, ... whose model position stays inside that cube.
__device__ __forceinline__ int empty_span_at(const VolumeView &v, const RayModel &m, int t, int level, int cx,
                                             int cy, int cz)
{
    const float ft = (float)t;
    const float gx = __builtin_fmaf(ft, m.bx, m.ax), gy = __builtin_fmaf(ft, m.by, m.ay), gz = __builtin_fmaf(ft, m.bz, m.az);
    const int shift = 3 + 2 * level;
    return span_inside(gx, gy, gz, (cx >> (2 * level)) << shift, (cy >> (2 * level)) << shift, (cz >> (2 * level)) << shift,
                       (float)(1 << shift), m);
}

// Taps t, t + dir, ... whose model position stays inside the 2^3-cell block that holds cell c.
__device__ __forceinline__ int block_span(const RayModel &m, int t, const Cell &c)
{
    const float ft = (float)t;
    const float gx = __builtin_fmaf(ft, m.bx, m.ax), gy = __builtin_fmaf(ft, m.by, m.ay), gz = __builtin_fmaf(ft, m.bz, m.az);
    return span_inside(gx, gy, gz, c.sx & ~1, c.sy & ~1, c.sz & ~1, 2.0f, m);
}

// The bet on material of the marches that skip (see k_march_fast): after a tap with alpha != 0 the next taps are
// fetched without a look at the occupancy tables.
#ifndef VR_MATERIAL_BET
#define VR_MATERIAL_BET true
#endif
constexpr bool kMaterialBet = VR_MATERIAL_BET;
#ifndef VR_CLEAR_RUN
#define VR_CLEAR_RUN 12
#endif
constexpr int kClearRun = VR_CLEAR_RUN;   // transparent taps in a row that end the bet on material

// ---- reference-order march ---------------------------------------------------------------------
// Back to front over the samples + 2 taps exactly as the shader loops (VolumeRendering.comp:
// 183-202), same blend expression, so the colour is bit-identical to it.  With SKIP, a tap that
// lies in an empty cube of the occupancy pyramid is not fetched: its alpha is exactly 0 and the
// blend a*c + (1-a)*C leaves C unchanged bit for bit.  The back-most tap (t == 0) is always
// fetched because the shader assigns its colour instead of blending.
template <typename T, bool SKIP, bool STATS, bool SHADE>
__global__ __launch_bounds__(256) void k_march_exact(VolumeView vol, FrameParams fp, const float4 *__restrict__ lut,
                                                    float4 border, uint32_t *__restrict__ out, Counters *ctr,
                                                    uint32_t *touched)
{
    __shared__ float4 s_lut[kLutLds];
    stage_lut(s_lut, lut, border);
    int px, py;
    uint32_t out_index;
    if (!lane_pixel(fp, px, py, out_index)) return;
    const Ray r = make_ray(fp, px, py);
    float cr = 0.0f, cg = 0.0f, cb = 0.0f;
    unsigned long long evaluated = 0, lookups = 0;
    if (r.hit) {
        const int last = fp.samples + 1;
        const float step = (r.tmax - r.tmin) / (float)(uint32_t)last;
        const RayModel rm = ray_model(r, vol, step, 1.0f);
        int t = 0;
        // SKIP: after a tap with alpha != 0 the following taps are fetched without asking the occupancy tables
        // (any tap may be fetched: one in empty space has alpha exactly 0 and blends as the identity), until
        // kClearRun transparent taps in a row end the bet.  Inside material -- all of a ray under a table of low
        // alpha everywhere -- a tap then costs what it costs the plain march, not two table look-ups more.
        bool material = false;
        int clear_run = 0;
        while (t <= last) {
            if (SKIP && t > 0) {
                if (t < rm.t_stop) t = rm.t_stop;      // nothing occupied before or beyond these taps
                if (t > rm.t_first || t > last) break;
                if (!material) {
                    if (STATS) lookups++;
                    const int n = empty_span(vol, rm, t);
                    if (n > 0) {
                        t += n;
                        continue;
                    }
                }
            }
            const Tap p = tap_position(r, vol, step, t);
            const Cell c = tap_cell(p, vol);
            const uint32_t brick = brick_of(c, vol);
            if (SKIP && t > 0 && !material && (brick_empty(brick, vol) || micro_empty(vol.micro[brick], c))) {
                t++;
                continue;
            }
            const float inten = sample_volume<T>(c, brick, vol);
            float4 tf = sample_tf(s_lut, inten);
            if (SKIP) {
                if (tf.w != 0.0f) {
                    material = kMaterialBet;
                    clear_run = 0;
                } else if (material && ++clear_run >= kClearRun) {
                    material = false;
                }
            }
            if (t == 0) {
                cr = tf.x; cg = tf.y; cb = tf.z;
            } else {
                if (SHADE && !(SKIP && tf.w == 0.0f && isfinite(tf.x) && isfinite(tf.y) && isfinite(tf.z))) {
                    const float sh = shade_tap<T, STATS>(vol, fp, r, step, t, touched, evaluated);
                    tf.x += sh; tf.y += sh; tf.z += sh;
                }
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
        atomicAdd(&ctr->rays_hit, r.hit ? 1ull : 0ull);
        atomicAdd(&ctr->samples_evaluated, evaluated);
        atomicAdd(&ctr->cells_read, lookups);
    }
}

// ---- deciding a pixel before the ray is finished -------------------------------------------------
// See k_march_fast for the argument.  decide() returns true and the final bytes when, for all
// three channels, both ends of the interval that must contain the shader's float result convert to
// the same unorm8 value.
#ifndef VR_DECIDE_GATE
#define VR_DECIDE_GATE (1.0f / 255.0f)
#endif
struct Decider {
    float cmin[3], cmax[3], kappa[3];
    // Transmittance below which the decision test is tried.  With all colours inside [0,1] both interval
    // ends can only convert to the same byte if they are less than 1/255 apart, and the interval is at
    // least tr * (cmax - cmin) wide.
    float tr_gate;
    bool ok;
    __device__ explicit Decider(const LutInfo *info)
    {
        float range = 0.0f;
        for (int c = 0; c < 3; c++) {
            cmin[c] = info->cmin[c];
            cmax[c] = info->cmax[c];
            kappa[c] = info->kappa[c];
            range = fmaxf(range, cmax[c] - cmin[c]);
        }
        tr_gate = range > 0.0f ? fminf(1.0f, (VR_DECIDE_GATE) / range) : 1.0f;
        // colours outside [0,1] (a table with such entries, or the local illumination): an interval of any
        // width is decided once it lies wholly above 1 or below 0, so the test is worth trying earlier
        for (int c = 0; c < 3; c++)
            if (cmax[c] > 1.0f || cmin[c] < 0.0f) tr_gate = fmaxf(tr_gate, 0.03125f);
        ok = info->fast_ok != 0;
    }
    __device__ __forceinline__ bool decide(double F0, double F1, double F2, double tr, float S, int taps_left, uint32_t &rgba) const
    {
        // The interval ends are formed in float (double arithmetic runs at half rate and this test is a
        // sizeable part of the wave-per-ray march), pushed outwards by more than that arithmetic can
        // lose (u = 2^-24): F and tr converted to float (u |F| + u tr |c|), the product e (u e), the fused
        // multiply-add and the two subtractions (u of their results each, which are at most
        // m = |F| + tr max|c| + e in magnitude) -- at most 4 u m in all; 5 u m is taken.  Every unit of
        // slack costs rays: an interval that straddles a rounding boundary sends its ray to the replay.
        const float trf = (float)tr;
        const float spread = (S * 1.001f + trf * (float)taps_left) * 1.000001f;
        uint32_t bytes = 0xff000000u;
        bool same = true;
        const double Fc[3] = {F0, F1, F2};
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float Ff = (float)Fc[c];
            const float e = kappa[c] * spread;
            const float m = fabsf(Ff) + trf * fmaxf(fabsf(cmin[c]), fabsf(cmax[c])) + e;
            const float slack = __builtin_fmaf(m, 5.0f * kUlpHalf, 1e-38f);
            const float lo = (__builtin_fmaf(trf, cmin[c], Ff) - e) - slack;
            const float hi = (__builtin_fmaf(trf, cmax[c], Ff) + e) + slack;
            const uint32_t bl = to_unorm8(lo), bh = to_unorm8(hi);
            same = same && (bl == bh);
            bytes |= bl << (8 * c);
        }
        if (same) rgba = bytes;
        return same;
    }
};

// Reference-order march of one ray by a whole wave: the fast march hands over the few rays whose
// bytes it could not decide, far too few to fill the chip with one lane each.  The 64 lanes fetch 64
// consecutive taps at once; the blend itself is inherently serial, so the lanes' (alpha, colour) are
// then folded in tap order with the shader's expression.  Taps with alpha exactly 0 and a finite
// colour are identities and are left out of the fold; the back-most tap assigns.
//
// The fold need not start at the back.  One blend step C' = fl(fl(a c) + fl(fl(1 - a) C)) is, for
// 0 <= a <= 1, a non-decreasing function of C in float arithmetic (every rounding is monotone), so two
// folds started at tap m from a lower and an upper bound of the colour behind it enclose the shader's
// colour after every tap.  The two differ by the start width times the transmittance of the taps
// folded, so behind enough material they arrive at the front as the same float: then that float is the
// shader's result, whatever lies behind m.  A first pass from the eye finds the tap where the
// transmittance drops below 2^-40; if the two folds from there do not meet, the ray is folded from
// the back after all.  Needs alphas in [0,1] and finite colours (Decider::ok).
constexpr double kReplayDark = 0x1p-40;

//
// With PARTIAL (sort-last) the fold covers the taps of this context's box only and starts from the
// colour `carried` in from the boxes behind; the result goes to `result` instead of the frame.  If the
// box's own taps make the ray dark the two bounding folds meet and what was carried in does not matter.
template <typename T, bool STATS, bool PARTIAL = false, bool SHADE = false>
__device__ __forceinline__ void replay_ray(const VolumeView &vol, const FrameParams &fp, const float4 *s_lut,
                                           const Decider &dec, uint32_t *__restrict__ out, Counters *ctr,
                                           uint32_t *touched, uint32_t out_index, const uint2 *__restrict__ tiles,
                                           int tiles_x, const float *carried = nullptr, float *result = nullptr, int dark_from = 0)
{
    const int lane = threadIdx.x & 63;
    int px, py;
    pixel_of_index(fp, out_index, px, py);
    const Ray r = make_ray(fp, px, py);          // identical in every lane
    const int last = fp.samples + 1;
    const float step = (r.tmax - r.tmin) / (float)(uint32_t)last;
    unsigned long long evaluated = 0;

    // Where does the ray go dark, seen from the eye?  The marches that queue a ray know: they hand over the
    // first tap they did not visit once the transmittance of the visited ones is below kReplayDark (0: the
    // ray never went dark, fold all of it).  A box of a split volume looks for the point among its own taps.
    int start = (!PARTIAL && dec.ok && r.hit && dark_from > 0) ? (dark_from < last ? dark_from : last) : 0;
    if (PARTIAL && dec.ok && r.hit) {
        RayModel fm = ray_model(r, vol, step, -1.0f);
        clip_to_tile(fm, r, step, tiles, tiles_x, px, py, fp);
        double tr = 1.0;
        int t = last < fm.t_first ? last : fm.t_first;
        while (t >= fm.t_stop && !(tr < kReplayDark)) {
            const int n = empty_span(vol, fm, t);   // wave-uniform
            if (n > 0) {
                t -= n;
                continue;
            }
            const int tt = t - lane;
            double keep = 1.0;
            if (tt >= fm.t_stop) {
                Cell c;
                bool mine = true;
                if (PARTIAL) mine = tap_cell_local(tap_position(r, vol, step, tt), vol, c);
                else c = tap_cell(tap_position(r, vol, step, tt), vol);
                if (!mine) c.sx = c.sy = c.sz = 0;
                const uint32_t brick = brick_of(c, vol);
                if (mine && !(brick_empty(brick, vol) || micro_empty(vol.micro[brick], c))) {
                    keep = 1.0 - (double)sample_tf(s_lut, sample_volume<T>(c, brick, vol)).w;
                    if (STATS) {
                        evaluated++;
                        atomicOr(&touched[brick >> 5], 1u << (brick & 31));
                    }
                }
            }
            for (int o = 32; o > 0; o >>= 1) keep *= __shfl_xor(keep, o);
            tr *= keep;
            t -= 64;
        }
        if (tr < kReplayDark) start = t + 1 > 1 ? t + 1 : 1;   // every tap in front of t + 1 was part of the product
    }

    RayModel rm = ray_model(r, vol, step, 1.0f);
    clip_to_tile(rm, r, step, tiles, tiles_x, px, py, fp);
    float lo[3] = {0.0f, 0.0f, 0.0f}, hi[3] = {0.0f, 0.0f, 0.0f};
    for (int attempt = 0; attempt < 2; attempt++) {
        if (start > 0) {
            // bounds of the colour behind tap `start`: colours stay inside the table's range up to the
            // rounding of the steps taken, at most 3 ulp of the largest colour each
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float mag = fmaxf(fabsf(dec.cmin[c]), fabsf(dec.cmax[c]));
                const float slack = mag * (2e-3f + 4.0f * kUlpHalf * (float)(last + 1)) + 1e-30f;
                lo[c] = dec.cmin[c] - slack;
                hi[c] = dec.cmax[c] + slack;
            }
        } else if (PARTIAL) {
            lo[0] = hi[0] = carried[0];
            lo[1] = hi[1] = carried[1];
            lo[2] = hi[2] = carried[2];
        } else {
            lo[0] = lo[1] = lo[2] = hi[0] = hi[1] = hi[2] = 0.0f;
        }
        int base = start;
        while (base <= last) {
            if (base > 0) {
                if (base < rm.t_stop) base = rm.t_stop;    // nothing occupied before or beyond these taps
                if (base > rm.t_first || base > last) break;
                const int n = empty_span(vol, rm, base);   // wave-uniform: whole runs of empty taps
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
                if (PARTIAL) fetch = tap_cell_local(p, vol, c);
                else c = tap_cell(p, vol);
                if (fetch) {
                    brick = brick_of(c, vol);
                    if (t > 0 && (brick_empty(brick, vol) || micro_empty(vol.micro[brick], c))) fetch = false;
                }
            }
            float4 tf = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (fetch) {
                tf = sample_tf(s_lut, sample_volume<T>(c, brick, vol));
                if (STATS) {
                    evaluated++;
                    atomicOr(&touched[brick >> 5], 1u << (brick & 31));
                }
            }
            const bool identity = tf.w == 0.0f && isfinite(tf.x) && isfinite(tf.y) && isfinite(tf.z);
            if (SHADE && fetch && t > 0 && !identity) {
                const float sh = shade_tap<T, STATS>(vol, fp, r, step, t, touched, evaluated);
                tf.x += sh; tf.y += sh; tf.z += sh;
            }
            unsigned long long todo = __ballot(fetch && (t == 0 || !identity));
            // each lane forms its tap's products once; the fold then reads them lane after lane.  The lane
            // index is the same in all lanes (it comes from the ballot), so the reads are scalar broadcasts.
            const float ia = 1.0f - tf.w;
            const float ax = tf.w * tf.x, ay = tf.w * tf.y, az = tf.w * tf.z;
            auto from_lane = [](float v, int lane_index) -> float {
                return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane_index));
            };
            const bool both = start > 0;     // folding from the back, the two bounds are one value
            while (todo) {
                const int src = __builtin_ctzll(todo);
                todo &= todo - 1;
                if (base + src == 0) {
                    lo[0] = hi[0] = from_lane(tf.x, src);
                    lo[1] = hi[1] = from_lane(tf.y, src);
                    lo[2] = hi[2] = from_lane(tf.z, src);
                } else {
                    const float sia = from_lane(ia, src), sx = from_lane(ax, src), sy = from_lane(ay, src), sz = from_lane(az, src);
                    lo[0] = sx + sia * lo[0];
                    lo[1] = sy + sia * lo[1];
                    lo[2] = sz + sia * lo[2];
                    if (both) {
                        hi[0] = sx + sia * hi[0];
                        hi[1] = sy + sia * hi[1];
                        hi[2] = sz + sia * hi[2];
                    }
                }
            }
            if (!both) { hi[0] = lo[0]; hi[1] = lo[1]; hi[2] = lo[2]; }
            base += 64;
        }
        const bool met = __float_as_uint(lo[0]) == __float_as_uint(hi[0]) && __float_as_uint(lo[1]) == __float_as_uint(hi[1]) &&
                         __float_as_uint(lo[2]) == __float_as_uint(hi[2]);
        if (start == 0 || met) break;
        start = 0;   // the two folds did not meet: fold the whole ray
    }
    if (PARTIAL) {
        for (int c = 0; c < 3; c++) result[c] = r.hit ? lo[c] : carried[c];
        return;
    }
    if (!r.hit) { lo[0] = lo[1] = lo[2] = 0.0f; }
    if (lane == 0) out[out_index] = pack_rgba8(lo[0], lo[1], lo[2]);
    if (STATS) {
        if (lane == 0) atomicAdd(&ctr->replay_rays, 1ull);
        atomicAdd(&ctr->samples_evaluated, evaluated);
        atomicAdd(&ctr->replay_samples, evaluated);
    }
}

// A ray neither march could decide, queued for the reference-order replay: its pixel and, when the ray went
// dark (transmittance of the taps visited below kReplayDark), the tap from which the replay has to fold:
// every tap in front of `dark_from` was visited.  0: the ray reached the back without going dark.
struct ReplayRay {
    uint32_t pixel_id;
    int dark_from;
};

// A ray the lane-per-ray march hands over to a whole wave: where it stands and what it has
// accumulated (k_march_fast -> k_march_long).
struct LongRay {
    uint32_t pixel_id;
    int t;        // next tap to visit (>= 1)
    float S;
    int t_stop;   // taps below it (the back tap aside) cannot matter: the lane march's clipped value
    double F0, F1, F2, Tr;
    // the ray as the lane march computed it, so that the wave that takes over need not repeat the
    // divisions of the slab test, the clip against the occupied bricks and the tile look-up
    float d[3], tmax, step, pad_;
};

// Fetch rounds a lane spends on its ray before it hands the rest over: bounds the serial,
// latency-bound chain a single slow ray (one that grazes the object without ever becoming opaque)
// can keep its wave alive for.
#ifndef VR_LANE_BUDGET
#define VR_LANE_BUDGET 128
#endif
constexpr int kLaneBudget = VR_LANE_BUDGET;
#ifndef VR_DECIDE_EVERY
#define VR_DECIDE_EVERY 4
#endif
constexpr int kDecideEvery = VR_DECIDE_EVERY;   // a power of two
#ifndef VR_MODEL_CELLS
#define VR_MODEL_CELLS false
#endif
// In empty space the cell of a tap may be taken from the line model (cheaper than its exact position,
// but a model cell that points into an occupied brick costs a second pass with the exact one).  With
// the per-tile depth ranges a ray starts next to the surface and spends few passes in open space, and
// the exact position every time is the faster choice (fast march 1.70 -> 1.48 ms); a box of a split
// volume keeps the model, which also walks it through the space of the other boxes.
constexpr bool kModelCells = VR_MODEL_CELLS;
#ifndef VR_TAP_GROUP
#define VR_TAP_GROUP 1
#endif
constexpr int kTapGroup = VR_TAP_GROUP;   // taps a lane inside material fetches per round

// ---- pieces shared with the sort-last path (vr_sortlast.hpp) ------------------------------------------
// There a context holds one box of the volume's bricks and marches only the taps whose filter
// footprint starts in that box; per pixel it produces kPartial floats (F0, F1, F2, Tr, S).
constexpr int kPartial = 5;             // floats per pixel of a partial result
// A box stops marching a ray once the transmittance of its own taps is below this: what it leaves
// out changes the ray's colour by at most 2 * kTrCut * (largest colour), which k_decide_partials
// adds to its error interval.
constexpr double kTrCut = 0x1p-20;

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

// ---- front-to-back march ------------------------------------------------------------------------
// Visits the taps from the eye towards the back, steps over empty cubes of the occupancy pyramid,
// and stops a ray as soon as its RGBA8 value is decided.
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
// the back tap is appended to a queue and re-marched in reference order by k_replay_exact.
// The loop leaves a wave when no lane has a tap left (wavefront ballot): lanes first step over
// empty space (cheap, divergent), then all lanes that found a tap fetch it together.
//
// One lane marches one ray; the 64 rays of a wave are the pixels of an 8x8 tile.
template <typename T, bool STATS, bool PARTIAL, bool SHADE = false>
__device__ __forceinline__ void march_lanes(const VolumeView &vol, const FrameParams &fp, const float4 *s_lut,
                                            const Decider &dec, uint32_t *__restrict__ out, Counters *ctr,
                                            uint32_t *touched, ReplayRay *__restrict__ queue,
                                            LongRay *__restrict__ next_queue, unsigned int *next_count, int budget,
                                            bool in_rect, int px, int py, uint32_t pixel_id, float *__restrict__ partial_out,
                                            const uint2 *__restrict__ tiles, int tiles_x)
{
#ifdef VR_DIAG
    const unsigned long long diag_t0 = wall_clock64();
    unsigned int diag_model = 0, diag_exact = 0, diag_rounds = 0, diag_sampled = 0, diag_wave_passes = 0, diag_wave_passes0 = 0, diag_pyramid = 0;
    long long diag_c0 = clock64(), diag_setup = 0, diag_skip = 0, diag_sample = 0;
    unsigned long long diag_hist = 0;
    unsigned int diag_preps = 0, diag_decides = 0;   // rounds in which the wave prepared taps of lanes inside material / ran the decision test
    long long diag_decide = 0;
#endif
    const Ray r = make_ray(fp, in_rect ? px : fp.x0, in_rect ? py : fp.y0);
    const bool fast_ok = dec.ok;

    const int last = fp.samples + 1;
    const float step = (r.tmax - r.tmin) / (float)(uint32_t)last;
    RayModel rm;
    // more than half of the tiles of a typical frame have no occupied brick behind them: their rays need
    // no line model, only the back-most tap
    const bool nothing = in_rect && tiles && !clip_disabled(tiles, tiles_x, fp) && step > 0.0f &&
                         tiles[(size_t)(py >> kRangeShift) * tiles_x + (px >> kRangeShift)].y == 0u;
    if (nothing) {
        rm = RayModel{};
        rm.t_first = 0;
        rm.t_stop = 0x7fffffff;
    } else {
        rm = ray_model(r, vol, step, -1.0f);
        if (in_rect) clip_to_tile(rm, r, step, tiles, tiles_x, px, py, fp);
    }

    double F0 = 0.0, F1 = 0.0, F2 = 0.0, Tr = 1.0;
    float S = 0.0f;
    int t = last < rm.t_first ? last : rm.t_first;   // front-most tap that can matter
    const int t_stop = rm.t_stop;       // taps below it (the back tap aside) cannot matter either
    bool marching = in_rect && r.hit;   // still has taps with index >= 1 to visit
    bool decided = false, gave_up = false;
    uint32_t rgba = 0xff000000u;        // miss: clear colour (VolumeRendering.comp:174,180)
    unsigned long long evaluated = 0, lookups = 0;

    auto try_decide = [&](int taps_left) -> bool { return dec.decide(F0, F1, F2, Tr, S, taps_left, rgba); };

    uint32_t solid = 0xffffffffu;       // last brick found occupied: its taps need no pyramid lookup
    unsigned long long solid_mask = 0;  // and its 2^3-cell block bits
    constexpr bool model_cells = PARTIAL || kModelCells;
    bool in_air = model_cells;          // the last tap looked at was in an empty cube: the line model leads
    // The last tap fetched had alpha != 0 (or only one transparent tap since): the next tap is fetched
    // without asking the occupancy tables.  Any tap may be fetched, so this is only a bet that saves the
    // look-ups while a ray is inside material; two transparent taps in a row end it.
    bool material = false;
    int clear_run = 0;
    // Work a lane has spent on its ray: taps fetched plus passes of the skipping loop.  Every pass and
    // every fetch is a memory round trip the whole wave waits for, so a ray that needs many of them is
    // handed to a wave of its own (k_march_long) instead of holding 63 other lanes.
    int work = 0;
#ifdef VR_DIAG
    diag_setup = clock64() - diag_c0;
#endif
    for (int round = 0;; round++) {
#ifdef VR_DIAG
        const long long diag_r0 = clock64();
#endif
        // taps this lane fetches this round: tap t in slot 0, and while it is inside material the
        // kTapGroup - 1 taps behind it as well, so that their loads are in flight together
        int taps = 0;
        Cell cs[kTapGroup];
        uint32_t bricks[kTapGroup];
        Cell &c = cs[0];
        uint32_t &brick = bricks[0];
        bool have = false;
        const bool active = marching && work < budget;
#ifdef VR_DIAG
        if (__ballot(material && active) != 0ull) diag_preps++;
#endif
        if (material && active) {
            if (t < t_stop) {
                marching = false;
            } else {
                taps = t - t_stop + 1 < kTapGroup ? t - t_stop + 1 : kTapGroup;
#pragma unroll
                for (int k = 0; k < kTapGroup; k++) {
                    cs[k] = tap_cell(tap_position(r, vol, step, t - k), vol);   // for k >= taps: a valid cell, never fetched
                    bricks[k] = brick_of(cs[k], vol);
                }
                have = true;
                if (PARTIAL) {
                    // sort-last: the tap may be another box's (one tap per round there)
                    static_assert(!PARTIAL || kTapGroup == 1, "the ownership test looks at one tap");
                    if (!tap_cell_local(tap_position(r, vol, step, t), vol, cs[0])) {
                        have = false;
                        taps = 0;
                        material = false;
                        in_air = false;
                    } else {
                        bricks[0] = brick_of(cs[0], vol);
                    }
                }
            }
        }
        // 1. lanes outside material step over empty space until they stand on a tap that must be
        //    fetched.  One pass of the loop looks at one tap: through the line model while in empty
        //    space (cheap), through its exact position otherwise; either way the answer is "inside an
        //    empty cube of the pyramid", "inside a transparent 2^3-cell block" (both: count how many taps
        //    stay inside and step over them) or "fetch it".
        while (marching && !have && work < budget) {
            if (t < t_stop) { marching = false; break; }
            work++;
#ifdef VR_DIAG
            if ((int)(threadIdx.x & 63) == __builtin_ctzll(__ballot(true))) {   // one lane counts for the wave
                diag_wave_passes++;
                if (round == 0) diag_wave_passes0++;
            }
#endif
            // position of tap t in the padded texel grid for the span counts below: the line model's, or,
            // when every pass takes the exact position anyway, that one (it differs from the model by
            // rounding only, which the margin covers just the same, and the model's six coefficients
            // need not stay in registers)
            float gx = 0.0f, gy = 0.0f, gz = 0.0f;
            if (model_cells) {
                const float ft = (float)t;
                gx = __builtin_fmaf(ft, rm.bx, rm.ax);
                gy = __builtin_fmaf(ft, rm.by, rm.ay);
                gz = __builtin_fmaf(ft, rm.bz, rm.az);
            }
            if (in_air) {
#ifdef VR_DIAG
                diag_model++;
#endif
                c.sx = min((int)fminf(fmaxf(gx, 0.0f), 1e7f), vol.lx[0] * kBrick - 1);
                c.sy = min((int)fminf(fmaxf(gy, 0.0f), 1e7f), vol.ly[0] * kBrick - 1);
                c.sz = min((int)fminf(fmaxf(gz, 0.0f), 1e7f), vol.lz[0] * kBrick - 1);
            } else {
#ifdef VR_DIAG
                diag_exact++;
#endif
                if (PARTIAL) {
                    if (!tap_cell_local(tap_position(r, vol, step, t), vol, c)) {
                        // another box's tap (only near the faces of this one): nothing to do with it
                        t -= 1;
                        in_air = true;
                        continue;
                    }
                } else {
                    const Tap p = tap_position(r, vol, step, t);
                    c = tap_cell(p, vol);
                    if (!model_cells) {
                        gx = p.x + (float)kPad;
                        gy = p.y + (float)kPad;
                        gz = p.z + (float)kPad;
                    }
                }
            }
            brick = brick_of(c, vol);
            int level = -1;
            if (brick != solid) {
                if (STATS) lookups++;
                // both tables are read at once (the block bits only matter for an occupied brick found
                // through an exact position): one memory round trip instead of two in a row
                const uint32_t code = skip_code(brick, vol);
                unsigned long long bits = 0;
                if (!in_air) bits = vol.micro[brick];
                level = (int)code - 1;
                if (level < 0) {
                    if (in_air) {
                        // the model points into an occupied brick: look again with the exact position
                        in_air = false;
                        continue;
                    }
                    solid = brick;
                    solid_mask = bits;
                }
            } else if (in_air) {
                in_air = false;
                continue;
            }
            int lox, loy, loz;
            float extent;
            if (level >= 0) {
#ifdef VR_DIAG
                diag_pyramid++;
#endif
                const int shift = 3 + 2 * level;
                lox = (c.sx >> shift) << shift;
                loy = (c.sy >> shift) << shift;
                loz = (c.sz >> shift) << shift;
                extent = (float)(1 << shift);
            } else if (micro_empty(solid_mask, c)) {
                lox = c.sx & ~1;
                loy = c.sy & ~1;
                loz = c.sz & ~1;
                extent = 2.0f;
            } else {
                have = true;
                taps = 1;
                // a block that is not transparent: bet on the taps behind this one as well (at a surface the
                // first of them are still air; each used to cost a pass of this loop and a fetch)
                material = kMaterialBet;
                clear_run = 0;
                continue;
            }
            const int n = span_inside(gx, gy, gz, lox, loy, loz, extent, rm);
            if (in_air) {
                // the cell came from the model: only a model position inside the shrunk cube proves anything
                if (n > 0) t -= n;
                else in_air = false;
            } else {
                // exact cell: tap t itself is transparent for certain, the model counts the taps after it
                t -= n > 1 ? n : 1;
                in_air = model_cells && level >= 0;
            }
        }
        // 2. the wave leaves the loop when no lane has a tap to fetch
#ifdef VR_DIAG
        const long long diag_r1 = clock64();
        diag_skip += diag_r1 - diag_r0;
#endif
        if (__ballot(have) == 0ull) break;
#ifdef VR_DIAG
        diag_rounds++;
        diag_sampled += (unsigned int)taps;
        {   // fetch rounds by how many lanes fetch in them: 1..8, 9..16, 17..32, 33..48 (the rest: 49..64)
            const int fetching = __popcll(__ballot(have));
            diag_hist += fetching <= 8 ? 1ull : fetching <= 16 ? (1ull << 8) : fetching <= 32 ? (1ull << 16) : fetching <= 48 ? (1ull << 24) : 0ull;
        }
#endif
        if (have) {
            work++;
            TapWords raw[kTapGroup];
#pragma unroll
            for (int k = 0; k < kTapGroup; k++)
                if (k < taps) raw[k] = fetch_volume<T>(cs[k], bricks[k], vol);
#pragma unroll
            for (int k = 0; k < kTapGroup; k++) {
                if (k >= taps) break;
                float4 tf = sample_tf(s_lut, filter_volume<T>(raw[k], cs[k], vol));
                if (STATS) {
                    evaluated++;
                    atomicOr(&touched[bricks[k] >> 5], 1u << (bricks[k] & 31));
                }
                if (tf.w != 0.0f) {
                    if (SHADE) {
                        static_assert(!SHADE || !PARTIAL, "a box's apron does not hold the neighbours the gradient reads");
                        const float sh = shade_tap<T, STATS>(vol, fp, r, step, t - k, touched, evaluated);
                        tf.x += sh; tf.y += sh; tf.z += sh;
                    }
                    const double a = (double)tf.w;
                    const double w = Tr * a;
                    F0 = __builtin_fma(w, (double)tf.x, F0);
                    F1 = __builtin_fma(w, (double)tf.y, F1);
                    F2 = __builtin_fma(w, (double)tf.z, F2);
                    S += (float)Tr;
                    Tr = Tr * (1.0 - a);
                    material = kMaterialBet;
                    clear_run = 0;
                    if (PARTIAL && Tr < kTrCut) marching = false;   // dark: whatever follows in this box cannot matter
                } else if (material) {
                    clear_run++;
                    if (clear_run >= kClearRun) {
                        material = false;
                        in_air = false;
                    }
                }
            }
            t -= taps;
        }
        // 3. early termination, tested every few rounds for all lanes at once: the test costs about as
        //    much as half a tap and the whole wave pays for it whenever a single lane runs it
#ifdef VR_DIAG
        const long long diag_d0 = clock64();
        if (__ballot(!PARTIAL && (round & (kDecideEvery - 1)) == kDecideEvery - 1 && fast_ok && marching && Tr < (double)dec.tr_gate) != 0ull) diag_decides++;
#endif
        if (!PARTIAL && (round & (kDecideEvery - 1)) == kDecideEvery - 1 && fast_ok && marching && Tr < (double)dec.tr_gate) {
            if (try_decide(t + 1)) {     // taps t .. 0 are still to come
                decided = true;
                marching = false;
            } else if (Tr < kReplayDark) {
                // dark and still on a rounding boundary: nothing behind can decide it any more (the interval
                // is as narrow as the rounding bound allows); the replay folds it from here to the front
                gave_up = true;
                marching = false;
            }
        }
#ifdef VR_DIAG
        diag_decide += clock64() - diag_d0;
        diag_sample += clock64() - diag_r1;
#endif
    }

#ifdef VR_DIAG
    // what the wave does behind the loop: bit 0 no occupied brick behind the tile, 1 hands rays over, 2 fetches back taps
    // (and runs the decision test on them), 3 queues rays for the replay
    const unsigned long long diag_flags = (nothing ? 1ull : 0ull) | (__ballot(marching) != 0ull ? 2ull : 0ull) |
                                          (__ballot(!marching && !PARTIAL && in_rect && r.hit && !gave_up && !decided) != 0ull ? 4ull : 0ull) |
                                          (__ballot(gave_up) != 0ull ? 8ull : 0ull);
#endif
    if (marching) {
        // out of budget with taps left: the next pass continues this ray
        LongRay e;
        e.pixel_id = pixel_id; e.t = t; e.S = S; e.t_stop = t_stop;
        e.F0 = F0; e.F1 = F1; e.F2 = F2; e.Tr = Tr;
        e.d[0] = r.d[0]; e.d[1] = r.d[1]; e.d[2] = r.d[2];
        e.tmax = r.tmax; e.step = step; e.pad_ = 0.0f;
        // one addition for all the rays the wave hands over (additions on one word take their turn), which
        // also keeps a tile's rays together in the queue
        const unsigned long long givers = __ballot(true);
        const int leader = __builtin_ctzll(givers), me = (int)(threadIdx.x & 63);
        uint32_t first = 0u;
        if (me == leader) first = atomicAdd(next_count, (uint32_t)__popcll(givers));
        first = (uint32_t)__builtin_amdgcn_readlane((int)first, leader);
        next_queue[first + (uint32_t)__popcll(givers & ((1ull << me) - 1ull))] = e;
    } else if (PARTIAL) {
        if (in_rect) {
            float *o = partial_out + (size_t)pixel_id * kPartial;
            if (r.hit) finish_partial<T>(vol, r, step, s_lut, F0, F1, F2, Tr, S, o);
            else o[0] = o[1] = o[2] = o[3] = o[4] = 0.0f;
        }
    } else if (in_rect) {
        if (r.hit && gave_up) {
            queue[atomicAdd(&ctr->replay_count, 1u)] = ReplayRay{pixel_id, t + 1};
        } else if (r.hit && !decided) {
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
            if (fast_ok && S == 0.0f) {
                // no tap in front of it had alpha != 0: every blend was an identity and the shader's colour
                // is this tap's, bit for bit (most pixels of tiles no occupied brick projects into)
                rgba = pack_rgba8(tf.x, tf.y, tf.z);
                decided = true;
            } else {
                decided = fast_ok && try_decide(0);
            }
            if (!decided) queue[atomicAdd(&ctr->replay_count, 1u)] = ReplayRay{pixel_id, 0};
        }
        if (!r.hit || decided) out[pixel_id] = rgba;
    }
    if (STATS && in_rect) {
        atomicAdd(&ctr->rays_hit, r.hit ? 1ull : 0ull);
        atomicAdd(&ctr->samples_evaluated, evaluated);
        atomicAdd(&ctr->cells_read, lookups);
        if (r.hit && !PARTIAL) atomicAdd(&ctr->hist_lane[63 - __builtin_clzll(evaluated + 1ull)], 1u);
    }
#ifdef VR_DIAG
    // diagnostic build only: wave start/end wall clock (100 MHz) and placement, one record per wave
    unsigned int diag_both = diag_model + diag_exact, diag_sum = diag_sampled, diag_lane_passes = diag_model + diag_exact;
    for (int o = 32; o > 0; o >>= 1) {
        diag_wave_passes += (unsigned int)__shfl_xor((int)diag_wave_passes, o);
        diag_wave_passes0 += (unsigned int)__shfl_xor((int)diag_wave_passes0, o);
        diag_lane_passes += (unsigned int)__shfl_xor((int)diag_lane_passes, o);
        diag_pyramid += (unsigned int)__shfl_xor((int)diag_pyramid, o);
        diag_model = max(diag_model, (unsigned int)__shfl_xor((int)diag_model, o));
        diag_exact = max(diag_exact, (unsigned int)__shfl_xor((int)diag_exact, o));
        diag_both = max(diag_both, (unsigned int)__shfl_xor((int)diag_both, o));
        diag_sum += (unsigned int)__shfl_xor((int)diag_sum, o);
    }
    if (g_vr_timeline && !PARTIAL && (threadIdx.x & 63) == 0) {
        const unsigned long long wave_id = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        unsigned int hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned int xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_vr_timeline[wave_id * kTimelineWords + 4] = (unsigned long long)diag_setup;
        g_vr_timeline[wave_id * kTimelineWords + 5] = (unsigned long long)diag_skip;
        g_vr_timeline[wave_id * kTimelineWords + 6] = (unsigned long long)diag_sample;
        g_vr_timeline[wave_id * kTimelineWords + 7] = ((unsigned long long)(clock64() - diag_c0) & 0xffffffffull) | (diag_hist << 32);
        g_vr_timeline[wave_id * kTimelineWords + 8] = (unsigned long long)diag_preps | ((unsigned long long)diag_decides << 16) | (diag_flags << 32);
        g_vr_timeline[wave_id * kTimelineWords + 9] = (unsigned long long)diag_decide;
        g_vr_timeline[wave_id * kTimelineWords + 0] = diag_t0;
        g_vr_timeline[wave_id * kTimelineWords + 1] = wall_clock64();
        (void)hw; (void)xcc;
        g_vr_timeline[wave_id * kTimelineWords + 2] = (unsigned long long)diag_wave_passes | ((unsigned long long)diag_wave_passes0 << 16) | ((unsigned long long)diag_lane_passes << 32) | ((unsigned long long)diag_pyramid << 48);
        g_vr_timeline[wave_id * kTimelineWords + 3] = (unsigned long long)diag_rounds | ((unsigned long long)diag_model << 12) |
                                         ((unsigned long long)diag_exact << 24) | ((unsigned long long)diag_both << 36) |
                                         ((unsigned long long)diag_sum << 48);
    }
#endif
}

// Longest tiles first.  The waves of a launch take very different times (a tile of background 6 us, a tile
// on the silhouette up to 250), and the launch ends when the last of them does: in launch order the last
// heavy tiles of the image run on a nearly empty chip.  Every wave therefore notes how long its tile took;
// k_order_tiles sorts the tiles into kOrderBuckets lists by that time, and the next launch over the same
// tiles hands them out list after list, longest first, so what is left for the end are the short ones.  The camera moves little from frame
// to frame, and any permutation of the tiles gives the same frame, so a stale order only costs time.
// Layout of an order buffer (uint32): [0, kOrderBuckets) entries per list, lists from kOrderLists on,
// `cap` entries apart.
constexpr int kOrderBuckets = 16;
constexpr int kOrderLists = 32;
constexpr int kOrderShift = 11;   // list = 15 - duration / 2^11 ticks of 10 ns: 20 us a list

// The tile wave `wave_id` of `tiles` renders: its own in launch order, or entry wave_id of the lists.
__device__ __forceinline__ uint32_t ordered_tile(const uint32_t *__restrict__ order_in, uint32_t cap, uint32_t wave_id, uint32_t tiles)
{
    if (!order_in || wave_id >= tiles) return wave_id;
    const uint32_t mine = order_in[threadIdx.x & (kOrderBuckets - 1)];
    uint32_t before = 0, tile = wave_id;
    bool found = false;
#pragma unroll
    for (int b = 0; b < kOrderBuckets; b++) {
        const uint32_t n = (uint32_t)__builtin_amdgcn_readlane((int)mine, b);
        if (!found && wave_id - before < n) {
            tile = order_in[kOrderLists + (size_t)b * cap + (wave_id - before)];
            found = true;
        }
        before += n;
    }
    // lists that do not hold exactly this launch's tiles are not used by any wave
    return before == tiles && tile < tiles ? tile : wave_id;
}

// The lists for the next launch out of the times the waves of this one noted (ticks of 10 ns per tile; a
// plain store per wave: 130 000 atomic additions on sixteen words at the end of the waves cost a third of
// the march).  One thread per tile; a block ranks its tiles per list in LDS and reserves room in each
// list with one addition, so the tiles of a list stay in image order block after block.
__global__ __launch_bounds__(1024) void k_order_tiles(const uint32_t *__restrict__ ticks, uint32_t tiles, uint32_t *__restrict__ order,
                                                      uint32_t cap)
{
    __shared__ uint32_t s_count[kOrderBuckets], s_base[kOrderBuckets];
    if (threadIdx.x < kOrderBuckets) s_count[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t b = 0, rank = 0;
    if (i < tiles) {
        const uint32_t t = ticks[i] >> kOrderShift;
        b = t >= (uint32_t)kOrderBuckets ? 0u : (uint32_t)(kOrderBuckets - 1) - t;
        rank = atomicAdd(&s_count[b], 1u);
    }
    __syncthreads();
    if (threadIdx.x < kOrderBuckets) s_base[threadIdx.x] = atomicAdd(&order[threadIdx.x], s_count[threadIdx.x]);
    __syncthreads();
    if (i < tiles) order[kOrderLists + (size_t)b * cap + s_base[b] + rank] = i;
}

template <typename T, bool STATS, bool SHADE>
__global__ __launch_bounds__(256, (STATS || SHADE) ? 1 : VR_FAST_MIN_WAVES) void k_march_fast(VolumeView vol, FrameParams fp, const float4 *__restrict__ lut,
                                                   float4 border, const LutInfo *__restrict__ info,
                                                   uint32_t *__restrict__ out, Counters *ctr, uint32_t *touched,
                                                   ReplayRay *__restrict__ queue, LongRay *__restrict__ long_queue, int budget,
                                                   const uint2 *__restrict__ tiles, int tiles_x,
                                                   const uint32_t *__restrict__ order_in, uint32_t *__restrict__ ticks_out,
                                                   uint32_t order_cap, uint32_t order_tiles)
{
    __shared__ float4 s_lut[kLutLds];
    const unsigned long long began = ticks_out ? wall_clock64() : 0ull;
    stage_lut(s_lut, lut, border);
    const Decider dec(info);
    int px, py;
    uint32_t pixel_id;
    const uint32_t wave_id = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t tile = ordered_tile(order_in, order_cap, wave_id, order_tiles);
    const bool in_rect = lane_pixel_of_tile(fp, (int)tile, px, py, pixel_id);
    march_lanes<T, STATS, false, SHADE>(vol, fp, s_lut, dec, out, ctr, touched, queue, long_queue, &ctr->long_count, budget, in_rect, px,
                                 py, pixel_id, nullptr, tiles, tiles_x);
    if (ticks_out && wave_id < order_tiles && (threadIdx.x & 63) == 0) {
        const unsigned long long ticks = wall_clock64() - began;
        ticks_out[tile] = ticks > 0xffffffffull ? 0xffffffffu : (uint32_t)ticks;
    }
}

// What a run of consecutive taps does to the light behind it, taken front to back from full
// transmittance: colour sum f, transmittance tr, and s = sum of the transmittance in front of each of
// its taps with a != 0 (the weight of their rounding errors).  Runs concatenate associatively.
struct Blend {
    double f0, f1, f2, tr;
    float s;
};
__device__ __forceinline__ Blend blend_of_tap(bool on, const float4 &tf)
{
    const double a = on ? (double)tf.w : 0.0;
    return Blend{a * (double)tf.x, a * (double)tf.y, a * (double)tf.z, 1.0 - a, on ? 1.0f : 0.0f};
}
__device__ __forceinline__ void blend_under(Blend &front, const Blend &back)
{
    front.f0 = __builtin_fma(front.tr, back.f0, front.f0);
    front.f1 = __builtin_fma(front.tr, back.f1, front.f1);
    front.f2 = __builtin_fma(front.tr, back.f2, front.f2);
    front.s = front.s + (float)front.tr * back.s;
    front.tr = front.tr * back.tr;
}
#ifndef VR_SERIAL_FOLD_MAX
#define VR_SERIAL_FOLD_MAX 8
#endif
constexpr int kSerialFoldMax = VR_SERIAL_FOLD_MAX;    // up to this many lanes' taps of a chunk are blended one lane after the other
#ifndef VR_LONG_TAPS
#define VR_LONG_TAPS 2
#endif
constexpr int kLongTaps = VR_LONG_TAPS;   // consecutive taps per lane and round in the wave-per-ray march

// Lanes that march one handed-over ray together (template parameter G of the wave-per-ray kernels): a whole
// wave, or half a wave so that a wave carries two rays at a time.  The march is a chain of dependent
// memory round trips; two rays in a wave share its instructions and its slot on the SIMD, which pays
// when there are many rays with few taps left each (small frames, where most rays are handed over early);
// a whole wave covers 128 taps per round trip, which pays for the long rays of large volumes.
// Continuation of a ray k_march_fast ran out of budget on, by a group of G lanes.  All control
// flow below is uniform within a group (the two groups of a wave may diverge); every shuffle and ballot
// stays inside the group.  The lanes fetch G * kLongTaps consecutive taps at once, then the taps
// with a non-zero alpha are blended front to back in tap order, the accumulation k_march_fast does one
// tap at a time; the decision test runs once per round.  While the ray is inside material a round is a
// single memory round trip: the block masks of a round's taps are requested one round ahead, together
// with the voxels of the round before, so the voxels of the taps that matter can be requested as soon
// as the round starts.  After a round without any tap that matters the group walks on through the
// occupancy pyramid, a cube at a time.  `valid`: the group has a ray at all.
// Returns whether the bytes are decided.
// With PARTIAL (sort-last) only the taps of this context's box are taken, nothing is decided, the
// march stops once the box's own taps have made the ray dark, and the sums go to partial_out.
template <typename T, int G, bool STATS, bool PARTIAL = false, bool SHADE = false>
__device__ __forceinline__ bool continue_ray(const VolumeView &vol, const FrameParams &fp, const float4 *s_lut,
                                             const Decider &dec, Counters *ctr, uint32_t *touched, const LongRay &e, bool valid,
                                             uint32_t &rgba, int &dark_from, float *__restrict__ partial_out = nullptr)
{
    static_assert(G == 16 || G == 32 || G == 64, "a group is a wave, a half or a quarter of one");
    const int lane = threadIdx.x & 63;
    const int sub = lane & (G - 1);          // lane within the group
    const int gbase = lane & ~(G - 1);       // first lane of the group
    auto group_ballot = [&](bool flag) -> unsigned long long {
        const unsigned long long m = __ballot(flag);
        return G == 64 ? m : ((m >> gbase) & ((1ull << (G & 63)) - 1ull));
    };
    // the ray and its clipped tap range as the lane march left them (identical in every lane of the group)
    Ray r;
    r.o[0] = fp.loc[0]; r.o[1] = fp.loc[1]; r.o[2] = fp.loc[2];
    r.d[0] = e.d[0]; r.d[1] = e.d[1]; r.d[2] = e.d[2];
    r.tmax = e.tmax;
    r.tmin = 0.0f;      // not needed any more: the step is carried
    r.hit = true;
    const float step = e.step;
    RayModel rm = ray_line(r, vol, step, -1.0f);
    rm.t_stop = e.t_stop;
    double F0 = e.F0, F1 = e.F1, F2 = e.F2, Tr = e.Tr;
    float S = e.S;
    int t = e.t;
    bool decided = false;
    rgba = 0xff000000u;
    dark_from = 0;
    unsigned long long evaluated = 0;
    // the back-most tap is needed when the march ends undecided: requested together with the last round's taps
    const Tap back_p = tap_position(r, vol, step, 0);
    const Cell back_c = tap_cell(back_p, vol);
    const uint32_t back_brick = brick_of(back_c, vol);
    TapWords back_raw{};
    bool back_requested = false;
    // One round's taps of a lane: cell, brick, whether the tap exists (and is this box's), and the brick's
    // block mask, which is requested a round ahead so that it has arrived when the round starts.
    struct Slot {
        Cell c;
        uint32_t brick;
        bool fetch;
        unsigned long long mask;
    };
    auto prepare = [&](int t0, Slot *slot) {
#pragma unroll
        for (int k = 0; k < kLongTaps; k++) {
            const int tt = t0 - (sub * kLongTaps + k);
            slot[k].fetch = valid && tt >= rm.t_stop;
            if (PARTIAL) {
                if (!tap_cell_local(tap_position(r, vol, step, slot[k].fetch ? tt : rm.t_stop), vol, slot[k].c)) slot[k].fetch = false;
                if (!slot[k].fetch) slot[k].c.sx = slot[k].c.sy = slot[k].c.sz = 0;   // any cell of the box: never used
            } else {
                slot[k].c = tap_cell(tap_position(r, vol, step, slot[k].fetch ? tt : rm.t_stop), vol);
            }
            slot[k].brick = brick_of(slot[k].c, vol);
            slot[k].mask = ~0ull;
            if (slot[k].fetch) slot[k].mask = vol.micro[slot[k].brick];   // all ones for a brick the pyramid calls empty
        }
    };
    Slot cur[kLongTaps], nxt[kLongTaps];
    int cur_t = t;
    prepare(t, cur);
    bool spec = true;   // the hand-over happens inside material: expect the next taps to matter
    while (valid && t >= rm.t_stop && !decided && !(PARTIAL && Tr < kTrCut)) {
        if (!spec) {
            const int n = empty_span(vol, rm, t);   // group-uniform
            if (n > 0) {
                t -= n;
                continue;
            }
        }
        if (cur_t != t) {      // the group has jumped: the masks requested ahead were for other taps
            prepare(t, cur);
            cur_t = t;
        }
        // lane l takes the kLongTaps consecutive taps t - l * kLongTaps - k; the voxels of those whose block is
        // not transparent and the masks of the next round's taps are requested together
        TapWords raw[kLongTaps];
        bool need[kLongTaps];
#pragma unroll
        for (int k = 0; k < kLongTaps; k++) {
            need[k] = cur[k].fetch && !micro_empty(cur[k].mask, cur[k].c);
            if (need[k]) raw[k] = fetch_volume<T>(cur[k].c, cur[k].brick, vol);
        }
        const int t_next = t - G * kLongTaps;
        if (!PARTIAL && !back_requested && t_next < rm.t_stop) {
            back_raw = fetch_volume<T>(back_c, back_brick, vol);
            back_requested = true;
        }
        prepare(t_next, nxt);
        Blend v{0.0, 0.0, 0.0, 1.0, 0.0f};
#pragma unroll
        for (int k = 0; k < kLongTaps; k++) {
            if (!need[k]) continue;
            float4 tf = sample_tf(s_lut, filter_volume<T>(raw[k], cur[k].c, vol));
            if (STATS) {
                evaluated++;
                atomicOr(&touched[cur[k].brick >> 5], 1u << (cur[k].brick & 31));
            }
            if (SHADE && tf.w != 0.0f) {
                const float sh = shade_tap<T, STATS>(vol, fp, r, step, t - (sub * kLongTaps + k), touched, evaluated);
                tf.x += sh; tf.y += sh; tf.z += sh;
            }
            if (tf.w != 0.0f) blend_under(v, blend_of_tap(true, tf));
        }
        bool mattered = false;
#pragma unroll
        for (int k = 0; k < kLongTaps; k++) mattered = mattered || need[k];
        spec = group_ballot(mattered) != 0ull;
        unsigned long long todo = group_ballot(v.s != 0.0f);
        if (__popcll(todo) > kSerialFoldMax) {
            // many taps to blend: "front over back" is associative, so the lanes reduce in log2(G)
            // shuffle steps instead of one lane at a time (lane 0 of the group holds the front-most taps)
            for (int o = 1; o < G; o <<= 1) {
                Blend back;
                back.f0 = __shfl_down(v.f0, o, G);
                back.f1 = __shfl_down(v.f1, o, G);
                back.f2 = __shfl_down(v.f2, o, G);
                back.tr = __shfl_down(v.tr, o, G);
                back.s = __shfl_down(v.s, o, G);
                blend_under(v, back);   // lanes whose partner lies beyond the group never feed its lane 0
            }
            todo = 1ull;
        }
        while (todo) {
            const int src = gbase + __builtin_ctzll(todo);   // lowest lane = front-most taps of the chunk
            todo &= todo - 1;
            Blend part;
            part.f0 = __shfl(v.f0, src);
            part.f1 = __shfl(v.f1, src);
            part.f2 = __shfl(v.f2, src);
            part.tr = __shfl(v.tr, src);
            part.s = __shfl(v.s, src);
            F0 = __builtin_fma(Tr, part.f0, F0);
            F1 = __builtin_fma(Tr, part.f1, F1);
            F2 = __builtin_fma(Tr, part.f2, F2);
            S += (float)Tr * part.s * 1.000001f;
            Tr = Tr * part.tr;
        }
        t = t_next;
#pragma unroll
        for (int k = 0; k < kLongTaps; k++) cur[k] = nxt[k];
        cur_t = t;
        if (!PARTIAL && dec.ok && Tr < (double)dec.tr_gate) {
            decided = dec.decide(F0, F1, F2, Tr, S, (t > 0 ? t : 0) + 1, rgba);
            if (!decided && Tr < kReplayDark) {
                dark_from = t + 1;     // dark and still on a rounding boundary: the replay folds it from here
                break;
            }
        }
    }
    if (PARTIAL) {
        float res[kPartial];
        finish_partial<T>(vol, r, step, s_lut, F0, F1, F2, Tr, S, res);
        if (valid && sub == 0) {
            float *o = partial_out + (size_t)e.pixel_id * kPartial;
            for (int k = 0; k < kPartial; k++) o[k] = res[k];
        }
        return false;
    }
    if (valid && !decided && dark_from == 0) {
        // back-most tap: its colour enters unweighted by its alpha (VolumeRendering.comp:189-191)
        if (!back_requested) back_raw = fetch_volume<T>(back_c, back_brick, vol);
        const float4 tf = sample_tf(s_lut, filter_volume<T>(back_raw, back_c, vol));
        if (STATS && sub == 0) {
            evaluated++;
            atomicOr(&touched[back_brick >> 5], 1u << (back_brick & 31));
        }
        F0 = __builtin_fma(Tr, (double)tf.x, F0);
        F1 = __builtin_fma(Tr, (double)tf.y, F1);
        F2 = __builtin_fma(Tr, (double)tf.z, F2);
        decided = dec.ok && dec.decide(F0, F1, F2, 0.0, S, 0, rgba);
    }
    if (STATS && valid) {
        atomicAdd(&ctr->samples_evaluated, evaluated);
        atomicAdd(&ctr->long_samples, evaluated);
        for (int o = G / 2; o > 0; o >>= 1) evaluated += __shfl_xor(evaluated, o);
        if (sub == 0) atomicAdd(&ctr->hist_long[63 - __builtin_clzll(evaluated + 1ull)], 1u);
    }
    return decided;
}

// Which entries of a hand-over queue the groups of a wave-per-ray kernel take (GROUPS groups of
// 64 / GROUPS lanes per wave, one ray each).  The lane march appends the rays of a wave (one 8x8 pixel
// tile) in one piece, so neighbours in the queue are neighbours on the screen and walk the same bricks.
// Workgroups are dealt round-robin to the chip's eight XCDs, each with an L2 of its own (blocks b and
// b + 8 share one): were consecutive entries handed to consecutive blocks, every XCD would fetch every
// tile's bricks from memory for itself.  Instead a run of consecutive entries stays on one XCD: four
// blocks of that XCD take a quarter of the run each, and the groups of a block work on neighbouring
// entries at any time.  A group takes up to four entries of a run in turn (one when the queue is short
// enough for every entry to have a group of its own).  Calls f(entry, valid) with all lanes of the wave,
// valid = false for a group without an entry (entry is then 0); the groups of the grid together cover
// [0, entries) once.  Correct for any placement of the blocks, only the cache hit rate depends on it.
// gridDim.x must be a multiple of 32.
template <int GROUPS, typename F>
__device__ __forceinline__ void for_queue_entries(uint32_t entries, F f)
{
    const uint32_t group = (threadIdx.x >> 6) * GROUPS + ((threadIdx.x & 63u) / (64u / GROUPS));   // within the block
    constexpr uint32_t per_block = 4u * GROUPS;
    const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;       // slot: index among the blocks of this XCD
    const uint32_t quads = gridDim.x >> 5;                               // runs one XCD works on at a time
    const uint32_t all_groups = gridDim.x * per_block;
    const uint32_t per_group = min(4u, max(1u, (entries + all_groups - 1u) / all_groups));
    const uint32_t run = 4u * per_block * per_group;
    const uint32_t runs = (entries + run - 1u) / run;
    for (uint32_t q = slot >> 2; q * 8u + xcd < runs; q += quads) {
        const uint32_t first = (q * 8u + xcd) * run + (slot & 3u) * per_block * per_group;
        for (uint32_t k = 0; k < per_group; k++) {
            const uint32_t entry = first + k * per_block + group;
            const bool valid = entry < entries;
            f(valid ? entry : 0u, valid);
        }
    }
}

// The rays k_march_fast ran out of lane budget on, one wave per ray; rays still undecided at the back
// join the replay queue.
#ifndef VR_LONG_MIN_WAVES
#define VR_LONG_MIN_WAVES 4
#endif
template <typename T, int G, bool STATS, bool SHADE>
__global__ __launch_bounds__(256, (STATS || SHADE) ? 1 : VR_LONG_MIN_WAVES) void k_march_long(VolumeView vol, FrameParams fp, const float4 *__restrict__ lut,
                                                   float4 border, const LutInfo *__restrict__ info,
                                                   uint32_t *__restrict__ out, Counters *ctr, uint32_t *touched,
                                                   ReplayRay *__restrict__ queue, const LongRay *__restrict__ long_queue)
{
    __shared__ float4 s_lut[kLutLds];
    stage_lut(s_lut, lut, border);
    const Decider dec(info);
    const int sub = threadIdx.x & (G - 1);
    const uint32_t entries = ctr->long_count;
    if (entries == 0u) return;
    for_queue_entries<64 / G>(entries, [&](uint32_t entry, bool valid) {
        const LongRay e = long_queue[entry];
        uint32_t rgba;
        int dark_from;
        const bool decided = continue_ray<T, G, STATS, false, SHADE>(vol, fp, s_lut, dec, ctr, touched, e, valid, rgba, dark_from);
        if (valid && sub == 0) {
            if (decided) out[e.pixel_id] = rgba;
            else queue[atomicAdd(&ctr->replay_count, 1u)] = ReplayRay{e.pixel_id, dark_from};
        }
    });
}

// The rays neither march could decide, re-marched in reference order, one wave per ray.
template <typename T, bool STATS, bool SHADE>
__global__ __launch_bounds__(256) void k_replay_exact(VolumeView vol, FrameParams fp, const float4 *__restrict__ lut,
                                                     float4 border, const LutInfo *__restrict__ info,
                                                     uint32_t *__restrict__ out, Counters *ctr, uint32_t *touched,
                                                     const ReplayRay *__restrict__ queue, const uint2 *__restrict__ tiles,
                                                     int tiles_x)
{
    __shared__ float4 s_lut[kLutLds];
    stage_lut(s_lut, lut, border);
    const Decider dec(info);
    for_queue_entries<1>(ctr->replay_count, [&](uint32_t entry, bool valid) {
        if (valid) {
            const ReplayRay e = queue[entry];
            replay_ray<T, STATS, false, SHADE>(vol, fp, s_lut, dec, out, ctr, touched, e.pixel_id, tiles, tiles_x, nullptr, nullptr, e.dark_from);
        }
    });
}

// Behind the last kernel of a fast frame: tells the host, through a word of page-locked host memory the GPU
// writes directly, how many of the frame's rays ended in the replay (the host reads the word whenever it
// next renders and never waits for it), and optionally clears the frame's counters for the next frame
// that uses them.  word = frame sequence number << 32 | rays queued for the replay.
__global__ __launch_bounds__(64) void k_finish_frame(Counters *ctr, unsigned long long *host_word, uint32_t seq, int clear)
{
    const unsigned long long replayed = ctr->replay_count;
    __syncthreads();   // every lane has read the count before any lane clears it
    if (threadIdx.x == 0)
        __hip_atomic_store(host_word, ((unsigned long long)seq << 32) | replayed, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (clear) {
        uint32_t *words = reinterpret_cast<uint32_t *>(ctr);
        for (uint32_t k = threadIdx.x; k < (uint32_t)(offsetof(Counters, lut_error) / sizeof(uint32_t)); k += 64u) words[k] = 0u;
    }
}

// Number of set bits of a bit map, added to *total (counting builds: bricks touched so far).
__global__ __launch_bounds__(256) void k_count_bits(const uint32_t *__restrict__ bits, size_t words, unsigned long long *total)
{
    unsigned long long n = 0;
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < words; i += (size_t)gridDim.x * 256u) n += (unsigned long long)__popc(bits[i]);
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o);
    if ((threadIdx.x & 63) == 0 && n) atomicAdd(total, n);
}

// Rays of the rectangle that hit the box (the reference marches samples + 2 taps on each).
__global__ __launch_bounds__(256) void k_count_hits(FrameParams fp, unsigned long long *hits)
{
    __shared__ unsigned int s_hits;
    if (threadIdx.x == 0) s_hits = 0;
    __syncthreads();
    int px, py;
    uint32_t out_index;
    const bool hit = lane_pixel(fp, px, py, out_index) && make_ray(fp, px, py).hit;
    const unsigned long long mask = __ballot(hit);
    if ((threadIdx.x & 63) == 0 && mask) atomicAdd(&s_hits, (unsigned int)__popcll(mask));
    __syncthreads();
    if (threadIdx.x == 0 && s_hits) atomicAdd(hits, (unsigned long long)s_hits);
}

}  // namespace vr
