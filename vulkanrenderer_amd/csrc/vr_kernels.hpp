// gfx950 kernels of the volume-rendering core.  Wave = 64 lanes; one lane marches one ray and a
// wave owns an 8x8 pixel tile, so neighbouring lanes walk neighbouring voxels.
//
// Every float expression that decides a sample value is written so that it rounds exactly like
// the shader it replaces (Applications/VolumeRendering/Shader/VolumeRendering.comp) evaluated
// with one rounding per operation; the file must be compiled with -ffp-contract=off and the
// only fused operations are the explicit __builtin_fmaf calls (texture-filter lerps and the
// exact u16/max quotient).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace vr {

// ---- volume layout ------------------------------------------------------------------------
// Voxel index i of an axis is stored at padded coordinate s = i + 2, so that the border texels
// -1 and size (and one more on each side, reached only through rounding) are ordinary zeros.
// The padded grid is cut into 8^3 bricks; a brick stores the 9^3 voxels s in [8b, 8b+8] of each
// axis (its own 8^3 plus a one-voxel apron), so the 8 corners of any trilinear footprint whose
// low corner lies in the brick come from that one brick.  Row pitch 9, slice pitch 81, brick
// pitch 768 elements (729 used; 1536 B for u16 keeps bricks 128-byte aligned).
constexpr int kBrick = 8;
constexpr int kRow = 9;
constexpr int kSlice = 81;
constexpr int kBrickElems = 768;
constexpr int kPad = 2;

struct VolumeView {
    const void *bricks;
    const uint32_t *cells;  // per brick: min | max << 16 over its 9^3 stored voxels
    int nx, ny, nz;         // voxels
    int nbx, nby, nbz;      // bricks
    float fx, fy, fz;       // (float) nx, ny, nz
    float inv_hi, inv_lo;   // 1/max split in two floats: v/max == fma(v, inv_hi, v*inv_lo) exactly
    float max_value;
    // Occupancy pyramid for the current transfer function.  Level l has one bit per cube of
    // (8 * 4^l)^3 padded voxels (level 0: a brick), set when no tap whose low corner lies inside
    // can receive a non-zero alpha.  Grid of level l: lx[l] x ly[l] x lz[l] cubes.
    const uint32_t *empty[4];
    int lx[4], ly[4], lz[4];
    const int *occ;         // bounding box of the occupied bricks: lo x,y,z then hi x,y,z (brick coordinates)
    const uint32_t *skip;   // 2 bits per brick: 0 occupied, else 1 + coarsest empty level (0..2) around it
    // Below the bricks: per brick one bit per 2^3-cell block (4 x 4 x 4 of them, bit z*16 + y*4 + x),
    // set when every tap whose low corner lies in the block gets alpha exactly 0.
    const unsigned long long *micro;
    // Sub-volume of a sort-last split: the bricks held are those of a box of the whole volume's
    // brick grid and (ox, oy, oz) is the padded coordinate of the box's first cell (a multiple of
    // 8).  nbx.., lx.., occ, skip and micro then describe the box, nx.., fx.. still the whole
    // volume.  All zero when the context holds the whole volume.
    int ox, oy, oz;
};
constexpr int kLevels = 4;

struct FrameParams {
    float m[16];      // toWorld, column-major (VolumeRendering.comp:20-22)
    float loc[4];     // viewPyramid.location (VolumeRendering.comp:14)
    int width, height;
    int x0, y0, x1, y1;  // pixel rectangle to render
    int samples;         // shader constant `samples` (VolumeRendering.comp:173)
    // Sort-first split: the rectangle's rows are cut into bands of 16 rows and this launch
    // renders the bands b with b % parts == part, written densely one after the other.
    int part, parts;
    float scene[4];      // push constants `Scene` (main.cpp:80-98): ambient, diffuse, specular, shininess
};

struct Counters {
    unsigned long long rays_hit;
    unsigned long long samples_evaluated;
    unsigned long long cells_read;
    unsigned long long replay_rays;
    unsigned int replay_count;  // fast mode: rays queued for the reference-order replay
    unsigned int long_count;    // fast mode: rays handed from the lane march to the wave march
    unsigned int pad1_;
    unsigned int pad0_;
    // counting builds only: where the taps of a fast frame are fetched
    unsigned long long long_samples;     // by the wave-per-ray continuation
    unsigned long long replay_samples;   // by the reference-order replay
    unsigned long long main_bricks;      // bricks touched by the lane march alone
    unsigned int hist_lane[24];          // rays by floor(log2(taps the lane march fetched for them + 1))
    unsigned int hist_long[24];          // handed-over rays by floor(log2(taps the wave march fetched + 1))
    // everything above is cleared at the start of a frame; what follows persists
    unsigned int lut_error;
    unsigned int pad_;
};

// LDS image of the transfer function: [0],[1] border, [2..257] the 256 texels, [258],[259] border.
constexpr int kLutLds = 260;

__device__ __forceinline__ float glsl_min(float x, float y) { return y < x ? y : x; }
__device__ __forceinline__ float glsl_max(float x, float y) { return x < y ? y : x; }
__device__ __forceinline__ float lerp_fma(float w, float a, float b) { return __builtin_fmaf(w, b - a, a); }

// floor()ed coordinate -> int, defined for any input: below -2 behaves like -2 and above size
// like size (both neighbours are border texels either way); NaN behaves like -2.
__device__ __forceinline__ int safe_index(float fl, int size)
{
    // one instruction: the median of (fl, -2, size) clamps, and with a NaN among its operands v_med3_f32
    // returns their minimum taken NaN-free, i.e. -2
    return (int)__builtin_amdgcn_fmed3f(fl, -2.0f, (float)size);
}

template <typename T>
struct Voxel;
template <>
struct Voxel<uint16_t> {
    // two neighbouring voxels of a brick row with one (possibly unaligned) 4-byte load
    static __device__ __forceinline__ uint32_t load_pair(const uint16_t *p)
    {
        uint32_t w;
        __builtin_memcpy(&w, p, 4);
        return w;
    }
    static __device__ __forceinline__ void split(uint32_t w, float &a, float &b)
    {
        a = (float)(w & 0xffffu);
        b = (float)(w >> 16);
    }
};
template <>
struct Voxel<uint8_t> {
    static __device__ __forceinline__ uint32_t load_pair(const uint8_t *p)
    {
        uint16_t w;
        __builtin_memcpy(&w, p, 2);
        return w;
    }
    static __device__ __forceinline__ void split(uint32_t w, float &a, float &b)
    {
        a = (float)(w & 0xffu);
        b = (float)(w >> 8);
    }
};

struct Ray {
    float o[3], d[3];
    float tmin, tmax;
    bool hit;
};

// Ray generation and slab test: VolumeRendering.comp:159-160, 177-178, 50-96.
__device__ __forceinline__ Ray make_ray(const FrameParams &fp, int px, int py)
{
    Ray r;
    const float fx = (float)px, fy = (float)py;
    float d[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float t = fp.m[4 + c] * fy + fp.m[12 + c];
        const float world = fp.m[c] * fx + t;
        d[c] = world - fp.loc[c];
        r.o[c] = fp.loc[c];
    }
    const float inv = 1.0f / sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
#pragma unroll
    for (int c = 0; c < 3; c++) r.d[c] = d[c] * inv;
    float t1 = (-0.5f - r.o[0]) / r.d[0], t2 = (0.5f - r.o[0]) / r.d[0];
    const float tminx = glsl_min(t1, t2), tmaxx = glsl_max(t1, t2);
    t1 = (-0.5f - r.o[1]) / r.d[1];
    t2 = (0.5f - r.o[1]) / r.d[1];
    const float tminy = glsl_min(t1, t2), tmaxy = glsl_max(t1, t2);
    t1 = (-1.0f - r.o[2]) / r.d[2];
    t2 = (0.0f - r.o[2]) / r.d[2];
    const float tminz = glsl_min(t1, t2), tmaxz = glsl_max(t1, t2);
    r.tmin = glsl_max(glsl_max(tminx, tminy), tminz);
    r.tmax = glsl_min(glsl_min(tmaxx, tmaxy), tmaxz);
    r.hit = !(r.tmin > r.tmax || r.tmax < 0.0f);
    return r;
}

// Position of tap t in unnormalised, padded texel space (continuous coordinate whose floor is
// the padded index of the low corner): VolumeRendering.comp:185-187 then u*size - 0.5.
struct Tap {
    float x, y, z;     // u*size - 0.5 per axis
};

// sampleLocation of tap t (VolumeRendering.comp:185-186)
__device__ __forceinline__ void tap_world(const Ray &r, float step, int t, float w[3])
{
    const float len = r.tmax - (float)t * step;
    w[0] = r.o[0] + len * r.d[0];
    w[1] = r.o[1] + len * r.d[1];
    w[2] = r.o[2] + len * r.d[2];
}

// abs(location - samplerOrigin) (VolumeRendering.comp:168,187) in unnormalised texel space
__device__ __forceinline__ Tap tap_of_world(float wx, float wy, float wz, const VolumeView &v)
{
    Tap p;
    p.x = fabsf(wx - -0.5f) * v.fx - 0.5f;
    p.y = fabsf(wy - -0.5f) * v.fy - 0.5f;
    p.z = fabsf(wz - 0.0f) * v.fz - 0.5f;
    return p;
}

__device__ __forceinline__ Tap tap_position(const Ray &r, const VolumeView &v, float step, int t)
{
    float w[3];
    tap_world(r, step, t, w);
    return tap_of_world(w[0], w[1], w[2], v);
}

struct Cell {
    int sx, sy, sz;       // padded index of the low corner, each in [0, size + 2]
    float wx, wy, wz;     // filter weights
};

__device__ __forceinline__ Cell tap_cell(const Tap &p, const VolumeView &v)
{
    const float xf = floorf(p.x), yf = floorf(p.y), zf = floorf(p.z);
    Cell c;
    c.wx = p.x - xf;
    c.wy = p.y - yf;
    c.wz = p.z - zf;
    c.sx = safe_index(xf, v.nx) + kPad;
    c.sy = safe_index(yf, v.ny) + kPad;
    c.sz = safe_index(zf, v.nz) + kPad;
    return c;
}

__device__ __forceinline__ uint32_t brick_of(const Cell &c, const VolumeView &v)
{
    return (uint32_t)(((c.sz >> 3) * v.nby + (c.sy >> 3)) * v.nbx + (c.sx >> 3));
}

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

// texture(volumeTexture, tc).x with linear filtering and clamp-to-border (border .r = 0):
// VolumeRendering.comp:187-188, sampler main.cpp:365-382.  In two halves so that a lane can have
// the loads of several taps in flight before it filters the first: the eight corners of a tap as
// four pairs of x-neighbours, then the arithmetic.
struct TapWords {
    uint32_t w[4];   // pairs (x, x+1) at (y, z), (y+1, z), (y, z+1), (y+1, z+1)
};
template <typename T>
__device__ __forceinline__ TapWords fetch_volume(const Cell &c, uint32_t brick, const VolumeView &v)
{
    const T *base = (const T *)v.bricks + (size_t)brick * kBrickElems +
                    (((c.sz & 7) * kRow + (c.sy & 7)) * kRow + (c.sx & 7));
    TapWords r;
    r.w[0] = Voxel<T>::load_pair(base);
    r.w[1] = Voxel<T>::load_pair(base + kRow);
    r.w[2] = Voxel<T>::load_pair(base + kSlice);
    r.w[3] = Voxel<T>::load_pair(base + kSlice + kRow);
    return r;
}
template <typename T>
__device__ __forceinline__ float filter_volume(const TapWords &raw, const Cell &c, const VolumeView &v)
{
    float r000, r100, r010, r110, r001, r101, r011, r111;
    Voxel<T>::split(raw.w[0], r000, r100);
    Voxel<T>::split(raw.w[1], r010, r110);
    Voxel<T>::split(raw.w[2], r001, r101);
    Voxel<T>::split(raw.w[3], r011, r111);
    const float hi = v.inv_hi, lo = v.inv_lo;
    const float c000 = __builtin_fmaf(r000, hi, r000 * lo), c100 = __builtin_fmaf(r100, hi, r100 * lo);
    const float c010 = __builtin_fmaf(r010, hi, r010 * lo), c110 = __builtin_fmaf(r110, hi, r110 * lo);
    const float c001 = __builtin_fmaf(r001, hi, r001 * lo), c101 = __builtin_fmaf(r101, hi, r101 * lo);
    const float c011 = __builtin_fmaf(r011, hi, r011 * lo), c111 = __builtin_fmaf(r111, hi, r111 * lo);
    const float p0 = lerp_fma(c.wy, lerp_fma(c.wx, c000, c100), lerp_fma(c.wx, c010, c110));
    const float p1 = lerp_fma(c.wy, lerp_fma(c.wx, c001, c101), lerp_fma(c.wx, c011, c111));
    return lerp_fma(c.wz, p0, p1);
}
template <typename T>
__device__ __forceinline__ float sample_volume(const Cell &c, uint32_t brick, const VolumeView &v)
{
    return filter_volume<T>(fetch_volume<T>(c, brick, v), c, v);
}

// texture(transferSampler, intensity): VolumeRendering.comp:98-106.
__device__ __forceinline__ float4 sample_tf(const float4 *s_lut, float intensity)
{
    const float x = intensity * 256.0f - 0.5f;
    const float xf = floorf(x);
    const float w = x - xf;
    const int i = safe_index(xf, 256) + 2;
    const float4 a = s_lut[i], b = s_lut[i + 1];
    return make_float4(lerp_fma(w, a.x, b.x), lerp_fma(w, a.y, b.y), lerp_fma(w, a.z, b.z), lerp_fma(w, a.w, b.w));
}

// ---- local illumination -----------------------------------------------------------------------------
// The shader carries a gradient + Blinn-Phong term that its author has commented out at the call site
// (VolumeRendering.comp:195-198; computeNormal :109-119, shading :123-152; push constants `Scene`
// main.cpp:80-98,808).  shade_tap returns what those lines would add to all three channels of a tap's
// colour (light colour and ambient light are grey).  normalize() of a zero vector, undefined in GLSL
// (a flat neighbourhood has gradient 0 and the pixel would turn NaN), gives the zero vector here: such
// a tap receives the ambient term only.  The value is finite for any input, so a tap whose alpha is
// exactly 0 still blends as the identity and the marches may leave it out as before.
__device__ __forceinline__ void normalize3(float v[3])
{
    const float d = (v[0] * v[0] + v[1] * v[1]) + v[2] * v[2];
    if (!(d > 0.0f)) {
        v[0] = v[1] = v[2] = 0.0f;
        return;
    }
    const float inv = 1.0f / sqrtf(d);
    v[0] *= inv;
    v[1] *= inv;
    v[2] *= inv;
}

// Largest value shade_tap can return for these constants (|light - sample| <= 1.2248 inside the box).
__host__ __device__ inline float shade_bound(const float scene[4])
{
    return 0.3f * scene[0] + 1.25f * scene[1] + 1.01f * scene[2];
}

template <typename T, bool STATS>
__device__ __forceinline__ float shade_tap(const VolumeView &vol, const FrameParams &fp, const Ray &r, float step, int t,
                                           uint32_t *touched, unsigned long long &evaluated)
{
    float w[3];
    tap_world(r, step, t, w);
    // central differences one ray step to either side of the tap (:111-113): all six loads first
    Cell cells[6];
    uint32_t bricks[6];
    TapWords raw[6];
#pragma unroll
    for (int k = 0; k < 6; k++) {
        const float off = (k & 1) ? -step : step;
        const float x = (k >> 1) == 0 ? w[0] + off : w[0], y = (k >> 1) == 1 ? w[1] + off : w[1],
                    z = (k >> 1) == 2 ? w[2] + off : w[2];
        cells[k] = tap_cell(tap_of_world(x, y, z, vol), vol);
        bricks[k] = brick_of(cells[k], vol);
        raw[k] = fetch_volume<T>(cells[k], bricks[k], vol);
        if (STATS) atomicOr(&touched[bricks[k] >> 5], 1u << (bricks[k] & 31));
    }
    if (STATS) evaluated += 6;
    float n[3];
#pragma unroll
    for (int a = 0; a < 3; a++)
        n[a] = (filter_volume<T>(raw[2 * a], cells[2 * a], vol) - filter_volume<T>(raw[2 * a + 1], cells[2 * a + 1], vol)) /
               (2.0f * step);
    normalize3(n);
    const float v[3] = {fp.loc[0] - w[0], fp.loc[1] - w[1], fp.loc[2] - w[2]};
    const float l[3] = {0.0f - w[0], 0.0f - w[1], -1.0f - w[2]};     // lightLocation (0, 0, -1), :169
    float h[3] = {l[0] + v[0], l[1] + v[1], l[2] + v[2]};
    normalize3(h);
    const float ambient = fp.scene[0] * 0.3f;
    const float diffuse_light = glsl_max((l[0] * n[0] + l[1] * n[1]) + l[2] * n[2], 0.0f);
    const float diffuse = fp.scene[1] * diffuse_light;
    float specular_light = powf(glsl_max((h[0] * n[0] + h[1] * n[1]) + h[2] * n[2], 0.0f), fp.scene[3]);
    if (diffuse_light <= 0.0f) specular_light = 0.0f;
    const float specular = fp.scene[2] * specular_light;
    return (ambient + diffuse) + specular;
}

__device__ __forceinline__ uint32_t to_unorm8(float x)
{
    // NaN and negatives to 0, >= 1 to 255, else round-half-even of x * 255; written without branches
    // (fmaxf returns the other operand for a NaN)
    return (uint32_t)__float2int_rn(fminf(fmaxf(x, 0.0f), 1.0f) * 255.0f);
}

__device__ __forceinline__ uint32_t pack_rgba8(float r, float g, float b)
{
    return to_unorm8(r) | (to_unorm8(g) << 8) | (to_unorm8(b) << 16) | 0xff000000u;
}

__device__ __forceinline__ void stage_lut(float4 *s_lut, const float4 *lut, const float4 border)
{
    for (int i = threadIdx.x; i < kLutLds; i += blockDim.x) {
        float4 v = border;
        if (i >= 2 && i < 258) v = lut[i - 2];
        s_lut[i] = v;
    }
    __syncthreads();
}

// Pixel owned by a lane.  Every wave owns one 8x8 pixel tile (lane l the pixel (l & 7, l >> 3) of
// it); tiles are numbered row-major over the rectangle's 8-row tile rows, one wave after the other
// through the grid whatever the block size.
// Returns false for lanes outside the rectangle.  out_index is the pixel's position in the output
// buffer: rectangle rows, or with a sort-first split the rows of this part's bands, contiguous.
__device__ __forceinline__ bool lane_pixel_of_tile(const FrameParams &fp, int tile, int &px, int &py, uint32_t &out_index)
{
    const int lane = threadIdx.x & 63;
    const int rect_w = fp.x1 - fp.x0;
    const int tiles_x = (rect_w + 7) >> 3;
    const int tx = tile % tiles_x, trow = tile / tiles_x;     // trow: 8-row tile row local to this part
    const int local_row = trow * 8 + (lane >> 3);             // row inside this part's dense output
    px = fp.x0 + tx * 8 + (lane & 7);
    py = fp.y0 + ((local_row >> 4) * fp.parts + fp.part) * 16 + (local_row & 15);
    out_index = (uint32_t)(local_row * rect_w + (px - fp.x0));
    return px < fp.x1 && py < fp.y1;
}
__device__ __forceinline__ bool lane_pixel(const FrameParams &fp, int &px, int &py, uint32_t &out_index)
{
    return lane_pixel_of_tile(fp, (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)), px, py, out_index);
}

// Inverse of out_index -> pixel, for the replay queue.
__device__ __forceinline__ void pixel_of_index(const FrameParams &fp, uint32_t out_index, int &px, int &py)
{
    const uint32_t rect_w = (uint32_t)(fp.x1 - fp.x0);
    const uint32_t row = out_index / rect_w;
    px = fp.x0 + (int)(out_index - row * rect_w);
    py = fp.y0 + (int)(((row >> 4) * (uint32_t)fp.parts + (uint32_t)fp.part) * 16u + (row & 15u));
}

// Sort-first assembly: `src` holds the parts' dense band buffers one after the other, `stride` pixels
// apart; row y of the frame is row (y / 16 / parts) * 16 + y % 16 of part (y / 16) % parts.
__global__ __launch_bounds__(256) void k_assemble_bands(const uint32_t *__restrict__ src, size_t stride, int parts, int width,
                                                       int height, uint32_t *__restrict__ dst)
{
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= (size_t)width * (size_t)height) return;
    const int y = (int)(i / (size_t)width), x = (int)(i % (size_t)width);
    const int band = y >> 4;
    const size_t row = (size_t)(band / parts) * 16u + (size_t)(y & 15);
    dst[i] = src[(size_t)(band % parts) * stride + row * (size_t)width + (size_t)x];
}

// ---- transfer function table ---------------------------------------------------------------
// updateTransferFunction, Applications/VolumeRendering/Source/main.cpp:643-686, one wave per
// channel.  Segments are applied in order (a later segment overwrites the pixel it shares with
// the previous one, main.cpp:660-681); inside a segment the written pixels are distinct, so the
// lanes of a wave can take one k each.  Table entries no segment covers keep their old value.
struct CtrlPt {
    float x, intensity;
};

// Control graphs small enough travel as a kernel argument (no staging buffer, no copy, nothing for the host
// to wait for): the editor's graphs hold a handful of points each.
constexpr int kLutArgPoints = 256;
struct LutPointsArg {
    CtrlPt p[kLutArgPoints];
};

// ARG: the points are read from `arg`, else from `pts` in device memory.
template <bool ARG>
__global__ __launch_bounds__(256) void k_lut_update(float *__restrict__ lut, const CtrlPt *__restrict__ pts, LutPointsArg arg,
                                                   int off0, int off1, int off2, int off3, int off4, Counters *ctr)
{
    __shared__ float s_chan[4][256];
    const int lane = threadIdx.x & 63, ch = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int offs[5] = {off0, off1, off2, off3, off4};
    const int begin = offs[ch], cnt = offs[ch + 1] - offs[ch];
    for (int i = lane; i < 256; i += 64) s_chan[ch][i] = lut[i * 4 + ch];
    if (cnt >= 1) {
        auto point = [&](int k) -> CtrlPt { return ARG ? arg.p[begin + k] : pts[begin + k]; };
        const float x_last = point(cnt - 1).x;
        for (int j = 1; j < cnt; j++) {
            const CtrlPt ga = point(j - 1), gb = point(j);
            const float xa = (ga.x / x_last) * 255.0f;
            const float xb = (gb.x / x_last) * 255.0f;
            const float dir = gb.intensity - ga.intensity;
            const float total = xb - xa + 1.0f;
            uint32_t count = 0;
            if (total >= 1.0f) count = total >= 4294967296.0f ? 0xFFFFFFFFu : (uint32_t)total;
            if (count > 1024u) {  // cannot happen for sorted points; keeps the loop bounded
                if (lane == 0) atomicOr(&ctr->lut_error, 1u);
                count = 0;
            }
            for (uint32_t k0 = 0; k0 < count; k0 += 64) {
                const uint32_t k = k0 + lane;
                if (k < count) {
                    const float val = ga.intensity + dir * ((float)k / (total - 1.0f));
                    const float pixel = xa + (float)k;
                    if (pixel >= 0.0f && pixel < 256.0f) s_chan[ch][(int)pixel] = val;
                    else atomicOr(&ctr->lut_error, 1u);
                }
            }
        }
    }
    for (int i = lane; i < 256; i += 64) lut[i * 4 + ch] = s_chan[ch][i];
}

// ---- ingest --------------------------------------------------------------------------------
// One pass from the linear volume (x fastest, then y, then slice: main.cpp:41,131-146) to bricks: the
// byte swap of big-endian input (main.cpp:141-143), the re-tiling and the (min,max) macrocell of each
// brick; the data maximum loadVolumeData tracks while reading (main.cpp:144) is the largest macrocell
// maximum (k_cells_max), so the raw voxels are read once and never written.
//
// A workgroup takes a strip of kStrip bricks along x and walks it through `z_steps` brick layers.  LDS
// holds the strip's bricks in the layout they have in memory.  Per layer every lane loads 8 voxels of a
// raw row at a time (one 16-byte load where the raw rows are aligned for it) and scatters them to the one
// or two bricks that store them; then each brick leaves LDS as it stands, 16 bytes per lane, one wave per
// brick, 1472 (u16) contiguous bytes per brick -- the 9^3 voxels; the 32 elements of padding behind them are
// never read and never written -- with the macrocell folded from the same registers.  The last voxel slice of a layer is the first of the next (the apron):
// the lanes that loaded it keep it in registers and scatter it again, so only 8 slices are read per
// further layer.  Voxel rows shared with the strips next to this one in x and y are read by both; the
// grid is dealt so that such neighbours run on one XCD (workgroups b and b + 8 sit on the same XCD).
//
// The bricks built are the nbx x nby x nbz box of the volume's brick grid that starts at brick `b0`;
// `raw` holds the voxels [v0, v0 + (rx, ry, rz)) of the volume, which must cover every voxel of the
// volume those bricks store (what lies outside raw is outside the volume: zeros).
struct BrickSource {
    int b0[3];       // first brick of the box
    int v0[3];       // first voxel held by raw
    int rx, ry, rz;  // extent of raw in voxels
};
constexpr int kStrip = 16;                // bricks along x per workgroup (32: LDS for half as many workgroups per CU, 40 % slower)
constexpr int kIngestLayers = 4;          // brick layers a workgroup walks (1: every layer's first slice read twice; 16 and more:
                                          // neighbours drift apart and the rows they share leave the L2 in between)
constexpr int kStripGroups = kStrip + 1;  // aligned groups of 8 raw voxels that hold a strip's 8 * kStrip + 1 columns
// LDS pitch of a brick: 8 elements more than in memory, so that the lanes of a wave, which scatter the same
// row of neighbouring bricks, do not all hit one bank
constexpr int kLdsBrick = kBrickElems + 8;

// 8 voxels as loaded: kWords dwords.  Element i of them.
template <typename T>
__device__ __forceinline__ uint32_t voxel_of(const uint32_t *w, int i)
{
    return sizeof(T) == 2 ? (w[i >> 1] >> (16 * (i & 1))) & 0xffffu : (w[i >> 2] >> (8 * (i & 3))) & 0xffu;
}

// XO: column of the strip's first stored voxel inside its aligned group of 8 raw voxels -- 6 when raw starts
// at voxel 0 of the volume (bricks start two voxels before a multiple of 8), 0 when raw starts at a box's
// first stored voxel.  Element i of group g then belongs to brick g - 1 (cell 8 - XO + i) for i <= XO and to
// brick g (cell i - XO) for i >= XO; element XO is the apron voxel both store.
// VEC: the rows of raw start on multiples of 8 voxels in memory (row length a multiple of 8, buffer aligned),
// so a group of 8 lies wholly inside raw or wholly outside and is one load; without it the voxels are loaded
// one by one (odd-sized volumes: small ones).
template <typename T, bool SWAP, int XO, bool VEC>
__global__ __launch_bounds__(256, VEC ? 6 : 1) void k_ingest_bricks(const T *__restrict__ raw, T *__restrict__ bricks,
                                                                   uint32_t *__restrict__ cells, int nbx, int nby, int nbz,
                                                                   BrickSource src, int strips, int z_steps, uint32_t blocks_total)
{
    constexpr int kWords = (int)sizeof(T) * 2;   // dwords that hold 8 voxels
    constexpr int kVoxels = kRow * kRow * kRow;
    // 8-element pieces of a brick that are written: those that hold voxels (the last has one voxel and 7 zeros);
    // the 32 elements of padding behind them are never read by anything and stay as allocated
    constexpr int kPieces = (kVoxels + 7) / 8;
    __shared__ __align__(16) T s_brick[kStrip * kLdsBrick];
    const uint32_t per_xcd = gridDim.x >> 3;   // the grid is a multiple of 8 workgroups
    const uint32_t logical = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (logical >= blocks_total) return;
    const int strip = (int)(logical % (uint32_t)strips);
    const int by = (int)((logical / (uint32_t)strips) % (uint32_t)nby);
    const int zc = (int)(logical / (uint32_t)(strips * nby));
    const int bx0 = strip * kStrip;
    const int kvalid = nbx - bx0 < kStrip ? nbx - bx0 : kStrip;
    const int bz_begin = zc * z_steps, bz_end = bz_begin + z_steps < nbz ? bz_begin + z_steps : nbz;
    const int y0 = (by + src.b0[1]) * kBrick - kPad - src.v0[1];              // raw row of the strip's first stored row
    const int xs = (bx0 + src.b0[0]) * kBrick - kPad - src.v0[0] - XO;        // raw column of the strip's first group (a multiple of 8)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;

    // the 39 elements of padding behind the 9^3 voxels of every brick are zero and stay so
    for (int i = threadIdx.x; i < kStrip * (kBrickElems - kVoxels); i += 256)
        s_brick[(i / (kBrickElems - kVoxels)) * kLdsBrick + kVoxels + i % (kBrickElems - kVoxels)] = (T)0;

    // 8 voxels of raw row (iy, iz) from column x on, zeros outside raw
    auto load_group = [&](int iy, int iz, int x, uint32_t *w) {
#pragma unroll
        for (int k = 0; k < kWords; k++) w[k] = 0u;
        if ((unsigned)iy < (unsigned)src.ry && (unsigned)iz < (unsigned)src.rz) {
            const T *row = raw + ((size_t)iz * (size_t)src.ry + (size_t)iy) * (size_t)src.rx;
            if (VEC) {
                if (x >= 0 && x + 8 <= src.rx) __builtin_memcpy(w, __builtin_assume_aligned(row + x, 8 * sizeof(T)), 8 * sizeof(T));
            } else {
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int xi = x + i;
                    const uint32_t v = (unsigned)xi < (unsigned)src.rx ? (uint32_t)row[xi] : 0u;
                    w[i * (int)sizeof(T) / 4] |= v << (8 * ((i * (int)sizeof(T)) & 3));
                }
            }
            if (SWAP) {
#pragma unroll
                for (int k = 0; k < kWords; k++) w[k] = ((w[k] & 0x00ff00ffu) << 8) | ((w[k] >> 8) & 0x00ff00ffu);
            }
        }
    };
    // the 8 voxels of group g into row r (= lz * 9 + ly) of the bricks that store them
    auto scatter_group = [&](const uint32_t *w, int r, int g) {
        T *to_prev = s_brick + (g - 1) * kLdsBrick + r * kRow + (kBrick - XO);   // element i -> cell 8 - XO + i of brick g - 1
        T *to_own = s_brick + g * kLdsBrick + r * kRow - XO;                       // element i -> cell i - XO of brick g
        if (g >= 1) {
#pragma unroll
            for (int i = 0; i <= XO; i++) to_prev[i] = (T)voxel_of<T>(w, i);
        }
        if (g < kvalid) {
#pragma unroll
            for (int i = XO; i < 8; i++) to_own[i] = (T)voxel_of<T>(w, i);
        }
    };

    // the rows of slice 8 come first in a layer's list of loads: 9 rows x kStripGroups groups, one per thread,
    // kept for the next layer, where they are slice 0
    const bool keeper = threadIdx.x < kRow * kStripGroups;
    const int keep_ly = threadIdx.x / kStripGroups, keep_g = threadIdx.x % kStripGroups;
    uint32_t keep[kWords];
    for (int bz = bz_begin, step = 0; bz < bz_end; bz++, step++) {
        const int z0 = (bz + src.b0[2]) * kBrick - kPad - src.v0[2];
        constexpr int kItems = (kBrick - 1) * kRow * kStripGroups;   // rows of slices 1..7
        if (keeper && keep_g <= kvalid) {
            if (step == 0) load_group(y0 + keep_ly, z0, xs + keep_g * 8, keep);
            scatter_group(keep, keep_ly, keep_g);
            load_group(y0 + keep_ly, z0 + kBrick, xs + keep_g * 8, keep);
            scatter_group(keep, kBrick * kRow + keep_ly, keep_g);
        }
        for (int idx = threadIdx.x; idx < kItems; idx += 256) {
            const int rr = idx / kStripGroups, g = idx - rr * kStripGroups;
            if (g > kvalid) continue;   // a short last strip
            uint32_t w[kWords];
            load_group(y0 + rr % kRow, z0 + 1 + rr / kRow, xs + g * 8, w);
            scatter_group(w, (1 + rr / kRow) * kRow + rr % kRow, g);
        }
        __syncthreads();
        for (int k = wave; k < kvalid; k += 4) {
            const uint32_t brick = ((uint32_t)bz * (uint32_t)nby + (uint32_t)by) * (uint32_t)nbx + (uint32_t)(bx0 + k);
            T *out = bricks + (size_t)brick * kBrickElems;
            uint32_t mn = 0xffffffffu, mx = 0u;
            typedef unsigned short pair_t __attribute__((ext_vector_type(2)));
            pair_t pair_mn = {0xffff, 0xffff}, pair_mx = {0, 0};   // u16: two voxels per dword folded at once
#pragma unroll
            for (int pass = 0; pass < 2; pass++) {
                const int j = pass * 64 + lane;      // 8-element piece of the brick's 768
                if (j < kPieces) {
                    uint32_t w[kWords];
                    __builtin_memcpy(w, __builtin_assume_aligned(s_brick + k * kLdsBrick + j * 8, 8 * sizeof(T)), 8 * sizeof(T));
                    // streamed past the L2, which then keeps the raw rows the neighbouring strips read again
                    typedef uint32_t vec_t __attribute__((ext_vector_type(kWords)));
                    vec_t v;
#pragma unroll
                    for (int q = 0; q < kWords; q++) v[q] = w[q];
                    __builtin_nontemporal_store(v, reinterpret_cast<vec_t *>(out + j * 8));
                    const int valid = kVoxels - j * 8;   // elements of the piece that are voxels, not padding
                    if (sizeof(T) == 2 && valid >= 8) {
#pragma unroll
                        for (int q = 0; q < kWords; q++) {
                            pair_t v;
                            __builtin_memcpy(&v, &w[q], 4);
                            pair_mn = __builtin_elementwise_min(pair_mn, v);
                            pair_mx = __builtin_elementwise_max(pair_mx, v);
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < 8; i++) {
                            const uint32_t v = voxel_of<T>(w, i);
                            if (i < valid) {
                                mn = v < mn ? v : mn;
                                mx = v > mx ? v : mx;
                            }
                        }
                    }
                }
            }
            if (sizeof(T) == 2) {
                const uint32_t a = pair_mn[0] < pair_mn[1] ? pair_mn[0] : pair_mn[1], b = pair_mx[0] > pair_mx[1] ? pair_mx[0] : pair_mx[1];
                mn = a < mn ? a : mn;
                mx = b > mx ? b : mx;
            }
            // one reduction for both: the halves of (max, 65535 - min) are folded by maxima
            uint32_t both = (mx << 16) | (0xffffu - (mn & 0xffffu));
            for (int o = 32; o > 0; o >>= 1) {
                const uint32_t other = __shfl_xor(both, o);
                const uint32_t hi = (other >> 16) > (both >> 16) ? (other >> 16) : (both >> 16);
                const uint32_t lo = (other & 0xffffu) > (both & 0xffffu) ? (other & 0xffffu) : (both & 0xffffu);
                both = (hi << 16) | lo;
            }
            if (lane == 0) cells[brick] = (0xffffu - (both & 0xffffu)) | (both & 0xffff0000u);
        }
        __syncthreads();
    }
}

// Largest macrocell maximum = the data maximum (every voxel is stored by some brick).
__global__ __launch_bounds__(256) void k_cells_max(const uint32_t *__restrict__ cells, uint32_t count, uint32_t *result)
{
    uint32_t mx = 0;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < count; i += gridDim.x * 256u) {
        const uint32_t v = cells[i] >> 16;
        mx = v > mx ? v : mx;
    }
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t other = __shfl_xor(mx, o);
        mx = other > mx ? other : mx;
    }
    if ((threadIdx.x & 63) == 0 && mx > 0u) atomicMax(result, mx);
}

// float(v) / float(max) for every voxel, back in linear order: what the reference keeps in
// `intensities` after main.cpp:161-164.  Uses the same two-term reciprocal as the sampler.
template <typename T>
__global__ __launch_bounds__(256) void k_read_normalized(VolumeView vol, float *__restrict__ out)
{
    const size_t count = (size_t)vol.nx * vol.ny * vol.nz;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        const int ix = (int)(i % vol.nx), iy = (int)((i / vol.nx) % vol.ny), iz = (int)(i / ((size_t)vol.nx * vol.ny));
        const int sx = ix + kPad, sy = iy + kPad, sz = iz + kPad;
        const size_t brick = ((size_t)(sz >> 3) * vol.nby + (sy >> 3)) * vol.nbx + (sx >> 3);
        const T raw = ((const T *)vol.bricks)[brick * kBrickElems + (((sz & 7) * kRow + (sy & 7)) * kRow + (sx & 7))];
        const float f = (float)raw;
        out[i] = __builtin_fmaf(f, vol.inv_hi, f * vol.inv_lo);
    }
}

// ---- intensity histogram ---------------------------------------------------------------------
// The reference's 100 bins (prepareHistogramVolumeData, main.cpp:166-181) are ranges of the sorted normalised
// values; float(v)/float(max) is monotone in v, so every bin is a range of raw values and the host hands the
// kernel a table value -> bin (kHistBins for the values between two bins).  The kernel streams the bricks'
// 8-element pieces that hold cells (the first 80 of a brick's 96; apron cells, which repeat a neighbour's voxels,
// and cells outside the volume are masked out per piece from three small tables), looks every voxel's bin up in
// LDS and counts it in one of kHistCopies LDS histograms (lanes that share a bin mostly do not share a counter;
// the odd pitch spreads the copies over the banks).  HBM-bound: 1280 B (u16) read per brick.
constexpr int kHistBins = 100;               // main.cpp:109
constexpr int kHistCopies = 32;
constexpr int kHistPitch = kHistBins + 1;    // entry kHistBins collects what no bin takes
constexpr int kHistPieces = 80;              // element 637 = cell (7, 7, 7) is the last that is not apron
constexpr int kHistThreads = 1024;

template <typename T>
__global__ __launch_bounds__(kHistThreads) void k_bin_histogram(VolumeView vol, const uint8_t *__restrict__ bin_of,
                                                                unsigned long long *__restrict__ out)
{
    constexpr int kValues = sizeof(T) == 2 ? 65536 : 256;
    constexpr int kWords = (int)sizeof(T) * 2;   // dwords that hold 8 voxels
    __shared__ __align__(16) uint8_t s_bin[kValues];
    __shared__ uint32_t s_hist[kHistCopies * kHistPitch];
    // per piece, the elements that are cells holding voxels of the volume, by where the brick sits on an axis:
    // 0 inside, 1 first brick, 2 last brick, 3 both.  s_yz: y and z together (index y * 4 + z), s_x: x.
    __shared__ uint8_t s_yz[16][kHistPieces], s_x[4][kHistPieces];
    const int tid = threadIdx.x;
    for (int i = tid * 16; i < kValues; i += kHistThreads * 16)
        *reinterpret_cast<uint4 *>(s_bin + i) = *reinterpret_cast<const uint4 *>(bin_of + i);
    for (int i = tid; i < kHistCopies * kHistPitch; i += kHistThreads) s_hist[i] = 0u;
    // all eight cells per axis, but from kPad on in the first brick and up to the volume's end in the last
    const int end_x = min(kBrick, vol.nx + kPad - kBrick * (vol.nbx - 1));
    const int end_y = min(kBrick, vol.ny + kPad - kBrick * (vol.nby - 1));
    const int end_z = min(kBrick, vol.nz + kPad - kBrick * (vol.nbz - 1));
    for (int i = tid; i < 20 * kHistPieces; i += kHistThreads) {
        const int variant = i / kHistPieces, p = i - variant * kHistPieces;
        uint32_t m = 0;
        for (int j = 0; j < 8; j++) {
            const int e = p * 8 + j, lz = e / (kRow * kRow), ly = (e / kRow) % kRow, lx = e % kRow;
            bool ok;
            if (variant < 16) {
                const int vy = variant >> 2, vz = variant & 3;
                ok = lx < kBrick && ly >= ((vy & 1) ? kPad : 0) && ly < ((vy & 2) ? end_y : kBrick) &&
                     lz >= ((vz & 1) ? kPad : 0) && lz < ((vz & 2) ? end_z : kBrick);
            } else {
                const int vx = variant - 16;
                ok = lx >= ((vx & 1) ? kPad : 0) && lx < ((vx & 2) ? end_x : kBrick);
            }
            m |= (ok ? 1u : 0u) << j;
        }
        if (variant < 16) s_yz[variant][p] = (uint8_t)m;
        else s_x[variant - 16][p] = (uint8_t)m;
    }
    __syncthreads();

    uint32_t *mine = s_hist + (tid & (kHistCopies - 1)) * kHistPitch;
    const uint32_t rows = (uint32_t)vol.nby * (uint32_t)vol.nbz;
    const int pieces = vol.nbx * kHistPieces;                                   // of a row of bricks along x
    const int turns = (pieces + 2 * kHistThreads - 1) / (2 * kHistThreads);     // two pieces per lane and turn
    // The workgroup walks rows blockIdx.x, blockIdx.x + gridDim.x, ...; the loads of the turn after this one are
    // issued in front of this turn's counting, across the end of a row too.
    uint32_t next_row = blockIdx.x;
    int next_turn = 0;
    auto fetch = [&](uint32_t (*w)[kWords], uint32_t *m) {
        m[0] = m[1] = 0u;
        if (next_row >= rows) return;
        const int by = (int)(next_row % (uint32_t)vol.nby), bz = (int)(next_row / (uint32_t)vol.nby);
        const uint8_t *yz = s_yz[((by == 0 ? 1 : 0) | (by == vol.nby - 1 ? 2 : 0)) * 4 + ((bz == 0 ? 1 : 0) | (bz == vol.nbz - 1 ? 2 : 0))];
        const T *bricks = (const T *)vol.bricks + (size_t)next_row * (size_t)vol.nbx * kBrickElems;
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int k = (next_turn * 2 + u) * kHistThreads + tid;
            if (k < pieces) {
                const int bx = k / kHistPieces, p = k - bx * kHistPieces;
                m[u] = yz[p] & s_x[(bx == 0 ? 1 : 0) | (bx == vol.nbx - 1 ? 2 : 0)][p];
                if (m[u]) __builtin_memcpy(w[u], __builtin_assume_aligned(bricks + (size_t)bx * kBrickElems + p * 8, 8 * sizeof(T)), 8 * sizeof(T));
            }
        }
        if (++next_turn == turns) {
            next_turn = 0;
            next_row += gridDim.x;
        }
    };
    uint32_t w[2][kWords], m[2];
    fetch(w, m);
    for (uint32_t row = blockIdx.x; row < rows; row += gridDim.x) {
        for (int turn = 0; turn < turns; turn++) {
            uint32_t w_next[2][kWords], m_next[2];
            fetch(w_next, m_next);
#pragma unroll
            for (int u = 0; u < 2; u++) {
                if (m[u] != 0u) {
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const uint32_t bin = s_bin[voxel_of<T>(w[u], j)];
                        atomicAdd(&mine[(m[u] >> j) & 1u ? bin : (uint32_t)kHistBins], 1u);
                    }
                }
                m[u] = m_next[u];
#pragma unroll
                for (int q = 0; q < kWords; q++) w[u][q] = w_next[u][q];
            }
        }
    }
    __syncthreads();
    if (tid < kHistBins) {
        unsigned long long total = 0;
        for (int c = 0; c < kHistCopies; c++) total += s_hist[c * kHistPitch + tid];
        if (total) atomicAdd(&out[tid], total);
    }
}

// ---- synthetic phantom ------------------------------------------------------------------------
// 3-D Shepp-Logan head (ten ellipsoids, contrast in tenths so the sum is an exact integer),
// voxel centres of [-1,1]^3, small hash noise inside the outer ellipsoid, exactly 0 outside it.
struct ellipsoid {
    int tenths;
    float ia2, ib2, ic2, x0, y0, z0, cp, sp;
};
#include "phantom_table.inc"
__constant__ ellipsoid d_phantom[10];

__device__ __forceinline__ uint32_t phantom_hash(uint32_t seed, uint64_t idx)
{
    uint32_t x = (uint32_t)idx ^ ((uint32_t)(idx >> 32) * 0x85EBCA6Bu) ^ seed;
    x *= 0x9E3779B1u;
    x ^= x << 13;
    x ^= x >> 17;
    x ^= x << 5;
    return x >> 8;
}

// Writes the voxels [lo, lo + ext) of the n^3 phantom, x fastest.
struct PhantomBox {
    uint32_t lo[3], ext[3];
};
template <typename T>
__global__ __launch_bounds__(256) void k_phantom(T *__restrict__ out, uint32_t n, uint32_t dmax, uint32_t seed, PhantomBox box)
{
    const uint64_t count = (uint64_t)box.ext[0] * box.ext[1] * box.ext[2];
    const float fn = (float)n;
    const uint32_t range = (dmax + 1) / 128;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < count; j += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t ix = box.lo[0] + (uint32_t)(j % box.ext[0]), iy = box.lo[1] + (uint32_t)((j / box.ext[0]) % box.ext[1]),
                       iz = box.lo[2] + (uint32_t)(j / ((uint64_t)box.ext[0] * box.ext[1]));
        const uint64_t i = ((uint64_t)iz * n + iy) * n + ix;   // index in the whole phantom: seeds the noise
        const float x = (float)(2 * ix + 1) / fn - 1.0f;
        const float y = (float)(2 * iy + 1) / fn - 1.0f;
        const float z = (float)(2 * iz + 1) / fn - 1.0f;
        int tenths = 0;
        bool inside_outer = false;
#pragma unroll
        for (int e = 0; e < 10; e++) {
            const ellipsoid E = d_phantom[e];
            const float dx = x - E.x0, dy = y - E.y0, dz = z - E.z0;
            const float xr = dx * E.cp + dy * E.sp;
            const float yr = dy * E.cp - dx * E.sp;
            const float q = ((xr * xr) * E.ia2 + (yr * yr) * E.ib2) + (dz * dz) * E.ic2;
            if (q <= 1.0f) {
                tenths += E.tenths;
                if (e == 0) inside_outer = true;
            }
        }
        uint32_t v = 0;
        if (inside_outer) {
            tenths = tenths < 0 ? 0 : (tenths > 10 ? 10 : tenths);
            v = (uint32_t)tenths * (dmax - (range - 1)) / 10 + phantom_hash(seed, i) % range;
        }
        out[j] = (T)v;
    }
}

}  // namespace vr
