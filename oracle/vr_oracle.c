/*
 * CPU oracle for the VolumeRendering ray-march path — TEST INFRASTRUCTURE ONLY.
 * See vr_oracle.h for the rules of use and the "parity unpinned" statement.
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (oracle/Makefile).  No expression
 * here may be contracted into an FMA by the compiler; the places where a fused multiply-add
 * is part of the restated behaviour call fmaf() explicitly.
 */
#define _GNU_SOURCE
#include "vr_oracle.h"

#include <dirent.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int vro_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------------------------
 * Transfer function LUT — main.cpp:643-686
 * ---------------------------------------------------------------------------------------- */

int vro_lut_update(float *lut, const vro_ctrl_pt *const pts[4], const int n[4])
{
    int status = 0;
    for (int i = 0; i < 4; i++) {                       /* main.cpp:647, channel i of RGBA */
        const int cnt = n[i];
        if (cnt < 1) continue;
        const vro_ctrl_pt *g = pts[i];
        const float x_last = g[cnt - 1].x;              /* controlGraph.back().position.x */
        for (int j = 1; j < cnt; j++) {                 /* main.cpp:660-662 */
            /* main.cpp:653: (x / x_last) * (size() - 1); size_t 255 converts to 255.0f */
            const float xa = (g[j - 1].x / x_last) * 255.0f;
            const float xb = (g[j].x / x_last) * 255.0f;
            const float dir = g[j].intensity - g[j - 1].intensity;      /* main.cpp:663 */
            const float total = xb - xa + 1;                             /* main.cpp:664 */
            /* main.cpp:665: iota(0u, uint32_t(total)).  A negative or NaN total is undefined in
             * the reference; here it runs zero iterations. */
            uint32_t count = 0;
            if (total >= 1.0f) count = total >= 4294967296.0f ? 0xFFFFFFFFu : (uint32_t)total;
            for (uint32_t k = 0; k < count; k++) {
                const float v = g[j - 1].intensity + dir * ((float)k / (total - 1.0f)); /* :667 */
                const float pixel = xa + (float)k;                                       /* :668 */
                if (!(pixel >= 0.0f && pixel < 256.0f)) { status = -1; break; }
                lut[(size_t)pixel * 4 + i] = v;          /* :669-680, other channels preserved */
            }
        }
    }
    return status;
}

float vro_control_point_intensity(float y, float child_h)
{
    return 1.0f - (y / (child_h - 1));                  /* main.cpp:952 */
}

/* ------------------------------------------------------------------------------------------
 * glm restated (glm is a vcpkg dependency of the reference, version not pinned, absent here).
 * Column-major: m[col][row].  Association orders follow glm's scalar (non-SIMD) code paths.
 * ---------------------------------------------------------------------------------------- */

typedef struct { float c[4][4]; } mat4;
typedef struct { float v[4]; } vec4;

static mat4 m_identity(void)
{
    mat4 m;
    memset(&m, 0, sizeof m);
    m.c[0][0] = m.c[1][1] = m.c[2][2] = m.c[3][3] = 1.0f;
    return m;
}

/* glm operator*(mat4, mat4): Result[j] = ((A0*B[j][0] + A1*B[j][1]) + A2*B[j][2]) + A3*B[j][3] */
static mat4 m_mul(const mat4 *a, const mat4 *b)
{
    mat4 r;
    for (int j = 0; j < 4; j++)
        for (int row = 0; row < 4; row++) {
            float t = a->c[0][row] * b->c[j][0];
            t = t + a->c[1][row] * b->c[j][1];
            t = t + a->c[2][row] * b->c[j][2];
            t = t + a->c[3][row] * b->c[j][3];
            r.c[j][row] = t;
        }
    return r;
}

/* glm operator*(mat4, vec4): (m0*v0 + m1*v1) + (m2*v2 + m3*v3) */
static vec4 m_mulv(const mat4 *m, const vec4 *v)
{
    vec4 r;
    for (int row = 0; row < 4; row++) {
        const float add0 = m->c[0][row] * v->v[0] + m->c[1][row] * v->v[1];
        const float add1 = m->c[2][row] * v->v[2] + m->c[3][row] * v->v[3];
        r.v[row] = add0 + add1;
    }
    return r;
}

/* glm::translate(v) = translate(mat4(1), v): Result[3] = m0*v0 + m1*v1 + m2*v2 + m3 */
static mat4 m_translate(float x, float y, float z)
{
    mat4 m = m_identity();
    for (int row = 0; row < 4; row++)
        m.c[3][row] = m.c[0][row] * x + m.c[1][row] * y + m.c[2][row] * z + m.c[3][row];
    return m;
}

/* glm::rotate(angle, axis) = rotate(mat4(1), angle, axis) (Rodrigues form of glm) */
static mat4 m_rotate(float angle, float ax, float ay, float az)
{
    const float c = cosf(angle);
    const float s = sinf(angle);
    const float inv = 1.0f / sqrtf(ax * ax + ay * ay + az * az);       /* normalize(v) */
    const float axis[3] = { ax * inv, ay * inv, az * inv };
    const float temp[3] = { (1.0f - c) * axis[0], (1.0f - c) * axis[1], (1.0f - c) * axis[2] };
    float R[3][3];
    R[0][0] = c + temp[0] * axis[0];
    R[0][1] = temp[0] * axis[1] + s * axis[2];
    R[0][2] = temp[0] * axis[2] - s * axis[1];
    R[1][0] = temp[1] * axis[0] - s * axis[2];
    R[1][1] = c + temp[1] * axis[1];
    R[1][2] = temp[1] * axis[2] + s * axis[0];
    R[2][0] = temp[2] * axis[0] + s * axis[1];
    R[2][1] = temp[2] * axis[1] - s * axis[0];
    R[2][2] = c + temp[2] * axis[2];
    const mat4 m = m_identity();
    mat4 r;
    for (int j = 0; j < 3; j++)
        for (int row = 0; row < 4; row++)
            r.c[j][row] = m.c[0][row] * R[j][0] + m.c[1][row] * R[j][1] + m.c[2][row] * R[j][2];
    for (int row = 0; row < 4; row++) r.c[3][row] = m.c[3][row];
    return r;
}

/* ------------------------------------------------------------------------------------------
 * Camera — main.cpp:59-77, 182-202, 737-799
 * ---------------------------------------------------------------------------------------- */

void vro_camera(float horizontal_rad, float vertical_rad, uint32_t width, uint32_t height,
                float view_pyramid[20], float to_world[16])
{
    /* main.cpp:70-76 defaults, then setupViewPyramid main.cpp:187-201 */
    mat4 basis = m_identity();
    basis.c[0][0] = 1; basis.c[0][1] = 0; basis.c[0][2] = 0; basis.c[0][3] = 0;   /* side    */
    basis.c[1][0] = 0; basis.c[1][1] = 1; basis.c[1][2] = 0; basis.c[1][3] = 0;   /* up      */
    basis.c[2][0] = 0; basis.c[2][1] = 0; basis.c[2][2] = -1; basis.c[2][3] = 0;  /* forward */
    basis.c[3][0] = 0; basis.c[3][1] = 0; basis.c[3][2] = 0; basis.c[3][3] = 1;
    {
        const mat4 t = m_translate(0.0f, -5.0f, -0.5f);                /* main.cpp:194 */
        const float rad90 = 90.0f * 0.01745329251994329576923690768489f; /* glm::radians */
        const mat4 r = m_rotate(rad90, 1.0f, 0.0f, 0.0f);              /* main.cpp:195 */
        const mat4 tr = m_mul(&t, &r);
        basis = m_mul(&tr, &basis);                                     /* main.cpp:193-196 */
    }
    const float pyramid_height = 4.5f;                                  /* main.cpp:197 */
    const vec4 location0 = { { basis.c[3][0], basis.c[3][1], basis.c[3][2], basis.c[3][3] } };
    const float centre[4] = { 0.0f, 0.0f, -0.5f, 0.0f };               /* main.cpp:68 */

    /* updatePixelTransformationUniforms, main.cpp:742-757 */
    mat4 b = basis;
    b.c[3][0] = 0; b.c[3][1] = 0; b.c[3][2] = 0; b.c[3][3] = 1;        /* main.cpp:746 */
    const mat4 vrot = m_rotate(vertical_rad, 1.0f, 0.0f, 0.0f);        /* main.cpp:752 */
    const mat4 hrot = m_rotate(horizontal_rad, 0.0f, 0.0f, 1.0f);      /* main.cpp:753 */
    const mat4 hv = m_mul(&hrot, &vrot);
    b = m_mul(&hv, &b);                                                 /* main.cpp:755 */
    const vec4 to_cam = { { location0.v[0] - centre[0], location0.v[1] - centre[1],
                            location0.v[2] - centre[2], 1.0f } };
    const vec4 rotated = m_mulv(&hv, &to_cam);                          /* main.cpp:756 */
    b.c[3][0] = centre[0] + rotated.v[0];                               /* main.cpp:757 */
    b.c[3][1] = centre[1] + rotated.v[1];
    b.c[3][2] = centre[2] + rotated.v[2];
    b.c[3][3] = 1.0f;

    /* main.cpp:759-765: {height, location, side, up, forward} */
    view_pyramid[0] = pyramid_height; view_pyramid[1] = 0; view_pyramid[2] = 0; view_pyramid[3] = 0;
    for (int r = 0; r < 4; r++) {
        view_pyramid[4 + r] = b.c[3][r];
        view_pyramid[8 + r] = b.c[0][r];
        view_pyramid[12 + r] = b.c[1][r];
        view_pyramid[16 + r] = b.c[2][r];
    }

    /* main.cpp:771-794 */
    mat4 normalize_components = m_identity();
    normalize_components.c[0][0] = 1.0f / ((float)width - 1.0f);
    normalize_components.c[1][1] = 1.0f / ((float)height - 1.0f);
    mat4 viewport_to_world = m_identity();
    viewport_to_world.c[1][1] = -1.0f;
    const mat4 pre = m_translate(-0.5f, 0.5f, 0.0f);                    /* main.cpp:783 */
    mat4 rot = b;
    rot.c[3][0] = 0; rot.c[3][1] = 0; rot.c[3][2] = 0; rot.c[3][3] = 1; /* main.cpp:784-789 */
    const mat4 post = m_translate(b.c[3][0] + pyramid_height * b.c[2][0],
                                  b.c[3][1] + pyramid_height * b.c[2][1],
                                  b.c[3][2] + pyramid_height * b.c[2][2]); /* main.cpp:790 */
    mat4 m = m_mul(&post, &rot);                                        /* main.cpp:794, left to right */
    m = m_mul(&m, &pre);
    m = m_mul(&m, &viewport_to_world);
    m = m_mul(&m, &normalize_components);
    for (int j = 0; j < 4; j++)
        for (int r = 0; r < 4; r++) to_world[j * 4 + r] = m.c[j][r];
}

/* ------------------------------------------------------------------------------------------
 * Volume ingest — main.cpp:112-165
 * ---------------------------------------------------------------------------------------- */

float vro_normalize_u16(const uint16_t *v, uint64_t count, float max_value, float *out)
{
    float mx = max_value;
    if (!(mx > 0.0f)) {
        mx = 0.0f;
        for (uint64_t i = 0; i < count; i++)
            if (mx < (float)v[i]) mx = (float)v[i];                    /* main.cpp:144 */
    }
    for (uint64_t i = 0; i < count; i++) out[i] = (float)v[i] / mx;    /* main.cpp:161-164 */
    return mx;
}

float vro_normalize_u8(const uint8_t *v, uint64_t count, float max_value, float *out)
{
    float mx = max_value;
    if (!(mx > 0.0f)) {
        mx = 0.0f;
        for (uint64_t i = 0; i < count; i++)
            if (mx < (float)v[i]) mx = (float)v[i];
    }
    for (uint64_t i = 0; i < count; i++) out[i] = (float)v[i] / mx;
    return mx;
}

typedef struct { char *path; int key; } slice_file;

static int slice_cmp(const void *a, const void *b)
{
    const int ka = ((const slice_file *)a)->key, kb = ((const slice_file *)b)->key;
    return (ka > kb) - (ka < kb);
}

int vro_load_slices(const char *dir, float **out, uint32_t *width, uint32_t *height,
                    uint32_t *slices, float *max_value)
{
    DIR *d = opendir(dir);
    if (!d) return -1;
    slice_file *files = NULL;
    size_t nfiles = 0, cap = 0;
    struct dirent *e;
    while ((e = readdir(d)) != NULL) {
        if (strcmp(e->d_name, ".") == 0 || strcmp(e->d_name, "..") == 0) continue;
        /* main.cpp:123-128: integer after the last dot, or the whole file name */
        const char *dot = strrchr(e->d_name, '.');
        const char *num = (dot && dot != e->d_name) ? dot + 1 : e->d_name;
        if (nfiles == cap) {
            cap = cap ? cap * 2 : 64;
            files = (slice_file *)realloc(files, cap * sizeof *files);
        }
        size_t len = strlen(dir) + strlen(e->d_name) + 2;
        files[nfiles].path = (char *)malloc(len);
        snprintf(files[nfiles].path, len, "%s/%s", dir, e->d_name);
        files[nfiles].key = atoi(num);
        nfiles++;
    }
    closedir(d);
    if (nfiles == 0) { free(files); return -1; }                       /* main.cpp:158 */
    qsort(files, nfiles, sizeof *files, slice_cmp);                    /* main.cpp:118-129 */

    uint16_t *raw = NULL;
    uint64_t count = 0, rcap = 0;
    uint32_t side = 0, nslices = 0;
    float mx = 0.0f;
    int status = 0;
    for (size_t f = 0; f < nfiles && status == 0; f++) {
        FILE *fp = fopen(files[f].path, "rb");
        if (!fp) { status = -1; break; }                               /* main.cpp:134 */
        unsigned char b[2];
        while (fread(b, 1, 2, fp) == 2) {                              /* main.cpp:138 */
            const uint16_t v = (uint16_t)((b[0] << 8) | b[1]);         /* big-endian, :141-143 */
            if (mx < (float)v) mx = (float)v;                          /* main.cpp:144 */
            if (count == rcap) {
                rcap = rcap ? rcap * 2 : (1u << 20);
                raw = (uint16_t *)realloc(raw, rcap * sizeof *raw);
            }
            raw[count++] = v;
        }
        fclose(fp);
        nslices++;
        if (nslices == 1) side = (uint32_t)sqrt((double)count);        /* main.cpp:150-155 */
    }
    for (size_t f = 0; f < nfiles; f++) free(files[f].path);
    free(files);
    if (status != 0) { free(raw); return status; }
    float *vol = (float *)malloc(count * sizeof(float));
    for (uint64_t i = 0; i < count; i++) vol[i] = (float)raw[i] / mx;  /* main.cpp:161-164 */
    free(raw);
    *out = vol; *width = side; *height = side; *slices = nslices; *max_value = mx;
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Histogram — main.cpp:166-181
 * ---------------------------------------------------------------------------------------- */

static int float_cmp(const void *a, const void *b)
{
    const float fa = *(const float *)a, fb = *(const float *)b;
    return (fa > fb) - (fa < fb);
}

void vro_histogram(const float *normalized, uint64_t count, float hist[100])
{
    float *sorted = (float *)malloc(count * sizeof(float));
    memcpy(sorted, normalized, count * sizeof(float));
    qsort(sorted, count, sizeof(float), float_cmp);                    /* main.cpp:171 */
    const float step = 1.0f / 100;                                      /* main.cpp:170 */
    size_t count_max = 0;
    for (int i = 0; i < 100; i++) {
        /* main.cpp:174: lower_bound(stepSize * i): float * int -> float, compared as float */
        const float lo = step * (float)i;
        /* main.cpp:175: upper_bound(stepSize * (i + 0.9)): float * double -> double */
        const double hi = (double)step * ((double)i + 0.9);
        uint64_t a = 0, b = count;
        while (a < b) { uint64_t m = (a + b) / 2; if (sorted[m] < lo) a = m + 1; else b = m; }
        const uint64_t lower = a;
        a = 0; b = count;
        while (a < b) { uint64_t m = (a + b) / 2; if (!(hi < (double)sorted[m])) a = m + 1; else b = m; }
        const uint64_t upper = a;
        /* main.cpp:176: distance may be negative when upper < lower; static_cast<float> */
        const float c = (float)((int64_t)upper - (int64_t)lower);
        hist[i] = c;
        if (c > (float)count_max) count_max = (size_t)c;               /* main.cpp:178 */
    }
    for (int i = 0; i < 100; i++) hist[i] /= (float)count_max;         /* main.cpp:180 */
    free(sorted);
}

/* ------------------------------------------------------------------------------------------
 * The shader — VolumeRendering.comp
 *
 * Choices where the result depends on the Vulkan implementation (restated from the behaviour
 * of Mesa's software rasteriser, the CPU baseline the reference is meant to run under):
 *   - shader-level float expressions are evaluated as written, one rounding per operation,
 *     no fused multiply-add (that driver lowers ffma to mul+add);
 *   - mat4 * vec4 accumulates from the last column: m0*x + (m1*y + (m2*z + m3*w)), and with
 *     z = 0, w = 1 known at compile time the inner term folds to m3;
 *   - normalize(v) = v * (1 / sqrt((x*x + y*y) + z*z)), with a true divide and sqrt;
 *   - min/max are the GLSL definitions (y < x ? y : x,  x < y ? y : x);
 *   - linear filtering: unnormalised coordinate u*size - 0.5 (two roundings), floor, fraction,
 *     texels outside the image read the border colour, and every lerp is
 *     fma(w, b - a, a) (that driver builds lerps with llvm.fmuladd, fused on an FMA host);
 *   - float -> unorm8: NaN and negatives to 0, >= 1 to 255, else round-half-even of x*255.
 * ---------------------------------------------------------------------------------------- */

static inline float glsl_min(float x, float y) { return y < x ? y : x; }
static inline float glsl_max(float x, float y) { return x < y ? y : x; }
static inline float lerp_fma(float w, float a, float b) { return fmaf(w, b - a, a); }

typedef struct {
    const float *vol;
    int nx, ny, nz;
    float fx, fy, fz;
    const float *lut;
    float border[4];
} sampler_state;

static inline float texel3(const sampler_state *s, int x, int y, int z)
{
    if ((unsigned)x >= (unsigned)s->nx || (unsigned)y >= (unsigned)s->ny || (unsigned)z >= (unsigned)s->nz)
        return 0.0f;                                    /* clamp-to-border, border .r = 0 */
    return s->vol[((size_t)z * s->ny + y) * s->nx + x];
}

/* Clamp a floored coordinate so that the int conversion is defined for any input, without
 * changing which texels are inside the image: everything below -2 behaves like -2 (both
 * neighbours are border), everything above size behaves like size. */
static inline int safe_index(float fl, int size)
{
    if (!(fl >= -2.0f)) fl = -2.0f;                     /* also catches NaN */
    if (fl > (float)size) fl = (float)size;
    return (int)fl;
}

/* texture(volumeTexture, tc).x — VolumeRendering.comp:187-188, sampler main.cpp:365-382 */
static inline float sample_volume(const sampler_state *s, float u, float v, float w)
{
    const float x = u * s->fx - 0.5f, y = v * s->fy - 0.5f, z = w * s->fz - 0.5f;
    const float xf = floorf(x), yf = floorf(y), zf = floorf(z);
    const float wx = x - xf, wy = y - yf, wz = z - zf;
    const int x0 = safe_index(xf, s->nx), y0 = safe_index(yf, s->ny), z0 = safe_index(zf, s->nz);
    const float c000 = texel3(s, x0, y0, z0),         c100 = texel3(s, x0 + 1, y0, z0);
    const float c010 = texel3(s, x0, y0 + 1, z0),     c110 = texel3(s, x0 + 1, y0 + 1, z0);
    const float c001 = texel3(s, x0, y0, z0 + 1),     c101 = texel3(s, x0 + 1, y0, z0 + 1);
    const float c011 = texel3(s, x0, y0 + 1, z0 + 1), c111 = texel3(s, x0 + 1, y0 + 1, z0 + 1);
    const float p0 = lerp_fma(wy, lerp_fma(wx, c000, c100), lerp_fma(wx, c010, c110));
    const float p1 = lerp_fma(wy, lerp_fma(wx, c001, c101), lerp_fma(wx, c011, c111));
    return lerp_fma(wz, p0, p1);
}

/* texture(transferSampler, intensity) — VolumeRendering.comp:98-106 */
static inline void sample_tf(const sampler_state *s, float intensity, float rgba[4])
{
    const float x = intensity * 256.0f - 0.5f;
    const float xf = floorf(x);
    const float wx = x - xf;
    const int i0 = safe_index(xf, 256);
    const float *t0 = ((unsigned)i0 < 256u) ? s->lut + 4 * i0 : s->border;
    const float *t1 = ((unsigned)(i0 + 1) < 256u) ? s->lut + 4 * (i0 + 1) : s->border;
    for (int c = 0; c < 4; c++) rgba[c] = lerp_fma(wx, t0[c], t1[c]);
}

static inline uint8_t to_unorm8(float x)
{
    if (!(x > 0.0f)) return 0;
    if (x >= 1.0f) return 255;
    return (uint8_t)lrintf(x * 255.0f);                 /* round-half-even in the default mode */
}

typedef struct { int hit; float tmin, tmax; float o[3], d[3]; } ray_t;

/* VolumeRendering.comp:159-160, 177-178, 50-96 */
static inline ray_t make_ray(const float *loc, const float *m, int px, int py)
{
    ray_t r;
    const float fx = (float)px, fy = (float)py;
    float world[3];
    for (int c = 0; c < 3; c++) {
        const float t = m[4 + c] * fy + m[12 + c];      /* m1*y + m3 (m2*0 folded away) */
        world[c] = m[c] * fx + t;                       /* m0*x + ... */
    }
    float d[3] = { world[0] - loc[0], world[1] - loc[1], world[2] - loc[2] };
    const float inv = 1.0f / sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
    for (int c = 0; c < 3; c++) { r.o[c] = loc[c]; r.d[c] = d[c] * inv; }

    /* volumeMin (-.5,-.5,-1), volumeMax (.5,.5,0): VolumeRendering.comp:166-167.
     * intersectSlab: dot(n, p - o) / dot(n, d) with axis normals reduces to (p_c - o_c) / d_c;
     * for the z slabs numerator and denominator are both negated, same quotient. */
    float t1 = (-0.5f - r.o[0]) / r.d[0], t2 = (0.5f - r.o[0]) / r.d[0];
    const float tminx = glsl_min(t1, t2), tmaxx = glsl_max(t1, t2);
    t1 = (-0.5f - r.o[1]) / r.d[1]; t2 = (0.5f - r.o[1]) / r.d[1];
    const float tminy = glsl_min(t1, t2), tmaxy = glsl_max(t1, t2);
    t1 = (-1.0f - r.o[2]) / r.d[2]; t2 = (0.0f - r.o[2]) / r.d[2];
    const float tminz = glsl_min(t1, t2), tmaxz = glsl_max(t1, t2);
    r.tmin = glsl_max(glsl_max(tminx, tminy), tminz);
    r.tmax = glsl_min(glsl_min(tmaxx, tmaxy), tmaxz);
    r.hit = !(r.tmin > r.tmax || r.tmax < 0.0f);        /* VolumeRendering.comp:91 */
    return r;
}

/* The local illumination the shader carries but has commented out (VolumeRendering.comp:195-198):
 * computeNormal (:109-119), shading (:123-152), with the push constants `Scene` (main.cpp:80-98,808).
 * scene = { ambient, diffuse, specular, shininess }.  Returns the one float the shader adds to all
 * three colour channels (light colour (1,1,1) and ambient light (0.3,0.3,0.3) are grey).
 * Conventions beyond the shader text:
 *   - normalize(v) = v * (1 / sqrt((x*x + y*y) + z*z)) as for the ray direction;
 *   - normalize of the zero vector, undefined in GLSL (a flat neighbourhood has gradient 0 and would
 *     turn the pixel into NaN), gives the zero vector: such a tap gets the ambient term only;
 *   - dot(a, b) = (a.x*b.x + a.y*b.y) + a.z*b.z; pow is powf. */
static inline void normalize3(float v[3])
{
    const float d = (v[0] * v[0] + v[1] * v[1]) + v[2] * v[2];
    if (!(d > 0.0f)) { v[0] = v[1] = v[2] = 0.0f; return; }
    const float inv = 1.0f / sqrtf(d);
    v[0] *= inv; v[1] *= inv; v[2] *= inv;
}

static inline float shade_tap(const sampler_state *s, const float loc[4], float x, float y, float z,
                              float step, const float scene[4])
{
    /* computeNormal: central differences one ray step to either side, :111-113 */
    const float w[3] = { x, y, z }, org[3] = { -0.5f, -0.5f, 0.0f };
    float n[3];
    for (int a = 0; a < 3; a++) {
        float tp[3], tm[3];
        for (int c = 0; c < 3; c++) {
            const float off = c == a ? step : 0.0f;
            tp[c] = fabsf((w[c] + off) - org[c]);
            tm[c] = fabsf((w[c] - off) - org[c]);
        }
        n[a] = (sample_volume(s, tp[0], tp[1], tp[2]) - sample_volume(s, tm[0], tm[1], tm[2])) / (2.0f * step);
    }
    normalize3(n);                                                   /* :116 */
    const float v[3] = { loc[0] - x, loc[1] - y, loc[2] - z };       /* sampleToView, :196 */
    const float l[3] = { 0.0f - x, 0.0f - y, -1.0f - z };            /* lightLocation (0,0,-1), :169,197 */
    float h[3] = { l[0] + v[0], l[1] + v[1], l[2] + v[2] };
    normalize3(h);                                                   /* :137 */
    const float ambient = scene[0] * 0.3f;                           /* :140 */
    const float diffuse_light = glsl_max((l[0] * n[0] + l[1] * n[1]) + l[2] * n[2], 0.0f);   /* :143 */
    const float diffuse = scene[1] * diffuse_light;                  /* :144 */
    float specular_light = powf(glsl_max((h[0] * n[0] + h[1] * n[1]) + h[2] * n[2], 0.0f), scene[3]);   /* :147 */
    if (diffuse_light <= 0.0f) specular_light = 0.0f;                /* :148 */
    const float specular = scene[2] * specular_light;                /* :149 */
    return (ambient + diffuse) + specular;                           /* :151 */
}

/* Two clones: one for hosts with FMA3 (fmaf becomes one instruction), one generic (libm fmaf,
 * which is exact as well).  Results are identical; only the speed differs. */
__attribute__((target_clones("arch=haswell", "default")))
static void render_rows(const sampler_state *s, const float loc[4], const float m[16],
                        uint32_t width, uint32_t samples, uint32_t row0, uint32_t row1,
                        uint8_t *out, float *colour, int threads, const float *scene)
{
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
#endif
    for (uint32_t py = row0; py < row1; py++) {
        for (uint32_t px = 0; px < width; px++) {
            float C[3] = { 0.0f, 0.0f, 0.0f };          /* clearColor, VolumeRendering.comp:174-175 */
            const ray_t r = make_ray(loc, m, (int)px, (int)py);
            if (r.hit) {
                const float step = (r.tmax - r.tmin) / (float)(samples + 1);    /* :183 */
                for (uint32_t t = 0; t <= samples + 1; t++) {                    /* :184 */
                    const float len = r.tmax - (float)t * step;                   /* :185 */
                    const float x = r.o[0] + len * r.d[0];
                    const float y = r.o[1] + len * r.d[1];
                    const float z = r.o[2] + len * r.d[2];                        /* :186 */
                    /* abs(sampleLocation - samplerOrigin), samplerOrigin (-.5,-.5,0): :168,187 */
                    const float I = sample_volume(s, fabsf(x - -0.5f), fabsf(y - -0.5f), fabsf(z - 0.0f));
                    float tf[4];
                    sample_tf(s, I, tf);
                    if (t == 0) {                                                 /* :189-191 */
                        C[0] = tf[0]; C[1] = tf[1]; C[2] = tf[2];
                    } else {                                                      /* :200 */
                        if (scene) {                                              /* :195-198, disabled in the shader */
                            const float sh = shade_tap(s, loc, x, y, z, step, scene);
                            tf[0] += sh; tf[1] += sh; tf[2] += sh;
                        }
                        const float a = tf[3];
                        const float ia = 1.0f - a;
                        C[0] = a * tf[0] + ia * C[0];
                        C[1] = a * tf[1] + ia * C[1];
                        C[2] = a * tf[2] + ia * C[2];
                    }
                }
            }
            uint8_t *o = out + ((size_t)py * width + px) * 4;
            o[0] = to_unorm8(C[0]); o[1] = to_unorm8(C[1]); o[2] = to_unorm8(C[2]);
            o[3] = 255;                                                           /* :205 */
            if (colour) {
                float *c = colour + ((size_t)py * width + px) * 3;
                c[0] = C[0]; c[1] = C[1]; c[2] = C[2];
            }
        }
    }
}

void vro_render_f(const float *volume, uint32_t nx, uint32_t ny, uint32_t nz,
                  const float *lut, const float tf_border[4],
                  const float location[4], const float to_world[16],
                  uint32_t width, uint32_t height, uint32_t samples,
                  uint32_t row0, uint32_t row1, uint8_t *out, float *colour, int threads)
{
    sampler_state s;
    s.vol = volume; s.nx = (int)nx; s.ny = (int)ny; s.nz = (int)nz;
    s.fx = (float)nx; s.fy = (float)ny; s.fz = (float)nz;
    s.lut = lut;
    memcpy(s.border, tf_border, sizeof s.border);
    if (row1 > height) row1 = height;
    render_rows(&s, location, to_world, width, samples, row0, row1, out, colour, threads, NULL);
}

void vro_render_shaded(const float *volume, uint32_t nx, uint32_t ny, uint32_t nz,
                       const float *lut, const float tf_border[4],
                       const float location[4], const float to_world[16],
                       uint32_t width, uint32_t height, uint32_t samples,
                       uint32_t row0, uint32_t row1, uint8_t *out, float *colour, int threads,
                       const float scene[4])
{
    sampler_state s;
    s.vol = volume; s.nx = (int)nx; s.ny = (int)ny; s.nz = (int)nz;
    s.fx = (float)nx; s.fy = (float)ny; s.fz = (float)nz;
    s.lut = lut;
    memcpy(s.border, tf_border, sizeof s.border);
    if (row1 > height) row1 = height;
    render_rows(&s, location, to_world, width, samples, row0, row1, out, colour, threads, scene);
}

void vro_render(const float *volume, uint32_t nx, uint32_t ny, uint32_t nz,
                const float *lut, const float tf_border[4],
                const float location[4], const float to_world[16],
                uint32_t width, uint32_t height, uint32_t samples,
                uint32_t row0, uint32_t row1, uint8_t *out, int threads)
{
    vro_render_f(volume, nx, ny, nz, lut, tf_border, location, to_world, width, height, samples,
                 row0, row1, out, NULL, threads);
}

uint64_t vro_count_hits_rows(const float location[4], const float to_world[16],
                             uint32_t width, uint32_t row0, uint32_t row1)
{
    uint64_t hits = 0;
    for (uint32_t py = row0; py < row1; py++)
        for (uint32_t px = 0; px < width; px++)
            hits += (uint64_t)make_ray(location, to_world, (int)px, (int)py).hit;
    return hits;
}

uint64_t vro_count_hits(const float location[4], const float to_world[16],
                        uint32_t width, uint32_t height)
{
    uint64_t hits = 0;
    for (uint32_t py = 0; py < height; py++)
        for (uint32_t px = 0; px < width; px++)
            hits += (uint64_t)make_ray(location, to_world, (int)px, (int)py).hit;
    return hits;
}

/* ------------------------------------------------------------------------------------------
 * Sort-last checkers: the shader loop restricted to the taps of one box of bricks.
 * ---------------------------------------------------------------------------------------- */

typedef struct { float rgba[4]; int owned; } tap_t;

/* Tap t of a ray (VolumeRendering.comp:185-188, 193) and whether the box owns it. */
static inline tap_t eval_tap(const sampler_state *s, const ray_t *r, float step, uint32_t t,
                             const int lo[3], const int hi[3])
{
    tap_t o;
    const float len = r->tmax - (float)t * step;
    const float u = fabsf((r->o[0] + len * r->d[0]) - -0.5f);
    const float v = fabsf((r->o[1] + len * r->d[1]) - -0.5f);
    const float w = fabsf((r->o[2] + len * r->d[2]) - 0.0f);
    const int cx = safe_index(floorf(u * s->fx - 0.5f), s->nx) + 2;
    const int cy = safe_index(floorf(v * s->fy - 0.5f), s->ny) + 2;
    const int cz = safe_index(floorf(w * s->fz - 0.5f), s->nz) + 2;
    o.owned = cx >= 8 * lo[0] && cx < 8 * hi[0] && cy >= 8 * lo[1] && cy < 8 * hi[1] &&
              cz >= 8 * lo[2] && cz < 8 * hi[2];
    o.rgba[0] = o.rgba[1] = o.rgba[2] = o.rgba[3] = 0.0f;
    if (o.owned) sample_tf(s, sample_volume(s, u, v, w), o.rgba);
    return o;
}

static sampler_state make_sampler(const float *volume, uint32_t nx, uint32_t ny, uint32_t nz,
                                  const float *lut, const float tf_border[4])
{
    sampler_state s;
    s.vol = volume; s.nx = (int)nx; s.ny = (int)ny; s.nz = (int)nz;
    s.fx = (float)nx; s.fy = (float)ny; s.fz = (float)nz;
    s.lut = lut;
    memcpy(s.border, tf_border, sizeof s.border);
    return s;
}

void vro_render_partial(const float *volume, uint32_t nx, uint32_t ny, uint32_t nz,
                        const float *lut, const float tf_border[4],
                        const float location[4], const float to_world[16],
                        uint32_t width, uint32_t height, uint32_t samples,
                        const int brick_lo[3], const int brick_hi[3], float *out, int threads)
{
    const sampler_state s = make_sampler(volume, nx, ny, nz, lut, tf_border);
    const double cut = 0x1p-20;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
#endif
    for (uint32_t py = 0; py < height; py++)
        for (uint32_t px = 0; px < width; px++) {
            float *o = out + ((size_t)py * width + px) * 5;
            o[0] = o[1] = o[2] = o[3] = o[4] = 0.0f;
            const ray_t r = make_ray(location, to_world, (int)px, (int)py);
            if (!r.hit) continue;
            const float step = (r.tmax - r.tmin) / (float)(samples + 1);
            double F[3] = { 0.0, 0.0, 0.0 }, Tr = 1.0;
            float S = 0.0f;
            for (uint32_t t = samples + 1; t >= 1 && !(Tr < cut); t--) {
                const tap_t tap = eval_tap(&s, &r, step, t, brick_lo, brick_hi);
                if (!tap.owned || tap.rgba[3] == 0.0f) continue;
                const double a = (double)tap.rgba[3], w = Tr * a;
                for (int c = 0; c < 3; c++) F[c] = fma(w, (double)tap.rgba[c], F[c]);
                S += (float)Tr;
                Tr = Tr * (1.0 - a);
            }
            if (!(Tr < cut)) {
                const tap_t tap = eval_tap(&s, &r, step, 0, brick_lo, brick_hi);
                if (tap.owned) {
                    for (int c = 0; c < 3; c++) F[c] = fma(Tr, (double)tap.rgba[c], F[c]);
                    Tr = 0.0;
                }
            }
            o[0] = (float)F[0]; o[1] = (float)F[1]; o[2] = (float)F[2]; o[3] = (float)Tr; o[4] = S;
        }
}

static inline void march_direction(const ray_t *r, float step, const sampler_state *s, int up[3])
{
    up[0] = -(step * r->d[0]) * s->fx >= 0.0f;
    up[1] = -(step * r->d[1]) * s->fy >= 0.0f;
    up[2] = (step * r->d[2]) * s->fz >= 0.0f;
}

void vro_march_direction(const float location[4], const float to_world[16],
                         uint32_t width, uint32_t height, uint32_t samples,
                         uint32_t nx, uint32_t ny, uint32_t nz, uint8_t *out)
{
    sampler_state s;
    memset(&s, 0, sizeof s);
    s.fx = (float)nx; s.fy = (float)ny; s.fz = (float)nz;
    for (uint32_t py = 0; py < height; py++)
        for (uint32_t px = 0; px < width; px++) {
            const ray_t r = make_ray(location, to_world, (int)px, (int)py);
            const float step = (r.tmax - r.tmin) / (float)(samples + 1);
            int up[3];
            march_direction(&r, step, &s, up);
            uint8_t *o = out + ((size_t)py * width + px) * 4;
            o[0] = (uint8_t)up[0]; o[1] = (uint8_t)up[1]; o[2] = (uint8_t)up[2]; o[3] = (uint8_t)r.hit;
        }
}

void vro_chain_stage(const float *volume, uint32_t nx, uint32_t ny, uint32_t nz,
                     const float *lut, const float tf_border[4],
                     const float location[4], const float to_world[16],
                     uint32_t width, uint32_t height, uint32_t samples,
                     const int brick_lo[3], const int brick_hi[3],
                     const uint32_t *pixels, uint32_t count, const float *colour_in, float *colour_out,
                     int stage, const int grid[3], const int box[3])
{
    const sampler_state s = make_sampler(volume, nx, ny, nz, lut, tf_border);
    for (uint32_t e = 0; e < count; e++) {
        float *o = colour_out + (size_t)e * 4;
        o[0] = o[1] = o[2] = o[3] = 0.0f;
        const uint32_t pixel = pixels[e];
        if (pixel >= width * height) continue;
        const ray_t r = make_ray(location, to_world, (int)(pixel % width), (int)(pixel / width));
        const float step = (r.tmax - r.tmin) / (float)(samples + 1);
        int up[3], m[3];
        march_direction(&r, step, &s, up);
        for (int k = 0; k < 3; k++) m[k] = up[k] ? box[k] : grid[k] - 1 - box[k];
        if ((m[2] * grid[1] + m[1]) * grid[0] + m[0] != stage) continue;
        float C[3] = { colour_in[e * 4], colour_in[e * 4 + 1], colour_in[e * 4 + 2] };
        if (r.hit)
            for (uint32_t t = 0; t <= samples + 1; t++) {
                const tap_t tap = eval_tap(&s, &r, step, t, brick_lo, brick_hi);
                if (!tap.owned) continue;
                if (t == 0) {
                    C[0] = tap.rgba[0]; C[1] = tap.rgba[1]; C[2] = tap.rgba[2];
                } else {
                    const float a = tap.rgba[3], ia = 1.0f - a;
                    for (int c = 0; c < 3; c++) C[c] = a * tap.rgba[c] + ia * C[c];
                }
            }
        o[0] = C[0]; o[1] = C[1]; o[2] = C[2];
    }
}

/* ------------------------------------------------------------------------------------------
 * Synthetic phantom (measurement input of this build, not part of the reference).
 * 3-D Shepp-Logan head: ten ellipsoids, Kak-Slaney geometry, "modified" contrast expressed in
 * tenths so the sum is an exact integer.  Voxel centres of [-1,1]^3.  Inside the outer
 * ellipsoid a small hash noise is added; outside it the value is exactly 0.
 * ---------------------------------------------------------------------------------------- */

typedef struct { int tenths; float ia2, ib2, ic2, x0, y0, z0, cp, sp; } ellipsoid;

#include "phantom_table.inc"

static inline uint32_t phantom_hash(uint32_t seed, uint64_t idx)
{
    uint32_t x = (uint32_t)idx ^ ((uint32_t)(idx >> 32) * 0x85EBCA6Bu) ^ seed;
    x *= 0x9E3779B1u;
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    return x >> 8;
}

static inline uint32_t phantom_voxel(uint32_t ix, uint32_t iy, uint32_t iz, uint32_t n,
                                     uint32_t dmax, uint32_t seed)
{
    const float fn = (float)n;
    const float x = (float)(2 * ix + 1) / fn - 1.0f;
    const float y = (float)(2 * iy + 1) / fn - 1.0f;
    const float z = (float)(2 * iz + 1) / fn - 1.0f;
    int tenths = 0, inside_outer = 0;
    for (int e = 0; e < 10; e++) {
        const ellipsoid *E = &PHANTOM[e];
        const float dx = x - E->x0, dy = y - E->y0, dz = z - E->z0;
        const float xr = dx * E->cp + dy * E->sp;
        const float yr = dy * E->cp - dx * E->sp;
        const float q = ((xr * xr) * E->ia2 + (yr * yr) * E->ib2) + (dz * dz) * E->ic2;
        if (q <= 1.0f) { tenths += E->tenths; if (e == 0) inside_outer = 1; }
    }
    if (!inside_outer) return 0;
    if (tenths < 0) tenths = 0;
    if (tenths > 10) tenths = 10;
    const uint32_t range = (dmax + 1) / 128;
    const uint64_t idx = ((uint64_t)iz * n + iy) * n + ix;
    return (uint32_t)tenths * (dmax - (range - 1)) / 10 + phantom_hash(seed, idx) % range;
}

void vro_phantom(void *out, uint32_t n, int dtype_bytes, uint32_t seed)
{
    const uint32_t dmax = dtype_bytes == 1 ? 255u : 65535u;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (uint32_t iz = 0; iz < n; iz++)
        for (uint32_t iy = 0; iy < n; iy++)
            for (uint32_t ix = 0; ix < n; ix++) {
                const uint32_t v = phantom_voxel(ix, iy, iz, n, dmax, seed);
                const uint64_t idx = ((uint64_t)iz * n + iy) * n + ix;
                if (dtype_bytes == 1) ((uint8_t *)out)[idx] = (uint8_t)v;
                else ((uint16_t *)out)[idx] = (uint16_t)v;
            }
}
