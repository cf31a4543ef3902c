/*
 * CPU oracle for the VolumeRendering ray-march path — TEST INFRASTRUCTURE ONLY.
 *
 * This is a scalar C restatement of the reference algorithm, written to be read next to the
 * reference sources it cites (paths relative to the reference checkout).  It is never linked
 * into, imported by, or called from the shipped library: only tests/, the smoke check and the
 * benchmark's cpu_baseline leg may use it, and there only as the checker.
 *
 * PARITY UNPINNED: the reference has no tests, golden images or known-answer vectors, and it
 * cannot be built here (needs a Vulkan driver, a GLSL compiler, glm, GLFW, ImGui, Crypto++ and
 * C++23 ranges).  Arithmetic that the reference delegates to absent third parties is restated
 * from their published behaviour and the choices are listed where they are made:
 *   - glm (unpinned vcpkg version): rotate/translate/mat*mat/mat*vec association orders;
 *   - the Vulkan implementation: texel filtering, border colour, float->unorm8 conversion,
 *     and how the GLSL compiler associates/fuses the shader's own float expressions.
 */
#ifndef VR_ORACLE_H
#define VR_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float x; float intensity; } vro_ctrl_pt;

/* Transfer function LUT: Applications/VolumeRendering/Source/main.cpp:643-686.
 * lut is 256 texels x RGBA (1024 floats), updated in place (state persists between calls,
 * main.cpp:110).  pts[c]/n[c] are the control points of channel c (R,G,B,A), sorted by x.
 * A channel with n[c] < 1 is left untouched.  Returns 0, or -1 if an index would leave the
 * table (undefined behaviour in the reference). */
int vro_lut_update(float *lut, const vro_ctrl_pt *const pts[4], const int n[4]);

/* getIntensity, main.cpp:948-953: intensity of a control point placed at pixel row y of a
 * histogram child window that is child_h pixels high. */
float vro_control_point_intensity(float y, float child_h);

/* Camera: main.cpp:59-77 (defaults), 182-202 (setupViewPyramid), 737-799
 * (updatePixelTransformationUniforms).  view_pyramid receives the 80-byte uniform block
 * {height, location, side, up, forward} (5 x vec4); to_world the column-major mat4. */
void vro_camera(float horizontal_rad, float vertical_rad, uint32_t width, uint32_t height,
                float view_pyramid[20], float to_world[16]);

/* Volume normalisation, main.cpp:137-164: float(v) / float(max) per voxel.
 * max_value <= 0 means "use the maximum of the data" as the reference does. Returns max used. */
float vro_normalize_u16(const uint16_t *v, uint64_t count, float max_value, float *out);
float vro_normalize_u8(const uint8_t *v, uint64_t count, float max_value, float *out);

/* Slice loader, main.cpp:112-165: every file of dir, ordered by the integer after the last
 * dot (or the whole name), big-endian u16, headerless; width = height = floor(sqrt(count of
 * the first slice)).  *out is malloc'ed normalised floats (caller frees).  Returns 0 or -1. */
int vro_load_slices(const char *dir, float **out, uint32_t *width, uint32_t *height,
                    uint32_t *slices, float *max_value);

/* Intensity histogram, main.cpp:166-181: 100 bins, bounds [i/100, (i+0.9)/100], normalised. */
void vro_histogram(const float *normalized, uint64_t count, float hist[100]);

/* The shader: Applications/VolumeRendering/Shader/VolumeRendering.comp:154-206 for the
 * pixel rows [row0,row1) of a width x height image.  volume is the normalised float 3-D image
 * (x fastest, then y, then slice; main.cpp:244-258), lut the 256-texel RGBA 1-D image,
 * tf_border the sampler border colour seen by the LUT fetch (main.cpp:380; the volume fetch
 * only reads .r, which is 0 under every reading of that enum), samples the shader constant at
 * VolumeRendering.comp:173 (200).  out is width*height*4 bytes RGBA8, only the requested rows
 * are written.  threads <= 0 means all cores. */
void vro_render(const float *volume, uint32_t nx, uint32_t ny, uint32_t nz,
                const float *lut, const float tf_border[4],
                const float location[4], const float to_world[16],
                uint32_t width, uint32_t height, uint32_t samples,
                uint32_t row0, uint32_t row1, uint8_t *out, int threads);

/* Same, but also returns per-pixel float colour before the unorm8 store (3 floats / pixel),
 * for diagnostics in tests.  colour may be NULL. */
void vro_render_f(const float *volume, uint32_t nx, uint32_t ny, uint32_t nz,
                  const float *lut, const float tf_border[4],
                  const float location[4], const float to_world[16],
                  uint32_t width, uint32_t height, uint32_t samples,
                  uint32_t row0, uint32_t row1, uint8_t *out, float *colour, int threads);

/* The same frame with the local illumination the shader has commented out switched on
 * (VolumeRendering.comp:109-152,195-198; push constants main.cpp:80-98,808):
 * scene = { ambient, diffuse, specular, shininess }, the reference's values are { 1, 1, 1, 100 }.
 * normalize() of a zero vector, undefined in GLSL, yields the zero vector here. */
void vro_render_shaded(const float *volume, uint32_t nx, uint32_t ny, uint32_t nz,
                       const float *lut, const float tf_border[4],
                       const float location[4], const float to_world[16],
                       uint32_t width, uint32_t height, uint32_t samples,
                       uint32_t row0, uint32_t row1, uint8_t *out, float *colour, int threads,
                       const float scene[4]);

/* Number of rays of the frame that hit the box (for the nominal sample count). */
uint64_t vro_count_hits(const float location[4], const float to_world[16],
                        uint32_t width, uint32_t height);
uint64_t vro_count_hits_rows(const float location[4], const float to_world[16],
                             uint32_t width, uint32_t row0, uint32_t row1);

/* Synthetic CT phantom (not part of the reference; the measurement input of this build):
 * 3-D Shepp-Logan, n^3 voxels, dtype_bytes 1 or 2, little-endian raw, x fastest. */
void vro_phantom(void *out, uint32_t n, int dtype_bytes, uint32_t seed);

/* ---- Checkers for the sort-last split (the split itself has no counterpart in the reference; these
 * restate the shader loop, VolumeRendering.comp:183-202, restricted to the taps of one box).
 * The volume's padded cell grid (cell = voxel index + 2) is cut into cubes of 8 cells ("bricks");
 * a box is the brick range [brick_lo, brick_hi).  A tap belongs to the box that holds the low corner
 * cell of its trilinear footprint. */

/* Per pixel five floats (F0, F1, F2, Tr, S): the box's taps taken front to back, F = sum of
 * T a c (double accumulation), Tr = product of (1 - a), S = sum of T over taps with a != 0; the
 * owner of tap 0 adds Tr * c0 and sets Tr = 0; marching stops below Tr = 2^-20; all zero for a
 * ray that misses the volume. */
void vro_render_partial(const float *volume, uint32_t nx, uint32_t ny, uint32_t nz,
                        const float *lut, const float tf_border[4],
                        const float location[4], const float to_world[16],
                        uint32_t width, uint32_t height, uint32_t samples,
                        const int brick_lo[3], const int brick_hi[3], float *out, int threads);

/* Per pixel four bytes: whether the x, y, z voxel index grows with the tap index (towards the
 * eye), and whether the ray hits the volume. */
void vro_march_direction(const float location[4], const float to_world[16],
                         uint32_t width, uint32_t height, uint32_t samples,
                         uint32_t nx, uint32_t ny, uint32_t nz, uint8_t *out);

/* One stage of the back-to-front pass over listed pixels: boxes form a grid[0] x grid[1] x grid[2]
 * grid, this one at `box`; for pixels whose back-to-front rank of this box (lexicographic, z
 * slowest, axes the ray descends along mirrored) equals stage, blend the box's taps in the
 * shader's order over colour_in (3 floats + 1 unused per pixel), else write zeros. */
void vro_chain_stage(const float *volume, uint32_t nx, uint32_t ny, uint32_t nz,
                     const float *lut, const float tf_border[4],
                     const float location[4], const float to_world[16],
                     uint32_t width, uint32_t height, uint32_t samples,
                     const int brick_lo[3], const int brick_hi[3],
                     const uint32_t *pixels, uint32_t count, const float *colour_in, float *colour_out,
                     int stage, const int grid[3], const int box[3]);

int vro_threads(void);

#ifdef __cplusplus
}
#endif
#endif
