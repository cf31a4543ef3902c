// Host-side camera uniforms of the VolumeRendering application, restated without glm.
//
// Produces what the reference uploads every frame: the ViewPyramid block and the pixel-index ->
// world matrix (Applications/VolumeRendering/Source/main.cpp:59-77, 182-202, 737-799).  The
// reference builds them with glm (a vcpkg dependency, version not pinned); the matrix helpers
// below follow glm's scalar code paths: column-major storage, mat*mat accumulating
// ((a0*b0 + a1*b1) + a2*b2) + a3*b3, mat*vec as (m0*v0 + m1*v1) + (m2*v2 + m3*v3),
// rotate() in glm's Rodrigues form applied to the identity.  This translation unit must be
// compiled with -ffp-contract=off so that every operation rounds once, as MSVC /fp:precise does.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>

namespace vr {
namespace cam {

struct Mat4 {
    float c[4][4];  // c[column][row]
};
struct Vec4 {
    float v[4];
};

inline Mat4 identity()
{
    Mat4 m;
    std::memset(&m, 0, sizeof m);
    for (int i = 0; i < 4; i++) m.c[i][i] = 1.0f;
    return m;
}

inline Mat4 mul(const Mat4 &a, const Mat4 &b)
{
    Mat4 r;
    for (int col = 0; col < 4; col++)
        for (int row = 0; row < 4; row++) {
            float acc = a.c[0][row] * b.c[col][0];
            acc = acc + a.c[1][row] * b.c[col][1];
            acc = acc + a.c[2][row] * b.c[col][2];
            acc = acc + a.c[3][row] * b.c[col][3];
            r.c[col][row] = acc;
        }
    return r;
}

inline Vec4 mul(const Mat4 &m, const Vec4 &x)
{
    Vec4 r;
    for (int row = 0; row < 4; row++) {
        const float lo = m.c[0][row] * x.v[0] + m.c[1][row] * x.v[1];
        const float hi = m.c[2][row] * x.v[2] + m.c[3][row] * x.v[3];
        r.v[row] = lo + hi;
    }
    return r;
}

inline Mat4 translate(float tx, float ty, float tz)
{
    Mat4 m = identity();
    for (int row = 0; row < 4; row++)
        m.c[3][row] = m.c[0][row] * tx + m.c[1][row] * ty + m.c[2][row] * tz + m.c[3][row];
    return m;
}

inline Mat4 rotate(float angle, float ax, float ay, float az)
{
    const float co = std::cos(angle);
    const float si = std::sin(angle);
    const float rlen = 1.0f / std::sqrt(ax * ax + ay * ay + az * az);
    const float n[3] = {ax * rlen, ay * rlen, az * rlen};
    const float t[3] = {(1.0f - co) * n[0], (1.0f - co) * n[1], (1.0f - co) * n[2]};
    float rot[3][3];
    rot[0][0] = co + t[0] * n[0];
    rot[0][1] = t[0] * n[1] + si * n[2];
    rot[0][2] = t[0] * n[2] - si * n[1];
    rot[1][0] = t[1] * n[0] - si * n[2];
    rot[1][1] = co + t[1] * n[1];
    rot[1][2] = t[1] * n[2] + si * n[0];
    rot[2][0] = t[2] * n[0] + si * n[1];
    rot[2][1] = t[2] * n[1] - si * n[0];
    rot[2][2] = co + t[2] * n[2];
    const Mat4 base = identity();
    Mat4 r;
    for (int col = 0; col < 3; col++)
        for (int row = 0; row < 4; row++)
            r.c[col][row] = base.c[0][row] * rot[col][0] + base.c[1][row] * rot[col][1] + base.c[2][row] * rot[col][2];
    for (int row = 0; row < 4; row++) r.c[3][row] = base.c[3][row];
    return r;
}

// The default pose, main.cpp:70-76 and setupViewPyramid main.cpp:187-201: columns are
// side, up, forward, location.
inline Mat4 default_view_pyramid()
{
    Mat4 basis = identity();
    basis.c[2][2] = -1.0f;  // forward (0,0,-1)
    const float quarter_turn = 90.0f * 0.01745329251994329576923690768489f;  // glm::radians(90.0f)
    const Mat4 pose = mul(translate(0.0f, -5.0f, -0.5f), rotate(quarter_turn, 1.0f, 0.0f, 0.0f));
    return mul(pose, basis);
}

constexpr float kPyramidHeight = 4.5f;  // main.cpp:197

// updatePixelTransformationUniforms, main.cpp:737-799.
// view_pyramid: {height, location, side, up, forward} as 5 vec4; to_world column-major.
inline void pixel_transform(float horizontal_rad, float vertical_rad, uint32_t width, uint32_t height,
                            float view_pyramid[20], float to_world[16])
{
    const Mat4 pose0 = default_view_pyramid();
    const float centre[3] = {0.0f, 0.0f, -0.5f};  // volumeGridCenter, main.cpp:68

    Mat4 basis = pose0;
    basis.c[3][0] = basis.c[3][1] = basis.c[3][2] = 0.0f;
    basis.c[3][3] = 1.0f;
    const Mat4 pitch = rotate(vertical_rad, 1.0f, 0.0f, 0.0f);   // main.cpp:752
    const Mat4 roll = rotate(horizontal_rad, 0.0f, 0.0f, 1.0f);  // main.cpp:753
    const Mat4 turn = mul(roll, pitch);
    basis = mul(turn, basis);  // main.cpp:755
    const Vec4 arm = {{pose0.c[3][0] - centre[0], pose0.c[3][1] - centre[1], pose0.c[3][2] - centre[2], 1.0f}};
    const Vec4 turned = mul(turn, arm);  // main.cpp:756
    for (int i = 0; i < 3; i++) basis.c[3][i] = centre[i] + turned.v[i];  // main.cpp:757
    basis.c[3][3] = 1.0f;

    view_pyramid[0] = kPyramidHeight;
    view_pyramid[1] = view_pyramid[2] = view_pyramid[3] = 0.0f;
    for (int row = 0; row < 4; row++) {
        view_pyramid[4 + row] = basis.c[3][row];
        view_pyramid[8 + row] = basis.c[0][row];
        view_pyramid[12 + row] = basis.c[1][row];
        view_pyramid[16 + row] = basis.c[2][row];
    }

    Mat4 to_unit = identity();  // main.cpp:771-776
    to_unit.c[0][0] = 1.0f / (static_cast<float>(width) - 1.0f);
    to_unit.c[1][1] = 1.0f / (static_cast<float>(height) - 1.0f);
    Mat4 flip = identity();  // main.cpp:777-782
    flip.c[1][1] = -1.0f;
    const Mat4 centre_grid = translate(-0.5f, 0.5f, 0.0f);  // main.cpp:783
    Mat4 orient = basis;                                     // main.cpp:784-789
    orient.c[3][0] = orient.c[3][1] = orient.c[3][2] = 0.0f;
    orient.c[3][3] = 1.0f;
    const Mat4 place = translate(basis.c[3][0] + kPyramidHeight * basis.c[2][0],
                                 basis.c[3][1] + kPyramidHeight * basis.c[2][1],
                                 basis.c[3][2] + kPyramidHeight * basis.c[2][2]);  // main.cpp:790
    Mat4 m = mul(place, orient);  // main.cpp:794, evaluated left to right
    m = mul(m, centre_grid);
    m = mul(m, flip);
    m = mul(m, to_unit);
    for (int col = 0; col < 4; col++)
        for (int row = 0; row < 4; row++) to_world[col * 4 + row] = m.c[col][row];
}

}  // namespace cam
}  // namespace vr
