// Host-side float matrix helpers with glm's conventions (column-major, post-multiplying transforms).
// They produce the matrices the reference builds with glm::translate / rotate / scale / inverse /
// perspective (Camera.cpp:22-35, Box.cpp:16-23, Mesh.cpp:47-55): same operation order, float precision.
#pragma once
#include <cmath>
#include <cstring>

namespace ptamd {

struct Mat4 {
    float m[4][4]; // m[column][row]
    static Mat4 identity()
    {
        Mat4 r; std::memset(&r, 0, sizeof r);
        for (int i = 0; i < 4; ++i) r.m[i][i] = 1.0f;
        return r;
    }
};

inline float deg2rad(float d) { return d * 0.01745329251994329576923690768489f; }

inline void axpy_col(float out[4], const float a[4], float s) { for (int r = 0; r < 4; ++r) out[r] = a[r] * s; }

inline Mat4 translate(const Mat4& a, const float v[3])
{
    Mat4 r = a;
    for (int k = 0; k < 4; ++k) r.m[3][k] = a.m[0][k] * v[0] + a.m[1][k] * v[1] + a.m[2][k] * v[2] + a.m[3][k];
    return r;
}

// Rotation about one of the principal axes (axis = 0, 1, 2), angle in radians; the general axis-angle
// form of glm::rotate evaluated for a unit axis.
inline Mat4 rotate_axis(const Mat4& a, float angle, int axis)
{
    const float c = std::cos(angle), s = std::sin(angle);
    float ax[3] = { 0.f, 0.f, 0.f };
    ax[axis] = 1.0f;
    // glm normalises the axis first: x * inversesqrt(dot(x,x)) leaves (1,0,0) unchanged
    float t[3] = { (1.0f - c) * ax[0], (1.0f - c) * ax[1], (1.0f - c) * ax[2] };
    float R[3][3];
    R[0][0] = c + t[0] * ax[0];           R[0][1] = t[0] * ax[1] + s * ax[2];   R[0][2] = t[0] * ax[2] - s * ax[1];
    R[1][0] = t[1] * ax[0] - s * ax[2];   R[1][1] = c + t[1] * ax[1];           R[1][2] = t[1] * ax[2] + s * ax[0];
    R[2][0] = t[2] * ax[0] + s * ax[1];   R[2][1] = t[2] * ax[1] - s * ax[0];   R[2][2] = c + t[2] * ax[2];
    Mat4 r;
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 4; ++k) r.m[i][k] = a.m[0][k] * R[i][0] + a.m[1][k] * R[i][1] + a.m[2][k] * R[i][2];
    for (int k = 0; k < 4; ++k) r.m[3][k] = a.m[3][k];
    return r;
}

// R = a * Rx * Ry * Rz, Euler angles in degrees.
inline Mat4 rotate_euler_xyz(Mat4 a, const float rot_deg[3])
{
    a = rotate_axis(a, deg2rad(rot_deg[0]), 0);
    a = rotate_axis(a, deg2rad(rot_deg[1]), 1);
    a = rotate_axis(a, deg2rad(rot_deg[2]), 2);
    return a;
}

inline Mat4 scale(const Mat4& a, const float v[3])
{
    Mat4 r;
    for (int k = 0; k < 4; ++k) { r.m[0][k] = a.m[0][k] * v[0]; r.m[1][k] = a.m[1][k] * v[1]; r.m[2][k] = a.m[2][k] * v[2]; r.m[3][k] = a.m[3][k]; }
    return r;
}

inline Mat4 mul(const Mat4& a, const Mat4& b)
{
    Mat4 r;
    for (int c = 0; c < 4; ++c)
        for (int k = 0; k < 4; ++k) {
            float t = a.m[0][k] * b.m[c][0];
            t = t + a.m[1][k] * b.m[c][1];
            t = t + a.m[2][k] * b.m[c][2];
            t = t + a.m[3][k] * b.m[c][3];
            r.m[c][k] = t;
        }
    return r;
}

// Cofactor inverse, the expansion glm::inverse(mat4) uses.
inline Mat4 inverse(const Mat4& in)
{
    const float (*m)[4] = in.m;
    float c00 = m[2][2] * m[3][3] - m[3][2] * m[2][3], c02 = m[1][2] * m[3][3] - m[3][2] * m[1][3], c03 = m[1][2] * m[2][3] - m[2][2] * m[1][3];
    float c04 = m[2][1] * m[3][3] - m[3][1] * m[2][3], c06 = m[1][1] * m[3][3] - m[3][1] * m[1][3], c07 = m[1][1] * m[2][3] - m[2][1] * m[1][3];
    float c08 = m[2][1] * m[3][2] - m[3][1] * m[2][2], c10 = m[1][1] * m[3][2] - m[3][1] * m[1][2], c11 = m[1][1] * m[2][2] - m[2][1] * m[1][2];
    float c12 = m[2][0] * m[3][3] - m[3][0] * m[2][3], c14 = m[1][0] * m[3][3] - m[3][0] * m[1][3], c15 = m[1][0] * m[2][3] - m[2][0] * m[1][3];
    float c16 = m[2][0] * m[3][2] - m[3][0] * m[2][2], c18 = m[1][0] * m[3][2] - m[3][0] * m[1][2], c19 = m[1][0] * m[2][2] - m[2][0] * m[1][2];
    float c20 = m[2][0] * m[3][1] - m[3][0] * m[2][1], c22 = m[1][0] * m[3][1] - m[3][0] * m[1][1], c23 = m[1][0] * m[2][1] - m[2][0] * m[1][1];
    const float f0[4] = { c00, c00, c02, c03 }, f1[4] = { c04, c04, c06, c07 }, f2[4] = { c08, c08, c10, c11 };
    const float f3[4] = { c12, c12, c14, c15 }, f4[4] = { c16, c16, c18, c19 }, f5[4] = { c20, c20, c22, c23 };
    const float v0[4] = { m[1][0], m[0][0], m[0][0], m[0][0] }, v1[4] = { m[1][1], m[0][1], m[0][1], m[0][1] };
    const float v2[4] = { m[1][2], m[0][2], m[0][2], m[0][2] }, v3[4] = { m[1][3], m[0][3], m[0][3], m[0][3] };
    const float sa[4] = { 1, -1, 1, -1 }, sb[4] = { -1, 1, -1, 1 };
    Mat4 inv;
    for (int i = 0; i < 4; ++i) {
        inv.m[0][i] = (v1[i] * f0[i] - v2[i] * f1[i] + v3[i] * f2[i]) * sa[i];
        inv.m[1][i] = (v0[i] * f0[i] - v2[i] * f3[i] + v3[i] * f4[i]) * sb[i];
        inv.m[2][i] = (v0[i] * f1[i] - v1[i] * f3[i] + v3[i] * f5[i]) * sa[i];
        inv.m[3][i] = (v0[i] * f2[i] - v1[i] * f4[i] + v2[i] * f5[i]) * sb[i];
    }
    float det = (m[0][0] * inv.m[0][0] + m[0][1] * inv.m[1][0]) + (m[0][2] * inv.m[2][0] + m[0][3] * inv.m[3][0]);
    float ood = 1.0f / det;
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) inv.m[c][r] *= ood;
    return inv;
}

// glm::perspective = perspectiveRH_NO (right-handed, depth -1..1); fovy in radians.
inline Mat4 perspective(float fovy, float aspect, float zn, float zf)
{
    const float th = std::tan(fovy / 2.0f);
    Mat4 r; std::memset(&r, 0, sizeof r);
    r.m[0][0] = 1.0f / (aspect * th);
    r.m[1][1] = 1.0f / th;
    r.m[2][2] = -(zf + zn) / (zf - zn);
    r.m[2][3] = -1.0f;
    r.m[3][2] = -(2.0f * zf * zn) / (zf - zn);
    return r;
}

} // namespace ptamd
