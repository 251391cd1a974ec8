// Host-side fp32 vector / matrix helpers of the product's C++ facade.
// Operation order follows glm's scalar code paths (the reference builds its
// matrices and boxes with glm), so arrays produced here are what a glm build of
// the reference's host code would hand to the device boundary.
#pragma once
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>

namespace pth {

struct Vec3 {
    float x = 0, y = 0, z = 0;
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator*(Vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float Dot(Vec3 a, Vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline Vec3 Cross(Vec3 a, Vec3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline Vec3 Normalize(Vec3 v) { return v * (1.0f / std::sqrt(Dot(v, v))); }
inline float Length(Vec3 v) { return std::sqrt(Dot(v, v)); }
inline float Min(float a, float b) { return b < a ? b : a; }  // glm::min
inline float Max(float a, float b) { return a < b ? b : a; }  // glm::max

// Axis-aligned box with the reference's empty state (math/aabb.h:8-64).
struct Box {
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
    float hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    void Grow(const float *p) {
        for (int a = 0; a < 3; ++a) {
            lo[a] = Min(lo[a], p[a]);
            hi[a] = Max(hi[a], p[a]);
        }
    }
    void Grow(const Box &o) {
        for (int a = 0; a < 3; ++a) {
            lo[a] = Min(lo[a], o.lo[a]);
            hi[a] = Max(hi[a], o.hi[a]);
        }
    }
    bool Empty() const { return lo[0] > hi[0] || lo[1] > hi[1] || lo[2] > hi[2]; }
    float Area() const {
        float dx = Max(hi[0] - lo[0], 0.0f), dy = Max(hi[1] - lo[1], 0.0f), dz = Max(hi[2] - lo[2], 0.0f);
        return 2.0f * (dx * dy + dy * dz + dz * dx);
    }
};

// Column-major 4x4 (glm::mat4 memory order): m[col*4 + row].
struct Mat4 {
    float m[16];
    static Mat4 Identity() {
        Mat4 r;
        std::memset(r.m, 0, sizeof(r.m));
        r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0f;
        return r;
    }
    float &at(int col, int row) { return m[col * 4 + row]; }
    float at(int col, int row) const { return m[col * 4 + row]; }
};

// mat4 * vec4 as glm evaluates it: (c0*x + c1*y) + (c2*z + c3*w).
inline void MulVec4(const Mat4 &M, const float v[4], float out[4]) {
    for (int r = 0; r < 4; ++r) {
        float a = M.at(0, r) * v[0] + M.at(1, r) * v[1];
        float b = M.at(2, r) * v[2] + M.at(3, r) * v[3];
        out[r] = a + b;
    }
}
inline Vec3 TransformPoint(const Mat4 &M, Vec3 p) {
    float v[4] = {p.x, p.y, p.z, 1.0f}, o[4];
    MulVec4(M, v, o);
    return {o[0], o[1], o[2]};
}
// mat4 * mat4 as glm evaluates it: column j = ((A0*b0 + A1*b1) + A2*b2) + A3*b3.
inline Mat4 Mul(const Mat4 &A, const Mat4 &B) {
    Mat4 R;
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 4; ++r) {
            float t = A.at(0, r) * B.at(j, 0);
            t = t + A.at(1, r) * B.at(j, 1);
            t = t + A.at(2, r) * B.at(j, 2);
            t = t + A.at(3, r) * B.at(j, 3);
            R.at(j, r) = t;
        }
    return R;
}

Mat4 Translate(const Mat4 &M, Vec3 v);                 // glm::translate
Mat4 Scale(const Mat4 &M, Vec3 v);                     // glm::scale
Mat4 Rotate(const Mat4 &M, float radians, Vec3 axis);  // glm::rotate
Mat4 FromQuat(float x, float y, float z, float w);     // glm::mat4_cast
Mat4 Inverse(const Mat4 &M);                           // glm::inverse
Mat4 InverseTranspose(const Mat4 &M);                  // glm::inverseTranspose
inline float Radians(float deg) { return deg * 0.01745329251994329576923690768489f; }

// Math::GetTransformMatrix, math/transform.cpp:5-14: T * (Rx * Ry * Rz) * S, angles in degrees.
Mat4 GetTransformMatrix(Vec3 translate, Vec3 rotate_deg, Vec3 scale);

}  // namespace pth
