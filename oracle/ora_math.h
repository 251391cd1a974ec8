// TEST INFRASTRUCTURE - CPU oracle, never linked into or called by the product.
// parity unpinned: the reference ships no tests or golden vectors, and its
// sources cannot be built in this image (glm is absent), so this restatement is
// checked by analytic known answers and brute-force cross-checks only.
//
// Minimal fp32 vector/matrix types whose operator order follows glm's scalar
// code paths, because the reference's arithmetic is written with glm and the
// order of additions decides the last bit:
//   dot(a,b)      = (a.x*b.x + a.y*b.y) + a.z*b.z
//   cross(a,b)    = (a.y*b.z - b.y*a.z, a.z*b.x - b.z*a.x, a.x*b.y - b.x*a.y)
//   normalize(v)  = v * (1 / sqrt(dot(v,v)))
//   mat4 * vec4   = (m0*x + m1*y) + (m2*z + m3*w)
// Build with -ffp-contract=off so that no a*b+c is fused.
#pragma once
#include <cfloat>
#include <cmath>
#include <cstdint>

namespace ora {

struct vec2 {
    float x, y;
};
struct vec3 {
    float x, y, z;
    float operator[](int i) const { return (&x)[i]; }
    float &operator[](int i) { return (&x)[i]; }
};
struct vec4 {
    float x, y, z, w;
};

inline vec3 V3(float s) { return {s, s, s}; }
inline vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator*(vec3 a, vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline vec3 operator/(vec3 a, vec3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
inline vec3 operator*(vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline vec3 operator*(float s, vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline vec3 operator/(vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline vec3 operator/(float s, vec3 a) { return {s / a.x, s / a.y, s / a.z}; }
inline vec3 operator-(vec3 a) { return {-a.x, -a.y, -a.z}; }
inline vec3 operator-(float s, vec3 a) { return {s - a.x, s - a.y, s - a.z}; }
inline vec3 &operator+=(vec3 &a, vec3 b) { a = a + b; return a; }
inline vec3 &operator*=(vec3 &a, vec3 b) { a = a * b; return a; }
inline vec3 &operator/=(vec3 &a, float s) { a = a / s; return a; }
inline vec2 operator*(float s, vec2 a) { return {s * a.x, s * a.y}; }
inline vec2 operator+(vec2 a, vec2 b) { return {a.x + b.x, a.y + b.y}; }

inline float dot(vec3 a, vec3 b) {
    vec3 t = a * b;
    return t.x + t.y + t.z;
}
inline vec3 cross(vec3 x, vec3 y) {
    return {x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y};
}
inline float length2(vec3 v) { return dot(v, v); }
inline float length(vec3 v) { return sqrtf(dot(v, v)); }
inline float distance(vec3 a, vec3 b) { return length(b - a); }
inline float distance2(vec3 a, vec3 b) { return length2(b - a); }
inline vec3 normalize(vec3 v) { return v * (1.0f / sqrtf(dot(v, v))); }
inline vec3 vmin(vec3 a, vec3 b) { return {b.x < a.x ? b.x : a.x, b.y < a.y ? b.y : a.y, b.z < a.z ? b.z : a.z}; }
inline vec3 vmax(vec3 a, vec3 b) { return {a.x < b.x ? b.x : a.x, a.y < b.y ? b.y : a.y, a.z < b.z ? b.z : a.z}; }
inline float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }
inline vec3 mix(vec3 a, vec3 b, float t) {  // glm::mix: x*(1-a) + y*a
    return a * (1.0f - t) + b * t;
}

// Column-major 4x4, c[j] is column j (glm::mat4 memory order).
struct mat4 {
    vec4 c[4];
};
inline vec4 operator*(vec4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
inline vec4 operator+(vec4 a, vec4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline vec4 mul(const mat4 &m, vec4 v) {
    vec4 a0 = m.c[0] * v.x + m.c[1] * v.y;
    vec4 a1 = m.c[2] * v.z + m.c[3] * v.w;
    return a0 + a1;
}
inline vec3 xform_point(const mat4 &m, vec3 p) {
    vec4 r = mul(m, {p.x, p.y, p.z, 1.0f});
    return {r.x, r.y, r.z};
}
inline vec3 xform_dir(const mat4 &m, vec3 d) {
    vec4 r = mul(m, {d.x, d.y, d.z, 0.0f});
    return {r.x, r.y, r.z};
}
inline mat4 mul(const mat4 &a, const mat4 &b) {
    mat4 r;
    for (int j = 0; j < 4; ++j) {
        vec4 t = a.c[0] * b.c[j].x;
        t = t + a.c[1] * b.c[j].y;
        t = t + a.c[2] * b.c[j].z;
        t = t + a.c[3] * b.c[j].w;
        r.c[j] = t;
    }
    return r;
}
inline mat4 identity() {
    return {{{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}}};
}

}  // namespace ora
