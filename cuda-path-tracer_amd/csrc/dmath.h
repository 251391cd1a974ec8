// Device-side fp32 helpers for the gfx950 kernels.
//
// The reference writes its arithmetic with glm; the order of additions inside dot,
// cross, mat*vec and normalize decides the last bit of every hit record, so the
// helpers below spell that order out (and the translation unit is compiled with
// -ffp-contract=off so no multiply-add pair is fused behind our back):
//   dot(a,b)     = (a.x*b.x + a.y*b.y) + a.z*b.z
//   cross(a,b)   = (a.y*b.z - b.y*a.z, a.z*b.x - b.z*a.x, a.x*b.y - b.x*a.y)
//   normalize(v) = v * (1 / sqrt(dot(v,v)))
//   M * (v,w)    = (c0*v.x + c1*v.y) + (c2*v.z + c3*w)
#pragma once
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>

namespace ptd {

struct V3 {
    float x, y, z;
};
struct V2 {
    float x, y;
};

#define PTD __device__ __forceinline__

PTD V3 mk3(float x, float y, float z) { return V3{x, y, z}; }
PTD V3 splat(float s) { return V3{s, s, s}; }
PTD V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
PTD V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
PTD V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
PTD V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
PTD V3 operator*(float s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
PTD V3 operator/(V3 a, float s) { return V3{a.x / s, a.y / s, a.z / s}; }
PTD V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
PTD V3 operator-(float s, V3 a) { return V3{s - a.x, s - a.y, s - a.z}; }
PTD V2 operator*(float s, V2 a) { return V2{s * a.x, s * a.y}; }
PTD V2 operator+(V2 a, V2 b) { return V2{a.x + b.x, a.y + b.y}; }

PTD float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
PTD V3 cross(V3 a, V3 b) { return V3{a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
PTD float length2(V3 v) { return dot(v, v); }
PTD float length(V3 v) { return sqrtf(dot(v, v)); }
PTD V3 normalize(V3 v) { return v * (1.0f / sqrtf(dot(v, v))); }
PTD float gmin(float a, float b) { return b < a ? b : a; }  // glm::min / std::min
PTD float gmax(float a, float b) { return a < b ? b : a; }  // glm::max / std::max
PTD V3 mix3(V3 a, V3 b, float t) { return a * (1.0f - t) + b * t; }

// sin and cos of an azimuth phi in [0, 2*pi] (math/sampling.h:33-35, :44-56 call sinf / cosf there).  Every
// implementation of those rounds differently in the last bit (CUDA's, glibc's, the ROCm device library's), and one
// differing bit in a sampled direction sends a path down another branch a few bounces later.  So the azimuth
// functions are defined here as a fixed sequence of correctly rounded fp32 operations, which a CPU can repeat bit for
// bit: quadrant k = int(phi * 2/pi + 0.5); r = (phi - k * 1.5703125) - k * 4.8382679e-4 (pi/2 split in two parts);
// with z = r * r: sin r = r + r * z * (S0 + z * (S1 + z * S2)), cos r = (1 - z/2) + z * z * (C0 + z * (C1 + z * C2))
// (single-precision minimax coefficients, error below 1.2e-7 over the quadrant), then the quadrant's rotation.
PTD void sincos_azimuth(float phi, float &sin_phi, float &cos_phi) {
    const int k = int(phi * 0.63661977236758134f + 0.5f);
    const float kf = float(k);
    const float r = (phi - kf * 1.5703125f) - kf * 4.83826794897e-4f;
    const float z = r * r;
    const float s = r + (r * z) * (-1.6666654611e-1f + z * (8.3321608736e-3f + z * -1.9515295891e-4f));
    const float c = (1.0f - 0.5f * z) + (z * z) * (4.166664568298827e-2f + z * (-1.388731625493765e-3f + z * 2.443315711809948e-5f));
    const bool swap = (k & 1) != 0;
    const float ss = swap ? c : s, cc = swap ? s : c;
    sin_phi = (k & 2) ? -ss : ss;
    cos_phi = ((k + 1) & 2) ? -cc : cc;
}

// 3x4 affine part of a column-major mat4: columns c0..c3 (xyz each).
struct M34 {
    V3 c0, c1, c2, c3;
};
PTD V3 xform(const M34 &m, V3 v, float w) {
    V3 a = m.c0 * v.x + m.c1 * v.y;
    V3 b = m.c2 * v.z + m.c3 * w;
    return a + b;
}

PTD float as_f(uint32_t u) { return __uint_as_float(u); }
PTD uint32_t as_u(float f) { return __float_as_uint(f); }

// ---- PCG stream of the reference's independent sampler (kernel/sampler/independent.h:10-32,
// math/hash.h:5-9); integer-exact.
PTD uint32_t pcg_hash(uint32_t input) {
    uint32_t state = input * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
PTD uint64_t seed_sampler(int px, int py, int iteration) {
    uint32_t key_xy = (uint32_t(px) & 0xFFFFu) | ((uint32_t(py) & 0xFFFFu) << 16);
    uint32_t key = key_xy ^ (uint32_t(iteration) * 0x9E3779B9u);
    uint32_t hashed = pcg_hash(key);
    return (uint64_t(hashed) << 32) | uint64_t(hashed ^ 0xdeadbeefu);
}
PTD float sample1d(uint64_t &state) {
    uint64_t old = state;
    state = old * 6364136223846793005ULL + 1;
    uint32_t x = uint32_t(((old >> 18) ^ old) >> 27);
    uint32_t r = uint32_t(old >> 59);
    uint32_t v = (x >> r) | (x << ((0u - r) & 31u));
    return float(v) * (1.0f / 4294967296.0f);
}

}  // namespace ptd
