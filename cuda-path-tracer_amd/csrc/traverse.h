// Two-level BVH traversal for gfx950, one ray per lane.
//
// Semantics follow IntersectScene / IntersectBLAS of the reference
// (kernel/intersection.h:30-235) and its slab / Moeller-Trumbore tests
// (math/intersection.h:9-71):
//   * ordered depth-first descent, the nearer child first; on equal entry distances the
//     right child first (the reference's `left_time < right_time` test);
//   * CLOSEST: entries whose box entry distance is >= the current best are skipped when
//     popped; a triangle replaces the best only when strictly closer, so the first one
//     found wins ties;
//   * inside an instance the search starts from "no hit" again and the candidate is compared
//     with the best of other instances by its world-space distance
//     dot(world_pos - origin, direction), exactly like the reference's nested call;
//   * each of the two levels owns a stack of 64 entries whose pushes are dropped when full
//     (utils/stack.h:8-11); the drop order of the reference is reproduced.
// Differences that do not change results: the near child is kept in a register instead of
// being pushed and popped again; both levels share one physical stack; ANY returns at the
// first accepted triangle.  Like the reference, ANY tests boxes against FLT_MAX, not against the
// ray's t_max: a box test and a triangle test can disagree within rounding when t is close to
// t_max (a grazing shadow ray ending on the light it was aimed at), and then skipping the box
// would change the answer.  Trace mode 1 does skip such boxes.
//
// The stack lives in LDS as [entry][lane] 8-byte records: a wave's 64 lanes at any mix of
// depths hit 64 different bank pairs, so pushes and pops never conflict.  Entries beyond
// kLdsDepth go to a per-lane global spill area.
#pragma once
#include "dscene.h"

namespace ptd {

#ifndef PT_LDS_DEPTH
#define PT_LDS_DEPTH 16
#endif
constexpr int kLdsDepth = PT_LDS_DEPTH;  // stack entries per lane held in LDS by the ray-query kernels
#ifndef PT_RENDER_LDS_DEPTH
#define PT_RENDER_LDS_DEPTH 8
#endif
// The kernels of a render pass share the CUs with the other pass in flight and are sensitive to
// LDS use (measured: 2 KB more per wave cost 4 % of the frame rate), so they keep fewer entries
// in LDS and spill a little more often.
constexpr int kRenderLdsDepth = PT_RENDER_LDS_DEPTH;
#ifndef PT_RENDER_LDS_DEPTH_WIDE
#define PT_RENDER_LDS_DEPTH_WIDE 12
#endif
// A 4-wide node pushes up to three entries at once and does so without a branch only while three
// more entries fit the LDS part of the stack, so the render kernels of trace mode 2 keep a few more.
constexpr int kRenderLdsDepthWide = PT_RENDER_LDS_DEPTH_WIDE;
constexpr int kRefStack = 64;  // capacity of each of the reference's two stacks
constexpr int kSpillDepth = 2 * kRefStack - (kLdsDepth < kRenderLdsDepth ? kLdsDepth : kRenderLdsDepth);  // sized for the shallower LDS part
// The spill area has one more row than that: the first entry of a wave's share of it holds the wave's two
// counters of rare events (LaneEvents).
constexpr int kSpillRows = kSpillDepth + 1;
constexpr int kWave = 64;

enum { MODE_CLOSEST = 0, MODE_ANY = 1 };

// LDS pointers keep their address space in the type, so stack traffic is ds_read_b64 /
// ds_write_b64 and never a flat access that has to pick an address space at run time.
typedef __attribute__((address_space(3))) unsigned long long *LdsU64;

typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4f *LdsF4;
// A wave's LDS: the stack (kLdsDepth entries for each of 64 lanes) and, behind it, 64 rays read
// ahead of their use (two float4 each).
constexpr int kRayStageEntries = 2 * kWave * 2;  // in 8-byte units
template <int DEPTH>
constexpr int wave_lds_entries() { return DEPTH * kWave + kRayStageEntries; }
constexpr int kWaveLdsEntries = wave_lds_entries<kLdsDepth>();

template <int DEPTH>
struct LaneStackT {
    static constexpr int kDepth = DEPTH;  // entries per lane in LDS
    LdsF4 stage;                // the wave's staged rays: (o.xyz, tmax) at [k], (d.xyz, valid) at [64 + k]
    LdsU64 lds;                 // this lane's column of the wave's LDS stack; entry k at lds[k * 64]
    unsigned long long *spill;  // this lane's column of the global spill area; entry k at spill[k * stride]
    uint32_t stride;            // lanes in the spill area
};

// `lds_array` is the kernel's __shared__ array of wave_lds_entries<DEPTH>() entries, `spill_base` the wave's
// first column of the global spill area.
template <int DEPTH = kLdsDepth>
PTD LaneStackT<DEPTH> make_lane_stack(unsigned long long *lds_array, uint2 *spill_base, uint32_t stride) {
    const uint32_t lane = threadIdx.x & 63u;
    return LaneStackT<DEPTH>{(LdsF4)(lds_array + DEPTH * kWave), (LdsU64)(lds_array + lane), reinterpret_cast<unsigned long long *>(spill_base) + lane, stride};
}

struct HitRec {
    float t, u, v;
    uint32_t rec;  // DTri index of the hit
    int inst;      // GeometryInstance index, -1 = miss
    // carried along from records the traversal reads anyway, so that finishing a ray needs no further gather:
    uint32_t prim;      // the triangle's index inside its mesh (DTri::c.y)
    uint32_t mat, type; // material id and material type of the instance (DInst::d.y, d.z)
};

struct TraceCounters {  // per-ray visit counts, filled only by the counting variants
    uint32_t interior, tlas_leaves, tris, blas_leaves;
};

// Rare events (an entry stored beyond the LDS part of the stack, a push dropped like the reference's full
// stack drops it) are counted per wave in global memory, not in registers: two words in the extra row of the
// wave's share of the spill area, zeroed by the wave when it starts and added to the context's statistics when
// it ends (flush_events).  Counters in registers are carried around the box-test loop as eight copies per round.
struct LaneEvents {
    uint32_t *wave = nullptr;  // {spills, drops} of this wave
    PTD void spill() const { atomicAdd(wave, 1u); }
    PTD void drop(uint32_t n = 1) const { atomicAdd(wave + 1, n); }
#ifdef PT_PHASE_PROF
    // diagnostic build only: per phase, rounds this lane took part in / rounds the wave ran
    mutable uint32_t lane_rounds[8] = {0, 0, 0, 0, 0, 0, 0, 0}, wave_rounds[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
};

#ifdef PT_PHASE_PROF
#define PT_PHASE(ev, k)                                                                        \
    do {                                                                                       \
        (ev).lane_rounds[k]++;                                                                 \
        if (int(threadIdx.x & 63) == __ffsll((unsigned long long)__ballot(true)) - 1) (ev).wave_rounds[k]++; \
    } while (0)
#else
#define PT_PHASE(ev, k) do { } while (0)
#endif

// One 8-byte record per entry: node reference in the low word, box entry distance in the high.
template <class STK>
PTD void stack_store(const STK &s, int sp, uint32_t ref, float t, const LaneEvents &ev) {
    const unsigned long long v = ((unsigned long long)as_u(t) << 32) | ref;
    if (__builtin_expect(sp < STK::kDepth, 1)) {
        s.lds[sp * kWave] = v;
    } else {
        s.spill[size_t(sp - STK::kDepth) * s.stride] = v;
        ev.spill();
    }
}
// The wave's event counters: `spill_base` is the wave's first column of the spill area.  Called by all lanes.
PTD LaneEvents make_events(uint2 *spill_base, uint32_t stride) {
    LaneEvents ev;
    ev.wave = reinterpret_cast<uint32_t *>(spill_base + size_t(kSpillDepth) * stride);
    if ((threadIdx.x & 63u) == 0) *reinterpret_cast<uint2 *>(ev.wave) = make_uint2(0u, 0u);
    return ev;
}
template <class STK>
PTD uint2 stack_load(const STK &s, int sp) {
    unsigned long long v;
    if (__builtin_expect(sp < STK::kDepth, 1)) {
        v = s.lds[sp * kWave];
    } else {
        v = __builtin_nontemporal_load(s.spill + size_t(sp - STK::kDepth) * s.stride);
    }
    return make_uint2(uint32_t(v), uint32_t(v >> 32));
}

// math/intersection.h:9-35 with inv_d hoisted (it is the same value on every call).
PTD bool slab(V3 o, V3 inv, float minx, float miny, float minz, float maxx, float maxy, float maxz, float t_max,
              float &t_out) {
    float t0 = (minx - o.x) * inv.x;
    float t1 = (maxx - o.x) * inv.x;
    float tnear = fminf(t0, t1);
    float tfar = fmaxf(t0, t1);
    t0 = (miny - o.y) * inv.y;
    t1 = (maxy - o.y) * inv.y;
    tnear = fmaxf(tnear, fminf(t0, t1));
    tfar = fminf(tfar, fmaxf(t0, t1));
    t0 = (minz - o.z) * inv.z;
    t1 = (maxz - o.z) * inv.z;
    tnear = fmaxf(tnear, fminf(t0, t1));
    tfar = fminf(tfar, fmaxf(t0, t1));
    t_out = tnear;
    return (tfar >= tnear) && (tnear < t_max) && (tfar > 0.0f);
}

// math/intersection.h:37-71 on a packed triangle record.
PTD bool tri_test(V3 o, V3 d, float4 ta, float4 tb, float4 tc, float t_max, float &t_out, float &u_out, float &v_out) {
    const V3 v0 = mk3(ta.x, ta.y, ta.z), v1 = mk3(ta.w, tb.x, tb.y), v2 = mk3(tb.z, tb.w, tc.x);
    const V3 edge1 = v1 - v0, edge2 = v2 - v0;
    const V3 p = cross(d, edge2);
    const float a = dot(edge1, p);
    if (fabsf(a) < 1e-9f) return false;
    const float inv_a = 1.0f / a;
    const V3 tvec = o - v0;
    const float u = dot(tvec, p) * inv_a;
    if (u < 0.0f || u > 1.0f) return false;
    const V3 q = cross(tvec, edge1);
    const float v = dot(d, q) * inv_a;
    if (v < 0.0f || u + v > 1.0f) return false;
    const float t = dot(edge2, q) * inv_a;
    if (t <= 1e-9f) return false;
    if (t > t_max) return false;
    t_out = t;
    u_out = u;
    v_out = v;
    return true;
}

PTD M34 load_inv(const DInst &in) {
    M34 m;
    m.c0 = mk3(in.a.x, in.a.y, in.a.z);
    m.c1 = mk3(in.a.w, in.b.x, in.b.y);
    m.c2 = mk3(in.b.z, in.b.w, in.c.x);
    m.c3 = mk3(in.c.y, in.c.z, in.c.w);
    return m;
}
PTD M34 load_m34(float4 q0, float4 q1, float4 q2) {
    M34 m;
    m.c0 = mk3(q0.x, q0.y, q0.z);
    m.c1 = mk3(q0.w, q1.x, q1.y);
    m.c2 = mk3(q1.z, q1.w, q2.x);
    m.c3 = mk3(q2.y, q2.z, q2.w);
    return m;
}

// Object-space position of barycentric (u,v) on a triangle record: w*p0 + u*p1 + v*p2
// (kernel/intersection.h:96-99).
PTD V3 tri_point(const DTri *tris, uint32_t rec, float u, float v) {
    const float4 *p = reinterpret_cast<const float4 *>(tris + rec);
    const float4 ta = p[0], tb = p[1], tc = p[2];
    const V3 p0 = mk3(ta.x, ta.y, ta.z), p1 = mk3(ta.w, tb.x, tb.y), p2 = mk3(tb.z, tb.w, tc.x);
    const float w = 1.0f - u - v;
    return w * p0 + u * p1 + v * p2;
}

}  // namespace ptd
