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
// first accepted triangle and culls boxes with the ray's t_max.
//
// The stack lives in LDS as [entry][lane] 8-byte records: a wave's 64 lanes at any mix of
// depths hit 64 different bank pairs, so pushes and pops never conflict.  Entries beyond
// kLdsDepth go to a per-lane global spill area.
#pragma once
#include "dscene.h"

namespace ptd {

constexpr int kLdsDepth = 24;  // stack entries per lane held in LDS
constexpr int kRefStack = 64;  // capacity of each of the reference's two stacks
constexpr int kSpillDepth = 2 * kRefStack - kLdsDepth;
constexpr int kWave = 64;

enum { MODE_CLOSEST = 0, MODE_ANY = 1 };

struct LaneStack {
    uint2 *lds;       // this lane's column of the wave's LDS stack; entry k at lds[k * 64]
    uint2 *spill;     // this lane's column of the global spill area; entry k at spill[k * stride]
    uint32_t stride;  // lanes in the spill area
};

struct HitRec {
    float t, u, v;
    uint32_t rec;  // DTri index of the hit
    int inst;      // GeometryInstance index, -1 = miss
};

struct TraceCounters {  // per-ray visit counts, filled only by the counting variants
    uint32_t interior, tlas_leaves, tris, blas_leaves;
};

struct LaneEvents {  // rare events, flushed to the context's statistics by the caller
    uint32_t spills = 0, drops = 0;
};

PTD void stack_store(const LaneStack &s, int sp, uint32_t ref, float t, LaneEvents &ev) {
    if (sp < kLdsDepth) {
        s.lds[sp * kWave] = make_uint2(ref, as_u(t));
    } else {
        s.spill[size_t(sp - kLdsDepth) * s.stride] = make_uint2(ref, as_u(t));
        ev.spills++;
    }
}
PTD uint2 stack_load(const LaneStack &s, int sp) {
    if (sp < kLdsDepth) return s.lds[sp * kWave];
    return s.spill[size_t(sp - kLdsDepth) * s.stride];
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

// Traces one ray.  Returns true on a hit; for MODE_CLOSEST `hit` then holds the winner.
template <int MODE, bool COUNT>
PTD bool trace_ray(const DScene &S, V3 o, V3 d, float t_max, const LaneStack &stk, HitRec &hit, LaneEvents &ev,
                   TraceCounters &ctr) {
    hit.t = -1.0f;
    hit.u = hit.v = 0.0f;
    hit.rec = 0;
    hit.inst = -1;
    const V3 iw = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    {
        float t;
        if (!slab(o, iw, S.root_min[0], S.root_min[1], S.root_min[2], S.root_max[0], S.root_max[1], S.root_max[2], t_max, t))
            return false;  // kernel/intersection.h:162
    }
    V3 co = o, cd = d, ci = iw;  // ray in the current space (world, or the instance's object space)
    float best = FLT_MAX;        // best world-space distance over finished instances
    float lbest = FLT_MAX;       // best object-space distance inside the current instance
    uint32_t lrec = 0;
    float lu = 0.0f, lv = 0.0f;
    int cur_inst = -1;
    int sp = 0, base = 0;
    bool in_blas = false;
    uint32_t cur = S.root_ref;

    for (;;) {
        // ---- descend through interior nodes
        while (cur < kNoRef) {
            if (COUNT) ctr.interior++;
            const float4 *np = reinterpret_cast<const float4 *>(S.nodes + cur);
            const float4 na = np[0], nb = np[1], nc = np[2], nd = np[3];
            const float bound = (MODE == MODE_ANY) ? t_max : (in_blas ? lbest : best);
            float tl, tr;
            const bool hl = slab(co, ci, na.x, na.y, na.z, nb.x, nb.y, nb.z, bound, tl);
            const bool hr = slab(co, ci, nc.x, nc.y, nc.z, nd.x, nd.y, nd.z, bound, tr);
            const uint32_t lref = as_u(na.w), rref = as_u(nb.w);
            const int depth = sp - base;
            if (hl && hr) {
                const bool left_first = tl < tr;
                const uint32_t near_ref = left_first ? lref : rref;
                const uint32_t far_ref = left_first ? rref : lref;
                const float far_t = left_first ? tr : tl;
                if (depth < kRefStack - 1) {
                    stack_store(stk, sp, far_ref, far_t, ev);
                    ++sp;
                    cur = near_ref;
                } else if (depth == kRefStack - 1) {  // the reference pushes far, then drops near
                    stack_store(stk, sp, far_ref, far_t, ev);
                    ++sp;
                    ev.drops++;
                    cur = kNoRef;
                } else {
                    ev.drops += 2;
                    cur = kNoRef;
                }
            } else if (hl || hr) {
                if (depth < kRefStack) {
                    cur = hl ? lref : rref;
                } else {
                    ev.drops++;
                    cur = kNoRef;
                }
            } else {
                cur = kNoRef;
            }
        }

        if (cur != kNoRef) {
            if (!in_blas) {
                // ---- TLAS leaf: enter the instance (kernel/intersection.h:183-188, :37-50)
                if (COUNT) ctr.tlas_leaves++;
                cur_inst = int(cur & 0x7FFFFFFFu);
                const float4 *ip = reinterpret_cast<const float4 *>(S.insts + cur_inst);
                DInst in;
                in.a = ip[0];
                in.b = ip[1];
                in.c = ip[2];
                in.d = ip[3];
                const M34 inv = load_inv(in);
                co = xform(inv, o, 1.0f);
                cd = xform(inv, d, 0.0f);
                ci = mk3(1.0f / cd.x, 1.0f / cd.y, 1.0f / cd.z);
                lbest = FLT_MAX;
                base = sp;
                in_blas = true;
                cur = as_u(in.d.x);
                continue;
            }
            // ---- BLAS leaf: test its triangles in order (kernel/intersection.h:63-112)
            if (COUNT) ctr.blas_leaves++;
            uint32_t first = cur & 0x0FFFFFFFu;
            uint32_t count = ((cur >> 28) & 7u) + 1u;
            if (count == 8u) {
                const uint2 big = S.big_leaves[first];
                first = big.x;
                count = big.y;
            }
            for (uint32_t i = 0; i < count; ++i) {
                if (COUNT) ctr.tris++;
                const float4 *tp = reinterpret_cast<const float4 *>(S.tris + first + i);
                const float4 ta = tp[0], tb = tp[1], tc = tp[2];
                float t, u, v;
                if (!tri_test(co, cd, ta, tb, tc, t_max, t, u, v)) continue;
                if (MODE == MODE_ANY) return true;
                if (t < lbest) {
                    lbest = t;
                    lrec = first + i;
                    lu = u;
                    lv = v;
                }
            }
            cur = kNoRef;
        }

        // ---- pop the next entry, leaving the instance / finishing when a level runs dry
        bool done = false;
        while (cur == kNoRef) {
            if (in_blas) {
                if (sp == base) {
                    in_blas = false;
                    if (lbest < FLT_MAX) {  // kernel/intersection.h:139-149, :190-194
                        const V3 lp = tri_point(S.tris, lrec, lu, lv);
                        const float4 *fp = reinterpret_cast<const float4 *>(S.fwd + cur_inst);
                        const M34 T = load_m34(fp[0], fp[1], fp[2]);
                        const V3 wp = xform(T, lp, 1.0f);
                        const float wt = dot(wp - o, d);
                        if (wt < best) {
                            best = wt;
                            hit.t = wt;
                            hit.u = lu;
                            hit.v = lv;
                            hit.rec = lrec;
                            hit.inst = cur_inst;
                        }
                    }
                    co = o;
                    cd = d;
                    ci = iw;
                    base = 0;
                    continue;
                }
            } else if (sp == 0) {
                done = true;
                break;
            }
            --sp;
            const uint2 e = stack_load(stk, sp);
            if (MODE == MODE_CLOSEST) {
                if (as_f(e.y) >= (in_blas ? lbest : best)) continue;
            }
            cur = e.x;
        }
        if (done) break;
    }
    if (best < 0.0f || best == FLT_MAX) {  // kernel/intersection.h:229-232
        hit.t = -1.0f;
        hit.inst = -1;
        return false;
    }
    return true;
}

}  // namespace ptd
