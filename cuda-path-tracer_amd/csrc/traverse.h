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

constexpr int kLdsDepth = 16;  // stack entries per lane held in LDS
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

// Per-lane traversal state.  begin() loads a ray and performs the root-box rejection;
// step() runs one round of "descend to a leaf, process it, pop the next entry" and returns
// true when the ray is finished.  Splitting the loop this way lets a persistent wave hand a
// finished lane a new ray while its neighbours are still traversing.
template <int MODE, bool COUNT>
struct Traversal {
    V3 o, d, iw;     // world-space ray and reciprocal direction
    V3 co, cd, ci;   // ray in the current space (world, or the current instance's object space)
    float t_max;
    float best;      // best world-space distance over finished instances
    float lbest;     // best object-space distance inside the current instance
    uint32_t lrec;
    float lu, lv;
    int cur_inst;
    int sp, base;
    bool in_blas;
    uint32_t cur;
    HitRec hit;
    bool found;      // MODE_ANY: an occluder was found
    TraceCounters ctr;

    // Returns false when the ray misses the scene's root box (kernel/intersection.h:162).
    PTD bool begin(const DScene &S, V3 origin, V3 dir, float tmax) {
        o = origin;
        d = dir;
        t_max = tmax;
        hit.t = -1.0f;
        hit.u = hit.v = 0.0f;
        hit.rec = 0;
        hit.inst = -1;
        found = false;
        if (COUNT) ctr = TraceCounters{0, 0, 0, 0};
        iw = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
        float t;
        if (!slab(o, iw, S.root_min[0], S.root_min[1], S.root_min[2], S.root_max[0], S.root_max[1], S.root_max[2], t_max, t)) return false;
        co = o;
        cd = d;
        ci = iw;
        best = FLT_MAX;
        lbest = FLT_MAX;
        lrec = 0;
        lu = lv = 0.0f;
        cur_inst = -1;
        sp = 0;
        base = 0;
        in_blas = false;
        cur = S.root_ref;
        return true;
    }

    // True when a closest hit exists (valid after the last step; kernel/intersection.h:229-232).
    PTD bool closest_valid() const { return !(best < 0.0f || best == FLT_MAX); }

    PTD bool step(const DScene &S, const LaneStack &stk, LaneEvents &ev) {
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
                return false;
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
                if (MODE == MODE_ANY) {
                    found = true;
                    return true;
                }
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
                return true;
            }
            --sp;
            const uint2 e = stack_load(stk, sp);
            if (MODE == MODE_CLOSEST) {
                if (as_f(e.y) >= (in_blas ? lbest : best)) continue;
            }
            cur = e.x;
        }
        return false;
    }
};

// Streams `n` rays through a persistent wave: the wave pulls chunks of kFetchChunk ray indices
// from a global cursor, and every lane that finishes its ray takes the next index of the
// chunk while its neighbours keep traversing (refill happens once at least kRefillMin lanes
// are idle, so the reload code runs for a meaningful share of the wave).
//   io.load(i, o, d, tmax)              ray i of the stream
//   io.done(i, trav, valid)             called once per ray by the lane that traced it
//   io.flush()                          called by all 64 lanes at every refill point
//   io.finish()                         called by all 64 lanes once, after the last flush
constexpr uint32_t kFetchChunk = 256;
constexpr int kRefillMin = 16;

template <int MODE, bool COUNT, class IO>
PTD void trace_stream(const DScene &S, IO &io, uint32_t n, uint32_t *cursor, const LaneStack &stk, LaneEvents &ev) {
    Traversal<MODE, COUNT> tr;
    bool active = false;
    uint32_t ray_index = 0;
    uint32_t pool_next = 0, pool_end = 0;  // wave-uniform
    bool exhausted = false;                // wave-uniform
    const uint32_t chunk = (n > kFetchChunk * 2u * gridDim.x) ? kFetchChunk : uint32_t(kWave);
    for (;;) {
        const uint64_t idle_mask = __ballot(!active);
        const int idle = __popcll(idle_mask);
        if (idle >= kRefillMin || idle == kWave) {
            io.flush();
            if (pool_next >= pool_end && !exhausted) {
                uint32_t b = 0;
                if ((threadIdx.x & 63) == 0) b = atomicAdd(cursor, chunk);
                b = __shfl(b, 0, 64);
                if (b >= n) {
                    exhausted = true;
                } else {
                    pool_next = b;
                    pool_end = (n - b < chunk) ? n : b + chunk;
                }
            }
            if (pool_next < pool_end) {
                const uint32_t prefix = __builtin_amdgcn_mbcnt_hi(uint32_t(idle_mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(idle_mask), 0u));
                const uint32_t mine = pool_next + prefix;
                if (!active && mine < pool_end) {
                    V3 o, d;
                    float tmax;
                    if (io.load(mine, o, d, tmax)) {  // false: an unused slot of the stream
                        ray_index = mine;
                        if (tr.begin(S, o, d, tmax)) active = true;
                        else io.done(ray_index, tr, false);
                    }
                }
                const uint32_t taken = uint32_t(idle);
                pool_next = (pool_end - pool_next < taken) ? pool_end : pool_next + taken;
            } else if (exhausted && idle == kWave) {
                break;
            }
        }
        if (active) {
            if (tr.step(S, stk, ev)) {
                active = false;
                io.done(ray_index, tr, MODE == MODE_ANY ? tr.found : tr.closest_valid());
            }
        }
    }
    io.flush();
    io.finish();
}

}  // namespace ptd
