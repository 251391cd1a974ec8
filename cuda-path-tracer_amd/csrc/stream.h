// Per-lane traversal state and the wave-level scheduler that streams rays through it.
//
// A busy lane is always in one of four situations:
//   wants_interior()  cur is an interior node: test its two child boxes
//   wants_triangle()  cur is a BLAS leaf: test the next triangle of it
//   wants_enter()     cur is a TLAS leaf: move the ray into the instance's space
//   wants_level()     cur is empty and the current level's stack ran dry: leave the instance
//                     (weighing its candidate hit against the best so far) or finish the ray
// plus the cheap case "cur is empty, the stack is not": pop.
//
// Running every code path for every lane on every round leaves most VALU lanes masked off
// (measured on bounce rays: 26 % of the lanes busy in the box tests).  So the wave counts the
// lanes that want each operation and runs an operation only while enough lanes want it; the
// others wait.  The box-test loop, by far the most frequent operation, holds nothing but the
// node fetch, the pop and the two slab tests; everything that changes the level of the
// traversal happens outside it.  Each lane still performs its own operations in the
// reference's order, so the results do not depend on the schedule.
#pragma once
#include "traverse.h"

// Trace modes 1 and 3 let any-hit searches skip boxes that start beyond t_max * (1 + margin).  The
// margin is there for the reference's shadow rays: they end 1e-4 before the sampled point of the light
// (path_trace.cu:221-227).  When such a ray grazes a flat light, the distance the triangle test computes to
// the light's own triangle is ill-conditioned (its error grows with 1 / sin of the grazing angle, 5e-3 at
// 0.006 degrees) and can land before t_max: the reference then counts the light as its own occluder.  The
// light's flat box is computed with another rounding and can land behind t_max, so a plain t_max cull
// pre-empts that decision.  Measured on 17.6 M captured shadow rays of five scenes: 12 800 flags differ from
// the reference's without a margin, 9 500 with 2^-16, none with 2^-12.
#ifndef PT_ANY_CULL_MARGIN
#define PT_ANY_CULL_MARGIN 2.44140625e-4f
#endif
// Diagnostic builds that ask which resource the traversal responds to (tools/sensitivity.sh):
//   -DPT_EXP_DOUBLE_LOAD  every 16-byte piece of a node of the box loop is requested a second time (one word,
//                         issued with the node's own loads, used after the box test; the repeats hit the L1)
//   -DPT_EXP_DOUBLE_VALU  the slab tests of a node are evaluated a second time on an opaque copy of the ray
// Results are unchanged in both.

namespace ptd {

// An accepted triangle closer than this does not end an any-hit search in trace modes 1 and 3 either: the
// distance the reference recomputes in world space for it may come out negative, so the search goes on
// by the reference's rule (Traversal::op_triangle).  Shadow rays start 1e-5 from the surface they leave.
constexpr float kAnyHitSettles = 1e-2f;

// Diagnostic build only (-DPT_TRACE_STAMPS): wave clock spent in the parts of the traversal loop, every
// stamp first waiting for all outstanding memory operations (so the waits are attributed to the part
// that issued them).  0 box round: votes + load issue + pop, 1 box round: wait for the node,
// 2 box round: box tests + push, 3 triangle rounds, 4 enter / level rounds, 5 refill, 6 rounds counted.
// Diagnostic build only (-DPT_NODE_HIST): visits per node record of the box loop (tools/node_hist.py).
#ifdef PT_NODE_HIST
constexpr uint32_t kNodeHistSize = 1u << 20;
__device__ unsigned int g_node_hist[kNodeHistSize];
#endif
#ifdef PT_TRACE_STAMPS
__device__ unsigned long long g_trace_stamps[8];
#define PT_TSTAMP(k)                                                                                         \
    do {                                                                                                     \
        unsigned long long t_;                                                                               \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
        tstamp_acc[k] += t_ - tstamp_last;                                                                   \
        tstamp_last = t_;                                                                                    \
    } while (0)
#define PT_TSTAMP_NOWAIT(k)                                                                                  \
    do {                                                                                                     \
        unsigned long long t_;                                                                               \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                             \
        tstamp_acc[k] += t_ - tstamp_last;                                                                   \
        tstamp_last = t_;                                                                                    \
    } while (0)
#else
#define PT_TSTAMP(k) do { } while (0)
#define PT_TSTAMP_NOWAIT(k) do { } while (0)
#endif

// WIDE (trace modes 2 and 3) walks 4-wide nodes (DNode4) instead of the binary ones: half as many
// dependent node visits, one 128-byte line for four children.  The boxes are the reference's and are
// tested with its arithmetic, so a box is entered here exactly when its own test passes in the binary
// tree; what differs is the visiting order, so exact ties (equal distances) may pick another
// triangle, and the winner's distance and barycentrics come from the same triangle test on the same
// record.  The wide layout also spends what the reference leaves on the table: inside an instance
// the search is bounded by the best hit of the instances before it (object-space and world-space
// distances are the same quantity, math/ray.h:19-21; the bound carries a margin for their different
// rounding, see op_level).
template <int MODE, bool COUNT, int DEPTH = kLdsDepth, bool WIDE = false>
struct Traversal {
    V3 o, d, iw;     // world-space ray and reciprocal direction
    V3 co, cd, ci;   // ray in the current space (world, or the current instance's object space)
    float t_max;
    float best;      // best world-space distance over finished instances
    float cbest;     // culling bound of the current level: `best` in the TLAS, the instance's own
                     // best object-space distance inside a BLAS; the ray's t_max in MODE_ANY
    uint32_t lrec, lprim;
    float lu, lv;
    uint32_t cur_mat, cur_type;  // of the instance being searched
    bool lfound;     // WIDE: the current instance has produced a candidate
    bool prov;       // WIDE, closest hit: `best` is still the winner's object-space distance (op_level)
    bool unit_dir;   // WIDE: |d| = 1 within rounding, so a world-space distance is also a bound on the ray parameter
    int cur_inst;
    int sp, base;
    bool in_blas;
    uint32_t cur;
    uint32_t tri_next, tri_left;  // progress inside the current BLAS leaf (tri_left == 0: not opened)
    HitRec hit;
    bool found;      // MODE_ANY: an occluder was found
    TraceCounters ctr;
    uint32_t tlas_interior;  // the TLAS's share of ctr.interior (counting variants only)

    // Returns false when the ray misses the scene's root box (kernel/intersection.h:162).
    PTD bool begin(const DScene &S, V3 origin, V3 dir, float tmax, bool root_tested = false) {
        o = origin;
        d = dir;
        t_max = tmax;
        hit.t = -1.0f;
        hit.u = hit.v = 0.0f;
        hit.rec = 0;
        hit.inst = -1;
        hit.prim = hit.mat = hit.type = 0xFFFFFFFFu;
        found = false;
        unit_dir = fabsf(dot(d, d) - 1.0f) <= 1e-4f;
        if (COUNT) {
            ctr = TraceCounters{0, 0, 0, 0};
            tlas_interior = 0;
        }
        iw = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
        float t;
        if (!root_tested && !slab(o, iw, S.root_min[0], S.root_min[1], S.root_min[2], S.root_max[0], S.root_max[1], S.root_max[2], t_max, t)) return false;
        co = o;
        cd = d;
        ci = iw;
        best = FLT_MAX;
        // ANY: the reference tests child boxes against "no hit yet" (FLT_MAX), never against the ray's
        // t_max (kernel/intersection.h:120-121, :209-210); only the scene's root box is (:162)
        cbest = (MODE == MODE_ANY && S.any_hit_culls_with_tmax) ? t_max + fabsf(t_max) * PT_ANY_CULL_MARGIN : FLT_MAX;
        lrec = 0;
        lu = lv = 0.0f;
        lfound = false;
        prov = false;
        cur_inst = -1;
        sp = 0;
        base = 0;
        in_blas = false;
        cur = WIDE ? S.root_ref4 : S.root_ref;
        tri_next = tri_left = 0;
        return true;
    }

    // True when a closest hit exists (valid once finished; kernel/intersection.h:229-232).
    PTD bool closest_valid() const { return !(best < 0.0f || best == FLT_MAX); }

    PTD bool wants_interior() const { return cur < kNoRef; }
    PTD bool wants_enter() const { return cur > kNoRef && !in_blas; }
    // cur keeps naming the leaf until its last triangle is done, so "cur is empty" alone says that no leaf is waiting
    PTD bool wants_triangle() const { return in_blas && cur > kNoRef; }
    PTD bool may_pop() const { return cur == kNoRef && sp != base; }
    PTD bool wants_pop() const { return may_pop() && sp <= DEPTH; }
    PTD bool wants_deep_pop() const { return may_pop() && sp > DEPTH; }  // entry is in the global spill area
    PTD bool wants_level() const { return cur == kNoRef && sp == base; }

    // One pop (kernel/intersection.h:54-60): the entry is dropped when its box starts beyond
    // the current bound; then cur stays empty and the lane pops again next round.
    // pop() only touches LDS, so the box-test loop never waits for a global load in it.
    template <class STK>
    PTD void pop(const STK &stk) {
        --sp;
        const unsigned long long v = stk.lds[sp * kWave];
        if (MODE == MODE_ANY || as_f(uint32_t(v >> 32)) < cbest) cur = uint32_t(v);
    }
    template <class STK>
    PTD void deep_pop(const STK &stk) {
        --sp;
        const uint2 e = stack_load(stk, sp);
        if (MODE == MODE_ANY || as_f(e.y) < cbest) cur = e.x;
    }

    // One interior node: both child boxes, near child next, far child pushed
    // (kernel/intersection.h:114-135, :203-226).
    template <class STK>
    PTD void op_interior(const STK &stk, const LaneEvents &ev, float4 na, float4 nb, float4 nc, float4 nd) {
        if (COUNT) {
            ctr.interior++;
            if (!in_blas) tlas_interior++;
        }
        float tl, tr;
        const bool hl = slab(co, ci, na.x, na.y, na.z, nb.x, nb.y, nb.z, cbest, tl);
        const bool hr = slab(co, ci, nc.x, nc.y, nc.z, nd.x, nd.y, nd.z, cbest, tr);
#ifdef PT_EXP_DOUBLE_VALU
        {
            V3 co2 = co;
            asm volatile("" : "+v"(co2.x), "+v"(co2.y), "+v"(co2.z));
            float t2l, t2r;
            const bool h2l = slab(co2, ci, na.x, na.y, na.z, nb.x, nb.y, nb.z, cbest, t2l);
            const bool h2r = slab(co2, ci, nc.x, nc.y, nc.z, nd.x, nd.y, nd.z, cbest, t2r);
            asm volatile("" ::"v"(t2l), "v"(t2r), "v"(int(h2l) + int(h2r)));
        }
#endif
        const uint32_t lref = as_u(na.w), rref = as_u(nb.w);
        const int depth = sp - base;
        const bool both = hl && hr;
        const bool left_first = tl < tr;
        // Usual case, decided for the wave as a whole: no lane is near the reference's 64-entry limit
        // or the end of the LDS part of the stack.  Then the push is straight-line code: the far
        // child is written above the top of the stack unconditionally and only counted when both
        // children were hit.
        const bool rare = depth >= kRefStack - 1 || sp >= DEPTH;
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(rare) == 0, 1)) {
            const uint32_t far_ref = left_first ? rref : lref;
            const float far_t = left_first ? tr : tl;
            stk.lds[sp * kWave] = ((unsigned long long)as_u(far_t) << 32) | far_ref;
            sp += both ? 1 : 0;
            const uint32_t near_ref = both ? (left_first ? lref : rref) : (hl ? lref : rref);
            cur = (hl || hr) ? near_ref : kNoRef;
            return;
        }
        if (__builtin_expect(depth < kRefStack - 1, 1)) {
            if (both) {
                stack_store(stk, sp, left_first ? rref : lref, left_first ? tr : tl, ev);
                ++sp;
                cur = left_first ? lref : rref;
            } else {
                cur = hl ? lref : (hr ? rref : kNoRef);
            }
        } else {
            // the reference's 64-entry stack is (nearly) full: reproduce which pushes it drops
            // (utils/stack.h:8-11; it pushes the far child, then the near one, then pops)
            if (both) {
                if (depth == kRefStack - 1) {
                    stack_store(stk, sp, left_first ? rref : lref, left_first ? tr : tl, ev);
                    ++sp;
                    ev.drop();
                } else {
                    ev.drop(2);
                }
                cur = kNoRef;
            } else if (hl || hr) {
                if (depth < kRefStack) {
                    cur = hl ? lref : rref;
                } else {
                    ev.drop();
                    cur = kNoRef;
                }
            } else {
                cur = kNoRef;
            }
        }
    }

    // One 4-wide node (trace modes 2 and 3): four slab tests with the reference's arithmetic and
    // acceptance rule (math/intersection.h:9-35) on the reference's own boxes, so each box decides as
    // it does in the binary tree; the children that are hit are ordered by entry distance, the nearest
    // is visited next and the others are pushed, farthest first.
    static PTD void order(uint32_t &ka, uint32_t &ra, uint32_t &kb, uint32_t &rb) {
        const bool swap = kb < ka;
        const uint32_t lo = swap ? kb : ka, hi = swap ? ka : kb;
        const uint32_t rlo = swap ? rb : ra, rhi = swap ? ra : rb;
        ka = lo; kb = hi; ra = rlo; rb = rhi;
    }
    static PTD float comp(const float4 &v, int k) { return k == 0 ? v.x : k == 1 ? v.y : k == 2 ? v.z : v.w; }
    template <class STK>
    PTD void op_interior4(const STK &stk, const LaneEvents &ev, const float4 (&q)[7]) {
        if (COUNT) {
            ctr.interior++;
            if (!in_blas) tlas_interior++;
        }
        constexpr uint32_t kMiss = 0x7F800000u;  // +inf: sorts behind every hit
        uint32_t key[4], ref[4] = {as_u(q[6].x), as_u(q[6].y), as_u(q[6].z), as_u(q[6].w)};
        int n = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float tn;
            const bool hit = slab(co, ci, comp(q[0], k), comp(q[2], k), comp(q[4], k), comp(q[1], k), comp(q[3], k), comp(q[5], k), cbest, tn);
            n += hit ? 1 : 0;
            // the entry distance, clamped at the origin (as an integer: the bits of a negative float are a
            // negative integer) and rounded down to make room for the slot number that makes the keys distinct
            const int bits = int(as_u(tn));
            key[k] = hit ? ((uint32_t(bits < 0 ? 0 : bits) & ~3u) | uint32_t(k)) : (kMiss | uint32_t(k));
        }
#ifdef PT_EXP_DOUBLE_VALU
        {
            V3 co2 = co;
            asm volatile("" : "+v"(co2.x), "+v"(co2.y), "+v"(co2.z));
            int n2 = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float tn;
                n2 += slab(co2, ci, comp(q[0], k), comp(q[2], k), comp(q[4], k), comp(q[1], k), comp(q[3], k), comp(q[5], k), cbest, tn) ? 1 : 0;
                asm volatile("" ::"v"(tn));
            }
            asm volatile("" ::"v"(n2));
        }
#endif
        order(key[0], ref[0], key[1], ref[1]);
        order(key[2], ref[2], key[3], ref[3]);
        order(key[0], ref[0], key[2], ref[2]);
        order(key[1], ref[1], key[3], ref[3]);
        order(key[1], ref[1], key[2], ref[2]);
        cur = n > 0 ? ref[0] : kNoRef;
        // Usual case, decided for the wave: every lane has room for three more entries in the LDS part
        // of its stack.  Then all three candidates are stored without a branch: child j of the sorted
        // list belongs at sp + n - 1 - j, and the ones that were not hit land above the new top.
        const bool rare = sp + 3 >= DEPTH;
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(rare) == 0, 1)) {
#pragma unroll
            for (int j = 1; j < 4; ++j) stk.lds[(sp + ((n - 1 - j) & 3)) * kWave] = ((unsigned long long)key[j] << 32) | ref[j];
            sp += n > 1 ? n - 1 : 0;
            return;
        }
#pragma unroll
        for (int j = 3; j >= 1; --j) {
            if (j < n) {
                if (sp < DEPTH + kSpillDepth) {
                    stack_store(stk, sp, ref[j], as_f(key[j]), ev);
                    ++sp;
                } else {
                    ev.drop();  // the physical stack is full: the subtree is lost (counted in pt_stats.stack_drops)
                }
            }
        }
    }

    // TLAS leaf: enter the instance (kernel/intersection.h:183-188, :37-50).
    PTD void op_enter(const DScene &S) {
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
        if (MODE == MODE_CLOSEST) cbest = WIDE ? wide_bound() : FLT_MAX;
        lfound = false;
        base = sp;
        in_blas = true;
        cur = WIDE ? S.inst_root4[cur_inst] : as_u(in.d.x);
        cur_mat = as_u(in.d.y);
        cur_type = as_u(in.d.z);
        tri_left = 0;
    }

    // The next triangle of the current leaf (kernel/intersection.h:63-112).  Returns true in
    // MODE_ANY when the ray is occluded (finished).
    PTD bool op_triangle(const DScene &S) {
        if (tri_left == 0) {  // open the leaf
            if (COUNT) ctr.blas_leaves++;
            const uint32_t leaf = cur;
            tri_next = leaf & 0x0FFFFFFFu;
            tri_left = ((leaf >> 28) & 7u) + 1u;
            if (tri_left == 8u) {
                const uint2 big = S.big_leaves[tri_next];
                tri_next = big.x;
                tri_left = big.y;
            }
        }
        if (COUNT) ctr.tris++;
        const float4 *tp = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(S.tris) + size_t(tri_next) * sizeof(DTri));
        const float4 ta = tp[0], tb = tp[1], tc = tp[2];
        float t, u, v;
        const bool ok = tri_test(co, cd, ta, tb, tc, t_max, t, u, v);
        if (MODE == MODE_ANY) {
            if (ok) {
                // Trace modes 1 and 3: an occluder at a distance that no rounding can turn negative settles it.
                // (What they give up: a later triangle of the same instance that touches the origin could
                // still have turned the reference's answer into "free".)
                if (S.any_hit_culls_with_tmax && t > kAnyHitSettles) {
                    found = true;
                    return true;
                }
                // The reference's any-hit search only leaves the leaf's loop here (kernel/intersection.h:101-109):
                // it goes on through the rest of the instance, testing boxes against the time of the latest
                // accepted triangle, every accepted triangle replaces the one before, and the instance reports
                // a hit only if the world-space distance recomputed from the last one is not negative (:139-149).
                // A triangle touching the ray's origin can fail that, and the ray then counts as unoccluded.
                cbest = t;
                lfound = true;
                lrec = tri_next;
                lu = u;
                lv = v;
                tri_left = 0;
                cur = kNoRef;
                return false;
            }
        } else if (ok && t < cbest) {
            cbest = t;
            lfound = true;
            lrec = tri_next;
            lprim = as_u(tc.y);
            lu = u;
            lv = v;
        }
        ++tri_next;
        --tri_left;
        if (tri_left == 0) cur = kNoRef;
        return false;
    }

    // The current level has run dry.  Inside an instance: its candidate, if any, competes by
    // world-space distance with the best of the other instances, and the ray returns to the
    // TLAS (kernel/intersection.h:139-149, :190-194).  In the TLAS: the ray is finished.
    // Returns true when finished.
    //
    // The 4-wide layout bounds the search inside an instance, and the pops of the top level, by the best
    // distance so far, which the reference does not (it starts every instance from "no hit" and compares
    // the candidates by the distance it recomputes in world space from each).  The two distances are the
    // same quantity but not the same bits (2e-5 apart for a strongly scaled instance), so the bound
    // carries a margin, and a candidate that survives it still competes by the recomputed distance, as
    // in the reference.  What the layout saves: the first candidate of a ray has nothing to compete with,
    // so its instance is left without touching memory and its recomputed distance (two dependent
    // gathers) is taken only when a second candidate turns up or when the ray finishes - there always
    // with EXACT_T, otherwise only for a hit so close to the origin that the recomputed distance could
    // come out negative, which the reference reports as a miss.
    static constexpr float kWideBoundMargin = 2.44140625e-4f;  // 2^-12
    // `best` is a distance along the ray, dot(p - o, d) = t * |d|^2, and bounds the ray parameter t only for unit
    // directions (pt_ray does not require them): any other ray searches every instance unbounded, as the
    // reference does.
    PTD float wide_bound() const { return unit_dir ? best + fabsf(best) * kWideBoundMargin : __builtin_inff(); }  // FLT_MAX stays "no bound" (+inf)
    PTD float world_distance(const DScene &S, uint32_t rec, float u, float v, int inst) const {
        const V3 lp = tri_point(S.tris, rec, u, v);
        const float4 *fp = reinterpret_cast<const float4 *>(S.fwd + inst);
        const M34 T = load_m34(fp[0], fp[1], fp[2]);
        return dot(xform(T, lp, 1.0f) - o, d);
    }
    PTD void take_candidate(float t) {
        best = t;
        hit.t = t;
        hit.u = lu;
        hit.v = lv;
        hit.rec = lrec;
        hit.inst = cur_inst;
        hit.prim = lprim;
        hit.mat = cur_mat;
        hit.type = cur_type;
    }
    template <bool EXACT_T = true>
    PTD bool op_level(const DScene &S) {
        if (in_blas) {
            if (MODE == MODE_ANY && lfound) {  // the first instance that reports a hit ends the search (kernel/intersection.h:196-200, :229-232)
                const float wt = world_distance(S, lrec, lu, lv, cur_inst);
                found = !(wt < 0.0f || wt == FLT_MAX);
                return true;
            }
            if (MODE == MODE_CLOSEST) {
                if constexpr (WIDE) {
                    if (lfound) {
                        if (hit.inst < 0) {
                            take_candidate(cbest);
                            prov = true;
                        } else {
                            const float wt = world_distance(S, lrec, lu, lv, cur_inst);
                            if (prov) {
                                best = world_distance(S, hit.rec, hit.u, hit.v, hit.inst);
                                hit.t = best;
                                prov = false;
                            }
                            if (wt < best) take_candidate(wt);
                        }
                    }
                    cbest = wide_bound();
                } else {
                    if (cbest < FLT_MAX) {
                        const float wt = world_distance(S, lrec, lu, lv, cur_inst);
                        if (wt < best) take_candidate(wt);
                    }
                    cbest = best;
                }
            }
            in_blas = false;
            co = o;
            cd = d;
            ci = iw;
            base = 0;
        }
        if (sp != 0) return false;
        if constexpr (MODE == MODE_CLOSEST && WIDE) {
            if (prov && (EXACT_T || best <= kAnyHitSettles)) {
                best = world_distance(S, hit.rec, hit.u, hit.v, hit.inst);
                hit.t = best;
            }
        }
        return true;
    }
};

// Streams `n` rays through a persistent wave: the wave pulls chunks of ray indices (fetch_rays)
// from a global cursor, and every lane that finishes its ray takes the next index of the
// chunk while its neighbours keep traversing (refill happens once at least kRefillMin lanes
// are idle, so the reload code runs for a meaningful share of the wave).
//   IO::kExactDistance                  done() reads the hit's distance (Traversal::op_level)
//   IO::kStageCull                      rays that miss the scene's box are answered by io.miss() when they are read
//   io.load(i, o, d, tmax, tag)         ray i of the stream; tag names the ray from here on (i itself, or a path slot)
//   io.done(tag, trav, valid)           called once per ray by the lane that traced it
//   io.miss(tag, tmax)                  kStageCull: the ray missed the scene's box (in place of done())
//   io.flush()                          called by all 64 lanes at every refill point
//   io.finish()                         called by all 64 lanes once, after the last flush
constexpr uint32_t kFetchChunk = 256;
constexpr uint32_t kNoRayBits = 0x7FC0DEADu;  // t_max bit pattern (a NaN no computation produces) of a staged entry that holds no ray
#ifndef PT_REFILL_MIN
#define PT_REFILL_MIN 12
#endif
#ifndef PT_VOTE_INT
#define PT_VOTE_INT 24   // lanes that keep the box-test loop running
#endif
#ifndef PT_VOTE_TRI
#define PT_VOTE_TRI 12   // lanes that keep the triangle loop running
#endif
#ifndef PT_DRAIN_ALL_OPS
#define PT_DRAIN_ALL_OPS 1
#endif
#ifndef PT_VOTE_RARE
#define PT_VOTE_RARE 8   // lanes that make an instance enter / level change round worth running
#endif
#ifndef PT_VOTE_LEVEL
#define PT_VOTE_LEVEL 12  // lanes that make a level change round worth running (8 / 12 / 16: 1.56 / 1.53 / 1.54 ms on the bench batch)
#endif
constexpr int kRefillMin = PT_REFILL_MIN;

// Lanes of the wave for which `pred` holds (the predicate's lane mask is counted directly).
PTD int count_lanes(bool pred) { return __builtin_popcountll(__builtin_amdgcn_ballot_w64(pred)); }

// Where a wave gets its rays from.  The stream is cut into chunks of `chunk` rays; chunk number
// t * groups + g is handed out by ticket t of cursor g.  With one cursor the chunks must be large
// for long streams, because the cursor is one address and same-address atomics serialise at
// ~12 ns each; a wave then holds hundreds of neighbouring (similarly expensive) rays when the
// others run dry.  With several cursors on separate cache lines the chunks are one wave's worth:
// each cursor serves the waves whose number is congruent to g, its chunks are interleaved with the
// other cursors' over the whole stream, and a wave whose cursor has run dry goes on to the next.
#ifndef PT_FEED_CHUNK
#define PT_FEED_CHUNK 64
#endif
constexpr uint32_t kCursorStride = 64;  // words between cursors: 256 bytes
struct WorkFeed {
    uint32_t *cursors;
    uint32_t groups, chunk;
};
PTD WorkFeed make_feed(uint32_t *cursors, uint32_t groups, uint32_t n) {
    if (groups > 1) return WorkFeed{cursors, groups, uint32_t(PT_FEED_CHUNK)};
    const uint32_t chunk = (n > kFetchChunk * 8u * gridDim.x) ? 4u * kFetchChunk : (n > kFetchChunk * 2u * gridDim.x) ? kFetchChunk : uint32_t(kWave);
    return WorkFeed{cursors, 1u, chunk};
}
// One work fetch: true and [base, end) when the wave got rays.  `dry` counts the cursors the wave
// has found empty (wave-uniform; the feed is exhausted once dry == groups).  A fetch probes one
// cursor, so a wave whose other lanes are still traversing waits for one round trip at most.
PTD bool fetch_rays(const WorkFeed &feed, uint32_t n, uint32_t &dry, uint32_t &base, uint32_t &end) {
    const uint32_t g = (blockIdx.x + dry) % feed.groups;
    uint32_t t = 0;
    if ((threadIdx.x & 63u) == 0) t = atomicAdd(feed.cursors + size_t(g) * kCursorStride, 1u);
    t = __shfl(t, 0, 64);
    const unsigned long long first = ((unsigned long long)t * feed.groups + g) * feed.chunk;
    if (first >= n) {
        ++dry;
        return false;
    }
    base = uint32_t(first);
    end = (n - base < feed.chunk) ? n : base + feed.chunk;
    return true;
}

template <int MODE, bool COUNT, bool WIDE = false, class IO, class STK>
PTD void trace_stream(const DScene &S, IO &io, uint32_t n, const WorkFeed &feed, const STK &stk, const LaneEvents &ev) {
    Traversal<MODE, COUNT, STK::kDepth, WIDE> tr;
    tr.cur = kNoRef;
    tr.in_blas = false;
    tr.sp = tr.base = 0;
    bool active = false;
    uint32_t ray_tag = 0;
    uint32_t pool_next = 0, pool_end = 0;  // wave-uniform
    uint32_t dry = 0;                      // wave-uniform: cursors found empty
    bool exhausted = false;                // wave-uniform
#ifdef PT_TRACE_STAMPS
    unsigned long long tstamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tstamp_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tstamp_last)::"memory");
#endif
    const char *node_base = WIDE ? reinterpret_cast<const char *>(S.nodes4) : reinterpret_cast<const char *>(S.nodes);
    constexpr int kNodeShift = WIDE ? 7 : 6;  // 128-byte wide nodes, 64-byte binary ones
    constexpr int kNodeRegs = WIDE ? 7 : 4;   // 16-byte words of a node that the box tests read
#ifdef PT_EXP_DOUBLE_LOAD
    float dup_word[kNodeRegs];
#endif
    // Rays are read 64 at a time, by all lanes together, into the wave's LDS, and handed to lanes as
    // they fall idle: the wave then waits for a gather from path state (thousands of cycles when the
    // state does not fit the caches) once per 64 rays instead of at every refill.
    uint32_t stage_next = 0, stage_end = 0;  // wave-uniform: staged entries [stage_next, stage_end) wait for a lane
    const uint32_t lane_id = threadIdx.x & 63u;
    for (;;) {
        int idle = count_lanes(!active);
        if (idle >= kRefillMin || idle == kWave) {
            io.flush();
            for (;;) {
                if (stage_next == stage_end) {  // restock
                    if (pool_next >= pool_end && !exhausted) {
                        if (!fetch_rays(feed, n, dry, pool_next, pool_end)) exhausted = dry >= feed.groups;
                    }
                    if (pool_next >= pool_end) break;
                    const uint32_t count = (pool_end - pool_next < uint32_t(kWave)) ? pool_end - pool_next : uint32_t(kWave);
                    V3 o = mk3(0.0f, 0.0f, 0.0f), d = o;
                    float tmax = 0.0f;
                    uint32_t tag = 0;
                    if constexpr (IO::kStageCull) {
                        // The test against the scene's box (kernel/intersection.h:162) runs here, for 64 rays at once, and
                        // only the rays that pass are staged: a stream in which many rays miss the scene (camera rays)
                        // then costs the lanes that refill no hand-outs that come to nothing.
                        bool enters = false;
                        if (lane_id < count && io.load(pool_next + lane_id, o, d, tmax, tag)) {
                            float t;
                            enters = slab(o, mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z), S.root_min[0], S.root_min[1], S.root_min[2], S.root_max[0], S.root_max[1],
                                          S.root_max[2], tmax, t);
                            if (!enters) io.miss(tag, tmax);
                        }
                        const uint64_t enter_mask = __ballot(enters);
                        const uint32_t at = __builtin_amdgcn_mbcnt_hi(uint32_t(enter_mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(enter_mask), 0u));
                        if (enters) {
                            stk.stage[at] = v4f{o.x, o.y, o.z, tmax};
                            stk.stage[kWave + at] = v4f{d.x, d.y, d.z, as_f(tag)};
                        }
                        stage_end = uint32_t(__popcll(enter_mask));
                    } else {
                        if (lane_id < count && !io.load(pool_next + lane_id, o, d, tmax, tag)) tmax = as_f(kNoRayBits);  // an unused slot of the stream
                        stk.stage[lane_id] = v4f{o.x, o.y, o.z, tmax};
                        stk.stage[kWave + lane_id] = v4f{d.x, d.y, d.z, as_f(tag)};
                        stage_end = count;
                    }
                    __builtin_amdgcn_wave_barrier();
                    stage_next = 0;
                    pool_next += count;
                }
                const uint64_t idle_mask = __ballot(!active);
                const uint32_t prefix = __builtin_amdgcn_mbcnt_hi(uint32_t(idle_mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(idle_mask), 0u));
                const uint32_t entry = stage_next + prefix;
                if (!active && entry < stage_end) {
                    PT_PHASE(ev, 5);
                    const v4f so = stk.stage[entry], sd = stk.stage[kWave + entry];
                    if (IO::kStageCull || as_u(so.w) != kNoRayBits) {
                        ray_tag = as_u(sd.w);
                        if (tr.begin(S, mk3(so.x, so.y, so.z), mk3(sd.x, sd.y, sd.z), so.w, IO::kStageCull)) active = true;
                        else io.done(ray_tag, tr, false);
                    }
                }
                const uint32_t wanted = uint32_t(__popcll(idle_mask));
                stage_next = (stage_end - stage_next < wanted) ? stage_end : stage_next + wanted;
                if (stage_next < stage_end) break;  // every idle lane was offered a ray
                idle = count_lanes(!active);
                if (idle < kRefillMin) break;
            }
            if (exhausted && pool_next >= pool_end && stage_next == stage_end && count_lanes(!active) == kWave) break;
        }
        const bool can_refill = !(exhausted && pool_next >= pool_end && stage_next == stage_end);  // wave-uniform
        idle = count_lanes(!active);
        PT_PHASE(ev, 7);
        PT_TSTAMP(5);

        // ---- box-test rounds, while enough lanes take part.  A lane without a node pops before the vote and takes part
        // in the round (popping between the issue of the node loads and their use, where the LDS round trip would be
        // hidden, makes the lane join one round later: 5-6 % slower on the bench batch)
        const unsigned long long active_mask = __builtin_amdgcn_ballot_w64(active);
        // a wave that can no longer refill keeps the loop going for half of the lanes it has left
        const int vote_int = (PT_DRAIN_ALL_OPS && !can_refill) ? max(1, (__builtin_popcountll(active_mask) + 1) >> 1) : PT_VOTE_INT;
        for (;;) {
            if (active && tr.wants_pop()) {
                PT_PHASE(ev, 3);
                tr.pop(stk);
            }
            const unsigned long long int_mask = __builtin_amdgcn_ballot_w64(tr.wants_interior()) & active_mask;
            const bool w_int = active && tr.wants_interior();
            const bool go = __builtin_popcountll(int_mask) >= vote_int;
            float4 nq[kNodeRegs];
            if (go && w_int) {
                const float4 *np = reinterpret_cast<const float4 *>(node_base + (size_t(tr.cur) << kNodeShift));
#pragma unroll
                for (int r = 0; r < kNodeRegs; ++r) nq[r] = np[r];
#ifdef PT_EXP_DOUBLE_LOAD
                // one more request per 16-byte piece, issued with the others and consumed only after the node has been
                // used, so that the round's chain of dependent waits stays what it was
#pragma unroll
                for (int r = 0; r < kNodeRegs; ++r) asm volatile("global_load_dword %0, %1, off offset:4" : "=v"(dup_word[r]) : "v"(np + r));  // a plain cached load the compiler cannot merge
#endif
            }
            PT_TSTAMP_NOWAIT(0);
            if (!go) break;
            PT_TSTAMP(1);
            if (w_int) {
                PT_PHASE(ev, 0);
#ifdef PT_NODE_HIST
                atomicAdd(&g_node_hist[tr.cur < kNodeHistSize ? tr.cur : kNodeHistSize - 1], 1u);
#endif
                if constexpr (WIDE) tr.op_interior4(stk, ev, nq);
                else tr.op_interior(stk, ev, nq[0], nq[1], nq[2], nq[3]);
#ifdef PT_EXP_DOUBLE_LOAD
#pragma unroll
                for (int r = 0; r < kNodeRegs; ++r) asm volatile("s_waitcnt vmcnt(0)" ::"v"(dup_word[r]));
#endif
            }
            PT_TSTAMP(2);
#ifdef PT_TRACE_STAMPS
            tstamp_acc[6]++;
#endif
        }

        // ---- the other operations, each only when enough lanes want it (or nothing else would run)
        const bool w_int = active && tr.wants_interior();
        const bool w_ent = active && tr.wants_enter();
        const bool w_lvl = active && tr.wants_level();
        const bool w_deep = active && tr.wants_deep_pop();
        const int n_int = count_lanes(w_int);
        int n_tri = count_lanes(active && tr.wants_triangle());
        const int n_ent = count_lanes(w_ent);
        const int n_lvl = count_lanes(w_lvl || w_deep);
        const int n_max = max(max(n_int, n_tri), max(n_ent, n_lvl));
        if (n_max == 0) continue;  // only lanes that are still popping
        if (n_int >= vote_int) continue;  // pops brought the box-test loop back above its threshold
        // Once the wave has no rays left to hand out, idle lanes stay idle whatever the others do:
        // every operation some lane waits for then runs in every round, so that no ray's last steps
        // are held back by a vote.
        const bool drain = PT_DRAIN_ALL_OPS && !can_refill;
        if (n_int == n_max || (drain && n_int > 0)) {  // fewer than PT_VOTE_INT lanes, but still the largest group
            if (w_int) {
                PT_PHASE(ev, 0);
#ifdef PT_NODE_HIST
                atomicAdd(&g_node_hist[tr.cur < kNodeHistSize ? tr.cur : kNodeHistSize - 1], 1u);
#endif
                const float4 *np = reinterpret_cast<const float4 *>(node_base + (size_t(tr.cur) << kNodeShift));
                float4 nq[kNodeRegs];
#pragma unroll
                for (int r = 0; r < kNodeRegs; ++r) nq[r] = np[r];
                if constexpr (WIDE) tr.op_interior4(stk, ev, nq);
                else tr.op_interior(stk, ev, nq[0], nq[1], nq[2], nq[3]);
            }
        }
        PT_TSTAMP(4);
        if (n_tri >= PT_VOTE_TRI || n_tri == n_max || (drain && n_tri > 0)) {
            do {
                if (active && tr.wants_triangle()) {
                    PT_PHASE(ev, 2);
                    if (tr.op_triangle(S)) {  // MODE_ANY only: occluded
                        active = false;
                        io.done(ray_tag, tr, true);
                    }
                }
                n_tri = count_lanes(active && tr.wants_triangle());
#ifdef PT_TRACE_STAMPS
                tstamp_acc[7]++;
#endif
            } while (n_tri >= PT_VOTE_TRI);
        }
        PT_TSTAMP(3);
        if (n_ent > 0 && (n_ent >= PT_VOTE_RARE || n_ent == n_max || drain)) {
            if (w_ent) {
                PT_PHASE(ev, 1);
                tr.op_enter(S);
            }
        }
        if (n_lvl > 0 && (n_lvl >= PT_VOTE_LEVEL || n_lvl == n_max || drain || (can_refill && n_lvl + idle >= kRefillMin))) {
            if (w_deep) tr.deep_pop(stk);
            if (w_lvl) {
                PT_PHASE(ev, 4);
                if (tr.template op_level<IO::kExactDistance>(S)) {
                    active = false;
                    io.done(ray_tag, tr, MODE == MODE_ANY ? tr.found : tr.closest_valid());
                }
            }
        }
    }
    io.flush();
    io.finish();
#ifdef PT_TRACE_STAMPS
    PT_TSTAMP(4);
    if ((threadIdx.x & 63u) == 0)
        for (int k = 0; k < 8; ++k) atomicAdd(&g_trace_stamps[k], tstamp_acc[k]);
#endif
}

}  // namespace ptd
