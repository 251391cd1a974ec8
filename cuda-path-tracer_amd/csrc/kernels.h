// Wavefront path-tracing kernels for gfx950 (wave64).  One sample per pixel per iteration:
//   k_generate -> [ k_extend -> k_shade -> k_shadow ] x max_depth -> k_film (-> k_tonemap)
// Queues hold path slots; appends are wave-aggregated (one atomic per wave and queue, ballot +
// mbcnt prefix), and every queue consumer is a persistent grid whose waves fetch 64-entry
// chunks through a global cursor, so no launch depends on a host-side read of a counter.
//
// Reference: kernel/integrators/path_trace.cu:76-434 (GenerateRaysKernel, IntersectClosestKernel,
// MaterialShadeKernel<M>, EmissiveHitKernel, FinalGather, PostprocessAndUpload).
#pragma once
#include "shade.h"
#include "stream.h"

namespace ptd {

#ifndef PT_STAGE_FLUSH
#define PT_STAGE_FLUSH 64
#endif
constexpr int kStageFlush = PT_STAGE_FLUSH;       // a class's staged slots are written out this many at a time
static_assert(kStageFlush >= 64 && kStageFlush % 64 == 0, "whole waves are flushed");
constexpr int kStageCap = kStageFlush + 64;    // per-class staging entries per wave: < kStageFlush carried over + <= 64 new
constexpr int kNumClasses = 5;  // 0 = emissive hit or miss, 1 lambertian, 2 mirror, 3 metallic-roughness, 4 dielectric

// Per-bounce counters, 64 B, zeroed once per iteration for all bounces.
struct BounceCounters {
    uint32_t ext_count;               // extension rays entering this bounce
    uint32_t class_count[kNumClasses];
    uint32_t shadow_count;
    uint32_t cur_extend, cur_shade, cur_shadow;  // work-fetch cursors
    uint32_t _pad[6];
};

struct DevStats {
    uint32_t spills, drops, nan_pixels, _pad;
};

// Per-path state, one array per field, indexed by path slot.  (One 128-byte record per slot was
// measured too: k_generate's partial-line writes made it 4x slower and k_shade gained nothing.)
constexpr uint32_t kThroughputNotFinite = 0x80000000u;  // flag in the bounce word of a path's ray

struct PathState {
    float4 *ray_o;     // origin.xyz, last_pdf
    float4 *ray_d;     // direction.xyz, bits(remaining_bounces | kThroughputNotFinite)
    float4 *thr;       // throughput.xyz
    float4 *accum;     // accumulated radiance.xyz
    float4 *last_pos;  // last surface position.xyz
    uint2 *rng;        // PCG state
    float4 *hit;       // bits(instance), u, v, bits(triangle record) of the last closest hit
    float4 *sh_o;      // shadow ray origin.xyz, tmax        (indexed by path slot, like everything above)
    float4 *sh_d;      // shadow ray direction.xyz
    float4 *sh_c;      // contribution added when unoccluded
    uint32_t *sh_q;    // path slots with a pending shadow ray
    uint32_t *ext_q[2];
    uint32_t *class_q[kNumClasses];
    uint32_t capacity;  // path slots = samples_in_flight * slots_per_sample
    uint32_t slots_per_sample;  // pixels of the render window padded to 8x8 tiles
    int win_x0, win_y0, win_w, win_h;  // the render window (pt_set_render_window): the pixels this context traces
};

// One atomic per wave: lanes with `pred` receive consecutive slots of the queue behind `counter`.
// Must be reached by all 64 lanes of the wave.
PTD uint32_t wave_append(uint32_t *counter, bool pred) {
    const uint64_t mask = __ballot(pred);
    if (mask == 0) return 0;
    const uint32_t prefix = __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u));
    const int leader = __ffsll((unsigned long long)mask) - 1;
    uint32_t base = 0;
    if (int(threadIdx.x & 63) == leader) base = atomicAdd(counter, uint32_t(__popcll(mask)));
    base = __shfl(base, leader, 64);
    return base + prefix;
}

// Fetches the next 64-entry chunk index for the calling wave (wave-uniform result).
PTD uint32_t wave_fetch(uint32_t *cursor, uint32_t amount) {
    uint32_t v = 0;
    if ((threadIdx.x & 63) == 0) v = atomicAdd(cursor, amount);
    return __shfl(v, 0, 64);
}

#ifdef PT_PHASE_PROF
__device__ unsigned long long g_phase_prof[16];
#endif
#ifdef PT_WAVE_TIMES
// Diagnostic build: start and end of every wave of the last closest-hit query (100 MHz clock).
__device__ unsigned long long g_wave_times[2 * 8192];
#define PT_WAVE_TIME(k) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_wave_times[2 * blockIdx.x + (k)] = wall_clock64(); } while (0)
#else
#define PT_WAVE_TIME(k) do { } while (0)
#endif

PTD void flush_events(DevStats *stats, const LaneEvents &ev) {
    if ((threadIdx.x & 63u) == 0) {  // the wave's own adds to these two words come first in its program order
        const uint32_t spills = atomicAdd(ev.wave, 0u), drops = atomicAdd(ev.wave + 1, 0u);
        if (spills) atomicAdd(&stats->spills, spills);
        if (drops) atomicAdd(&stats->drops, drops);
    }
#ifdef PT_PHASE_PROF
    for (int k = 0; k < 8; ++k) {
        if (ev.lane_rounds[k]) atomicAdd(&g_phase_prof[k], (unsigned long long)ev.lane_rounds[k]);
        if (ev.wave_rounds[k]) atomicAdd(&g_phase_prof[8 + k], (unsigned long long)ev.wave_rounds[k]);
    }
#endif
}

// ------------------------------------------------------------- batch queries
template <bool COUNT>
struct ClosestIO {
    static constexpr bool kExactDistance = true;
    static constexpr bool kStageCull = true;
    const DScene &S;
    const pt_ray *rays;
    pt_hit *hits;
    TraceCounters *counters;
    PTD bool load(uint32_t i, V3 &o, V3 &d, float &tmax, uint32_t &tag) const {
        const float4 *rp = reinterpret_cast<const float4 *>(rays + i);
        const float4 ro = rp[0], rd = rp[1];
        o = mk3(ro.x, ro.y, ro.z);
        d = mk3(rd.x, rd.y, rd.z);
        tmax = rd.w;
        tag = i;
        return true;
    }
    // the record is not read again by this launch: stored past the caches' retention (2.5 % of the launch time
    // against plain stores, measured on the bench's batch)
    PTD void store(uint32_t i, float t, float u, float v, int prim, int inst, int mat) const {
        v4f *np = reinterpret_cast<v4f *>(hits + i);
        __builtin_nontemporal_store(v4f{t, u, v, as_f(uint32_t(prim))}, np);
        __builtin_nontemporal_store(v4f{as_f(uint32_t(inst)), as_f(uint32_t(mat)), 0.0f, 0.0f}, np + 1);
    }
    template <class TR>
    PTD void done(uint32_t i, const TR &tr, bool hit) const {
        // one store site for hits and misses: the two halves of a record leave back to back and are merged on their way
        // to memory (with a site per case the halves went out apart, and the memory side counted 61 bytes per record)
        store(i, hit ? tr.hit.t : -1.0f, hit ? tr.hit.u : 0.0f, hit ? tr.hit.v : 0.0f, hit ? int(tr.hit.prim) : -1, hit ? tr.hit.inst : -1,
              hit ? int(tr.hit.mat) : -1);
        if (COUNT) counters[i] = tr.ctr;
    }
    PTD void miss(uint32_t i, float) const {
        store(i, -1.0f, 0.0f, 0.0f, -1, -1, -1);
        if (COUNT) counters[i] = TraceCounters{0, 0, 0, 0};
    }
    PTD bool has_pending() const { return false; }
    PTD void flush() const {}
    PTD void finish() const {}
};

#ifndef PT_QUERY_WAVES_PER_SIMD
#define PT_QUERY_WAVES_PER_SIMD 4  // the 4-wide variants would otherwise take 134 registers: three waves per SIMD, 15 % slower
#endif
template <bool COUNT, bool WIDE = false>
__global__ __launch_bounds__(64, PT_QUERY_WAVES_PER_SIMD) void k_trace_closest(DScene S, const pt_ray *rays, uint32_t n, pt_hit *hits, uint32_t *cursors, uint32_t cursor_count,
                                                      uint2 *spill, uint32_t spill_stride, DevStats *stats, TraceCounters *counters) {
    LaneEvents ev = make_events(spill + size_t(blockIdx.x) * kWave, spill_stride);
    ClosestIO<COUNT> io{S, rays, hits, counters};
    __shared__ unsigned long long lds[kWaveLdsEntries];
    const auto stk = make_lane_stack(lds, spill + size_t(blockIdx.x) * kWave, spill_stride);
    PT_WAVE_TIME(0);
    trace_stream<MODE_CLOSEST, COUNT, WIDE>(S, io, n, make_feed(cursors, cursor_count, n), stk, ev);
    PT_WAVE_TIME(1);
    flush_events(stats, ev);
}

template <bool COUNT>
struct AnyIO {
    static constexpr bool kExactDistance = true;
    static constexpr bool kStageCull = true;
    const pt_ray *rays;
    uint8_t *occluded;
    TraceCounters *counters;
    PTD bool load(uint32_t i, V3 &o, V3 &d, float &tmax, uint32_t &tag) const {
        const float4 *rp = reinterpret_cast<const float4 *>(rays + i);
        const float4 ro = rp[0], rd = rp[1];
        o = mk3(ro.x, ro.y, ro.z);
        d = mk3(rd.x, rd.y, rd.z);
        tmax = rd.w;
        tag = i;
        return true;
    }
    template <class TR>
    PTD void done(uint32_t i, const TR &tr, bool hit) const {
        occluded[i] = hit ? 1 : 0;
        if (COUNT) counters[i] = tr.ctr;
    }
    PTD void miss(uint32_t i, float) const {
        occluded[i] = 0;
        if (COUNT) counters[i] = TraceCounters{0, 0, 0, 0};
    }
    PTD bool has_pending() const { return false; }
    PTD void flush() const {}
    PTD void finish() const {}
};

template <bool WIDE = false, bool COUNT = false>
__global__ __launch_bounds__(64) void k_trace_any(DScene S, const pt_ray *rays, uint32_t n, uint8_t *occluded, uint32_t *cursors, uint32_t cursor_count,
                                                  uint2 *spill, uint32_t spill_stride, DevStats *stats, TraceCounters *counters) {
    LaneEvents ev = make_events(spill + size_t(blockIdx.x) * kWave, spill_stride);
    AnyIO<COUNT> io{rays, occluded, counters};
    __shared__ unsigned long long lds[kWaveLdsEntries];
    const auto stk = make_lane_stack(lds, spill + size_t(blockIdx.x) * kWave, spill_stride);
    trace_stream<MODE_ANY, COUNT, WIDE>(S, io, n, make_feed(cursors, cursor_count, n), stk, ev);
    flush_events(stats, ev);
}

__global__ void k_resolve_hits(DScene S, const pt_ray *rays, const pt_hit *hits, uint32_t n, pt_intersection *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const pt_hit h = hits[i];
    pt_intersection r;
    r.time = -1.0f;
    r.pos[0] = r.pos[1] = r.pos[2] = 0.0f;
    r.normal[0] = r.normal[1] = r.normal[2] = 0.0f;
    r.uv[0] = r.uv[1] = 0.0f;
    r.material_id = -1;
    r.geometry_id = -1;
    if (h.inst >= 0 && uint32_t(h.inst) < S.geometry_count) {
        // find the triangle record of (instance, prim): records of a mesh are a permutation, so
        // the resolve kernel interpolates from the mesh arrays directly
        const float4 id = reinterpret_cast<const float4 *>(S.insts + h.inst)[3];
        const uint32_t first_index = as_u(id.w);
        const uint32_t i0 = S.indices[first_index + uint32_t(h.prim) * 3u + 0], i1 = S.indices[first_index + uint32_t(h.prim) * 3u + 1],
                       i2 = S.indices[first_index + uint32_t(h.prim) * 3u + 2];
        const V3 p0 = ld3(S.positions, i0), p1 = ld3(S.positions, i1), p2 = ld3(S.positions, i2);
        const V3 n0 = ld3(S.normals, i0), n1 = ld3(S.normals, i1), n2 = ld3(S.normals, i2);
        const V2 uv0 = ld2(S.uvs, i0), uv1 = ld2(S.uvs, i1), uv2 = ld2(S.uvs, i2);
        const float w = 1.0f - h.u - h.v;
        const V3 bn = normalize(w * n0 + h.u * n1 + h.v * n2);
        const V2 buv = w * uv0 + h.u * uv1 + h.v * uv2;
        const V3 bp = w * p0 + h.u * p1 + h.v * p2;
        const float4 *fp = reinterpret_cast<const float4 *>(S.fwd + h.inst);
        const M34 T = load_m34(fp[0], fp[1], fp[2]);
        const M34 N = load_m34(fp[3], fp[4], fp[5]);
        const V3 pos = xform(T, bp, 1.0f);
        const V3 nrm = normalize(xform(N, bn, 0.0f));
        const V3 ro = mk3(rays[i].origin[0], rays[i].origin[1], rays[i].origin[2]);
        const V3 rd = mk3(rays[i].direction[0], rays[i].direction[1], rays[i].direction[2]);
        r.time = dot(pos - ro, rd);
        r.pos[0] = pos.x; r.pos[1] = pos.y; r.pos[2] = pos.z;
        r.normal[0] = nrm.x; r.normal[1] = nrm.y; r.normal[2] = nrm.z;
        r.uv[0] = buv.x; r.uv[1] = buv.y;
        r.material_id = int(as_u(id.y));
        r.geometry_id = h.inst;
    }
    out[i] = r;
}

// Incoherent rays: origin uniform in the root box scaled 1.1x about its centre, direction
// uniform on the sphere, PCG state (i << 32) | 0xdeadbeef after two discarded draws.
__global__ void k_gen_incoherent(DScene S, pt_ray *rays, uint32_t n, uint64_t first_index) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    uint64_t rng = ((first_index + k) << 32) | 0xdeadbeefull;
    sample1d(rng);  // the outputs of an unmixed counter state are ~0: two LCG steps mix it first
    sample1d(rng);
    float o[3];
    for (int a = 0; a < 3; ++a) {
        const float c = 0.5f * (S.root_min[a] + S.root_max[a]);
        const float hw = 0.55f * (S.root_max[a] - S.root_min[a]);
        const float xi = sample1d(rng);
        o[a] = c + hw * (2.0f * xi - 1.0f);
    }
    const float z = 1.0f - 2.0f * sample1d(rng);
    const float phi = 2.0f * kPi * sample1d(rng);
    const float r = sqrtf(fmaxf(0.0f, 1.0f - z * z));
    float4 *rp = reinterpret_cast<float4 *>(rays + k);
    rp[0] = make_float4(o[0], o[1], o[2], 0.0f);
    float sin_phi, cos_phi;
    sincos_azimuth(phi, sin_phi, cos_phi);
    rp[1] = make_float4(r * cos_phi, r * sin_phi, z, FLT_MAX);
}

// ------------------------------------------------------------ path pipeline
// Path slot <-> pixel: slots are laid out in 8x8 pixel tiles (one tile per wave) so that the
// camera rays of a wave form a compact bundle.  When several samples (iterations) are in flight
// at once, sample k owns the slots [k * slots_per_sample, (k+1) * slots_per_sample).
// The path pipeline lays its slots over the render window only; a pixel keeps its image coordinates
// (and with them its sampler seed), so a window renders exactly what the full frame renders there.
PTD bool slot_to_pixel(uint32_t slot, int width, int height, int &x, int &y);
PTD bool slot_to_pixel(uint32_t slot, const PathState &P, int &x, int &y) {
    const bool inside = slot_to_pixel(slot, P.win_w, P.win_h, x, y);
    x += P.win_x0;
    y += P.win_y0;
    return inside;
}
PTD bool slot_to_pixel(uint32_t slot, int width, int height, int &x, int &y) {
    const uint32_t tiles_x = uint32_t(width + 7) >> 3;
    const uint32_t tiles_y = uint32_t(height + 7) >> 3;
    slot %= tiles_x * tiles_y * 64u;
    const uint32_t tile = slot >> 6, within = slot & 63u;
    x = int((tile % tiles_x) * 8u + (within & 7u));
    y = int((tile / tiles_x) * 8u + (within >> 3));
    return x < width && y < height;
}

// GenerateRaysKernel, path_trace.cu:76-108.
__global__ __launch_bounds__(64) void k_generate(DScene S, PathState P, BounceCounters *cnt, int first_iteration, int width, int height) {
    const uint32_t slot = blockIdx.x * kWave + threadIdx.x;
    int x, y;
    const bool active = slot_to_pixel(slot, P, x, y);
    if (active) {
        const int iteration = first_iteration + int(slot / P.slots_per_sample);
        uint64_t rng = seed_sampler(x, y, iteration);
        const float sx = sample1d(rng), sy = sample1d(rng);
        sample1d(rng);  // Sample4D draws four values; the lens pair is unused by the pinhole camera
        sample1d(rng);
        const DCamera &c = S.camera;  // SpawnCameraRay, kernel/camera.h:10-22
        const float fx = ((float(x) + sx) - 0.5f * float(c.res_x)) * c.px;
        const float fy = ((float(y) + sy) - 0.5f * float(c.res_y)) * c.py;
        const V3 film_point = c.position + c.front * c.focal_length + c.right * fx - c.up * fy;
        const V3 dir = normalize(film_point - c.position);
        P.ray_o[slot] = make_float4(c.position.x, c.position.y, c.position.z, -1.0f);
        P.ray_d[slot] = make_float4(dir.x, dir.y, dir.z, as_f(S.max_depth));
        P.thr[slot] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
        P.accum[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        P.rng[slot] = make_uint2(uint32_t(rng), uint32_t(rng >> 32));
    }
    // bounce 0 needs no queue: its extension "queue" is the identity over all path slots
    if (slot == 0) cnt[0].ext_count = gridDim.x * kWave;
}

// IntersectClosestKernel, path_trace.cu:110-179: closest hit + classification by material type.
// A miss ends the path on the spot (the reference routes it through EmissiveHitKernel, which adds
// throughput * 0, path_trace.cu:351-356: a no-op unless the throughput is not finite).
struct ExtendIO {
    static constexpr bool kExactDistance = false;  // a path keeps (instance, u, v, record) of its hit, not the distance
    static constexpr bool kStageCull = true;
    const DScene &S;
    const PathState &P;
    BounceCounters *bc;
    const uint32_t *queue;  // null: identity (bounce 0)
    int width, height;
    bool pending = false;
    uint32_t pend_slot = 0;
    int pend_cls = 0;
    // The path slot travels with the ray as its tag.  A throughput that is not finite (flagged in the
    // bounce word by k_shade) is passed on as t_max = +infinity instead of FLT_MAX: no distance lies
    // between the two, so the traversal is the same, and done() can tell without reading the path.
    PTD bool load(uint32_t i, V3 &o, V3 &d, float &tmax, uint32_t &tag) {
        uint32_t slot;
        if (queue) {
            slot = queue[i];
        } else {
            slot = i;
            int x, y;
            if (!slot_to_pixel(slot, P, x, y)) return false;
        }
        const float4 ro = P.ray_o[slot], rd = P.ray_d[slot];
        o = mk3(ro.x, ro.y, ro.z);
        d = mk3(rd.x, rd.y, rd.z);
        tmax = (as_u(rd.w) & kThroughputNotFinite) ? __builtin_inff() : FLT_MAX;
        tag = slot;
        return true;
    }
    PTD bool has_pending() const { return pending; }
    // a miss ends the path on the spot: it enters no class queue, so no hit record is kept
    PTD void miss(uint32_t slot, float tmax) const {
        if (tmax > FLT_MAX) {  // the black environment adds throughput * 0: NaN iff the throughput is not finite
            const float4 t = P.thr[slot];
            float4 a = P.accum[slot];
            a.x += t.x * 0.0f;
            a.y += t.y * 0.0f;
            a.z += t.z * 0.0f;
            P.accum[slot] = a;
        }
    }
    template <class TR>
    PTD void done(uint32_t slot, const TR &tr, bool hit) {
        if (!hit) {
            miss(slot, tr.t_max);
            return;
        }
        P.hit[slot] = make_float4(as_f(uint32_t(tr.hit.inst)), tr.hit.u, tr.hit.v, as_f(tr.hit.rec));
        const uint32_t type = tr.hit.type;
        pend_cls = type == PT_MAT_LAMBERTIAN ? 1 : type == PT_MAT_MIRROR ? 2 : type == PT_MAT_METALLIC_ROUGHNESS ? 3
                   : type == PT_MAT_DIELECTRIC ? 4 : 0;
        pend_slot = slot;
        pending = true;
    }
    // Finished paths are staged per class in LDS and written out 64 at a time, so each class
    // queue costs one atomic per 64 entries (same-address atomics serialise at ~12 ns each) and
    // its stores are full 256-byte rows.
    uint32_t *stage;                      // [kNumClasses][kStageCap], this wave's LDS staging area
    uint32_t staged[kNumClasses] = {};    // wave-uniform fill levels
    PTD void flush() {
        if (__ballot(pending) == 0) return;
        const uint32_t lane = threadIdx.x & 63;
#pragma unroll
        for (int k = 0; k < kNumClasses; ++k) {
            const bool mine = pending && pend_cls == k;
            const uint64_t mask = __ballot(mine);
            if (mask == 0) continue;
            const uint32_t prefix = __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u));
            if (mine) stage[k * kStageCap + staged[k] + prefix] = pend_slot;
            staged[k] += uint32_t(__popcll(mask));
            if (staged[k] >= kStageFlush) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&bc->class_count[k], uint32_t(kStageFlush));
                base = __shfl(base, 0, 64);
#pragma unroll
                for (int r = 0; r < kStageFlush / kWave; ++r) P.class_q[k][base + r * kWave + lane] = stage[k * kStageCap + r * kWave + lane];
                const uint32_t rest = staged[k] - kStageFlush;  // < 64
                const uint32_t moved = (lane < rest) ? stage[k * kStageCap + kStageFlush + lane] : 0u;
                if (lane < rest) stage[k * kStageCap + lane] = moved;
                staged[k] = rest;
            }
        }
        pending = false;
    }
    PTD void finish() {
        const uint32_t lane = threadIdx.x & 63;
#pragma unroll
        for (int k = 0; k < kNumClasses; ++k) {
            if (staged[k] == 0) continue;
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&bc->class_count[k], staged[k]);
            base = __shfl(base, 0, 64);
            for (uint32_t i = lane; i < staged[k]; i += kWave) P.class_q[k][base + i] = stage[k * kStageCap + i];  // up to kStageFlush - 1 entries
            staged[k] = 0;
        }
    }
};

#ifndef PT_EXTEND_WAVES_PER_SIMD
#define PT_EXTEND_WAVES_PER_SIMD 4
#endif
template <bool WIDE = false>
__global__ __launch_bounds__(64, PT_EXTEND_WAVES_PER_SIMD) void k_extend(DScene S, PathState P, BounceCounters *cnt, int bounce, int width, int height, uint2 *spill,
                                               uint32_t spill_stride, DevStats *stats) {
    BounceCounters *bc = cnt + bounce;
    LaneEvents ev = make_events(spill + size_t(blockIdx.x) * kWave, spill_stride);
    __shared__ uint32_t stage[kNumClasses * kStageCap];
    ExtendIO io{S, P, bc, bounce == 0 ? nullptr : P.ext_q[bounce & 1], width, height};
    io.stage = stage;
    constexpr int kDepth = WIDE ? kRenderLdsDepthWide : kRenderLdsDepth;
    __shared__ unsigned long long lds[wave_lds_entries<kDepth>()];
    const auto stk = make_lane_stack<kDepth>(lds, spill + size_t(blockIdx.x) * kWave, spill_stride);
    trace_stream<MODE_CLOSEST, false, WIDE>(S, io, bc->ext_count, make_feed(&bc->cur_extend, 1u, bc->ext_count), stk, ev);
    flush_events(stats, ev);
}

// MaterialShadeKernel<M> (path_trace.cu:181-290) for classes 1..4 and EmissiveHitKernel
// (:337-389) for class 0, as one persistent launch whose waves each work on a single class.
// Next-event estimation is split off: the light sample and its full contribution are computed
// here, the visibility test runs in k_shadow, which adds the contribution when unoccluded.
//
// Same-address atomics serialise at ~12 ns each, so a wave takes kShadeGroup chunks of 64 paths
// per work fetch when the queues are long, stages the slots it emits (next bounce's extension
// rays, shadow rays) in LDS and appends them with one atomic per queue and group.
#ifndef PT_SHADE_GROUP
#define PT_SHADE_GROUP 16
#endif
constexpr uint32_t kShadeGroup = PT_SHADE_GROUP;

// Diagnostic build only (-DPT_PHASE_PROF -DPT_SHADE_STAMPS): wave clock spent between the marked
// points of k_shade, every stamp first waiting for all outstanding memory operations.
#ifdef PT_SHADE_STAMPS
#define PT_STAMP(k)                                                                                          \
    do {                                                                                                     \
        unsigned long long t_;                                                                               \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
        stamp_acc[k] += t_ - stamp_last;                                                                     \
        stamp_last = t_;                                                                                     \
    } while (0)
#else
#define PT_STAMP(k) do { } while (0)
#endif
constexpr int kShadeWavesPerCu = 16;

// The kernel is bound by the latency of its dependent gathers (path state, vertex data, texels:
// ~2500 cycles per L2 miss measured), not by VALU work (forcing more waves per SIMD through
// register spilling made it slower).  So the loads are organised in levels: everything whose
// address is known is requested before anything is waited for, and the hit chain (triangle ->
// indices -> vertex attributes -> textures) advances level by level together with the light
// chain (light pick -> instance -> triangle pick -> indices -> vertices -> texture).
__global__ __launch_bounds__(64) void k_shade(DScene S, PathState P, BounceCounters *cnt, int bounce) {
    __shared__ uint32_t stage_ext[kShadeGroup * kWave], stage_sh[kShadeGroup * kWave];
    const uint32_t lane = threadIdx.x;
    BounceCounters *bc = cnt + bounce;
    uint32_t chunk_start[kNumClasses + 1];
    chunk_start[0] = 0;
#pragma unroll
    for (int k = 0; k < kNumClasses; ++k) chunk_start[k + 1] = chunk_start[k] + ((bc->class_count[k] + kWave - 1) >> 6);
    const uint32_t total_chunks = chunk_start[kNumClasses];
    const uint32_t group = (total_chunks >= 4u * kShadeGroup * gridDim.x) ? kShadeGroup : (total_chunks >= 8u * gridDim.x ? 2u : 1u);
#ifdef PT_SHADE_STAMPS
    unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_last)::"memory");
#endif
    for (;;) {
      PT_STAMP(7);
      const uint32_t first_chunk = wave_fetch(&bc->cur_shade, group);
      if (first_chunk >= total_chunks) break;
      const uint32_t end_chunk = (total_chunks - first_chunk < group) ? total_chunks : first_chunk + group;
      uint32_t n_ext = 0, n_sh = 0;  // wave-uniform fill levels of the staging areas
      for (uint32_t chunk = first_chunk; chunk < end_chunk; ++chunk) {
        int cls = 0;
#pragma unroll
        for (int k = 1; k < kNumClasses; ++k)
            if (chunk >= chunk_start[k]) cls = k;
        const uint32_t qi = (chunk - chunk_start[cls]) * kWave + lane;
        const bool active = qi < bc->class_count[cls];
        bool want_shadow = false, want_extend = false;
        uint32_t slot = 0;
        float4 sh_o = make_float4(0, 0, 0, 0), sh_d = sh_o, sh_c = sh_o;
        if (active) {
            PT_STAMP(0);
            slot = P.class_q[cls][qi];
            const float4 ro4 = P.ray_o[slot], rd4 = P.ray_d[slot];
            const V3 ro = mk3(ro4.x, ro4.y, ro4.z), rd = mk3(rd4.x, rd4.y, rd4.z);
            const float last_pdf = ro4.w;
            int remaining_bounces = int(as_u(rd4.w) & ~kThroughputNotFinite);
            const float4 thr4 = P.thr[slot];
            V3 throughput = mk3(thr4.x, thr4.y, thr4.z);
            const float4 h4 = P.hit[slot];
            const int inst = int(as_u(h4.x));
            if (cls == 0) {
                const float4 acc4 = P.accum[slot];  // only emissive hits add radiance here
                V3 accum = mk3(acc4.x, acc4.y, acc4.z);
                if (inst < 0) {
                    accum = accum + throughput * splat(0.0f);  // black environment, path_trace.cu:351-356
                } else {
                    const Isect is = resolve_hit(S, ro, rd, h4.y, h4.z, as_u(h4.w), inst);
                    const DMaterial m = S.materials[is.material_id];
                    const float4 fm = reinterpret_cast<const float4 *>(S.fwd + inst)[6];
                    const V3 color = sample_tex3(S, m.tex0, is.uv);
                    const V3 radiance = as_f(m.w1) * color;
                    float weight_bsdf = 1.0f;
                    if (last_pdf > 0.0f) {
                        const float4 lp4 = P.last_pos[slot];
                        const V3 surface_point = mk3(lp4.x, lp4.y, lp4.z);
                        const pt_area_light light = S.lights[int(as_u(fm.y))];
                        const float dist2 = length2(is.pos - surface_point);
                        const V3 w_in = normalize(surface_point - is.pos);
                        const float cos_light = dot(is.normal, w_in);
                        if (cos_light > 0.0f) {
                            const float light_pdf = (light.power / S.total_light_power) * (1.0f / fm.x) * dist2 / cos_light;
                            weight_bsdf = power_heuristic(last_pdf, light_pdf);
                        }
                    }
                    accum = accum + throughput * radiance * weight_bsdf;
                }
                P.accum[slot] = make_float4(accum.x, accum.y, accum.z, 0.0f);
            } else {
                // ---- level A: the rest of the path state (requested together with the loads above)
                const uint2 r2 = P.rng[slot];
                uint64_t rng = uint64_t(r2.x) | (uint64_t(r2.y) << 32);
                const float hu = h4.y, hv = h4.z;
                const uint32_t rec = as_u(h4.w);
                const bool specular = (cls == 2 || cls == 4);  // the class is the material type (k_extend)
                const bool nee = !specular && S.light_count > 0;
                // kernel/light.h:15-59 draws: light pick, triangle pick, point (2).  An invalid light
                // returns after the first draw, so the later three are taken from a copy.
                AliasPick pick{0, 0.0f};
                uint64_t rng_l = rng;
                if (nee) pick = alias_begin(S.light_sampler.size, sample1d(rng_l));
                const uint64_t rng_after_pick = rng_l;
                float x_tri = 0.0f, lsx = 0.0f, lsy = 0.0f;
                if (nee) {
                    x_tri = sample1d(rng_l);
                    lsx = sample1d(rng_l);
                    lsy = sample1d(rng_l);
                }
                // Loads that miss the caches (path state above, triangle records, texels, light
                // triangles) are requested in batches; the small tables in between (instances,
                // materials, texture and light descriptors) are cache hits and are read where needed,
                // which keeps the register count, and with it the waves per SIMD, in check.
                PT_STAMP(1);
                // ---- level B: triangle corners and vertex attributes
                const float4 *tp = reinterpret_cast<const float4 *>(S.tris + rec);
                const float4 ta = tp[0], tb = tp[1], tc = tp[2];
                const float4 *sp = reinterpret_cast<const float4 *>(S.tri_shade + rec);
                const float4 s0 = sp[0], s1 = sp[1], s2 = sp[2], s3 = sp[3];
                // light pick -> light descriptor -> triangle pick (small tables, then one pair per triangle)
                DLight light;
                light.power = 0.0f;
                light.gi = -1;
                light.tri_base = 0;
                light.tri_count = 1;
                light.total_area = 1.0f;
                light.tex0 = -1;
                light.emittance = 0.0f;
                if (nee) {
                    const uint32_t light_id = alias_end(pick, S.light_sampler.prob[pick.i], S.light_sampler.alias[pick.i]);
                    light = S.dlights[light_id];
                }
                const bool lvalid = nee && light.gi != -1;
                if (nee) rng = lvalid ? rng_l : rng_after_pick;
                const AliasPick tpick = alias_begin(light.tri_count, x_tri);
                uint2 la = make_uint2(0u, 0u);
                if (lvalid) la = S.light_alias[light.tri_base + tpick.i];
                const float4 idv = reinterpret_cast<const float4 *>(S.insts + inst)[3];
                const int mid = int(as_u(idv.y));
                const DMaterial m = S.materials[mid];
                const uint32_t mtype = m.type & 0xFFu;
                Isect is;  // resolve_hit(), shade.h
                {
                    const float4 *fp = reinterpret_cast<const float4 *>(S.fwd + inst);
                    const M34 T = load_m34(fp[0], fp[1], fp[2]);
                    const M34 N = load_m34(fp[3], fp[4], fp[5]);
                    const V3 p0 = mk3(ta.x, ta.y, ta.z), p1 = mk3(ta.w, tb.x, tb.y), p2 = mk3(tb.z, tb.w, tc.x);
                    const V3 n0 = mk3(s0.x, s0.y, s0.z), n1 = mk3(s0.w, s1.x, s1.y), n2 = mk3(s1.z, s1.w, s2.x);
                    const V2 uv0 = V2{s2.y, s2.z}, uv1 = V2{s2.w, s3.x}, uv2 = V2{s3.y, s3.z};
                    const float w = 1.0f - hu - hv;
                    const V3 best_normal = normalize(w * n0 + hu * n1 + hv * n2);
                    const V2 best_uv = w * uv0 + hu * uv1 + hv * uv2;
                    const V3 best_pos = w * p0 + hu * p1 + hv * p2;
                    is.pos = xform(T, best_pos, 1.0f);
                    is.uv = best_uv;
                    is.time = dot(is.pos - ro, rd);
                    is.normal = normalize(xform(N, best_normal, 0.0f));
                    is.material_id = mid;
                    is.geometry_id = inst;
                }
                PT_STAMP(2);
                // ---- level C: texels of the material's textures; the picked light triangle
                const bool is_mr = (mtype == PT_MAT_METALLIC_ROUGHNESS);
                const TexReq R0 = tex_request(S.tex3, S.tex3_count, m.tex0, is.uv);
                const TexReq R1 = tex_request(S.tex1, is_mr ? S.tex1_count : 0u, int(m.w1), is.uv);
                const TexReq R2 = tex_request(S.tex1, is_mr ? S.tex1_count : 0u, m.w2, is.uv);
                float4 a00 = make_float4(0, 0, 0, 0), a10 = a00, a01 = a00, a11 = a00;
                if (R0.p.mode == 3) {
                    const float4 *d = static_cast<const float4 *>(R0.data);
                    a00 = d[R0.p.o00]; a10 = d[R0.p.o10]; a01 = d[R0.p.o01]; a11 = d[R0.p.o11];
                }
                float b00 = 0, b10 = 0, b01 = 0, b11 = 0, c00 = 0, c10 = 0, c01 = 0, c11 = 0;
                if (R1.p.mode == 3) {
                    const float *d = static_cast<const float *>(R1.data);
                    b00 = d[R1.p.o00]; b10 = d[R1.p.o10]; b01 = d[R1.p.o01]; b11 = d[R1.p.o11];
                }
                if (R2.p.mode == 3) {
                    const float *d = static_cast<const float *>(R2.data);
                    c00 = d[R2.p.o00]; c10 = d[R2.p.o10]; c01 = d[R2.p.o01]; c11 = d[R2.p.o11];
                }
                const uint32_t ltri = alias_end(tpick, as_f(la.x), la.y);
                float4 g0 = make_float4(0, 0, 0, 0), g1 = g0, g2 = g0, g3 = g0, g4 = g0, g5 = g0;
                if (lvalid) {
                    const float4 *lt = reinterpret_cast<const float4 *>(S.light_tris + (light.tri_base + ltri));
                    g0 = lt[0]; g1 = lt[1]; g2 = lt[2]; g3 = lt[3]; g4 = lt[4]; g5 = lt[5];
                }
                // the material's textures are looked up once per hit; the reference's sample / eval /
                // pdf routines each repeat the same lookups (metallic_roughness.h:8-177)
                PT_STAMP(3);
                const V3 albedo = tex3_finish(R0, a00, a10, a01, a11);
                MRParams mr;
                mr.base_color = albedo;
                mr.metallic = tex1_finish(R1, b00, b10, b01, b11);
                {
                    float alpha = tex1_finish(R2, c00, c10, c01, c11);
                    alpha = gmin(gmax(alpha, 0.02f), 1.0f);
                    mr.alpha = alpha * alpha;
                }
                const V3 w_out = -normalize(rd);
                LightSample ls;
                ls.position = ls.normal = ls.radiance = splat(0.0f);
                ls.pdf = 0.0f;
                if (lvalid) {
                    const V3 p0 = mk3(g0.x, g0.y, g0.z), p1 = mk3(g0.w, g1.x, g1.y), p2 = mk3(g1.z, g1.w, g2.x);
                    const V3 n0 = mk3(g2.y, g2.z, g2.w), n1 = mk3(g3.x, g3.y, g3.z), n2 = mk3(g3.w, g4.x, g4.y);
                    const V2 t0 = V2{g4.z, g4.w}, t1 = V2{g5.x, g5.y}, t2 = V2{g5.z, g5.w};
                    const float sqr = sqrtf(lsx);  // Math::SampleTriangle, math/sampling.h:61-68
                    const float u = 1.0f - sqr;
                    const float v = lsy * sqr;
                    const float w = 1.0f - u - v;
                    ls.position = u * p0 + v * p1 + w * p2;
                    ls.normal = u * n0 + v * n1 + w * n2;  // un-normalised on purpose (kernel/light.h:47,54)
                    const V2 texcoord = u * t0 + v * t1 + w * t2;
                    // ---- level D: the light's texels
                    const TexReq LR = tex_request(S.tex3, S.tex3_count, light.tex0, texcoord);
                    float4 e00 = make_float4(0, 0, 0, 0), e10 = e00, e01 = e00, e11 = e00;
                    if (LR.p.mode == 3) {
                        const float4 *d = static_cast<const float4 *>(LR.data);
                        e00 = d[LR.p.o00]; e10 = d[LR.p.o10]; e01 = d[LR.p.o01]; e11 = d[LR.p.o11];
                    }
                    ls.radiance = tex3_finish(LR, e00, e10, e01, e11) * light.emittance;
                    ls.pdf = (light.power / S.total_light_power) * (1.0f / light.total_area);
                }
                if (nee) {
                    const V3 w_in = normalize(ls.position - is.pos);
                    const float dist = length(is.pos - ls.position);
                    const float cos_surf = dot(is.normal, w_in);
                    const float cos_light = dot(ls.normal, -w_in);
                    if (cos_surf > 0.0f && cos_light > 0.0f) {
                        V3 bsdf;
                        float bsdf_pdf;
                        if (mtype == PT_MAT_LAMBERTIAN) {
                            bsdf = albedo * kInvPi;
                            bsdf_pdf = fmaxf(0.0f, dot(is.normal, w_in)) * kInvPi;
                        } else {
                            bsdf = eval_metallic_roughness(mr, is, w_in, w_out);
                            bsdf_pdf = metallic_roughness_pdf(mr, is, w_in, w_out);
                        }
                        const float light_pdf = ls.pdf * dist * dist / cos_light;
                        const float light_weight = power_heuristic(light_pdf, bsdf_pdf);
                        const V3 contrib = throughput * light_weight * bsdf * cos_surf * ls.radiance / light_pdf;
                        const V3 so = is.pos + kEps * w_in;
                        const float tmax = float(double(dist) - 1e-4);  // path_trace.cu:228
                        sh_o = make_float4(so.x, so.y, so.z, tmax);
                        sh_d = make_float4(w_in.x, w_in.y, w_in.z, 0.0f);
                        sh_c = make_float4(contrib.x, contrib.y, contrib.z, 0.0f);
                        want_shadow = true;
                    }
                }
                PT_STAMP(4);
                BsdfSample sample;
                if (mtype == PT_MAT_LAMBERTIAN) sample = sample_lambertian(albedo, is, rng);
                else if (mtype == PT_MAT_MIRROR) sample = sample_mirror(albedo, is, w_out);
                else if (mtype == PT_MAT_DIELECTRIC) sample = sample_dielectric(albedo, m, is, w_out, rng);
                else sample = sample_metallic_roughness(mr, is, w_out, rng);
                const float cos_surf = fabsf(dot(is.normal, sample.w_in));
                bool alive = !(sample.pdf < kEps || cos_surf < kEps || sample.type == S_Invalid);
                float new_last_pdf = last_pdf;
                if (alive) {
                    throughput = throughput * (sample.bsdf * cos_surf / sample.pdf);
                    new_last_pdf = specular ? -1.0f : sample.pdf;
                    P.last_pos[slot] = make_float4(is.pos.x, is.pos.y, is.pos.z, 0.0f);
                    const float bounces = float(S.max_depth - uint32_t(remaining_bounces));
                    if (bounces >= 3) {  // Russian roulette, path_trace.cu:263-277
                        const float survival = gmin(gmax(luminance(throughput), 0.05f), 0.95f);
                        if (sample1d(rng) >= survival) {
                            remaining_bounces = 0;
                            alive = false;
                        } else {
                            throughput = throughput / survival;
                        }
                    }
                }
                V3 no = ro, nd = rd;
                if (alive) {
                    no = is.pos + kEps * sample.w_in;
                    nd = sample.w_in;
                    remaining_bounces--;
                    want_extend = remaining_bounces > 0;
                }
                P.thr[slot] = make_float4(throughput.x, throughput.y, throughput.z, 0.0f);
                P.ray_o[slot] = make_float4(no.x, no.y, no.z, new_last_pdf);
                const float tz = throughput.x * 0.0f + throughput.y * 0.0f + throughput.z * 0.0f;  // NaN iff a component is not finite
                P.ray_d[slot] = make_float4(nd.x, nd.y, nd.z, as_f(uint32_t(remaining_bounces) | (tz != tz ? kThroughputNotFinite : 0u)));
                P.rng[slot] = make_uint2(uint32_t(rng), uint32_t(rng >> 32));
            }
        }
        if (want_shadow) {
            P.sh_o[slot] = sh_o;
            P.sh_d[slot] = sh_d;
            P.sh_c[slot] = sh_c;
        }
        PT_STAMP(5);
        const uint64_t m_sh = __builtin_amdgcn_ballot_w64(want_shadow);
        if (want_shadow) stage_sh[n_sh + __builtin_amdgcn_mbcnt_hi(uint32_t(m_sh >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m_sh), 0u))] = slot;
        n_sh += uint32_t(__builtin_popcountll(m_sh));
        const uint64_t m_ext = __builtin_amdgcn_ballot_w64(want_extend);
        if (want_extend) stage_ext[n_ext + __builtin_amdgcn_mbcnt_hi(uint32_t(m_ext >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m_ext), 0u))] = slot;
        n_ext += uint32_t(__builtin_popcountll(m_ext));
      }
      // ---- append the group's output to the global queues: one atomic each
      __syncthreads();  // one wave per block: orders the staging writes before the reads below
      if (n_sh > 0) {
          const uint32_t base = wave_fetch(&bc->shadow_count, n_sh);
          for (uint32_t i = lane; i < n_sh; i += kWave) P.sh_q[base + i] = stage_sh[i];
      }
      if (n_ext > 0) {
          const uint32_t base = wave_fetch(&bc[1].ext_count, n_ext);
          uint32_t *q = P.ext_q[(bounce + 1) & 1];
          for (uint32_t i = lane; i < n_ext; i += kWave) q[base + i] = stage_ext[i];
      }
      __syncthreads();  // the staging areas are reused by the next group
      PT_STAMP(6);
    }
#ifdef PT_SHADE_STAMPS
    if (lane == 0)
        for (int k = 0; k < 8; ++k) atomicAdd(&g_phase_prof[k], stamp_acc[k]);
#endif
}

// The visibility half of next-event estimation (path_trace.cu:221-243): IntersectScene<ANY> with
// t_max = dist - 1e-4; an unoccluded sample adds its contribution to the path's radiance.
struct ShadowIO {
    static constexpr bool kExactDistance = true;
    static constexpr bool kStageCull = true;
    const PathState &P;
    PTD bool load(uint32_t i, V3 &o, V3 &d, float &tmax, uint32_t &tag) {
        const uint32_t slot = P.sh_q[i];
        tag = slot;
        const float4 so = P.sh_o[slot], sd = P.sh_d[slot];
        o = mk3(so.x, so.y, so.z);
        d = mk3(sd.x, sd.y, sd.z);
        tmax = so.w;
        return true;
    }
    PTD void miss(uint32_t slot, float) const {  // nothing between the surface and the light
        const float4 contrib = P.sh_c[slot];
        float4 a = P.accum[slot];
        a.x += contrib.x;
        a.y += contrib.y;
        a.z += contrib.z;
        P.accum[slot] = a;
    }
    template <class TR>
    PTD void done(uint32_t slot, const TR &, bool occluded) const {
        if (!occluded) miss(slot, 0.0f);
    }
    PTD bool has_pending() const { return false; }
    PTD void flush() const {}
    PTD void finish() const {}
};

#ifndef PT_SHADOW_WAVES_PER_SIMD
#define PT_SHADOW_WAVES_PER_SIMD 4
#endif
template <bool WIDE = false>
__global__ __launch_bounds__(64, PT_SHADOW_WAVES_PER_SIMD) void k_shadow(DScene S, PathState P, BounceCounters *cnt, int bounce, uint2 *spill,
                                               uint32_t spill_stride, DevStats *stats) {
    BounceCounters *bc = cnt + bounce;
    LaneEvents ev = make_events(spill + size_t(blockIdx.x) * kWave, spill_stride);
    ShadowIO io{P};
    constexpr int kDepth = WIDE ? kRenderLdsDepthWide : kRenderLdsDepth;
    __shared__ unsigned long long lds[wave_lds_entries<kDepth>()];
    const auto stk = make_lane_stack<kDepth>(lds, spill + size_t(blockIdx.x) * kWave, spill_stride);
    trace_stream<MODE_ANY, false, WIDE>(S, io, bc->shadow_count, make_feed(&bc->cur_shadow, 1u, bc->shadow_count), stk, ev);
    flush_events(stats, ev);
}

// FinalGather, path_trace.cu:391-407: film += sample, NaN samples are dropped (and counted).
// With several samples in flight the samples of a pixel are added in iteration order, which is
// the order separate Render calls would add them, so the film is bit-identical either way.
__global__ __launch_bounds__(64) void k_film(PathState P, float *film, int width, int height, int samples, DevStats *stats) {
    const uint32_t slot = blockIdx.x * kWave + threadIdx.x;
    int x, y;
    if (!slot_to_pixel(slot, P, x, y)) return;
    float *px = film + 3 * (size_t(y) * width + x);
    float r = px[0], g = px[1], b = px[2];
    for (int k = 0; k < samples; ++k) {
        float4 a = P.accum[size_t(k) * P.slots_per_sample + slot];
        if (isnan(a.x) || isnan(a.y) || isnan(a.z)) {
            atomicAdd(&stats->nan_pixels, 1u);
            a = make_float4(0, 0, 0, 0);
        }
        r += a.x;
        g += a.y;
        b += a.z;
    }
    px[0] = r;
    px[1] = g;
    px[2] = b;
}

// PostprocessAndUpload, path_trace.cu:409-434: film / iterations -> ACES -> (int)(p * 255.0), no gamma.
__global__ void k_tonemap(const float *film, uint8_t *rgba, int width, int height, int iterations) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= uint32_t(width) * uint32_t(height)) return;
    const float inv = float(iterations);
    uchar4 o;
    uint8_t *c = reinterpret_cast<uint8_t *>(&o);
    for (int k = 0; k < 3; ++k) {
        const float radiance = film[3 * size_t(i) + k] / inv;
        const float p = aces(radiance);
        int v = int(double(p) * 255.0);
        v = v < 0 ? 0 : (v > 255 ? 255 : v);
        c[k] = uint8_t(v);
    }
    c[3] = 255;
    reinterpret_cast<uchar4 *>(rgba)[i] = o;
}

// Copies the extension rays of one bounce into a pt_ray buffer (queue order).
__global__ void k_gather_rays(PathState P, const BounceCounters *cnt, int bounce, pt_ray *out, uint32_t capacity, int width, int height) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n = cnt[bounce].ext_count;
    if (i >= n) return;
    uint32_t slot;
    if (bounce == 0) {  // camera rays: every path slot, written in pixel order
        slot = i;
        int x, y;
        if (!slot_to_pixel(slot, P, x, y)) return;
        i = uint32_t(y - P.win_y0) * uint32_t(P.win_w) + uint32_t(x - P.win_x0);  // pixel order inside the render window
    } else {
        slot = P.ext_q[bounce & 1][i];
    }
    if (i >= capacity) return;
    const float4 ro = P.ray_o[slot], rd = P.ray_d[slot];
    float4 *rp = reinterpret_cast<float4 *>(out + i);
    rp[0] = make_float4(ro.x, ro.y, ro.z, 0.0f);
    rp[1] = make_float4(rd.x, rd.y, rd.z, FLT_MAX);
}

// Copies the shadow rays of one bounce into a pt_ray buffer (queue order).
__global__ void k_gather_shadow_rays(PathState P, const BounceCounters *cnt, int bounce, pt_ray *out, uint32_t capacity) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cnt[bounce].shadow_count || i >= capacity) return;
    const uint32_t slot = P.sh_q[i];
    const float4 so = P.sh_o[slot], sd = P.sh_d[slot];
    float4 *rp = reinterpret_cast<float4 *>(out + i);
    rp[0] = make_float4(so.x, so.y, so.z, 0.0f);
    rp[1] = make_float4(sd.x, sd.y, sd.z, so.w);
}

}  // namespace ptd
