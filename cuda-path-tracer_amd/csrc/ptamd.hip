// libptamd.so: implementation of include/ptamd.h for gfx950 (MI355X).
// Context and device memory management, scene validation and re-packing, kernel launches.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <system_error>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/ptamd.h"
#include "kernels.h"
#include "debug_view.h"
#include "bvh_build.h"
#include "wide_bvh.h"
#ifndef PT_WAVES_PER_CU
#define PT_WAVES_PER_CU 16  // persistent waves per CU (4 per SIMD) of a ray query
#endif
// The same for the kernels of a render pass, two of which run side by side.  k_shade, whose waves
// mostly wait for gathers and for the queue counters, does best with few waves (measured: 1065 spp/s
// with 16 / 16 / 16, 1295 with 10 / 2 / 10 and 1024 paths per queue fetch).
#ifndef PT_LANES
#define PT_LANES 3  // render passes in flight (each with its own stream, path state, counters and spill area)
#endif
#ifndef PT_EXTEND_WAVES_PER_CU
#define PT_EXTEND_WAVES_PER_CU 10
#endif
#ifndef PT_SHADE_WAVES_PER_CU
#define PT_SHADE_WAVES_PER_CU 2
#endif
#ifndef PT_SHADOW_WAVES_PER_CU
#define PT_SHADOW_WAVES_PER_CU 10
#endif
#ifndef PT_QUERY_CURSORS
#define PT_QUERY_CURSORS 8   // work-fetch cursors of a batch query (one per XCD's share of the waves)
#endif
constexpr uint32_t kMaxQueryCursors = 64;

using namespace ptd;

// Scratch of pt_trace_closest_static, the entry point that allocates nothing: the work-feed cursors,
// the event counters and the traversal-stack spill area of a grid of kStaticWaves persistent waves.
// Every GPU the library runs on holds its own copy of this array until the process ends; the name
// below is only the host-side handle of the symbol, the device address is looked up once per device
// (StaticScratch) and that pointer is what kernels and memsets receive.
constexpr uint32_t kStaticWaves = 1024;
constexpr size_t kStaticCursorBytes = size_t(kMaxQueryCursors) * kCursorStride * sizeof(uint32_t);
constexpr size_t kStaticStatsOffset = kStaticCursorBytes;
constexpr size_t kStaticSpillOffset = kStaticCursorBytes + 256;
constexpr size_t kStaticSpillBytes = size_t(kStaticWaves) * kWave * kSpillRows * sizeof(uint2);
constexpr size_t kStaticScratchBytes = kStaticSpillOffset + kStaticSpillBytes;  // 60.5 MiB + 16.25 KiB
static_assert(sizeof(DevStats) <= 256, "the counters fit their slot");
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

namespace {

thread_local std::string g_create_error;

constexpr int kMaxDevices = 64;
std::mutex g_static_mutex;
unsigned char *g_static_ptr[kMaxDevices] = {};  // device address of `scratch` on each GPU, filled by pt_ctx_create

// The address of this device's copy of `scratch`; `device` must be the current device.
hipError_t StaticScratch(int device, unsigned char **out) {
    std::lock_guard<std::mutex> lock(g_static_mutex);
    if (!g_static_ptr[device]) {
        void *p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        g_static_ptr[device] = static_cast<unsigned char *>(p);
    }
    *out = g_static_ptr[device];
    return hipSuccess;
}

struct EventPair {
    hipEvent_t start, stop;
    int cls;
};
enum { EV_CLOSEST = 0, EV_ANY, EV_GENERATE, EV_SHADE, EV_FILM, EV_COUNT };

}  // namespace

struct pt_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string error;
    int num_cus = 256;
    uint32_t grid_blocks = 0;  // persistent grid size (blocks of one wave)
    uint32_t render_blocks[3] = {0, 0, 0};  // grids of k_extend / k_shade / k_shadow inside a render pass

    // scene
    bool scene_ready = false;
    DScene scene;
    std::vector<void *> scene_allocs;

    // traversal scratch
    uint32_t spill_stride = 0;
    uint32_t *cursor = nullptr;  // work-fetch cursors of the batch queries, kCursorStride words apart
    uint32_t query_cursors = PT_QUERY_CURSORS;
    DevStats *dev_stats = nullptr;
    unsigned char *static_scratch = nullptr;  // this device's copy of the file-scope `scratch` array

    // integrator.  Everything a render pass mutates besides the film exists twice ("lanes"):
    // consecutive pt_render_batch calls alternate between the lanes and run on the lanes' own
    // streams, so the latency-bound tail of one pass (late bounces with a few thousand rays
    // each) overlaps the head of the next.  The film passes stay in submission order.
    struct Lane {
        hipStream_t stream = nullptr;
        PathState paths;
        std::vector<void *> state_allocs;  // per-path state and queues (re-sized with the batch)
        uint32_t samples_alloc = 0;        // samples (iterations) the path state can hold in flight
        BounceCounters *counters = nullptr;
        uint32_t counters_capacity = 0;  // bounces + 2
        uint2 *spill = nullptr;          // traversal-stack spill area of this lane's kernels
        hipEvent_t done = nullptr;       // recorded after the lane's last film pass
        bool pending = false;            // `done` has been recorded and not yet joined
    } lane[PT_LANES];
    bool lane_retired[PT_LANES] = {};  // true once the lane's path state did not fit in device memory: fewer passes in flight
    int cur_lane = 0;            // the lane of the last render
    bool overlap_passes = true;  // off while profiling (events bracket kernels on one stream)
    hipEvent_t fork_event = nullptr;
    bool integrator_ready = false;
    int width = 0, height = 0;
    int win[4] = {0, 0, 0, 0};  // render window x0, y0, w, h (pt_set_render_window); the whole frame after pt_integrator_init
    std::vector<void *> path_allocs;   // film, counters
    float *film = nullptr;
    uint32_t *pinned_word = nullptr;
    int last_render_bounces = 0;
    int last_render_samples = 1;

    // profiling
    bool profiling = false;
    std::vector<EventPair> pending;
    std::vector<EventPair> free_events;
    pt_stats stats;
    int trace_mode = 0;
    int texture_filter = 0;  // kTexFixedPointWeights | kTexSrgbDecode, applied by the next pt_scene_upload
    bool host_may_block = false;  // set by the synchronous entry points around RunBounces
};

namespace {

int Fail(pt_ctx *ctx, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->error = buf;
    else g_create_error = buf;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                                  \
    do {                                                                                                    \
        hipError_t e__ = (expr);                                                                            \
        if (e__ != hipSuccess) return Fail(ctx, PT_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e__)); \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) {
        (void)hipGetDevice(&prev);
        if (prev != dev) (void)hipSetDevice(dev);
        else prev = -1;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

template <typename T>
int Upload(pt_ctx *ctx, std::vector<void *> &owner, const T *host, size_t count, const T **out) {
    *out = nullptr;
    if (count == 0) {
        // keep a valid (non-null) pointer so kernels never see nullptr arithmetic
        void *p = nullptr;
        HIP_TRY(ctx, hipMalloc(&p, 256));
        owner.push_back(p);
        *out = static_cast<const T *>(p);
        return PT_OK;
    }
    void *p = nullptr;
    HIP_TRY(ctx, hipMalloc(&p, count * sizeof(T)));
    owner.push_back(p);
    HIP_TRY(ctx, hipMemcpyAsync(p, host, count * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *out = static_cast<const T *>(p);
    return PT_OK;
}

template <typename T>
int Alloc(pt_ctx *ctx, std::vector<void *> &owner, size_t count, T **out) {
    void *p = nullptr;
    HIP_TRY(ctx, hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)));
    owner.push_back(p);
    *out = static_cast<T *>(p);
    return PT_OK;
}

void FreeAll(std::vector<void *> &owner) {
    for (void *p : owner) (void)hipFree(p);
    owner.clear();
}

// ---------------------------------------------------------------- profiling
void BeginEvent(pt_ctx *ctx, int cls, EventPair &ev, bool &active) {
    active = false;
    if (!ctx->profiling) return;
    if (!ctx->free_events.empty()) {
        ev = ctx->free_events.back();
        ctx->free_events.pop_back();
    } else {
        if (hipEventCreate(&ev.start) != hipSuccess) return;
        if (hipEventCreate(&ev.stop) != hipSuccess) return;
    }
    ev.cls = cls;
    (void)hipEventRecord(ev.start, ctx->stream);
    active = true;
}
void EndEvent(pt_ctx *ctx, EventPair &ev, bool active) {
    if (!active) return;
    (void)hipEventRecord(ev.stop, ctx->stream);
    ctx->pending.push_back(ev);
}
void DrainEvents(pt_ctx *ctx) {
    for (EventPair &ev : ctx->pending) {
        (void)hipEventSynchronize(ev.stop);
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, ev.start, ev.stop) == hipSuccess) {
            switch (ev.cls) {
                case EV_CLOSEST: ctx->stats.ms_closest += ms; ctx->stats.n_closest++; break;
                case EV_ANY: ctx->stats.ms_any += ms; ctx->stats.n_any++; break;
                case EV_GENERATE: ctx->stats.ms_generate += ms; ctx->stats.n_generate++; break;
                case EV_SHADE: ctx->stats.ms_shade += ms; ctx->stats.n_shade++; break;
                case EV_FILM: ctx->stats.ms_film += ms; ctx->stats.n_film++; break;
            }
        }
        ctx->free_events.push_back(ev);
    }
    ctx->pending.clear();
}
struct ScopedEvent {
    pt_ctx *ctx;
    EventPair ev;
    bool active;
    ScopedEvent(pt_ctx *c, int cls) : ctx(c), active(false) {  // null context: no timing
        if (c) BeginEvent(c, cls, ev, active);
    }
    ~ScopedEvent() { EndEvent(ctx, ev, active); }
};

// ------------------------------------------------------------ scene re-pack
bool FiniteBox(const float *mn, const float *mx) {
    for (int a = 0; a < 3; ++a)
        if (!(mn[a] == mn[a]) || !(mx[a] == mx[a])) return false;
    return true;
}

int Validate(pt_ctx *ctx, const pt_scene_desc *d) {
    if (!d) return Fail(ctx, PT_ERR_INVALID, "scene description is null");
    if (d->max_depth < 1) return Fail(ctx, PT_ERR_INVALID, "max_depth must be >= 1");
    if (d->max_depth > 4096) return Fail(ctx, PT_ERR_LIMIT, "max_depth %u exceeds the limit of 4096", d->max_depth);
    if (d->vertex_count == 0 || !d->positions || !d->normals || !d->uvs) return Fail(ctx, PT_ERR_INVALID, "scene has no vertices");
    if (d->index_count == 0 || !d->indices) return Fail(ctx, PT_ERR_INVALID, "scene has no indices");
    if (d->geometry_count == 0 || !d->geometries) return Fail(ctx, PT_ERR_INVALID, "scene has no geometry instances");
    if (d->tlas_node_count == 0 || !d->tlas_nodes) return Fail(ctx, PT_ERR_INVALID, "scene has no TLAS nodes");
    if (d->blas_count == 0 || !d->blases || d->blas_node_count == 0 || !d->blas_nodes) return Fail(ctx, PT_ERR_INVALID, "scene has no BLAS");
    if (!d->tri_indices || d->tri_index_count == 0) return Fail(ctx, PT_ERR_INVALID, "scene has no BLAS triangle indices");
    if (d->tri_index_count >= kMaxTris) return Fail(ctx, PT_ERR_LIMIT, "more than 2^28 BLAS triangles");
    if (d->material_count == 0 || !d->materials) return Fail(ctx, PT_ERR_INVALID, "scene has no materials");
    if (d->light_count > 0 && (!d->lights || !d->light_sampler || !d->triangle_samplers))
        return Fail(ctx, PT_ERR_INVALID, "lights without sampler tables");
    if (d->camera.resolution[0] <= 0 || d->camera.resolution[1] <= 0) return Fail(ctx, PT_ERR_INVALID, "camera resolution must be positive");
    for (uint32_t i = 0; i < d->index_count; ++i)
        if (d->indices[i] >= d->vertex_count) return Fail(ctx, PT_ERR_INVALID, "vertex index %u out of range at %u", d->indices[i], i);
    for (uint32_t i = 0; i < d->tlas_node_count; ++i) {
        const pt_tlas_node &n = d->tlas_nodes[i];
        if (n.left_right == 0) {
            if (n.instance >= d->geometry_count) return Fail(ctx, PT_ERR_INVALID, "TLAS leaf %u names instance %u", i, n.instance);
        } else {
            const uint32_t l = n.left_right & 0xFFFFu, r = n.left_right >> 16;
            if (l >= d->tlas_node_count || r >= d->tlas_node_count || l == 0 || r == 0 || l == i || r == i)
                return Fail(ctx, PT_ERR_INVALID, "TLAS node %u has bad children", i);
        }
    }
    // A node may be the child of one node only: children that are shared make the structure a graph whose
    // unfolding (the 4-wide collapse walks it as a tree) can be exponentially larger than the arrays.
    std::vector<uint8_t> has_parent(d->blas_node_count, 0);
    for (uint32_t i = 0; i < d->blas_count; ++i) {
        const pt_blas &b = d->blases[i];
        if (b.node_count == 0) continue;
        if (uint64_t(b.first_node) + b.node_count > d->blas_node_count || uint64_t(b.first_index) + b.index_count > d->tri_index_count)
            return Fail(ctx, PT_ERR_INVALID, "BLAS %u exceeds the node / index arrays", i);
        for (uint32_t k = b.first_node; k < b.first_node + b.node_count; ++k) {
            const pt_bvh_node &n = d->blas_nodes[k];
            if (n.primitive_count > 0) {
                if (uint64_t(n.first_primitive) + n.primitive_count > b.index_count)
                    return Fail(ctx, PT_ERR_INVALID, "BLAS %u leaf %u exceeds its triangle range", i, k);
            } else if (n.first_primitive <= k || uint64_t(n.first_primitive) + 1 >= uint64_t(b.first_node) + b.node_count) {
                return Fail(ctx, PT_ERR_INVALID, "BLAS %u node %u has bad children", i, k);
            } else {
                for (uint32_t c = n.first_primitive; c <= n.first_primitive + 1; ++c) {
                    if (has_parent[c]) return Fail(ctx, PT_ERR_INVALID, "BLAS %u node %u is the child of more than one node", i, c);
                    has_parent[c] = 1;
                }
            }
        }
    }
    for (uint32_t i = 0; i < d->geometry_count; ++i) {
        const pt_geometry_instance &g = d->geometries[i];
        if (g.material_id < 0 || uint32_t(g.material_id) >= d->material_count) return Fail(ctx, PT_ERR_INVALID, "instance %u: bad material", i);
        if (g.blas_index < 0 || uint32_t(g.blas_index) >= d->blas_count || d->blases[g.blas_index].node_count == 0)
            return Fail(ctx, PT_ERR_INVALID, "instance %u: bad blas_index", i);
        if (g.index_count == 0 || g.index_count % 3 || uint64_t(g.first_index) + g.index_count > d->index_count)
            return Fail(ctx, PT_ERR_INVALID, "instance %u: bad index range", i);
        if (g.index_count / 3 != d->blases[g.blas_index].index_count)
            return Fail(ctx, PT_ERR_INVALID, "instance %u: triangle count differs from its BLAS", i);
        const uint8_t mt = d->materials[g.material_id].type;
        if (mt == PT_MAT_EMISSIVE) {
            if (g.area_light_id < 0 || uint32_t(g.area_light_id) >= d->light_count || g.triangle_sampler_index < 0 ||
                uint32_t(g.triangle_sampler_index) >= d->triangle_sampler_count)
                return Fail(ctx, PT_ERR_INVALID, "emissive instance %u lacks light tables", i);
        }
    }
    for (uint32_t i = 0; i < d->material_count; ++i)
        if (d->materials[i].type > PT_MAT_EMISSIVE) return Fail(ctx, PT_ERR_INVALID, "material %u: unknown type", i);
    for (uint32_t i = 0; i < d->light_count; ++i) {
        const int gi = d->lights[i].geometry_instance_index;
        // -1 is a light without geometry: SampleSceneLight returns a black sample after one draw (kernel/light.h:18-19)
        if (gi < -1 || (gi >= 0 && uint32_t(gi) >= d->geometry_count)) return Fail(ctx, PT_ERR_INVALID, "light %u: bad instance", i);
    }
    if (d->light_count > 0) {
        if (d->light_sampler->size != d->light_count) return Fail(ctx, PT_ERR_INVALID, "light sampler size differs from light count");
        for (uint32_t i = 0; i < d->triangle_sampler_count; ++i)
            if (d->triangle_samplers[i].size == 0 || !d->triangle_samplers[i].alias_table || !d->triangle_samplers[i].probability_table)
                return Fail(ctx, PT_ERR_INVALID, "triangle sampler %u is empty", i);
    }
    if (!FiniteBox(d->tlas_nodes[0].min, d->tlas_nodes[0].max)) return Fail(ctx, PT_ERR_INVALID, "TLAS root bounds are NaN");
    return PT_OK;
}

uint32_t EncodeLeaf(uint32_t first, uint32_t count, std::vector<uint2> &big) {
    if (count <= 7) return kLeafBit | ((count - 1) << 28) | first;
    big.push_back(make_uint2(first, count));
    return kLeafBit | (7u << 28) | uint32_t(big.size() - 1);
}

void SetNodeChild(DNode &n, int side, const float *mn, const float *mx, uint32_t ref) {
    if (side == 0) {
        n.a = make_float4(mn[0], mn[1], mn[2], __builtin_bit_cast(float, ref));
        n.b.x = mx[0]; n.b.y = mx[1]; n.b.z = mx[2];
    } else {
        n.c = make_float4(mn[0], mn[1], mn[2], 0.0f);
        n.d = make_float4(mx[0], mx[1], mx[2], 0.0f);
        n.b.w = __builtin_bit_cast(float, ref);
    }
}

int FreeScene(pt_ctx *ctx) {
    FreeAll(ctx->scene_allocs);
    ctx->scene_ready = false;
    std::memset(&ctx->scene, 0, sizeof(ctx->scene));
    return PT_OK;
}

void FillCamera(DCamera &c, const pt_camera &h) {
    c.res_x = h.resolution[0];
    c.res_y = h.resolution[1];
    c.px = h.pixel_length[0];
    c.py = h.pixel_length[1];
    c.focal_length = h.focal_length;
    c.position = V3{h.position[0], h.position[1], h.position[2]};
    c.front = V3{h.front[0], h.front[1], h.front[2]};
    c.up = V3{h.up[0], h.up[1], h.up[2]};
    c.right = V3{h.right[0], h.right[1], h.right[2]};
    c.near_plane = h.near_plane;
    c.far_plane = h.far_plane;
}

// Makes ctx->stream wait for the render passes still running on the lanes' streams.
hipError_t Join(pt_ctx *ctx) {
    for (pt_ctx::Lane &l : ctx->lane) {
        if (!l.pending) continue;
        const hipError_t e = hipStreamWaitEvent(ctx->stream, l.done, 0);
        if (e != hipSuccess) return e;
        l.pending = false;
    }
    return hipSuccess;
}
// Host waits for everything the context has in flight.
hipError_t SyncAll(pt_ctx *ctx) {
    const hipError_t e = Join(ctx);
    if (e != hipSuccess) return e;
    return hipStreamSynchronize(ctx->stream);
}

int EnsureLaneSpill(pt_ctx *ctx, pt_ctx::Lane &l) {
    if (l.spill) return PT_OK;
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&l.spill), size_t(ctx->spill_stride) * kSpillRows * sizeof(uint2)));
    return PT_OK;
}

int EnsureScratch(pt_ctx *ctx) {
    if (ctx->cursor) return PT_OK;
    ctx->spill_stride = ctx->grid_blocks * kWave;
    const int st = EnsureLaneSpill(ctx, ctx->lane[0]);
    if (st != PT_OK) return st;
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->cursor), size_t(kMaxQueryCursors) * kCursorStride * sizeof(uint32_t)));
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->dev_stats), sizeof(DevStats)));
    HIP_TRY(ctx, hipMemsetAsync(ctx->dev_stats, 0, sizeof(DevStats), ctx->stream));
    HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void **>(&ctx->pinned_word), 64, hipHostMallocDefault));
    return PT_OK;
}

}  // namespace

// =========================================================================== API
extern "C" {

const char *pt_version(void) { return "ptamd 0.1 (gfx950)"; }

const char *pt_last_error(const pt_ctx *ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

int pt_ctx_create(int device, pt_ctx **out) {
    if (!out) return Fail(nullptr, PT_ERR_INVALID, "pt_ctx_create: out is null");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0) return Fail(nullptr, PT_ERR_DEVICE, "no HIP device available: %s", hipGetErrorString(e));
    if (device < 0 || device >= count) return Fail(nullptr, PT_ERR_INVALID, "device %d out of range (%d devices)", device, count);
    if (device >= kMaxDevices) return Fail(nullptr, PT_ERR_LIMIT, "device %d: at most %d devices are supported", device, kMaxDevices);
    pt_ctx *ctx = new pt_ctx();
    ctx->device = device;
    std::memset(&ctx->scene, 0, sizeof(ctx->scene));
    for (pt_ctx::Lane &l : ctx->lane) std::memset(&l.paths, 0, sizeof(l.paths));
    std::memset(&ctx->stats, 0, sizeof(ctx->stats));
    DeviceGuard guard(device);
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) {
        delete ctx;
        return Fail(nullptr, PT_ERR_DEVICE, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    }
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        delete ctx;
        return Fail(nullptr, PT_ERR_DEVICE, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    }
    ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    e = StaticScratch(device, &ctx->static_scratch);  // `device` is current here (DeviceGuard above)
    if (e != hipSuccess) {
        delete ctx;
        return Fail(nullptr, PT_ERR_DEVICE, "hipGetSymbolAddress(scratch): %s", hipGetErrorString(e));
    }
    // one wave per block; a wave's LDS (stack + ray stage) bounds the waves that fit a CU's 160 KiB
    int waves_per_cu = std::min(PT_WAVES_PER_CU, int((160 * 1024) / (kWaveLdsEntries * sizeof(uint2))));
    if (const char *e = getenv("PTAMD_WAVES_PER_CU")) waves_per_cu = std::max(1, std::min(32, atoi(e)));  // tuning aid
    ctx->grid_blocks = uint32_t(ctx->num_cus * waves_per_cu);
    if (const char *e = getenv("PTAMD_QUERY_CURSORS")) ctx->query_cursors = uint32_t(std::max(1, std::min(int(kMaxQueryCursors), atoi(e))));  // tuning aid
    // Two render passes run side by side, so each pass's kernels take a smaller share of a CU than a
    // lone query does.
    static const char *const kRenderEnv[3] = {"PTAMD_EXTEND_WAVES_PER_CU", "PTAMD_SHADE_WAVES_PER_CU", "PTAMD_SHADOW_WAVES_PER_CU"};
    static const int kRenderWaves[3] = {PT_EXTEND_WAVES_PER_CU, PT_SHADE_WAVES_PER_CU, PT_SHADOW_WAVES_PER_CU};
    for (int k = 0; k < 3; ++k) {
        int w = std::min(kRenderWaves[k], waves_per_cu);
        if (const char *e = getenv(kRenderEnv[k])) w = std::max(1, std::min(waves_per_cu, atoi(e)));  // tuning aid
        ctx->render_blocks[k] = uint32_t(ctx->num_cus * w);
    }
    e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete ctx;
        return Fail(nullptr, PT_ERR_DEVICE, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    ctx->stream = ctx->own_stream;
    bool ok = hipEventCreateWithFlags(&ctx->fork_event, hipEventDisableTiming) == hipSuccess;
    for (pt_ctx::Lane &l : ctx->lane) {
        ok = ok && hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&l.done, hipEventDisableTiming) == hipSuccess;
    }
    if (!ok) {
        pt_ctx_destroy(ctx);
        return Fail(nullptr, PT_ERR_DEVICE, "could not create the render streams");
    }
    if (const char *v = getenv("PTAMD_OVERLAP_PASSES")) ctx->overlap_passes = atoi(v) != 0;  // tuning aid
    *out = ctx;
    return PT_OK;
}

void pt_ctx_destroy(pt_ctx *ctx) {
    if (!ctx) return;
    DeviceGuard guard(ctx->device);
    (void)SyncAll(ctx);
    DrainEvents(ctx);
    for (EventPair &ev : ctx->free_events) {
        (void)hipEventDestroy(ev.start);
        (void)hipEventDestroy(ev.stop);
    }
    pt_integrator_destroy(ctx);
    FreeScene(ctx);
    for (pt_ctx::Lane &l : ctx->lane) {
        if (l.spill) (void)hipFree(l.spill);
        if (l.stream) (void)hipStreamDestroy(l.stream);
        if (l.done) (void)hipEventDestroy(l.done);
    }
    if (ctx->fork_event) (void)hipEventDestroy(ctx->fork_event);
    if (ctx->cursor) (void)hipFree(ctx->cursor);
    if (ctx->dev_stats) (void)hipFree(ctx->dev_stats);
    if (ctx->pinned_word) (void)hipHostFree(ctx->pinned_word);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

int pt_set_stream(pt_ctx *ctx, void *hip_stream) {
    if (!ctx) return PT_ERR_INVALID;
    DeviceGuard guard(ctx->device);
    HIP_TRY(ctx, SyncAll(ctx));
    ctx->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->own_stream;
    return PT_OK;
}

int pt_synchronize(pt_ctx *ctx) {
    if (!ctx) return PT_ERR_INVALID;
    DeviceGuard guard(ctx->device);
    HIP_TRY(ctx, SyncAll(ctx));
    return PT_OK;
}

int pt_set_profiling(pt_ctx *ctx, int enabled) {
    if (!ctx) return PT_ERR_INVALID;
    ctx->profiling = enabled != 0;
    return PT_OK;
}

#ifdef PT_PHASE_PROF
// Diagnostic build only: reads and clears the phase counters of the traversal loop.
int pt_debug_phase_prof(unsigned long long *out16) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(ptd::g_phase_prof), 16 * sizeof(unsigned long long)) != hipSuccess) return PT_ERR_DEVICE;
    unsigned long long zero[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(ptd::g_phase_prof), zero, sizeof(zero)) != hipSuccess) return PT_ERR_DEVICE;
    return PT_OK;
}
#endif

#ifdef PT_TRACE_STAMPS
// Diagnostic build only: reads and clears the clock totals of the traversal loop's parts.
int pt_debug_trace_stamps(unsigned long long *out8) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(ptd::g_trace_stamps), 8 * sizeof(unsigned long long)) != hipSuccess) return PT_ERR_DEVICE;
    unsigned long long zero[8] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(ptd::g_trace_stamps), zero, sizeof(zero)) != hipSuccess) return PT_ERR_DEVICE;
    return PT_OK;
}
#endif

#ifdef PT_NODE_HIST
// Diagnostic build only: reads and clears the per-node visit counts of the box loop.
int pt_debug_node_hist(unsigned *out, unsigned count) {
    if (count > ptd::kNodeHistSize) return PT_ERR_INVALID;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(ptd::g_node_hist), size_t(count) * sizeof(unsigned)) != hipSuccess) return PT_ERR_DEVICE;
    std::vector<unsigned> zero(ptd::kNodeHistSize, 0u);
    if (hipMemcpyToSymbol(HIP_SYMBOL(ptd::g_node_hist), zero.data(), zero.size() * sizeof(unsigned)) != hipSuccess) return PT_ERR_DEVICE;
    return PT_OK;
}
#endif

#ifdef PT_WAVE_TIMES
// Diagnostic build only: (start, end) clock pairs of the waves of the last closest-hit query.
int pt_debug_wave_times(unsigned long long *out, unsigned waves) {
    if (waves > 8192) return PT_ERR_INVALID;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(ptd::g_wave_times), size_t(waves) * 2 * sizeof(unsigned long long)) != hipSuccess) return PT_ERR_DEVICE;
    return PT_OK;
}
#endif

int pt_set_trace_mode(pt_ctx *ctx, int mode) {
    if (!ctx) return PT_ERR_INVALID;
    if (mode < 0 || mode > 3) return Fail(ctx, PT_ERR_INVALID, "trace mode %d (bit 0: any-hit culls boxes with t_max, bit 1: 4-wide BVH)", mode);
    ctx->trace_mode = mode;
    ctx->scene.any_hit_culls_with_tmax = (mode & 1) ? 1u : 0u;
    return PT_OK;
}

int pt_set_texture_filter(pt_ctx *ctx, int options) {
    if (!ctx) return PT_ERR_INVALID;
    if (options < 0 || options > 3) return Fail(ctx, PT_ERR_INVALID, "texture filter options %d (bit 0: 8-bit fixed-point weights, bit 1: sRGB decode of tagged images)", options);
    ctx->texture_filter = options;
    return PT_OK;
}

int pt_reset_stats(pt_ctx *ctx) {
    if (!ctx) return PT_ERR_INVALID;
    DeviceGuard guard(ctx->device);
    HIP_TRY(ctx, SyncAll(ctx));
    DrainEvents(ctx);
    std::memset(&ctx->stats, 0, sizeof(ctx->stats));
    if (ctx->dev_stats) HIP_TRY(ctx, hipMemsetAsync(ctx->dev_stats, 0, sizeof(DevStats), ctx->stream));
    return PT_OK;
}

int pt_get_stats(pt_ctx *ctx, pt_stats *out) {
    if (!ctx || !out) return PT_ERR_INVALID;
    DeviceGuard guard(ctx->device);
    HIP_TRY(ctx, SyncAll(ctx));
    DrainEvents(ctx);
    if (ctx->dev_stats) {
        DevStats ds;
        HIP_TRY(ctx, hipMemcpy(&ds, ctx->dev_stats, sizeof(ds), hipMemcpyDeviceToHost));
        ctx->stats.stack_spills = ds.spills;
        ctx->stats.stack_drops = ds.drops;
        ctx->stats.nan_pixels = ds.nan_pixels;
    }
    const BounceCounters *last_counters = ctx->lane[ctx->cur_lane].counters;
    if (ctx->integrator_ready && last_counters && ctx->last_render_bounces > 0) {
        std::vector<BounceCounters> bc(size_t(ctx->last_render_bounces) + 1);
        HIP_TRY(ctx, hipMemcpy(bc.data(), last_counters, bc.size() * sizeof(BounceCounters), hipMemcpyDeviceToHost));
        ctx->stats.closest_rays = 0;
        ctx->stats.shadow_rays = 0;
        ctx->stats.shaded_paths = 0;
        ctx->stats.bounces_run = 0;
        std::memset(ctx->stats.queue_sizes, 0, sizeof(ctx->stats.queue_sizes));
        // bounce 0 streams all path slots (8x8-tile padded); its real ray count is one per pixel
        bc[0].ext_count = uint32_t(ctx->win[2]) * uint32_t(ctx->win[3]) * uint32_t(ctx->last_render_samples);
        for (int b = 0; b < ctx->last_render_bounces; ++b) {
            ctx->stats.closest_rays += bc[b].ext_count;
            ctx->stats.shadow_rays += bc[b].shadow_count;
            for (int k = 0; k < kNumClasses; ++k) ctx->stats.shaded_paths += bc[b].class_count[k];
            if (bc[b].ext_count) ctx->stats.bounces_run = uint32_t(b + 1);
            if (b < PT_MAX_TRACKED_BOUNCES) ctx->stats.queue_sizes[b] = bc[b].ext_count;
        }
        ctx->stats.paths = bc[0].ext_count;
    }
    *out = ctx->stats;
    return PT_OK;
}

// ------------------------------------------------------------------ scene upload
int pt_scene_free(pt_ctx *ctx) {
    if (!ctx) return PT_ERR_INVALID;
    DeviceGuard guard(ctx->device);
    HIP_TRY(ctx, SyncAll(ctx));
    return FreeScene(ctx);
}

int pt_camera_update(pt_ctx *ctx, const pt_camera *camera) {
    if (!ctx || !camera) return PT_ERR_INVALID;
    if (!ctx->scene_ready) return Fail(ctx, PT_ERR_STATE, "pt_camera_update before pt_scene_upload");
    if (camera->resolution[0] <= 0 || camera->resolution[1] <= 0) return Fail(ctx, PT_ERR_INVALID, "camera resolution must be positive");
    FillCamera(ctx->scene.camera, *camera);
    return PT_OK;
}

static int UploadScene(pt_ctx *ctx, const pt_scene_desc *d);

int pt_scene_upload(pt_ctx *ctx, const pt_scene_desc *d) {
    if (!ctx) return PT_ERR_INVALID;
    try {
        return UploadScene(ctx, d);
    } catch (const std::bad_alloc &) {  // nothing may be thrown across the C boundary
        DeviceGuard guard(ctx->device);
        FreeScene(ctx);
        return Fail(ctx, PT_ERR_LIMIT, "out of host memory while re-packing the scene");
    }
}

static int UploadScene(pt_ctx *ctx, const pt_scene_desc *d) {
    DeviceGuard guard(ctx->device);
    int st = Validate(ctx, d);
    if (st != PT_OK) return st;
    HIP_TRY(ctx, SyncAll(ctx));
    FreeScene(ctx);
    st = EnsureScratch(ctx);
    if (st != PT_OK) return st;

    // which mesh (index range) does each BLAS describe?
    std::vector<int> blas_owner(d->blas_count, -1);
    for (uint32_t i = 0; i < d->geometry_count; ++i) {
        const pt_geometry_instance &g = d->geometries[i];
        int &owner = blas_owner[g.blas_index];
        if (owner < 0) {
            owner = int(i);
        } else {
            const pt_geometry_instance &o = d->geometries[owner];
            if (o.first_index != g.first_index || o.index_count != g.index_count)
                return Fail(ctx, PT_ERR_INVALID, "instances %d and %u share BLAS %d but not the mesh range", owner, i, g.blas_index);
        }
    }

    // ---- triangle records in leaf order
    std::vector<DTri> tris(d->tri_index_count);
    std::memset(tris.data(), 0, tris.size() * sizeof(DTri));
    // ... and, under the same record index, the vertex attributes a hit needs for shading: one
    // 64-byte record instead of three index reads and six scattered attribute reads
    std::vector<DTriShade> tri_shade(d->tri_index_count);
    std::memset(tri_shade.data(), 0, tri_shade.size() * sizeof(DTriShade));
    for (uint32_t b = 0; b < d->blas_count; ++b) {
        const pt_blas &blas = d->blases[b];
        if (blas.node_count == 0 || blas_owner[b] < 0) continue;
        const pt_geometry_instance &g = d->geometries[blas_owner[b]];
        for (uint32_t j = 0; j < blas.index_count; ++j) {
            const uint32_t tri = d->tri_indices[blas.first_index + j];
            if (tri >= g.index_count / 3) return Fail(ctx, PT_ERR_INVALID, "BLAS %u: triangle index %u out of range", b, tri);
            const float *p[3];
            for (int k = 0; k < 3; ++k) p[k] = d->positions + 3 * size_t(d->indices[g.first_index + tri * 3 + k]);
            DTri &r = tris[blas.first_index + j];
            r.a = make_float4(p[0][0], p[0][1], p[0][2], p[1][0]);
            r.b = make_float4(p[1][1], p[1][2], p[2][0], p[2][1]);
            r.c = make_float4(p[2][2], __builtin_bit_cast(float, tri), 0.0f, 0.0f);
            DTriShade &sh = tri_shade[blas.first_index + j];
            for (int k = 0; k < 3; ++k) {
                const size_t v = d->indices[g.first_index + tri * 3 + k];
                for (int c = 0; c < 3; ++c) sh.n[3 * k + c] = d->normals[3 * v + c];
                for (int c = 0; c < 2; ++c) sh.uv[2 * k + c] = d->uvs[2 * v + c];
            }
        }
    }

    // ---- interior nodes: TLAS first, then every BLAS
    std::vector<DNode> nodes;
    std::vector<uint2> big_leaves;
    std::vector<uint32_t> tlas_map(d->tlas_node_count, kNoRef);
    for (uint32_t i = 0; i < d->tlas_node_count; ++i)
        if (d->tlas_nodes[i].left_right != 0) {
            tlas_map[i] = uint32_t(nodes.size());
            nodes.emplace_back();
        }
    auto tlas_ref = [&](uint32_t i) -> uint32_t {
        const pt_tlas_node &n = d->tlas_nodes[i];
        return n.left_right == 0 ? (kLeafBit | n.instance) : tlas_map[i];
    };
    for (uint32_t i = 0; i < d->tlas_node_count; ++i) {
        const pt_tlas_node &n = d->tlas_nodes[i];
        if (n.left_right == 0) continue;
        const uint32_t l = n.left_right & 0xFFFFu, r = n.left_right >> 16;
        DNode &dn = nodes[tlas_map[i]];
        std::memset(&dn, 0, sizeof(dn));
        SetNodeChild(dn, 0, d->tlas_nodes[l].min, d->tlas_nodes[l].max, tlas_ref(l));
        SetNodeChild(dn, 1, d->tlas_nodes[r].min, d->tlas_nodes[r].max, tlas_ref(r));
    }
    const uint32_t root_ref = tlas_ref(0);

    std::vector<uint32_t> blas_root(d->blas_count, kNoRef);
    std::vector<uint32_t> node_map(d->blas_node_count, kNoRef);
    std::vector<uint32_t> leaf_ref(d->blas_node_count, kNoRef);  // reference word of every BLAS leaf, shared by both node layouts
    for (uint32_t b = 0; b < d->blas_count; ++b) {
        const pt_blas &blas = d->blases[b];
        if (blas.node_count == 0) continue;
        for (uint32_t k = blas.first_node; k < blas.first_node + blas.node_count; ++k)
            if (d->blas_nodes[k].primitive_count == 0) {
                node_map[k] = uint32_t(nodes.size());
                nodes.emplace_back();
            }
        for (uint32_t k = blas.first_node; k < blas.first_node + blas.node_count; ++k) {
            const pt_bvh_node &n = d->blas_nodes[k];
            if (n.primitive_count > 0) leaf_ref[k] = EncodeLeaf(blas.first_index + n.first_primitive, n.primitive_count, big_leaves);
        }
        auto ref_of = [&](uint32_t k) -> uint32_t { return d->blas_nodes[k].primitive_count > 0 ? leaf_ref[k] : node_map[k]; };
        for (uint32_t k = blas.first_node; k < blas.first_node + blas.node_count; ++k) {
            const pt_bvh_node &n = d->blas_nodes[k];
            if (n.primitive_count > 0) continue;
            const uint32_t l = n.first_primitive, r = l + 1;
            DNode &dn = nodes[node_map[k]];
            std::memset(&dn, 0, sizeof(dn));
            SetNodeChild(dn, 0, d->blas_nodes[l].min, d->blas_nodes[l].max, ref_of(l));
            SetNodeChild(dn, 1, d->blas_nodes[r].min, d->blas_nodes[r].max, ref_of(r));
        }
        blas_root[b] = ref_of(blas.first_node);
    }
    if (nodes.size() >= kNoRef) return Fail(ctx, PT_ERR_LIMIT, "too many BVH nodes");

    // ---- the same trees as 4-wide nodes with quantised boxes (trace mode 2)
    std::vector<DNode4> nodes4;
    struct BlasTree {
        const pt_bvh_node *n;
        const uint32_t *leaf_refs;
        bool is_leaf(uint32_t i) const { return n[i].primitive_count > 0; }
        uint32_t left(uint32_t i) const { return n[i].first_primitive; }
        uint32_t right(uint32_t i) const { return n[i].first_primitive + 1; }
        ptw::Box box(uint32_t i) const { return ptw::Box{{n[i].min[0], n[i].min[1], n[i].min[2]}, {n[i].max[0], n[i].max[1], n[i].max[2]}}; }
        uint32_t leaf_ref(uint32_t i) const { return leaf_refs[i]; }
    };
    // The top level is rebuilt over the reference's TLAS leaves (instance world boxes, bvh.cpp:273-280), each
    // clipped to the world box of the transformed mesh itself (wide_bvh.h, TightInstanceBox).
    struct MeshTree : BlasTree {
        const pt_bvh_node *nodes;
        const DTri *records;  // the BLAS's triangle records (object space), leaf order
        uint32_t first_index;
        void for_each_vertex(uint32_t i, const std::function<void(float, float, float)> &f) const {
            for (uint32_t k = 0; k < nodes[i].primitive_count; ++k) {
                const DTri &t = records[first_index + nodes[i].first_primitive + k];
                f(t.a.x, t.a.y, t.a.z);
                f(t.a.w, t.b.x, t.b.y);
                f(t.b.z, t.b.w, t.c.x);
            }
        }
    };
    const bool tight_boxes = !getenv("PTAMD_LOOSE_INSTANCE_BOXES");
    std::vector<ptw::Box> inst_boxes;
    std::vector<uint32_t> inst_ids;
    for (uint32_t i = 0; i < d->tlas_node_count; ++i) {
        const pt_tlas_node &n = d->tlas_nodes[i];
        if (n.left_right != 0 || (i == 0 && d->tlas_node_count > 1)) continue;  // interior node, or the root's copy of a lone leaf
        inst_boxes.push_back(ptw::Box{{n.min[0], n.min[1], n.min[2]}, {n.max[0], n.max[1], n.max[2]}});
        inst_ids.push_back(n.instance);
    }
    double scene_extent = 0.0;  // diagonal of the scene's root box
    for (int a = 0; a < 3; ++a) {
        const double e = double(d->tlas_nodes[0].max[a]) - double(d->tlas_nodes[0].min[a]);
        if (e == e && e > 0.0 && e < 1e30) scene_extent += e * e;
    }
    scene_extent = std::sqrt(scene_extent);
    if (tight_boxes) {
        // Six support searches per instance (≈ 35 µs for a 70 k-triangle mesh): shared among host threads when there
        // are thousands of instances.  Every search reads the scene only and writes its own box.
        auto clip = [&](size_t first, size_t step) {
            for (size_t k = first; k < inst_boxes.size(); k += step) {
                const pt_geometry_instance &g = d->geometries[inst_ids[k]];
                const pt_blas &blas = d->blases[g.blas_index];
                if (blas.node_count == 0) continue;
                const MeshTree mesh{{d->blas_nodes, leaf_ref.data()}, d->blas_nodes, tris.data(), blas.first_index};
                try {
                    inst_boxes[k] = ptw::TightInstanceBox(mesh, blas.first_node, g.transform, inst_boxes[k], scene_extent);
                } catch (const std::bad_alloc &) {
                    // the reference's box stays
                }
            }
        };
        const unsigned hw = std::thread::hardware_concurrency();
        const size_t workers = inst_boxes.size() < 512 ? 1 : std::min<size_t>(hw ? hw : 1, 16);
        std::vector<std::thread> pool;
        try {
            for (size_t t = 1; t < workers; ++t) pool.emplace_back(clip, t, workers);
        } catch (const std::system_error &) {
            // no more threads to be had: the ones that started keep their shares, the rest is done here
        }
        const size_t started = pool.size() + 1;
        clip(0, workers);
        for (size_t t = started; t < workers; ++t) clip(t, workers);
        for (std::thread &th : pool) th.join();
    }
    const ptw::TopLevel top = ptw::BuildTopLevel(inst_boxes.data(), uint32_t(inst_boxes.size()));
    struct TopTree {
        const ptw::TopLevel *t;
        const uint32_t *ids;
        bool is_leaf(uint32_t i) const { return t->nodes[i].left == 0xFFFFFFFFu; }
        uint32_t left(uint32_t i) const { return t->nodes[i].left; }
        uint32_t right(uint32_t i) const { return t->nodes[i].right; }
        ptw::Box box(uint32_t i) const { return t->nodes[i].box; }
        uint32_t leaf_ref(uint32_t i) const { return kLeafBit | ids[t->nodes[i].item]; }
    };
    const uint32_t root_ref4 = top.nodes.empty() ? root_ref : ptw::Collapse(TopTree{&top, inst_ids.data()}, 0u, nodes4);
    std::vector<uint32_t> blas_root4(d->blas_count, kNoRef);
    for (uint32_t b = 0; b < d->blas_count; ++b)
        if (d->blases[b].node_count > 0) blas_root4[b] = ptw::Collapse(BlasTree{d->blas_nodes, leaf_ref.data()}, d->blases[b].first_node, nodes4);
    std::vector<uint32_t> inst_root4(d->geometry_count);
    for (uint32_t i = 0; i < d->geometry_count; ++i) inst_root4[i] = blas_root4[d->geometries[i].blas_index];

    // ---- instances
    std::vector<DInst> insts(d->geometry_count);
    std::vector<DInstFwd> fwd(d->geometry_count);
    for (uint32_t i = 0; i < d->geometry_count; ++i) {
        const pt_geometry_instance &g = d->geometries[i];
        auto col = [](const float *m, int c, int r) { return m[c * 4 + r]; };
        const float *I = g.inv_transform, *T = g.transform, *N = g.inv_transpose;
        DInst &in = insts[i];
        in.a = make_float4(col(I, 0, 0), col(I, 0, 1), col(I, 0, 2), col(I, 1, 0));
        in.b = make_float4(col(I, 1, 1), col(I, 1, 2), col(I, 2, 0), col(I, 2, 1));
        in.c = make_float4(col(I, 2, 2), col(I, 3, 0), col(I, 3, 1), col(I, 3, 2));
        in.d = make_float4(__builtin_bit_cast(float, blas_root[g.blas_index]), __builtin_bit_cast(float, uint32_t(g.material_id)),
                           __builtin_bit_cast(float, uint32_t(d->materials[g.material_id].type)), __builtin_bit_cast(float, g.first_index));
        DInstFwd &f = fwd[i];
        f.t0 = make_float4(col(T, 0, 0), col(T, 0, 1), col(T, 0, 2), col(T, 1, 0));
        f.t1 = make_float4(col(T, 1, 1), col(T, 1, 2), col(T, 2, 0), col(T, 2, 1));
        f.t2 = make_float4(col(T, 2, 2), col(T, 3, 0), col(T, 3, 1), col(T, 3, 2));
        f.n0 = make_float4(col(N, 0, 0), col(N, 0, 1), col(N, 0, 2), col(N, 1, 0));
        f.n1 = make_float4(col(N, 1, 1), col(N, 1, 2), col(N, 2, 0), col(N, 2, 1));
        f.n2 = make_float4(col(N, 2, 2), col(N, 3, 0), col(N, 3, 1), col(N, 3, 2));
        f.m = make_float4(g.total_area, __builtin_bit_cast(float, uint32_t(g.area_light_id)),
                          __builtin_bit_cast(float, uint32_t(g.triangle_sampler_index)), __builtin_bit_cast(float, uint32_t(g.material_id)));
    }

    // ---- materials, textures, images
    std::vector<DMaterial> mats(d->material_count);
    for (uint32_t i = 0; i < d->material_count; ++i) {
        const pt_material &m = d->materials[i];
        mats[i].type = m.type;
        mats[i].tex0 = m.tex0;
        std::memcpy(&mats[i].w1, &m.w1, 4);
        mats[i].w2 = m.w2;
    }
    std::vector<DTex> tex1(d->texture1_count), tex3(d->texture3_count);
    for (uint32_t i = 0; i < d->texture1_count; ++i) {
        const pt_texture1 &t = d->textures1[i];
        DTex &o = tex1[i];
        std::memset(&o, 0, sizeof(o));
        o.type = t.type;
        o.image_id = -1;
        if (t.type == PT_TEX_CONSTANT) {
            o.c0[0] = t.u.constant.value;
        } else if (t.type == PT_TEX_IMAGE) {
            o.image_id = t.u.image.image_id;
            o.width = t.u.image.width;
            o.height = t.u.image.height;
            std::memcpy(o.scale, t.u.image.scale, 8);
            std::memcpy(o.offset, t.u.image.offset, 8);
        } else if (t.type == PT_TEX_CHECKERBOARD) {
            o.c0[0] = t.u.checkerboard.color0;
            o.c1[0] = t.u.checkerboard.color1;
            std::memcpy(o.scale, t.u.checkerboard.scale, 8);
            std::memcpy(o.offset, t.u.checkerboard.offset, 8);
        } else {
            return Fail(ctx, PT_ERR_INVALID, "scalar texture %u: unknown type", i);
        }
    }
    for (uint32_t i = 0; i < d->texture3_count; ++i) {
        const pt_texture3 &t = d->textures3[i];
        DTex &o = tex3[i];
        std::memset(&o, 0, sizeof(o));
        o.type = t.type;
        o.image_id = -1;
        if (t.type == PT_TEX_CONSTANT) {
            std::memcpy(o.c0, t.u.constant.value, 12);
        } else if (t.type == PT_TEX_IMAGE) {
            o.image_id = t.u.image.image_id;
            o.width = t.u.image.width;
            o.height = t.u.image.height;
            std::memcpy(o.scale, t.u.image.scale, 8);
            std::memcpy(o.offset, t.u.image.offset, 8);
        } else if (t.type == PT_TEX_CHECKERBOARD) {
            std::memcpy(o.c0, t.u.checkerboard.color0, 12);
            std::memcpy(o.c1, t.u.checkerboard.color1, 12);
            std::memcpy(o.scale, t.u.checkerboard.scale, 8);
            std::memcpy(o.offset, t.u.checkerboard.offset, 8);
        } else {
            return Fail(ctx, PT_ERR_INVALID, "colour texture %u: unknown type", i);
        }
    }
    DScene &S = ctx->scene;
    std::vector<DImage> img1(d->image1_count), img3(d->image3_count);
    for (uint32_t i = 0; i < d->image1_count; ++i) {
        const pt_image &im = d->images1[i];
        if (!im.data || im.width <= 0 || im.height <= 0 || im.channels != 1) return Fail(ctx, PT_ERR_INVALID, "scalar image %u is malformed", i);
        const float *dev = nullptr;
        st = Upload(ctx, ctx->scene_allocs, im.data, size_t(im.width) * im.height, &dev);
        if (st != PT_OK) return st;
        img1[i] = DImage{dev, im.width, im.height, ctx->texture_filter & kTexFixedPointWeights, 0};
    }
    for (uint32_t i = 0; i < d->image3_count; ++i) {
        const pt_image &im = d->images3[i];
        if (!im.data || im.width <= 0 || im.height <= 0 || im.channels != 3) return Fail(ctx, PT_ERR_INVALID, "colour image %u is malformed", i);
        std::vector<float4> rgba(size_t(im.width) * im.height);  // float4 texels as texture.cpp:40-53
        for (size_t k = 0; k < rgba.size(); ++k) rgba[k] = make_float4(im.data[3 * k], im.data[3 * k + 1], im.data[3 * k + 2], 1.0f);
        const float4 *dev = nullptr;
        st = Upload(ctx, ctx->scene_allocs, rgba.data(), rgba.size(), &dev);
        if (st != PT_OK) return st;
        img3[i] = DImage{dev, im.width, im.height, (ctx->texture_filter & kTexFixedPointWeights) | ((im.srgb && (ctx->texture_filter & kTexSrgbDecode)) ? kTexSrgbDecode : 0), 0};
    }
    for (DTex &t : tex1)
        if (t.type == PT_TEX_IMAGE && t.image_id >= 0 && uint32_t(t.image_id) < d->image1_count) {
            t.img_data = img1[t.image_id].data;
            t.img_width = img1[t.image_id].width;
            t.img_height = img1[t.image_id].height;
            t.filter = img1[t.image_id].filter;
        }
    for (DTex &t : tex3)
        if (t.type == PT_TEX_IMAGE && t.image_id >= 0 && uint32_t(t.image_id) < d->image3_count) {
            t.img_data = img3[t.image_id].data;
            t.img_width = img3[t.image_id].width;
            t.img_height = img3[t.image_id].height;
            t.filter = img3[t.image_id].filter;
        }

    // ---- alias tables
    std::vector<DAlias> tri_samplers(d->triangle_sampler_count);
    for (uint32_t i = 0; i < d->triangle_sampler_count; ++i) {
        const pt_alias_table &t = d->triangle_samplers[i];
        const float *p = nullptr;
        const uint32_t *a = nullptr;
        if ((st = Upload(ctx, ctx->scene_allocs, t.probability_table, t.size, &p)) != PT_OK) return st;
        if ((st = Upload(ctx, ctx->scene_allocs, t.alias_table, t.size, &a)) != PT_OK) return st;
        tri_samplers[i] = DAlias{p, a, t.size, 0};
    }
    S.light_sampler = DAlias{nullptr, nullptr, 0, 0};
    if (d->light_count > 0) {
        const float *p = nullptr;
        const uint32_t *a = nullptr;
        if ((st = Upload(ctx, ctx->scene_allocs, d->light_sampler->probability_table, d->light_sampler->size, &p)) != PT_OK) return st;
        if ((st = Upload(ctx, ctx->scene_allocs, d->light_sampler->alias_table, d->light_sampler->size, &a)) != PT_OK) return st;
        S.light_sampler = DAlias{p, a, d->light_sampler->size, 0};
    }

    // ---- light sampling records: what kernel/light.h:15-59 gathers per sample through
    // light -> instance -> alias table -> indices -> 9 vertex attributes is laid out per light as
    // one descriptor, one {probability, alias} pair per triangle and one 96-byte record per
    // triangle holding the world-space corners, transformed normals and uvs.  The transforms
    // are evaluated here with the device's operation order, so the values are the same bits.
    std::vector<DLight> dlights(d->light_count);
    std::vector<uint2> light_alias;
    std::vector<DLightTri> light_tris;
    for (uint32_t li = 0; li < d->light_count; ++li) {
        DLight &L = dlights[li];
        std::memset(&L, 0, sizeof(L));
        L.power = d->lights[li].power;
        L.gi = d->lights[li].geometry_instance_index;
        L.tex0 = -1;
        if (L.gi < 0) {
            L.gi = -1;
            continue;
        }
        const pt_geometry_instance &g = d->geometries[L.gi];
        if (g.triangle_sampler_index < 0 || uint32_t(g.triangle_sampler_index) >= d->triangle_sampler_count)
            return Fail(ctx, PT_ERR_INVALID, "light %u: instance %d has no triangle sampler", li, L.gi);
        const pt_alias_table &t = d->triangle_samplers[g.triangle_sampler_index];
        if (t.size != g.index_count / 3) return Fail(ctx, PT_ERR_INVALID, "light %u: sampler size %u differs from the mesh's %u triangles", li, t.size, g.index_count / 3);
        L.tri_base = uint32_t(light_tris.size());
        L.tri_count = t.size;
        L.total_area = g.total_area;
        L.tex0 = d->materials[g.material_id].tex0;
        std::memcpy(&L.emittance, &d->materials[g.material_id].w1, 4);
        const DInstFwd &f = fwd[L.gi];
        const float Tm[12] = {f.t0.x, f.t0.y, f.t0.z, f.t0.w, f.t1.x, f.t1.y, f.t1.z, f.t1.w, f.t2.x, f.t2.y, f.t2.z, f.t2.w};
        const float Nm[12] = {f.n0.x, f.n0.y, f.n0.z, f.n0.w, f.n1.x, f.n1.y, f.n1.z, f.n1.w, f.n2.x, f.n2.y, f.n2.z, f.n2.w};
        auto xf = [](const float *m, const float *v, float w, float *out) {  // xform(), dmath.h
            for (int c = 0; c < 3; ++c) {
                const float a = m[c] * v[0] + m[3 + c] * v[1];
                const float b = m[6 + c] * v[2] + m[9 + c] * w;
                out[c] = a + b;
            }
        };
        for (uint32_t tri = 0; tri < t.size; ++tri) {
            light_alias.push_back(make_uint2(__builtin_bit_cast(uint32_t, t.probability_table[tri]), t.alias_table[tri]));
            DLightTri r;
            for (int k = 0; k < 3; ++k) {
                const size_t v = d->indices[g.first_index + tri * 3 + k];
                xf(Tm, d->positions + 3 * v, 1.0f, r.p + 3 * k);
                xf(Nm, d->normals + 3 * v, 0.0f, r.n + 3 * k);
                r.uv[2 * k] = d->uvs[2 * v];
                r.uv[2 * k + 1] = d->uvs[2 * v + 1];
            }
            light_tris.push_back(r);
        }
    }

#define UP(field, vec)                                                                             \
    if ((st = Upload(ctx, ctx->scene_allocs, (vec).data(), (vec).size(), &S.field)) != PT_OK) return st
    UP(tri_shade, tri_shade);
    UP(dlights, dlights);
    UP(light_alias, light_alias);
    UP(light_tris, light_tris);
    if (nodes.empty()) nodes.emplace_back();  // lanes without work read record 0 of this array
    UP(nodes, nodes);
    if (nodes4.empty()) nodes4.emplace_back();
    UP(nodes4, nodes4);
    UP(inst_root4, inst_root4);
    UP(tris, tris);
    UP(insts, insts);
    UP(fwd, fwd);
    UP(big_leaves, big_leaves);
    UP(materials, mats);
    UP(tex1, tex1);
    UP(tex3, tex3);
    UP(images1, img1);
    UP(images3, img3);
    UP(tri_samplers, tri_samplers);
#undef UP
    if ((st = Upload(ctx, ctx->scene_allocs, d->indices, d->index_count, &S.indices)) != PT_OK) return st;
    if ((st = Upload(ctx, ctx->scene_allocs, d->positions, size_t(d->vertex_count) * 3, &S.positions)) != PT_OK) return st;
    if ((st = Upload(ctx, ctx->scene_allocs, d->normals, size_t(d->vertex_count) * 3, &S.normals)) != PT_OK) return st;
    if ((st = Upload(ctx, ctx->scene_allocs, d->uvs, size_t(d->vertex_count) * 2, &S.uvs)) != PT_OK) return st;
    if ((st = Upload(ctx, ctx->scene_allocs, d->lights, d->light_count, &S.lights)) != PT_OK) return st;
    S.tex1_count = d->texture1_count;
    S.tex3_count = d->texture3_count;
    S.image1_count = d->image1_count;
    S.image3_count = d->image3_count;
    S.light_count = d->light_count;
    S.material_count = d->material_count;
    S.geometry_count = d->geometry_count;
    S.root_ref = root_ref;
    S.root_ref4 = root_ref4;
    for (int a = 0; a < 3; ++a) {
        S.root_min[a] = d->tlas_nodes[0].min[a];
        S.root_max[a] = d->tlas_nodes[0].max[a];
    }
    S.max_depth = d->max_depth;
    S.any_hit_culls_with_tmax = (ctx->trace_mode & 1) ? 1u : 0u;
    S.total_light_power = d->total_light_power;
    FillCamera(S.camera, d->camera);
    ctx->scene_ready = true;
    return PT_OK;
}

// ------------------------------------------------------------------ integrator
// (Re)allocates a lane's per-path state for `samples` iterations in flight at the context's resolution.
static int EnsurePaths(pt_ctx *ctx, pt_ctx::Lane &l, uint32_t samples) {
    if (l.samples_alloc >= samples) return PT_OK;
    HIP_TRY(ctx, SyncAll(ctx));
    FreeAll(l.state_allocs);
    l.samples_alloc = 0;
    const uint32_t tiles = uint32_t((ctx->win[2] + 7) / 8) * uint32_t((ctx->win[3] + 7) / 8);
    const size_t per_sample = size_t(tiles) * 64;
    const size_t cap = per_sample * samples;
    if (cap >= (1ull << 31)) return Fail(ctx, PT_ERR_LIMIT, "%u samples in flight at %dx%d exceed 2^31 path slots", samples, ctx->win[2], ctx->win[3]);
    PathState &P = l.paths;
    std::memset(&P, 0, sizeof(P));
    int st = PT_OK;
    bool ok = true;
    if (const char *e = getenv("PTAMD_PATH_STATE_LIMIT_MB")) {  // testing aid: a smaller device, as far as path state goes
        size_t held = cap;
        for (const pt_ctx::Lane &o : ctx->lane) held += size_t(o.samples_alloc) * per_sample;
        if (double(held) * 184.0 > atof(e) * 1e6) {
            ok = false;
            st = Fail(ctx, PT_ERR_DEVICE, "PTAMD_PATH_STATE_LIMIT_MB=%s", e);
        }
    }
#define AL(field) ok = ok && (st = Alloc(ctx, l.state_allocs, cap, &P.field)) == PT_OK
    AL(ray_o); AL(ray_d); AL(thr); AL(accum); AL(last_pos); AL(rng); AL(hit); AL(sh_o); AL(sh_d); AL(sh_c); AL(sh_q);
    AL(ext_q[0]); AL(ext_q[1]);
    for (int k = 0; k < kNumClasses; ++k) AL(class_q[k]);
#undef AL
    if (!ok) {  // give back what was allocated and leave the lane empty, so that a smaller request can succeed
        FreeAll(l.state_allocs);
        std::memset(&P, 0, sizeof(P));
        (void)hipGetLastError();  // the failed allocation's error is reported here, not by a later launch check
        const std::string why = ctx->error;
        return Fail(ctx, st, "path state for %u samples per pass at %dx%d (%.1f GB for each pass in flight) does not fit: %s", samples, ctx->win[2],
                    ctx->win[3], double(cap) * 184.0 / 1e9, why.c_str());
    }
    P.capacity = uint32_t(cap);
    P.slots_per_sample = uint32_t(per_sample);
    P.win_x0 = ctx->win[0];
    P.win_y0 = ctx->win[1];
    P.win_w = ctx->win[2];
    P.win_h = ctx->win[3];
    l.samples_alloc = samples;
    return PT_OK;
}

int pt_integrator_destroy(pt_ctx *ctx) {
    if (!ctx) return PT_ERR_INVALID;
    DeviceGuard guard(ctx->device);
    (void)SyncAll(ctx);
    FreeAll(ctx->path_allocs);
    for (pt_ctx::Lane &l : ctx->lane) {
        FreeAll(l.state_allocs);
        l.samples_alloc = 0;
        l.counters = nullptr;
        l.counters_capacity = 0;
        std::memset(&l.paths, 0, sizeof(l.paths));
    }
    ctx->film = nullptr;
    ctx->integrator_ready = false;
    ctx->width = ctx->height = 0;
    for (bool &r : ctx->lane_retired) r = false;
    return PT_OK;
}

int pt_integrator_init(pt_ctx *ctx, int width, int height) {
    if (!ctx) return PT_ERR_INVALID;
    if (width <= 0 || height <= 0) return Fail(ctx, PT_ERR_INVALID, "resolution must be positive");
    if (uint64_t(width) * uint64_t(height) > (1ull << 28)) return Fail(ctx, PT_ERR_LIMIT, "more than 2^28 pixels");
    DeviceGuard guard(ctx->device);
    pt_integrator_destroy(ctx);
    int st = EnsureScratch(ctx);
    if (st != PT_OK) return st;
    if ((st = Alloc(ctx, ctx->path_allocs, size_t(width) * height * 3, &ctx->film)) != PT_OK) return st;
    HIP_TRY(ctx, hipMemsetAsync(ctx->film, 0, size_t(width) * height * 3 * sizeof(float), ctx->stream));
    ctx->width = width;
    ctx->height = height;
    ctx->win[0] = ctx->win[1] = 0;
    ctx->win[2] = width;
    ctx->win[3] = height;
    if ((st = EnsurePaths(ctx, ctx->lane[0], 1)) != PT_OK) return st;
    ctx->cur_lane = 0;
    ctx->integrator_ready = true;
    return PT_OK;
}

int pt_integrator_resize(pt_ctx *ctx, int width, int height) { return pt_integrator_init(ctx, width, height); }

int pt_set_render_window(pt_ctx *ctx, int x0, int y0, int w, int h) {
    if (!ctx) return PT_ERR_INVALID;
    if (!ctx->integrator_ready) return Fail(ctx, PT_ERR_STATE, "pt_set_render_window before pt_integrator_init");
    if (w == 0 && h == 0 && x0 == 0 && y0 == 0) {  // back to the whole frame
        w = ctx->width;
        h = ctx->height;
    }
    if (x0 < 0 || y0 < 0 || w <= 0 || h <= 0 || x0 > ctx->width - w || y0 > ctx->height - h)
        return Fail(ctx, PT_ERR_INVALID, "render window %d,%d %dx%d does not lie inside the %dx%d frame", x0, y0, w, h, ctx->width, ctx->height);
    if (x0 == ctx->win[0] && y0 == ctx->win[1] && w == ctx->win[2] && h == ctx->win[3]) return PT_OK;
    DeviceGuard guard(ctx->device);
    HIP_TRY(ctx, SyncAll(ctx));
    // path state is laid out over the window: every lane's is dropped and allocated again, for the new window, by its next pass
    for (pt_ctx::Lane &l : ctx->lane) {
        FreeAll(l.state_allocs);
        l.samples_alloc = 0;
        std::memset(&l.paths, 0, sizeof(l.paths));
    }
    for (bool &r : ctx->lane_retired) r = false;
    ctx->win[0] = x0;
    ctx->win[1] = y0;
    ctx->win[2] = w;
    ctx->win[3] = h;
    return EnsurePaths(ctx, ctx->lane[0], 1);
}

int pt_integrator_reset(pt_ctx *ctx, int width, int height) {
    if (!ctx) return PT_ERR_INVALID;
    if (!ctx->integrator_ready) return Fail(ctx, PT_ERR_STATE, "pt_integrator_reset before pt_integrator_init");
    if (width != ctx->width || height != ctx->height) return Fail(ctx, PT_ERR_INVALID, "reset resolution differs from the integrator's");
    DeviceGuard guard(ctx->device);
    HIP_TRY(ctx, Join(ctx));
    HIP_TRY(ctx, hipMemsetAsync(ctx->film, 0, size_t(width) * height * 3 * sizeof(float), ctx->stream));
    return PT_OK;
}

static int EnsureCounters(pt_ctx *ctx, pt_ctx::Lane &l, uint32_t bounces) {
    if (l.counters_capacity >= bounces + 2) return PT_OK;
    BounceCounters *c = nullptr;
    int st = Alloc(ctx, ctx->path_allocs, size_t(bounces) + 2, &c);
    if (st != PT_OK) return st;
    l.counters = c;
    l.counters_capacity = bounces + 2;
    return PT_OK;
}

// Runs generate + `bounces` bounce rounds of iteration `iteration` (no film pass) with lane `l`'s
// state on `stream`.
static int RunBounces(pt_ctx *ctx, pt_ctx::Lane &l, hipStream_t stream, int width, int height, int iteration, int samples, int bounces,
                      bool shade_last) {
    const DScene &S = ctx->scene;
    int st = EnsurePaths(ctx, l, uint32_t(samples));
    if (st != PT_OK) return st;
    if ((st = EnsureLaneSpill(ctx, l)) != PT_OK) return st;
    PathState &P = l.paths;
    st = EnsureCounters(ctx, l, S.max_depth);
    if (st != PT_OK) return st;
    HIP_TRY(ctx, hipMemsetAsync(l.counters, 0, size_t(l.counters_capacity) * sizeof(BounceCounters), stream));
    const uint32_t tiles = P.slots_per_sample / 64 * uint32_t(samples);
    const bool timed = stream == ctx->stream;  // kernel timing events bracket launches on ctx->stream only
    {
        ScopedEvent ev(timed ? ctx : nullptr, EV_GENERATE);
        hipLaunchKernelGGL(k_generate, dim3(tiles), dim3(kWave), 0, stream, S, P, l.counters, iteration, width, height);
    }
    const dim3 block(kWave);
    const bool wide = (ctx->trace_mode & 2) != 0;
    const dim3 grid_extend(ctx->render_blocks[0]), grid_shade(ctx->render_blocks[1]), grid_shadow(ctx->render_blocks[2]);
    ctx->last_render_bounces = bounces;
    for (int b = 0; b < bounces; ++b) {
        const bool last = (b == bounces - 1);
        if (last && !shade_last) break;
        {
            ScopedEvent ev(timed ? ctx : nullptr, EV_CLOSEST);
            if (wide) hipLaunchKernelGGL(k_extend<true>, grid_extend, block, 0, stream, S, P, l.counters, b, width, height, l.spill, ctx->spill_stride, ctx->dev_stats);
            else hipLaunchKernelGGL(k_extend<false>, grid_extend, block, 0, stream, S, P, l.counters, b, width, height, l.spill, ctx->spill_stride, ctx->dev_stats);
        }
        {
            ScopedEvent ev(timed ? ctx : nullptr, EV_SHADE);
            hipLaunchKernelGGL(k_shade, grid_shade, block, 0, stream, S, P, l.counters, b);
        }
        {
            ScopedEvent ev(timed ? ctx : nullptr, EV_ANY);
            if (wide) hipLaunchKernelGGL(k_shadow<true>, grid_shadow, block, 0, stream, S, P, l.counters, b, l.spill, ctx->spill_stride, ctx->dev_stats);
            else hipLaunchKernelGGL(k_shadow<false>, grid_shadow, block, 0, stream, S, P, l.counters, b, l.spill, ctx->spill_stride, ctx->dev_stats);
        }
        // Long depth limits: stop launching once the extension queue has drained.  Only a call that blocks
        // the host anyway (pt_render, the ray captures) reads the counter back for that; an asynchronous
        // pass is submitted whole, and the persistent kernels of a drained bounce return at once.
        if (ctx->host_may_block && (b & 7) == 7 && b + 1 < bounces) {
            HIP_TRY(ctx, hipMemcpyAsync(ctx->pinned_word, &l.counters[b + 1].ext_count, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            HIP_TRY(ctx, hipStreamSynchronize(stream));
            if (*ctx->pinned_word == 0) {
                ctx->last_render_bounces = b + 1;
                break;
            }
        }
    }
    HIP_TRY(ctx, hipGetLastError());
    return PT_OK;
}

int pt_render_batch(pt_ctx *ctx, int width, int height, int iteration, int samples, uint8_t *rgba_device) {
    if (!ctx) return PT_ERR_INVALID;
    if (samples < 1 || samples > 1024) return Fail(ctx, PT_ERR_LIMIT, "samples per pass must be in [1, 1024]");
    if (!ctx->scene_ready) return Fail(ctx, PT_ERR_STATE, "pt_render before pt_scene_upload");
    if (!ctx->integrator_ready) return Fail(ctx, PT_ERR_STATE, "pt_render before pt_integrator_init");
    if (width != ctx->width || height != ctx->height) return Fail(ctx, PT_ERR_INVALID, "render resolution %dx%d differs from the integrator's %dx%d", width, height, ctx->width, ctx->height);
    if (width != ctx->scene.camera.res_x || height != ctx->scene.camera.res_y)
        return Fail(ctx, PT_ERR_INVALID, "render resolution differs from the camera's");
    if (iteration < 0) return Fail(ctx, PT_ERR_INVALID, "iteration must be >= 0");
    DeviceGuard guard(ctx->device);
    // Serial mode runs on ctx->stream with lane 0.  Overlapped mode alternates the lanes: the pass
    // starts after everything submitted to ctx->stream so far, shares nothing mutable with the
    // pass on the other lane except the film, and takes its turn at the film after that pass.
    const bool overlap = ctx->overlap_passes && !ctx->profiling;
    // The next usable lane after the one that ran the previous pass.
    auto next_lane = [&]() {
        for (int k = 1; k <= PT_LANES; ++k) {
            const int cand = (ctx->cur_lane + k) % PT_LANES;
            if (!ctx->lane_retired[cand]) return cand;
        }
        return ctx->cur_lane;
    };
    int li = overlap ? next_lane() : 0;
    if (ctx->lane[li].samples_alloc < uint32_t(samples)) {
        // Path state is allocated before anything of the pass is submitted.  When it does not fit, the
        // lane is taken out of the rotation and the pass runs on a lane that already holds enough
        // (fewer passes in flight); the error is reported only when there is no such lane.
        int st = EnsurePaths(ctx, ctx->lane[li], uint32_t(samples));
        if (st != PT_OK && overlap) {
            int fallback = -1;
            for (int k = 1; k <= PT_LANES && fallback < 0; ++k) {
                const int cand = (ctx->cur_lane + k) % PT_LANES;
                if (cand != li && !ctx->lane_retired[cand] && ctx->lane[cand].samples_alloc >= uint32_t(samples)) fallback = cand;
            }
            if (fallback >= 0) {
                ctx->lane_retired[li] = true;
                li = fallback;
                st = PT_OK;
            }
        }
        if (st != PT_OK) return st;
    }
    pt_ctx::Lane &l = ctx->lane[li], &other = ctx->lane[ctx->cur_lane];  // `other` ran the previous pass
    hipStream_t stream = ctx->stream;
    if (overlap) {
        stream = l.stream;
        HIP_TRY(ctx, hipEventRecord(ctx->fork_event, ctx->stream));
        HIP_TRY(ctx, hipStreamWaitEvent(stream, ctx->fork_event, 0));
    } else {
        HIP_TRY(ctx, Join(ctx));
    }
    if (iteration == 0) {  // path_trace.cu:438-439
        if (overlap && other.pending) HIP_TRY(ctx, hipStreamWaitEvent(stream, other.done, 0));
        HIP_TRY(ctx, hipMemsetAsync(ctx->film, 0, size_t(width) * height * 3 * sizeof(float), stream));
    }
    int st = RunBounces(ctx, l, stream, width, height, iteration, samples, int(ctx->scene.max_depth), true);
    if (st != PT_OK) return st;
    if (overlap && other.pending) HIP_TRY(ctx, hipStreamWaitEvent(stream, other.done, 0));  // film passes in submission order
    {
        ScopedEvent ev(overlap ? nullptr : ctx, EV_FILM);
        hipLaunchKernelGGL(k_film, dim3(l.paths.slots_per_sample / 64), dim3(kWave), 0, stream, l.paths, ctx->film, width, height, samples, ctx->dev_stats);
    }
    if (rgba_device) {
        const uint32_t n = uint32_t(width) * uint32_t(height);
        hipLaunchKernelGGL(k_tonemap, dim3((n + 255) / 256), dim3(256), 0, stream, ctx->film, rgba_device, width, height, iteration + samples);
    }
    HIP_TRY(ctx, hipGetLastError());
    if (overlap) {
        HIP_TRY(ctx, hipEventRecord(l.done, stream));
        l.pending = true;
    }
    ctx->cur_lane = li;
    ctx->last_render_samples = samples;
    if (getenv("PTAMD_DEBUG_QUEUES")) {  // tuning aid: per-bounce queue sizes of this pass on stderr
        std::vector<BounceCounters> h(size_t(ctx->last_render_bounces) + 1);
        HIP_TRY(ctx, SyncAll(ctx));
        HIP_TRY(ctx, hipMemcpy(h.data(), l.counters, h.size() * sizeof(BounceCounters), hipMemcpyDeviceToHost));
        for (size_t b = 0; b < h.size(); ++b)
            fprintf(stderr, "[ptamd] bounce %zu: extend %u, classes %u %u %u %u %u, shadow %u\n", b, h[b].ext_count, h[b].class_count[0],
                    h[b].class_count[1], h[b].class_count[2], h[b].class_count[3], h[b].class_count[4], h[b].shadow_count);
    }
    return PT_OK;
}

int pt_render_async(pt_ctx *ctx, int width, int height, int iteration, uint8_t *rgba_device) {
    return pt_render_batch(ctx, width, height, iteration, 1, rgba_device);
}

int pt_render(pt_ctx *ctx, int width, int height, int iteration, uint8_t *rgba_device) {
    if (!ctx) return PT_ERR_INVALID;
    ctx->host_may_block = true;
    int st = pt_render_async(ctx, width, height, iteration, rgba_device);
    ctx->host_may_block = false;
    if (st != PT_OK) return st;
    DeviceGuard guard(ctx->device);
    HIP_TRY(ctx, SyncAll(ctx));  // path_trace.cu:481
    return PT_OK;
}

int pt_film_download(pt_ctx *ctx, float *rgb_host) {
    if (!ctx || !rgb_host) return PT_ERR_INVALID;
    if (!ctx->integrator_ready) return Fail(ctx, PT_ERR_STATE, "no film: integrator not initialised");
    DeviceGuard guard(ctx->device);
    HIP_TRY(ctx, Join(ctx));  // the copy must come after the film passes still running on the lanes' streams
    HIP_TRY(ctx, hipMemcpyAsync(rgb_host, ctx->film, size_t(ctx->width) * ctx->height * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, SyncAll(ctx));
    return PT_OK;
}

int pt_film_copy_to_device(pt_ctx *ctx, float *rgb_device) {
    if (!ctx || !rgb_device) return PT_ERR_INVALID;
    if (!ctx->integrator_ready) return Fail(ctx, PT_ERR_STATE, "no film: integrator not initialised");
    DeviceGuard guard(ctx->device);
    HIP_TRY(ctx, Join(ctx));
    HIP_TRY(ctx, hipMemcpyAsync(rgb_device, ctx->film, size_t(ctx->width) * ctx->height * 3 * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(ctx, SyncAll(ctx));
    return PT_OK;
}

int pt_film_copy_from_device(pt_ctx *ctx, const float *rgb_device) {
    if (!ctx || !rgb_device) return PT_ERR_INVALID;
    if (!ctx->integrator_ready) return Fail(ctx, PT_ERR_STATE, "no film: integrator not initialised");
    DeviceGuard guard(ctx->device);
    HIP_TRY(ctx, Join(ctx));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->film, rgb_device, size_t(ctx->width) * ctx->height * 3 * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(ctx, SyncAll(ctx));
    return PT_OK;
}

int pt_film_upload(pt_ctx *ctx, const float *rgb_host) {
    if (!ctx || !rgb_host) return PT_ERR_INVALID;
    if (!ctx->integrator_ready) return Fail(ctx, PT_ERR_STATE, "no film: integrator not initialised");
    DeviceGuard guard(ctx->device);
    HIP_TRY(ctx, Join(ctx));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->film, rgb_host, size_t(ctx->width) * ctx->height * 3 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, SyncAll(ctx));
    return PT_OK;
}

// ------------------------------------------------------------------ multi-GPU film reduce (RCCL)
// RCCL is bound at run time, the first time one of these entry points is called: a process that
// never reduces a film (one GPU) needs no librccl, and a process that already holds a copy (PyTorch
// brings its own) keeps using that one instead of loading a second.
namespace {
struct RcclId {
    char internal[128];  // ncclUniqueId
};
struct RcclApi {
    void *lib = nullptr;
    int (*GetUniqueId)(RcclId *) = nullptr;
    int (*CommInitRank)(void **, int, RcclId, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*Reduce)(const void *, void *, size_t, int, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string error;
};
std::mutex g_rccl_mutex;
RcclApi g_rccl;

const RcclApi *Rccl() {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.lib) return &g_rccl;
    static const char *const kNames[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    void *lib = nullptr;
    for (const char *n : kNames)
        if ((lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;  // a copy the process already holds
    for (size_t k = 0; !lib && k < sizeof(kNames) / sizeof(kNames[0]); ++k) lib = dlopen(kNames[k], RTLD_NOW | RTLD_LOCAL);
    if (!lib) {
        const char *why = dlerror();  // one call: dlerror() clears the message it returns
        g_rccl.error = std::string("librccl could not be loaded: ") + (why ? why : "not found");
        return nullptr;
    }
    RcclApi api;
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
    api.Reduce = reinterpret_cast<decltype(api.Reduce)>(dlsym(lib, "ncclReduce"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.Reduce || !api.GetErrorString) {
        g_rccl.error = "librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclReduce / ncclGetErrorString";
        dlclose(lib);
        return nullptr;
    }
    api.lib = lib;
    g_rccl = api;
    return &g_rccl;
}
constexpr int kNcclFloat = 7, kNcclSum = 0;  // ncclFloat32, ncclSum (rccl.h)
}  // namespace

int pt_rccl_unique_id(pt_ctx *ctx, void *id128) {
    if (!id128) return Fail(ctx, PT_ERR_INVALID, "pt_rccl_unique_id: null buffer");
    const RcclApi *r = Rccl();
    if (!r) return Fail(ctx, PT_ERR_DEVICE, "%s", g_rccl.error.c_str());
    const int rc = r->GetUniqueId(static_cast<RcclId *>(id128));
    if (rc != 0) return Fail(ctx, PT_ERR_DEVICE, "ncclGetUniqueId: %s", r->GetErrorString(rc));
    return PT_OK;
}

int pt_rccl_comm_create(pt_ctx *ctx, int nranks, const void *id128, int rank, void **comm_out) {
    if (!ctx || !id128 || !comm_out) return Fail(ctx, PT_ERR_INVALID, "pt_rccl_comm_create: null argument");
    *comm_out = nullptr;
    if (nranks < 1 || rank < 0 || rank >= nranks) return Fail(ctx, PT_ERR_INVALID, "pt_rccl_comm_create: rank %d outside %d ranks", rank, nranks);
    const RcclApi *r = Rccl();
    if (!r) return Fail(ctx, PT_ERR_DEVICE, "%s", g_rccl.error.c_str());
    DeviceGuard guard(ctx->device);  // the communicator belongs to the context's GPU
    RcclId id;
    std::memcpy(&id, id128, sizeof(id));
    const int rc = r->CommInitRank(comm_out, nranks, id, rank);
    if (rc != 0) return Fail(ctx, PT_ERR_DEVICE, "ncclCommInitRank: %s", r->GetErrorString(rc));
    return PT_OK;
}

int pt_rccl_comm_destroy(pt_ctx *ctx, void *comm) {
    if (!comm) return PT_OK;
    const RcclApi *r = Rccl();
    if (!r) return Fail(ctx, PT_ERR_DEVICE, "%s", g_rccl.error.c_str());
    const int rc = r->CommDestroy(comm);
    if (rc != 0) return Fail(ctx, PT_ERR_DEVICE, "ncclCommDestroy: %s", r->GetErrorString(rc));
    return PT_OK;
}

// film(root) = sum over ranks of film(rank), in place, on the context's stream after the passes in flight.
int pt_film_reduce(pt_ctx *ctx, void *comm, int root) {
    if (!ctx || !comm) return Fail(ctx, PT_ERR_INVALID, "pt_film_reduce: null argument");
    if (!ctx->integrator_ready) return Fail(ctx, PT_ERR_STATE, "no film: integrator not initialised");
    const RcclApi *r = Rccl();
    if (!r) return Fail(ctx, PT_ERR_DEVICE, "%s", g_rccl.error.c_str());
    DeviceGuard guard(ctx->device);
    HIP_TRY(ctx, Join(ctx));
    const int rc = r->Reduce(ctx->film, ctx->film, size_t(ctx->width) * ctx->height * 3, kNcclFloat, kNcclSum, root, comm, ctx->stream);
    if (rc != 0) return Fail(ctx, PT_ERR_DEVICE, "ncclReduce: %s", r->GetErrorString(rc));
    return PT_OK;
}

// The displayed image of the context's film after `iterations` samples, in host memory.
int pt_display_download(pt_ctx *ctx, int iterations, uint8_t *rgba_host) {
    if (!ctx || !rgba_host) return PT_ERR_INVALID;
    if (!ctx->integrator_ready) return Fail(ctx, PT_ERR_STATE, "no film: integrator not initialised");
    if (iterations <= 0) return Fail(ctx, PT_ERR_INVALID, "pt_display_download: iterations must be positive");
    DeviceGuard guard(ctx->device);
    const size_t bytes = size_t(ctx->width) * ctx->height * 4;
    uint8_t *staging = nullptr;
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&staging), bytes));
    int st = pt_tonemap(ctx, nullptr, ctx->width, ctx->height, iterations, staging);
    hipError_t e = hipSuccess;
    if (st == PT_OK) e = hipMemcpy(rgba_host, staging, bytes, hipMemcpyDeviceToHost);
    (void)hipFree(staging);
    if (e != hipSuccess) return Fail(ctx, PT_ERR_DEVICE, "pt_display_download: %s", hipGetErrorString(e));
    return st;
}

int pt_tonemap(pt_ctx *ctx, const float *film_device, int width, int height, int iterations, uint8_t *rgba_device) {
    if (!ctx || !rgba_device) return PT_ERR_INVALID;
    if (width <= 0 || height <= 0 || iterations <= 0) return Fail(ctx, PT_ERR_INVALID, "pt_tonemap: bad size or iteration count");
    if (!film_device) {
        if (!ctx->integrator_ready || width != ctx->width || height != ctx->height) return Fail(ctx, PT_ERR_STATE, "pt_tonemap: no matching film");
        film_device = ctx->film;
    }
    DeviceGuard guard(ctx->device);
    HIP_TRY(ctx, Join(ctx));  // the context's film may still be receiving samples on a lane's stream
    const uint32_t n = uint32_t(width) * uint32_t(height);
    hipLaunchKernelGGL(k_tonemap, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, film_device, rgba_device, width, height, iterations);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, SyncAll(ctx));
    return PT_OK;
}

// ------------------------------------------------------------------ batch queries
static int CheckQuery(pt_ctx *ctx, const void *a, const void *b, size_t n) {
    if (!ctx) return PT_ERR_INVALID;
    if (!ctx->scene_ready) return Fail(ctx, PT_ERR_STATE, "ray query before pt_scene_upload");
    if (n > 0 && (!a || !b)) return Fail(ctx, PT_ERR_INVALID, "ray query: null buffer");
    if (n >= (1ull << 31)) return Fail(ctx, PT_ERR_LIMIT, "ray query: more than 2^31-1 rays per call");
    if ((reinterpret_cast<uintptr_t>(a) & 15u) != 0) return Fail(ctx, PT_ERR_INVALID, "ray buffer must be 16-byte aligned");
    return PT_OK;
}

int pt_trace_closest(pt_ctx *ctx, const pt_ray *rays, size_t n, pt_hit *hits) {
    int st = CheckQuery(ctx, rays, hits, n);
    if (st != PT_OK || n == 0) return st;
    if ((reinterpret_cast<uintptr_t>(hits) & 15u) != 0) return Fail(ctx, PT_ERR_INVALID, "hit buffer must be 16-byte aligned");
    DeviceGuard guard(ctx->device);
    HIP_TRY(ctx, Join(ctx));  // lane 0's spill area is shared with render passes
    HIP_TRY(ctx, hipMemsetAsync(ctx->cursor, 0, size_t(ctx->query_cursors) * kCursorStride * sizeof(uint32_t), ctx->stream));
    const uint32_t blocks = uint32_t(std::min<size_t>(ctx->grid_blocks, (n + kWave - 1) / kWave));
    {
        ScopedEvent ev(ctx, EV_CLOSEST);
        if (ctx->trace_mode & 2)
            hipLaunchKernelGGL((k_trace_closest<false, true>), dim3(blocks), dim3(kWave), 0, ctx->stream, ctx->scene, rays, uint32_t(n), hits, ctx->cursor, ctx->query_cursors,
                               ctx->lane[0].spill, ctx->spill_stride, ctx->dev_stats, static_cast<TraceCounters *>(nullptr));
        else
            hipLaunchKernelGGL((k_trace_closest<false, false>), dim3(blocks), dim3(kWave), 0, ctx->stream, ctx->scene, rays, uint32_t(n), hits, ctx->cursor, ctx->query_cursors,
                               ctx->lane[0].spill, ctx->spill_stride, ctx->dev_stats, static_cast<TraceCounters *>(nullptr));
    }
    HIP_TRY(ctx, hipGetLastError());
    return PT_OK;
}

// pt_trace_closest with every piece of scratch in this device's copy of the static array: no
// allocation, no event creation, nothing shared with the render passes (hence no join with them).
int pt_trace_closest_static(pt_ctx *ctx, const pt_ray *rays, size_t n, pt_hit *hits) {
    int st = CheckQuery(ctx, rays, hits, n);
    if (st != PT_OK) return st;
    if (n > PT_STATIC_MAX_RAYS) return Fail(ctx, PT_ERR_LIMIT, "pt_trace_closest_static: %zu rays exceed the limit of %zu per call", n, size_t(PT_STATIC_MAX_RAYS));
    if (n == 0) return PT_OK;
    if ((reinterpret_cast<uintptr_t>(hits) & 15u) != 0) return Fail(ctx, PT_ERR_INVALID, "hit buffer must be 16-byte aligned");
    DeviceGuard guard(ctx->device);
    unsigned char *base = ctx->static_scratch;
    uint32_t *cursors = reinterpret_cast<uint32_t *>(base);
    DevStats *stats = reinterpret_cast<DevStats *>(base + kStaticStatsOffset);
    uint2 *spill = reinterpret_cast<uint2 *>(base + kStaticSpillOffset);
    // cursors and counters start at zero; the spill area is written before it is read
    HIP_TRY(ctx, hipMemsetAsync(base, 0, kStaticSpillOffset, ctx->stream));
    const uint32_t blocks = uint32_t(std::min<size_t>(std::min<uint32_t>(ctx->grid_blocks, kStaticWaves), (n + kWave - 1) / kWave));
    if (ctx->trace_mode & 2)
        hipLaunchKernelGGL((k_trace_closest<false, true>), dim3(blocks), dim3(kWave), 0, ctx->stream, ctx->scene, rays, uint32_t(n), hits, cursors, ctx->query_cursors, spill,
                           kStaticWaves * uint32_t(kWave), stats, static_cast<TraceCounters *>(nullptr));
    else
        hipLaunchKernelGGL((k_trace_closest<false, false>), dim3(blocks), dim3(kWave), 0, ctx->stream, ctx->scene, rays, uint32_t(n), hits, cursors, ctx->query_cursors, spill,
                           kStaticWaves * uint32_t(kWave), stats, static_cast<TraceCounters *>(nullptr));
    HIP_TRY(ctx, hipGetLastError());
    return PT_OK;
}

static int TraceAny(pt_ctx *ctx, const pt_ray *rays, size_t n, uint8_t *occluded, uint32_t *counters4) {
    int st = CheckQuery(ctx, rays, occluded, n);
    if (st != PT_OK || n == 0) return st;
    DeviceGuard guard(ctx->device);
    HIP_TRY(ctx, Join(ctx));  // lane 0's spill area is shared with render passes
    HIP_TRY(ctx, hipMemsetAsync(ctx->cursor, 0, size_t(ctx->query_cursors) * kCursorStride * sizeof(uint32_t), ctx->stream));
    const uint32_t blocks = uint32_t(std::min<size_t>(ctx->grid_blocks, (n + kWave - 1) / kWave));
    TraceCounters *ctr = reinterpret_cast<TraceCounters *>(counters4);
    const bool wide = (ctx->trace_mode & 2) != 0;
#define PT_LAUNCH_ANY(W, C)                                                                                                                     \
    hipLaunchKernelGGL((k_trace_any<W, C>), dim3(blocks), dim3(kWave), 0, ctx->stream, ctx->scene, rays, uint32_t(n), occluded, ctx->cursor, \
                       ctx->query_cursors, ctx->lane[0].spill, ctx->spill_stride, ctx->dev_stats, ctr)
    {
        ScopedEvent ev(ctx, EV_ANY);
        if (wide && ctr) PT_LAUNCH_ANY(true, true);
        else if (wide) PT_LAUNCH_ANY(true, false);
        else if (ctr) PT_LAUNCH_ANY(false, true);
        else PT_LAUNCH_ANY(false, false);
    }
#undef PT_LAUNCH_ANY
    HIP_TRY(ctx, hipGetLastError());
    return PT_OK;
}

int pt_trace_any(pt_ctx *ctx, const pt_ray *rays, size_t n, uint8_t *occluded) { return TraceAny(ctx, rays, n, occluded, nullptr); }

// Visit counters of an any-hit query, as pt_trace_closest_counted has them for closest hits.
int pt_trace_any_counted(pt_ctx *ctx, const pt_ray *rays, size_t n, uint8_t *occluded, uint32_t *counters4) {
    if (ctx && !counters4) return Fail(ctx, PT_ERR_INVALID, "null counter buffer");
    return TraceAny(ctx, rays, n, occluded, counters4);
}

int pt_resolve_hits(pt_ctx *ctx, const pt_ray *rays, const pt_hit *hits, size_t n, pt_intersection *out) {
    int st = CheckQuery(ctx, rays, hits, n);
    if (st != PT_OK || n == 0) return st;
    if (!out) return Fail(ctx, PT_ERR_INVALID, "pt_resolve_hits: null output");
    DeviceGuard guard(ctx->device);
    hipLaunchKernelGGL(k_resolve_hits, dim3(uint32_t((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->scene, rays, hits, uint32_t(n), out);
    HIP_TRY(ctx, hipGetLastError());
    return PT_OK;
}

// Visit counters per ray (interior nodes, TLAS leaves, triangles, BLAS leaves): the counting
// build of the closest-hit kernel, used to compare traversal work with a CPU traversal.
int pt_trace_closest_counted(pt_ctx *ctx, const pt_ray *rays, size_t n, pt_hit *hits, uint32_t *counters4) {
    int st = CheckQuery(ctx, rays, hits, n);
    if (st != PT_OK || n == 0) return st;
    if (!counters4) return Fail(ctx, PT_ERR_INVALID, "null counter buffer");
    DeviceGuard guard(ctx->device);
    HIP_TRY(ctx, Join(ctx));  // lane 0's spill area is shared with render passes
    HIP_TRY(ctx, hipMemsetAsync(ctx->cursor, 0, size_t(ctx->query_cursors) * kCursorStride * sizeof(uint32_t), ctx->stream));
    const uint32_t blocks = uint32_t(std::min<size_t>(ctx->grid_blocks, (n + kWave - 1) / kWave));
    if (ctx->trace_mode & 2)
        hipLaunchKernelGGL((k_trace_closest<true, true>), dim3(blocks), dim3(kWave), 0, ctx->stream, ctx->scene, rays, uint32_t(n), hits, ctx->cursor, ctx->query_cursors, ctx->lane[0].spill,
                           ctx->spill_stride, ctx->dev_stats, reinterpret_cast<TraceCounters *>(counters4));
    else
        hipLaunchKernelGGL((k_trace_closest<true, false>), dim3(blocks), dim3(kWave), 0, ctx->stream, ctx->scene, rays, uint32_t(n), hits, ctx->cursor, ctx->query_cursors, ctx->lane[0].spill,
                           ctx->spill_stride, ctx->dev_stats, reinterpret_cast<TraceCounters *>(counters4));
    HIP_TRY(ctx, hipGetLastError());
    return PT_OK;
}

// DebugIntegrator::Render, kernel/integrators/debug.cu:198-239: both views in one streamed pass.
int pt_debug_render(pt_ctx *ctx, int width, int height, const pt_integrator_state *state, float *image_device, uint8_t *rgba_device) {
    if (!ctx) return PT_ERR_INVALID;
    if (!ctx->scene_ready) return Fail(ctx, PT_ERR_STATE, "pt_debug_render before pt_scene_upload");
    if (!state) return Fail(ctx, PT_ERR_INVALID, "pt_debug_render: null state");
    if (width <= 0 || height <= 0 || width != ctx->scene.camera.res_x || height != ctx->scene.camera.res_y)
        return Fail(ctx, PT_ERR_INVALID, "pt_debug_render: %dx%d does not match the camera's %dx%d", width, height, ctx->scene.camera.res_x,
                    ctx->scene.camera.res_y);
    if (state->debug_mode < PT_DEBUG_NONE || state->debug_mode > PT_DEBUG_ROUGHNESS) return Fail(ctx, PT_ERR_INVALID, "pt_debug_render: unknown debug_mode %d", state->debug_mode);
    const bool bvh = state->visualize_bvh != 0 && state->debug_mode == PT_DEBUG_NONE;
    if (bvh) {
        if (state->bvh_visualize_type < PT_BVH_BOUNDS_HIT || state->bvh_visualize_type > PT_BVH_TRIANGLES_HIT)
            return Fail(ctx, PT_ERR_INVALID, "pt_debug_render: unknown bvh_visualize_type %d", state->bvh_visualize_type);
        if (state->bvh_visualize_type != PT_BVH_TRIANGLES_HIT && state->bvh_visualize_depth <= 0)
            return Fail(ctx, PT_ERR_INVALID, "pt_debug_render: bvh_visualize_depth must be positive");
    }
    if (!bvh && state->debug_mode == PT_DEBUG_NONE) return PT_OK;  // the reference's Render writes nothing either
    if (!image_device && !rgba_device) return PT_OK;
    if (rgba_device && (reinterpret_cast<uintptr_t>(rgba_device) & 3u) != 0) return Fail(ctx, PT_ERR_INVALID, "rgba buffer must be 4-byte aligned");
    DeviceGuard guard(ctx->device);
    HIP_TRY(ctx, Join(ctx));  // lane 0's spill area is shared with render passes
    HIP_TRY(ctx, hipMemsetAsync(ctx->cursor, 0, size_t(ctx->query_cursors) * kCursorStride * sizeof(uint32_t), ctx->stream));
    const uint32_t tiles = (uint32_t(width + 7) >> 3) * (uint32_t(height + 7) >> 3);
    const uint32_t n = tiles * 64u;
    const uint32_t blocks = std::min<uint32_t>(ctx->grid_blocks, tiles);
    const DebugParams p{state->visualize_bvh, state->bvh_visualize_type, state->bvh_visualize_depth, state->debug_mode, width, height};
    hipLaunchKernelGGL(k_debug_view, dim3(blocks), dim3(kWave), 0, ctx->stream, ctx->scene, p, image_device, rgba_device, n, ctx->cursor, ctx->query_cursors,
                       ctx->lane[0].spill, ctx->spill_stride, ctx->dev_stats);
    HIP_TRY(ctx, hipGetLastError());
    return PT_OK;
}

// ------------------------------------------------------------------ device BVH build
struct pt_bvh {
    ptb::BuildResult r;
};

// TLAS::Build / BLAS::Build (bvh.cpp:5-332) on the device; host arrays in, host arrays out.
int pt_bvh_build(pt_ctx *ctx, const float *positions, uint32_t vertex_count, const uint32_t *indices, uint32_t index_count,
                 const pt_geometry_instance *geometries, uint32_t geometry_count, pt_bvh **out) {
    if (!ctx) return PT_ERR_INVALID;
    if (!out) return Fail(ctx, PT_ERR_INVALID, "pt_bvh_build: null output");
    *out = nullptr;
    if (!positions || !indices || !geometries) return Fail(ctx, PT_ERR_INVALID, "pt_bvh_build: null input array");
    DeviceGuard guard(ctx->device);
    pt_bvh *h = new pt_bvh();
    std::string err;
    const int st = ptb::BuildOnDevice(ctx->stream, positions, vertex_count, indices, index_count, geometries, geometry_count, h->r, err);
    if (st != PT_OK) {
        delete h;
        return Fail(ctx, st, "pt_bvh_build: %s", err.c_str());
    }
    *out = h;
    return PT_OK;
}
int pt_bvh_sizes(const pt_bvh *bvh, uint32_t *tlas_nodes, uint32_t *blases, uint32_t *blas_nodes, uint32_t *tri_indices) {
    if (!bvh) return PT_ERR_INVALID;
    if (tlas_nodes) *tlas_nodes = uint32_t(bvh->r.tlas_nodes.size());
    if (blases) *blases = uint32_t(bvh->r.blases.size());
    if (blas_nodes) *blas_nodes = uint32_t(bvh->r.blas_nodes.size());
    if (tri_indices) *tri_indices = uint32_t(bvh->r.tri_indices.size());
    return PT_OK;
}
int pt_bvh_get(const pt_bvh *bvh, pt_tlas_node *tlas_nodes, pt_blas *blases, pt_bvh_node *blas_nodes, uint32_t *tri_indices) {
    if (!bvh) return PT_ERR_INVALID;
    if (tlas_nodes) std::memcpy(tlas_nodes, bvh->r.tlas_nodes.data(), bvh->r.tlas_nodes.size() * sizeof(pt_tlas_node));
    if (blases) std::memcpy(blases, bvh->r.blases.data(), bvh->r.blases.size() * sizeof(pt_blas));
    if (blas_nodes) std::memcpy(blas_nodes, bvh->r.blas_nodes.data(), bvh->r.blas_nodes.size() * sizeof(pt_bvh_node));
    if (tri_indices) std::memcpy(tri_indices, bvh->r.tri_indices.data(), bvh->r.tri_indices.size() * sizeof(uint32_t));
    return PT_OK;
}
int pt_bvh_timing(const pt_bvh *bvh, float *ms_blas, float *ms_tlas, uint32_t *levels) {
    if (!bvh) return PT_ERR_INVALID;
    if (ms_blas) *ms_blas = bvh->r.ms_blas;
    if (ms_tlas) *ms_tlas = bvh->r.ms_tlas;
    if (levels) *levels = bvh->r.levels;
    return PT_OK;
}
void pt_bvh_free(pt_bvh *bvh) { delete bvh; }

// The shadow rays (next-event estimation) that the shading of `bounce` emits, with their t_max.
int pt_capture_shadow_rays(pt_ctx *ctx, int width, int height, int iteration, int bounce, pt_ray *rays_device, size_t capacity, size_t *count) {
    if (!ctx || !count) return PT_ERR_INVALID;
    *count = 0;
    if (!ctx->scene_ready || !ctx->integrator_ready) return Fail(ctx, PT_ERR_STATE, "capture needs an uploaded scene and an initialised integrator");
    if (width != ctx->width || height != ctx->height) return Fail(ctx, PT_ERR_INVALID, "capture resolution differs from the integrator's");
    if (bounce < 0 || uint32_t(bounce) >= ctx->scene.max_depth) return Fail(ctx, PT_ERR_INVALID, "bounce %d outside [0, max_depth)", bounce);
    if (!rays_device || (reinterpret_cast<uintptr_t>(rays_device) & 15u)) return Fail(ctx, PT_ERR_INVALID, "ray buffer must be a 16-byte aligned device pointer");
    DeviceGuard guard(ctx->device);
    HIP_TRY(ctx, Join(ctx));
    pt_ctx::Lane &l = ctx->lane[0];
    ctx->cur_lane = 0;
    ctx->host_may_block = true;
    int st = RunBounces(ctx, l, ctx->stream, width, height, iteration, 1, bounce + 1, true);
    ctx->host_may_block = false;
    ctx->last_render_samples = 1;
    if (st != PT_OK) return st;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->pinned_word, &l.counters[bounce].shadow_count, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, SyncAll(ctx));
    const size_t produced = *ctx->pinned_word;
    if (produced > capacity) return Fail(ctx, PT_ERR_LIMIT, "capture buffer holds %zu rays, %zu needed", capacity, produced);
    if (produced > 0) {
        hipLaunchKernelGGL(k_gather_shadow_rays, dim3(uint32_t((produced + 255) / 256)), dim3(256), 0, ctx->stream, l.paths, l.counters, bounce, rays_device,
                           uint32_t(capacity));
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, SyncAll(ctx));
    }
    *count = produced;
    return PT_OK;
}

}  // extern "C"

template <typename In, typename Out, typename F>
static int Staged(pt_ctx *ctx, const In *in, size_t n, Out *out, F &&launch) {
    if (!ctx) return PT_ERR_INVALID;
    if (n == 0) return PT_OK;
    if (!in || !out) return Fail(ctx, PT_ERR_INVALID, "null host buffer");
    DeviceGuard guard(ctx->device);
    In *din = nullptr;
    Out *dout = nullptr;
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&din), n * sizeof(In)));
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&dout), n * sizeof(Out));
    if (e != hipSuccess) {
        (void)hipFree(din);
        return Fail(ctx, PT_ERR_DEVICE, "hipMalloc failed: %s", hipGetErrorString(e));
    }
    int st = PT_OK;
    e = hipMemcpyAsync(din, in, n * sizeof(In), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) st = launch(din, dout);
    if (e == hipSuccess && st == PT_OK) e = hipMemcpyAsync(out, dout, n * sizeof(Out), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(din);
    (void)hipFree(dout);
    if (e != hipSuccess) return Fail(ctx, PT_ERR_DEVICE, "staged query failed: %s", hipGetErrorString(e));
    return st;
}

extern "C" {

int pt_trace_closest_host(pt_ctx *ctx, const pt_ray *rays, size_t n, pt_hit *hits) {
    return Staged(ctx, rays, n, hits, [&](pt_ray *dr, pt_hit *dh) { return pt_trace_closest(ctx, dr, n, dh); });
}
int pt_trace_any_host(pt_ctx *ctx, const pt_ray *rays, size_t n, uint8_t *occluded) {
    return Staged(ctx, rays, n, occluded, [&](pt_ray *dr, uint8_t *d_out) { return pt_trace_any(ctx, dr, n, d_out); });
}
int pt_resolve_hits_host(pt_ctx *ctx, const pt_ray *rays, const pt_hit *hits, size_t n, pt_intersection *out) {
    if (!ctx) return PT_ERR_INVALID;
    if (n == 0) return PT_OK;
    if (!rays || !hits || !out) return Fail(ctx, PT_ERR_INVALID, "null host buffer");
    DeviceGuard guard(ctx->device);
    pt_hit *dh = nullptr;
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&dh), n * sizeof(pt_hit)));
    hipError_t e = hipMemcpyAsync(dh, hits, n * sizeof(pt_hit), hipMemcpyHostToDevice, ctx->stream);
    int st = PT_OK;
    if (e == hipSuccess)
        st = Staged(ctx, rays, n, out, [&](pt_ray *dr, pt_intersection *d_out) { return pt_resolve_hits(ctx, dr, dh, n, d_out); });
    (void)hipFree(dh);
    if (e != hipSuccess) return Fail(ctx, PT_ERR_DEVICE, "hipMemcpyAsync failed: %s", hipGetErrorString(e));
    return st;
}

int pt_generate_incoherent_rays(pt_ctx *ctx, pt_ray *rays_device, size_t n, uint64_t first_index) {
    if (!ctx) return PT_ERR_INVALID;
    if (!ctx->scene_ready) return Fail(ctx, PT_ERR_STATE, "ray generation before pt_scene_upload");
    if (n == 0) return PT_OK;
    if (!rays_device || (reinterpret_cast<uintptr_t>(rays_device) & 15u)) return Fail(ctx, PT_ERR_INVALID, "ray buffer must be a 16-byte aligned device pointer");
    if (n >= (1ull << 31)) return Fail(ctx, PT_ERR_LIMIT, "more than 2^31-1 rays per call");
    DeviceGuard guard(ctx->device);
    hipLaunchKernelGGL(k_gen_incoherent, dim3(uint32_t((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->scene, rays_device, uint32_t(n), first_index);
    HIP_TRY(ctx, hipGetLastError());
    return PT_OK;
}

int pt_capture_bounce_rays(pt_ctx *ctx, int width, int height, int iteration, int bounce, pt_ray *rays_device, size_t capacity, size_t *count) {
    if (!ctx || !count) return PT_ERR_INVALID;
    *count = 0;
    if (!ctx->scene_ready || !ctx->integrator_ready) return Fail(ctx, PT_ERR_STATE, "capture needs an uploaded scene and an initialised integrator");
    if (width != ctx->width || height != ctx->height) return Fail(ctx, PT_ERR_INVALID, "capture resolution differs from the integrator's");
    if (bounce < 0 || uint32_t(bounce) >= ctx->scene.max_depth) return Fail(ctx, PT_ERR_INVALID, "bounce %d outside [0, max_depth)", bounce);
    if (!rays_device || (reinterpret_cast<uintptr_t>(rays_device) & 15u)) return Fail(ctx, PT_ERR_INVALID, "ray buffer must be a 16-byte aligned device pointer");
    DeviceGuard guard(ctx->device);
    HIP_TRY(ctx, Join(ctx));
    pt_ctx::Lane &l = ctx->lane[0];
    ctx->cur_lane = 0;
    ctx->host_may_block = true;
    int st = RunBounces(ctx, l, ctx->stream, width, height, iteration, 1, bounce + 1, false);
    ctx->host_may_block = false;
    ctx->last_render_samples = 1;
    if (st != PT_OK) return st;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->pinned_word, &l.counters[bounce].ext_count, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, SyncAll(ctx));
    // bounce 0 streams every path slot (padded to 8x8 tiles) and writes the render window's camera rays in pixel order
    const size_t launch_n = *ctx->pinned_word;
    const size_t produced = bounce == 0 ? size_t(ctx->win[2]) * size_t(ctx->win[3]) : launch_n;
    if (produced > capacity) return Fail(ctx, PT_ERR_LIMIT, "capture buffer holds %zu rays, %zu needed", capacity, produced);
    if (launch_n > 0) {
        hipLaunchKernelGGL(k_gather_rays, dim3(uint32_t((launch_n + 255) / 256)), dim3(256), 0, ctx->stream, l.paths, l.counters, bounce, rays_device,
                           uint32_t(capacity), width, height);
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, SyncAll(ctx));
    }
    *count = produced;
    return PT_OK;
}

}  // extern "C"
