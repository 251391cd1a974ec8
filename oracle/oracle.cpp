// TEST INFRASTRUCTURE - CPU oracle for the ray-scene hot path.
//
// A scalar restatement of the reference's algorithms (file:line citations are
// relative to the reference's src/ directory).  Only tests/, the smoke check and
// the cpu_baseline leg of bench.py may load this library; the product never does.
//
// parity unpinned, except for two pieces: the reference holds no tests, golden vectors
// or fixtures, and the code of this path cannot be compiled in this image (its glm
// dependency is absent and may not be replaced by a stand-in), so the arithmetic here
// was not checked against reference output.  The two exceptions are the files of the
// reference that compile unmodified on their own: PCG_Hash (math/hash.h) and Stack
// (utils/stack.h) below are checked against known answers produced by those headers
// (oracle/ref_probe.cpp -> tests/golden/ref_pins.npz, tests/test_ref_pins.py).  What
// pins the rest: analytic known answers, the brute-force all-triangles search (the
// reference's own cross-check, IntersectSceneNaive) and structural BVH invariants,
// all in tests/.
//
// Where the reference relies on hardware this target lacks (CUDA texture units),
// the definition used here is stated at the function.
#include <algorithm>
#include <array>
#include <atomic>
#include <climits>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <queue>
#include <thread>
#include <string>
#include <vector>

#include "../include/ptamd.h"
#include "ora_math.h"

namespace ora {

// ---------------------------------------------------------------- constants
// math/constants.h:4-13
constexpr float SMALL_EPSILON = 1e-9f;
constexpr float EPSILON = 1e-5f;
constexpr float PI = 3.14159265358979323846f;
constexpr float INV_PI = 0.31830988618379067154f;
constexpr float INV_SQRT3 = 0.57735026919;

struct Ray {  // math/ray.h:8-21
    vec3 origin, direction;
};
inline Ray xform_ray(const mat4 &m, const Ray &r) {  // math/ray.h:18-21
    return {xform_point(m, r.origin), xform_dir(m, r.direction)};
}

struct AABB {  // math/aabb.h:8-64
    vec3 min{FLT_MAX, FLT_MAX, FLT_MAX};
    vec3 max{-FLT_MAX, -FLT_MAX, -FLT_MAX};
    void Reset() { *this = AABB(); }
    void AddPoint(vec3 p) { min = vmin(min, p); max = vmax(max, p); }
    void Union(const AABB &o) { min = vmin(min, o.min); max = vmax(max, o.max); }
    bool IsEmpty() const { return (min.x > max.x) || (min.y > max.y) || (min.z > max.z); }
    vec3 Extent() const { return max - min; }
    float SurfaceArea() const {
        vec3 d = max - min;
        d = vmax(d, V3(0.0f));
        return 2.0f * (d.x * d.y + d.y * d.z + d.z * d.x);
    }
    AABB Transform(const mat4 &t) const {  // math/aabb.h:50-63
        AABB out;
        for (int c = 0; c < 8; ++c) {
            vec3 p = {(c & 1) ? max.x : min.x, (c & 2) ? max.y : min.y, (c & 4) ? max.z : min.z};
            out.AddPoint(xform_point(t, p));
        }
        return out;
    }
};
static inline AABB box_of(const float *mn, const float *mx) {
    AABB b;
    b.min = {mn[0], mn[1], mn[2]};
    b.max = {mx[0], mx[1], mx[2]};
    return b;
}

// ------------------------------------------------- primitive intersection
// math/intersection.h:9-35
static inline bool IntersectAABB(const Ray &ray, const AABB &aabb, float &t_out, float t_max) {
    const vec3 inv_d = 1.0f / ray.direction;
    const vec3 origin = ray.origin;
    float t0 = (aabb.min.x - origin.x) * inv_d.x;
    float t1 = (aabb.max.x - origin.x) * inv_d.x;
    float tnear = fminf(t0, t1);
    float tfar = fmaxf(t0, t1);
    t0 = (aabb.min.y - origin.y) * inv_d.y;
    t1 = (aabb.max.y - origin.y) * inv_d.y;
    tnear = fmaxf(tnear, fminf(t0, t1));
    tfar = fminf(tfar, fmaxf(t0, t1));
    t0 = (aabb.min.z - origin.z) * inv_d.z;
    t1 = (aabb.max.z - origin.z) * inv_d.z;
    tnear = fmaxf(tnear, fminf(t0, t1));
    tfar = fminf(tfar, fmaxf(t0, t1));
    const bool hit = (tfar >= tnear) && (tnear < t_max) && (tfar > 0.0f);
    if (hit) t_out = tnear;
    return hit;
}

// math/intersection.h:37-71
static inline bool IntersectTriangle(const Ray &ray, vec3 v0, vec3 v1, vec3 v2, vec2 &uv,
                                     float &t_out, float t_max) {
    t_out = -1.0f;
    vec3 edge1 = v1 - v0;
    vec3 edge2 = v2 - v0;
    vec3 p = cross(ray.direction, edge2);
    float a = dot(edge1, p);
    if (fabsf(a) < SMALL_EPSILON) return false;
    float inv_a = 1.0f / a;
    vec3 tvec = ray.origin - v0;
    float u = dot(tvec, p) * inv_a;
    if (u < 0.0f || u > 1.0f) return false;
    vec3 q = cross(tvec, edge1);
    float v = dot(ray.direction, q) * inv_a;
    if (v < 0.0f || u + v > 1.0f) return false;
    t_out = dot(edge2, q) * inv_a;
    if (t_out <= SMALL_EPSILON) return false;
    uv = {u, v};
    if (t_out > t_max) return false;
    return true;
}

// ------------------------------------------------------------------ scene
struct Image {
    const float *data;
    int w, h, ch;
};
struct Scene {
    const pt_scene_desc *d;
    const vec3 *positions, *normals;
    const vec2 *uvs;
    const uint32_t *indices;
    const pt_geometry_instance *geometries;
};
static inline Scene make_scene(const pt_scene_desc *d) {
    Scene s;
    s.d = d;
    s.positions = reinterpret_cast<const vec3 *>(d->positions);
    s.normals = reinterpret_cast<const vec3 *>(d->normals);
    s.uvs = reinterpret_cast<const vec2 *>(d->uvs);
    s.indices = d->indices;
    s.geometries = d->geometries;
    return s;
}
static inline const mat4 &M(const float *m) { return *reinterpret_cast<const mat4 *>(m); }

struct Intersection {  // kernel/intersection.h:10-17 (+ prim / barycentrics for id parity)
    float time;
    vec3 pos, normal;
    vec2 uv;
    int material_id = -1;
    int geometry_id = -1;
    int prim = -1;
    vec2 bary{0, 0};
};

struct StackItem {  // kernel/intersection.h:19-22
    uint32_t node_index;
    float aabb_time;
};
struct Stack {  // utils/stack.h:3-22
    StackItem data[64];
    uint32_t top = 0;
    uint32_t *drops;
    uint32_t *maxdepth;
    void Push(StackItem it) {
        if (top < 64) {
            data[top++] = it;
            if (maxdepth && top > *maxdepth) *maxdepth = top;
        } else if (drops) {
            ++*drops;
        }
    }
    StackItem Pop() {
        if (top > 0) return data[--top];
        return {};
    }
    bool Empty() const { return top == 0; }
};

struct Counters {
    uint64_t interior = 0;     // interior nodes expanded (TLAS + BLAS)
    uint64_t tlas_leaves = 0;  // TLAS leaves entered
    uint64_t tris = 0;         // triangles tested
    uint64_t blas_leaves = 0;  // BLAS leaves visited
    uint64_t tlas_interior = 0;  // the TLAS's share of `interior`
    uint32_t drops = 0;        // pushes dropped at depth 64
    uint32_t max_depth = 0;    // deepest stack seen (either level)
    // when set: the ray's traversal as a string of steps (ora_trace_ops), one character per step:
    // B interior node expanded, T triangle tested, E instance entered, L instance left / ray finished,
    // C stack entry popped and skipped because its box starts beyond the best hit
    std::string *ops = nullptr;
};

enum Mode { CLOSEST, ANY };

// kernel/intersection.h:30-152
template <Mode mode>
static bool IntersectBLAS(const Scene &scene, const pt_geometry_instance &geometry, const Ray &ray,
                          Intersection &intersection, float t_max, Counters *ctr) {
    const pt_scene_desc &d = *scene.d;
    if (d.blas_node_count <= 0) {
        intersection.time = -1.0f;
        return false;
    }
    const pt_blas &blas = d.blases[geometry.blas_index];
    Stack stack;
    stack.drops = ctr ? &ctr->drops : nullptr;
    stack.maxdepth = ctr ? &ctr->max_depth : nullptr;
    stack.Push({blas.first_node, 0.0f});
    const pt_bvh_node *nodes = d.blas_nodes;
    const Ray local_ray = xform_ray(M(geometry.inv_transform), ray);

    float best_time = FLT_MAX;
    vec2 best_uv{0, 0}, best_bary{0, 0};
    vec3 best_pos{0, 0, 0}, best_normal{0, 0, 0};
    int best_prim = -1;

    while (!stack.Empty()) {
        StackItem cur = stack.Pop();
        const pt_bvh_node &node = nodes[cur.node_index];
        if (mode == CLOSEST) {
            if (cur.aabb_time >= best_time) {
                if (ctr && ctr->ops) ctr->ops->push_back('C');
                continue;
            }
        }
        if (node.primitive_count > 0) {
            if (ctr) ctr->blas_leaves++;
            for (uint32_t i = 0; i < node.primitive_count; ++i) {
                uint32_t tri_index = d.tri_indices[blas.first_index + node.first_primitive + i];
                uint32_t base_index = geometry.first_index + tri_index * 3u;
                uint32_t i0 = scene.indices[base_index + 0];
                uint32_t i1 = scene.indices[base_index + 1];
                uint32_t i2 = scene.indices[base_index + 2];
                vec3 p0 = scene.positions[i0], p1 = scene.positions[i1], p2 = scene.positions[i2];
                vec2 bary;
                float time;
                if (ctr) ctr->tris++;
                if (ctr && ctr->ops) ctr->ops->push_back('T');
                if (!IntersectTriangle(local_ray, p0, p1, p2, bary, time, t_max)) continue;
                bool take = (mode == CLOSEST) ? (time < best_time) : true;
                if (take) {
                    best_time = time;
                    vec3 n0 = scene.normals[i0], n1 = scene.normals[i1], n2 = scene.normals[i2];
                    vec2 uv0 = scene.uvs[i0], uv1 = scene.uvs[i1], uv2 = scene.uvs[i2];
                    float w = 1.0f - bary.x - bary.y;
                    best_normal = normalize(w * n0 + bary.x * n1 + bary.y * n2);
                    best_uv = w * uv0 + bary.x * uv1 + bary.y * uv2;
                    best_pos = w * p0 + bary.x * p1 + bary.y * p2;
                    best_prim = int(tri_index);
                    best_bary = bary;
                }
                if (mode == ANY) break;  // leaves only the leaf loop (kernel/intersection.h:109)
            }
        } else {
            if (ctr) ctr->interior++;
            if (ctr && ctr->ops) ctr->ops->push_back('B');
            uint32_t left = node.first_primitive;
            uint32_t right = left + 1;
            float left_time, right_time;
            bool hit_left = IntersectAABB(local_ray, box_of(nodes[left].min, nodes[left].max), left_time, best_time);
            bool hit_right = IntersectAABB(local_ray, box_of(nodes[right].min, nodes[right].max), right_time, best_time);
            if (hit_left && hit_right) {
                if (left_time < right_time) {
                    stack.Push({right, right_time});
                    stack.Push({left, left_time});
                } else {
                    stack.Push({left, left_time});
                    stack.Push({right, right_time});
                }
            } else if (hit_left) {
                stack.Push({left, left_time});
            } else if (hit_right) {
                stack.Push({right, right_time});
            }
        }
    }
    if (best_time < 0.0f || best_time == FLT_MAX) {
        intersection.time = -1.0f;
        return false;
    }
    intersection.pos = xform_point(M(geometry.transform), best_pos);
    intersection.uv = best_uv;
    intersection.time = dot(intersection.pos - ray.origin, ray.direction);
    intersection.normal = normalize(xform_dir(M(geometry.inv_transpose), best_normal));
    intersection.material_id = geometry.material_id;
    intersection.prim = best_prim;
    intersection.bary = best_bary;
    return true;
}

// kernel/intersection.h:154-235
template <Mode mode>
static bool IntersectScene(const Scene &scene, const Ray &ray, Intersection &intersection,
                           float t_max, Counters *ctr) {
    const pt_scene_desc &d = *scene.d;
    if (d.tlas_node_count <= 0) {
        intersection.time = -1.0f;
        return false;
    }
    const pt_tlas_node *nodes = d.tlas_nodes;
    {
        float t = 0;
        if (!IntersectAABB(ray, box_of(nodes[0].min, nodes[0].max), t, t_max)) {
            intersection.time = -1.0f;
            return false;
        }
    }
    Stack stack;
    stack.drops = ctr ? &ctr->drops : nullptr;
    stack.maxdepth = ctr ? &ctr->max_depth : nullptr;
    stack.Push({0, 0.0f});
    float best_time = FLT_MAX;
    while (!stack.Empty()) {
        StackItem cur = stack.Pop();
        const pt_tlas_node &node = nodes[cur.node_index];
        if (mode == CLOSEST) {
            if (cur.aabb_time >= best_time) {
                if (ctr && ctr->ops) ctr->ops->push_back('C');
                continue;
            }
        }
        if (node.left_right == 0u) {
            if (ctr) ctr->tlas_leaves++;
            if (ctr && ctr->ops) ctr->ops->push_back('E');
            const pt_geometry_instance &geometry = scene.geometries[node.instance];
            Intersection temp;
            bool hit = IntersectBLAS<mode>(scene, geometry, ray, temp, t_max, ctr);
            if (ctr && ctr->ops) ctr->ops->push_back('L');
            temp.geometry_id = int(node.instance);
            if (mode == CLOSEST) {
                if (temp.time < best_time && hit) {
                    best_time = temp.time;
                    intersection = temp;
                }
            } else if (hit) {
                best_time = temp.time;
                intersection = temp;
                break;
            }
        } else {
            if (ctr) { ctr->interior++; ctr->tlas_interior++; }
            if (ctr && ctr->ops) ctr->ops->push_back('B');
            uint32_t left = node.left_right & 0xFFFFu;
            uint32_t right = node.left_right >> 16;
            float left_time, right_time;
            bool hit_left = IntersectAABB(ray, box_of(nodes[left].min, nodes[left].max), left_time, best_time);
            bool hit_right = IntersectAABB(ray, box_of(nodes[right].min, nodes[right].max), right_time, best_time);
            if (hit_left && hit_right) {
                if (left_time < right_time) {
                    stack.Push({right, right_time});
                    stack.Push({left, left_time});
                } else {
                    stack.Push({left, left_time});
                    stack.Push({right, right_time});
                }
            } else if (hit_left) {
                stack.Push({left, left_time});
            } else if (hit_right) {
                stack.Push({right, right_time});
            }
        }
    }
    if (best_time < 0.0f || best_time == FLT_MAX) {
        intersection.time = -1.0f;
        return false;
    }
    return true;
}

// kernel/intersection.h:237-314 (closest only; reports distance() as time like the original)
static bool IntersectSceneNaive(const Scene &scene, const Ray &ray, Intersection &intersection, float t_max) {
    const pt_scene_desc &d = *scene.d;
    float best_time = FLT_MAX;
    vec3 best_pos{0, 0, 0}, best_normal{0, 0, 0};
    vec2 best_uv{0, 0}, best_bary{0, 0};
    int best_geom = -1, best_prim = -1;
    for (uint32_t gi = 0; gi < d.geometry_count; ++gi) {
        const pt_geometry_instance &geom = scene.geometries[gi];
        Ray local_ray = xform_ray(M(geom.inv_transform), ray);
        uint32_t start = geom.first_index, end = start + geom.index_count;
        for (uint32_t idx = start; idx < end; idx += 3) {
            uint32_t i0 = scene.indices[idx], i1 = scene.indices[idx + 1], i2 = scene.indices[idx + 2];
            vec3 p0 = scene.positions[i0], p1 = scene.positions[i1], p2 = scene.positions[i2];
            vec2 bary;
            float t;
            if (!IntersectTriangle(local_ray, p0, p1, p2, bary, t, t_max)) continue;
            if (t < best_time) {
                best_time = t;
                float w = 1.0f - bary.x - bary.y;
                best_normal = normalize(w * scene.normals[i0] + bary.x * scene.normals[i1] + bary.y * scene.normals[i2]);
                best_uv = w * scene.uvs[i0] + bary.x * scene.uvs[i1] + bary.y * scene.uvs[i2];
                best_pos = w * p0 + bary.x * p1 + bary.y * p2;
                best_geom = int(gi);
                best_prim = int((idx - start) / 3);
                best_bary = bary;
            }
        }
    }
    if (best_time == FLT_MAX || best_geom < 0) {
        intersection.time = -1.0f;
        return false;
    }
    const pt_geometry_instance &g = scene.geometries[best_geom];
    intersection.pos = xform_point(M(g.transform), best_pos);
    intersection.uv = best_uv;
    intersection.time = distance(intersection.pos, ray.origin);
    intersection.normal = normalize(xform_dir(M(g.inv_transpose), best_normal));
    intersection.material_id = g.material_id;
    intersection.geometry_id = best_geom;
    intersection.prim = best_prim;
    intersection.bary = best_bary;
    return true;
}

// ---------------------------------------------------------------- BVH build
struct Primitive {  // bvh.h:16-19
    AABB aabb;
    vec3 centroid;
};
struct Bin {  // bvh.h:11-14
    AABB bounds{};
    uint32_t prims{};
};

struct BLASBuilder {  // bvh.cpp:5-204 (SAH strategy, the only reachable one)
    std::vector<pt_bvh_node> nodes;
    std::vector<uint32_t> tri_indices;
    std::vector<Primitive> prims;

    static AABB B(const pt_bvh_node &n) { return box_of(n.min, n.max); }
    static void setB(pt_bvh_node &n, const AABB &b) {
        n.min[0] = b.min.x; n.min[1] = b.min.y; n.min[2] = b.min.z;
        n.max[0] = b.max.x; n.max[1] = b.max.y; n.max[2] = b.max.z;
    }

    void Build(const vec3 *positions, const uint32_t *indices, uint32_t first_index, uint32_t index_count) {
        nodes.clear();
        tri_indices.clear();
        const uint32_t primitive_count = index_count / 3u;
        if (primitive_count == 0) return;
        nodes.reserve(2 * primitive_count - 1);
        tri_indices.resize(primitive_count);
        for (uint32_t i = 0; i < primitive_count; ++i) tri_indices[i] = i;
        prims.resize(primitive_count);
        for (uint32_t t = 0; t < primitive_count; ++t) {
            vec3 v0 = positions[indices[first_index + t * 3 + 0]];
            vec3 v1 = positions[indices[first_index + t * 3 + 1]];
            vec3 v2 = positions[indices[first_index + t * 3 + 2]];
            prims[t].centroid = (v0 + v1 + v2) * (1.0f / 3.0f);
            prims[t].aabb.Reset();
            prims[t].aabb.AddPoint(v0);
            prims[t].aabb.AddPoint(v1);
            prims[t].aabb.AddPoint(v2);
        }
        nodes.emplace_back();
        std::memset(&nodes[0], 0, sizeof(pt_bvh_node));
        nodes[0].first_primitive = 0;
        nodes[0].primitive_count = primitive_count;
        UpdateNodeBounds(0);
        Subdivide(0);
    }
    void UpdateNodeBounds(uint32_t index) {  // bvh.cpp:53-65
        pt_bvh_node &node = nodes[index];
        AABB b;
        if (node.primitive_count > 0) {
            for (uint32_t i = 0; i < node.primitive_count; ++i)
                b.Union(prims[tri_indices[node.first_primitive + i]].aabb);
        } else {
            b = B(nodes[node.first_primitive]);
            b.Union(B(nodes[node.first_primitive + 1]));
        }
        setB(node, b);
    }
    // bvh.cpp:115-200; returns false when no candidate plane exists (cost = FLT_MAX).
    void FindSplitPlane(uint32_t index, float &cost, uint32_t &best_axis, float &best_pos) {
        const pt_bvh_node &node = nodes[index];
        constexpr int bin_count = 8;
        float best_cost = FLT_MAX;
        best_axis = 0;
        best_pos = 0.0f;
        AABB cb;
        for (uint32_t i = 0; i < node.primitive_count; ++i)
            cb.AddPoint(prims[tri_indices[node.first_primitive + i]].centroid);
        if (cb.IsEmpty()) {
            cost = FLT_MAX;
            return;
        }
        vec3 extent = cb.Extent();
        vec3 inv_bin_width = 1.0f / vmax(extent, V3(1e-30f));
        for (uint32_t axis = 0; axis < 3; ++axis) {
            if (extent[axis] <= 0.0f) continue;
            std::array<Bin, bin_count> bins{};
            for (uint32_t i = 0; i < node.primitive_count; ++i) {
                uint32_t p = tri_indices[node.first_primitive + i];
                float scaled_offset = (prims[p].centroid[axis] - cb.min[axis]) * inv_bin_width[axis];
                int bin_index = std::min(std::max(static_cast<int>(scaled_offset * bin_count), 0), bin_count - 1);
                bins[bin_index].prims++;
                bins[bin_index].bounds.Union(prims[p].aabb);
            }
            std::array<float, bin_count - 1> left_area{}, right_area{};
            std::array<uint32_t, bin_count - 1> left_count{}, right_count{};
            AABB left_box, right_box;
            uint32_t left_sum = 0, right_sum = 0;
            for (int i = 0; i < bin_count - 1; ++i) {
                left_sum += bins[i].prims;
                left_count[i] = left_sum;
                left_box.Union(bins[i].bounds);
                left_area[i] = left_box.SurfaceArea();
                const int right_bin = bin_count - 1 - i;
                right_sum += bins[right_bin].prims;
                right_count[right_bin - 1] = right_sum;
                right_box.Union(bins[right_bin].bounds);
                right_area[right_bin - 1] = right_box.SurfaceArea();
            }
            for (int i = 0; i < bin_count - 1; ++i) {
                float c = left_count[i] * left_area[i] + right_count[i] * right_area[i];
                if (c < best_cost) {
                    best_cost = c;
                    best_axis = axis;
                    best_pos = cb.min[axis] + extent[axis] * (float(i + 1) / bin_count);
                }
            }
        }
        cost = best_cost;
    }
    void Subdivide(uint32_t index) {  // bvh.cpp:67-113
        if (nodes[index].primitive_count <= 2) return;
        float cost, split_pos;
        uint32_t axis;
        FindSplitPlane(index, cost, axis, split_pos);
        float parent_cost = nodes[index].primitive_count * B(nodes[index]).SurfaceArea();
        if (cost >= parent_cost) return;
        uint32_t i = nodes[index].first_primitive;
        uint32_t j = i + nodes[index].primitive_count - 1;
        while (i <= j) {
            uint32_t prim = tri_indices[i];
            if (prims[prim].centroid[int(axis)] < split_pos) ++i;
            else std::swap(tri_indices[i], tri_indices[j--]);
        }
        uint32_t left_count = i - nodes[index].first_primitive;
        if (left_count == 0 || left_count == nodes[index].primitive_count) return;
        uint32_t left_index = uint32_t(nodes.size());
        nodes.emplace_back();
        nodes.emplace_back();
        std::memset(&nodes[left_index], 0, 2 * sizeof(pt_bvh_node));
        uint32_t right_index = left_index + 1;
        nodes[left_index].first_primitive = nodes[index].first_primitive;
        nodes[left_index].primitive_count = left_count;
        UpdateNodeBounds(left_index);
        nodes[right_index].first_primitive = nodes[index].first_primitive + left_count;
        nodes[right_index].primitive_count = nodes[index].primitive_count - left_count;
        UpdateNodeBounds(right_index);
        nodes[index].first_primitive = left_index;
        nodes[index].primitive_count = 0;
        Subdivide(left_index);
        Subdivide(right_index);
    }
};

struct TLASBuilder {  // bvh.cpp:206-332
    std::vector<pt_tlas_node> tlas_nodes;
    std::vector<pt_blas> device_blases;
    std::vector<pt_bvh_node> blas_nodes;
    std::vector<uint32_t> tri_indices;

    void Build(const vec3 *positions, const uint32_t *indices, const pt_geometry_instance *instances, size_t n) {
        int max_mesh_id = 0;
        for (size_t i = 0; i < n; ++i) max_mesh_id = std::max(max_mesh_id, instances[i].blas_index);
        std::vector<BLASBuilder> blases(max_mesh_id + 1);
        std::vector<AABB> instance_bounds(n);
        for (size_t i = 0; i < n; ++i) {
            const auto &inst = instances[i];
            uint32_t blas_id = inst.blas_index;
            if (blases[blas_id].nodes.empty())
                blases[blas_id].Build(positions, indices, inst.first_index, inst.index_count);
            AABB obj = BLASBuilder::B(blases[blas_id].nodes[0]);
            instance_bounds[i] = obj.Transform(M(inst.transform));
        }
        device_blases.assign(blases.size(), pt_blas{0, 0, 0, 0});
        for (uint32_t id = 0; id < blases.size(); ++id) {
            const auto &blas = blases[id];
            if (blas.nodes.empty()) continue;
            uint32_t node_offset = uint32_t(blas_nodes.size());
            for (pt_bvh_node dst : blas.nodes) {
                if (!(dst.primitive_count > 0)) dst.first_primitive += node_offset;
                blas_nodes.push_back(dst);
            }
            tri_indices.insert(tri_indices.end(), blas.tri_indices.begin(), blas.tri_indices.end());
            device_blases[id] = {node_offset, uint32_t(blas.nodes.size()),
                                 uint32_t(tri_indices.size() - blas.tri_indices.size()),
                                 uint32_t(blas.tri_indices.size())};
        }
        tlas_nodes.reserve(2 * n);
        std::vector<uint32_t> instance_indices(n);
        int node_count = int(n);
        auto blank = []() {
            pt_tlas_node t;
            std::memset(&t, 0, sizeof(t));
            t.min[0] = t.min[1] = t.min[2] = FLT_MAX;
            t.max[0] = t.max[1] = t.max[2] = -FLT_MAX;
            return t;
        };
        auto TB = [](const pt_tlas_node &t) { return box_of(t.min, t.max); };
        auto setTB = [](pt_tlas_node &t, const AABB &b) {
            t.min[0] = b.min.x; t.min[1] = b.min.y; t.min[2] = b.min.z;
            t.max[0] = b.max.x; t.max[1] = b.max.y; t.max[2] = b.max.z;
        };
        tlas_nodes.push_back(blank());
        for (uint32_t i = 0; i < n; ++i) {
            instance_indices[i] = uint32_t(tlas_nodes.size());
            tlas_nodes.push_back(blank());
            setTB(tlas_nodes.back(), instance_bounds[i]);
            tlas_nodes.back().instance = i;
            tlas_nodes.back().left_right = 0;
        }
        auto find_best_match = [&](int instance) {
            float best_area = FLT_MAX;
            int best_j = -1;
            for (int j = 0; j < node_count; ++j) {
                if (j == instance) continue;
                AABB merged = TB(tlas_nodes[instance_indices[instance]]);
                merged.Union(TB(tlas_nodes[instance_indices[j]]));
                float area = merged.SurfaceArea();
                if (area < best_area) {
                    best_area = area;
                    best_j = j;
                }
            }
            return best_j;
        };
        int a = 0;
        int b = find_best_match(a);
        while (node_count > 1) {
            int c = find_best_match(b);
            if (a == c) {
                int node_index_a = int(instance_indices[a]);
                int node_index_b = int(instance_indices[b]);
                pt_tlas_node nn = blank();
                nn.left_right = uint32_t(node_index_a | (node_index_b << 16));
                AABB nb = TB(tlas_nodes[node_index_a]);
                nb.Union(TB(tlas_nodes[node_index_b]));
                setTB(nn, nb);
                tlas_nodes.push_back(nn);
                instance_indices[a] = uint32_t(tlas_nodes.size() - 1);
                instance_indices[b] = instance_indices[node_count - 1];
                --node_count;
                b = find_best_match(a);
            } else {
                a = b;
                b = c;
            }
        }
        tlas_nodes[0] = tlas_nodes[instance_indices[a]];
    }
};

// ------------------------------------------------------------ alias tables
// discrete_sampler.cpp:14-76
static int AliasBuild(const float *weights, uint32_t n, float *prob, uint32_t *alias) {
    float sum = std::accumulate(weights, weights + n, 0.0f);
    if (sum == 0.0f) return 1;
    for (uint32_t i = 0; i < n; ++i) {
        prob[i] = 0.0f;
        alias[i] = UINT_MAX;
    }
    for (uint32_t i = 0; i < n; ++i) prob[i] = n * weights[i] / sum;
    auto greater = [&](uint32_t a, uint32_t b) { return prob[a] < prob[b]; };
    auto lesser = [&](uint32_t a, uint32_t b) { return prob[a] > prob[b]; };
    std::priority_queue<uint32_t, std::vector<uint32_t>, decltype(greater)> overfull(greater);
    std::priority_queue<uint32_t, std::vector<uint32_t>, decltype(lesser)> underfull(lesser);
    for (uint32_t i = 0; i < n; ++i) {
        if (prob[i] > 1.0f + EPSILON) overfull.push(i);
        else if (prob[i] < 1.0f - EPSILON) underfull.push(i);
        else prob[i] = 1.0f;
    }
    while (!overfull.empty() && !underfull.empty()) {
        uint32_t i = overfull.top(); overfull.pop();
        uint32_t j = underfull.top(); underfull.pop();
        alias[j] = i;
        prob[i] -= 1.0f - prob[j];
        if (prob[i] > 1.0f + EPSILON) overfull.push(i);
        else if (prob[i] < 1.0f - EPSILON) underfull.push(i);
        else prob[i] = 1.0f;
    }
    while (!overfull.empty()) { prob[overfull.top()] = 1.0f; overfull.pop(); }
    while (!underfull.empty()) { prob[underfull.top()] = 1.0f; underfull.pop(); }
    return 0;
}

// ----------------------------------------------------------------- sampler
// math/hash.h:5-9
static inline uint32_t PCG_Hash(uint32_t input) {
    uint32_t state = input * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
struct Sampler {  // kernel/sampler/independent.h:10-12
    uint64_t state;
};
static inline void SeedSampler(Sampler &s, int px, int py, int iteration) {  // independent.h:14-23
    uint32_t key_xy = (uint32_t(px) & 0xFFFFu) | ((uint32_t(py) & 0xFFFFu) << 16);
    uint32_t key = key_xy ^ (uint32_t(iteration) * 0x9E3779B9u);
    uint32_t hashed = PCG_Hash(key);
    s.state = uint64_t(hashed) << 32 | uint64_t(hashed ^ 0xdeadbeefu);
}
static inline float Sample1D(Sampler &s) {  // independent.h:25-32
    uint64_t old = s.state;
    s.state = old * 6364136223846793005ULL + 1;
    uint32_t x = uint32_t(((old >> 18) ^ old) >> 27);
    uint32_t r = uint32_t(old >> 59);
    uint32_t v = (x >> r) | (x << ((-int(r)) & 31));
    return v * (1.0f / 4294967296.0f);
}
static inline vec2 Sample2D(Sampler &s) {
    float a = Sample1D(s);
    float b = Sample1D(s);
    return {a, b};
}
// kernel/sampler.h:57-70
static inline uint32_t DiscreteSample1D(const pt_alias_table &t, Sampler &s) {
    float x = Sample1D(s);
    float scaled = x * float(t.size);
    uint32_t i = static_cast<uint32_t>(scaled);
    if (scaled >= float(t.size)) {
        i = t.size - 1;
        scaled = float(i);
    }
    float y = scaled - static_cast<float>(i);
    return (y < t.probability_table[i]) ? i : t.alias_table[i];
}

// ----------------------------------------------------------------- camera
// camera.cpp:6-20
static void CameraUpdate(const float eye[3], const float look_at[3], const float up_in[3], float fovy_deg,
                         int width, int height, pt_camera *out) {
    vec3 position = {eye[0], eye[1], eye[2]};
    vec3 look = {look_at[0], look_at[1], look_at[2]};
    vec3 up = {up_in[0], up_in[1], up_in[2]};
    vec3 front = normalize(look - position);
    vec3 right = normalize(cross(front, up));
    up = normalize(cross(right, front));
    float theta = fovy_deg * 0.01745329251994329576923690768489f;  // glm::radians
    float half_height = std::tan(theta * 0.5f);
    float aspect = float(width) / float(height);
    float half_width = aspect * half_height;
    out->resolution[0] = width;
    out->resolution[1] = height;
    out->position[0] = position.x; out->position[1] = position.y; out->position[2] = position.z;
    out->front[0] = front.x; out->front[1] = front.y; out->front[2] = front.z;
    out->up[0] = up.x; out->up[1] = up.y; out->up[2] = up.z;
    out->right[0] = right.x; out->right[1] = right.y; out->right[2] = right.z;
    out->pixel_length[0] = (2.0f * half_width) / float(width);
    out->pixel_length[1] = (2.0f * half_height) / float(height);
    out->focal_length = 1.0f;  // camera.cpp:46
    out->fovy = fovy_deg;
}
// kernel/camera.h:10-22
static inline Ray SpawnCameraRay(const pt_camera &c, int i, int j, float sx, float sy) {
    float x = ((i + sx) - 0.5f * c.resolution[0]) * c.pixel_length[0];
    float y = ((j + sy) - 0.5f * c.resolution[1]) * c.pixel_length[1];
    vec3 position = {c.position[0], c.position[1], c.position[2]};
    vec3 front = {c.front[0], c.front[1], c.front[2]};
    vec3 right = {c.right[0], c.right[1], c.right[2]};
    vec3 up = {c.up[0], c.up[1], c.up[2]};
    vec3 film_point = position + front * c.focal_length + right * x - up * y;
    vec3 direction = normalize(film_point - position);
    return {position, direction};
}

// ---------------------------------------------------------------- textures
// kernel/texture.h:41-93.  Image lookups: the reference uses a CUDA texture
// object (bilinear, wrap, normalised coordinates; texture.cpp:60-69).  gfx950
// has no texture unit, so the filter is DEFINED here in fp32:
//   x = s*W - 0.5, i0 = floor(x), fx = x - i0, texel columns i0 mod W and
//   (i0+1) mod W (same for rows); value = lerp_y(lerp_x(T00,T10), lerp_x(T01,T11))
//   with lerp(a,b,f) = a*(1-f) + b*f.  Non-finite coordinates return 0.
// Two properties of the CUDA unit are options here as in the product (pt_set_texture_filter): bit 0
// rounds the bilinear weights to 8 fractional bits, bit 1 linearises texels of images tagged sRGB.
static int g_texture_filter = 0;
static inline float SrgbToLinear(float c) { return c <= 0.04045f ? c * (1.0f / 12.92f) : powf((c + 0.055f) * (1.0f / 1.055f), 2.4f); }
static inline int wrapi(int i, int n) {
    int m = i % n;
    return m < 0 ? m + n : m;
}
template <int CH>
static inline void SampleImage(const pt_image &img, float s, float t, float *out) {
    for (int c = 0; c < CH; ++c) out[c] = 0.0f;
    if (!std::isfinite(s) || !std::isfinite(t) || img.width <= 0 || img.height <= 0) return;
    float x = s * float(img.width) - 0.5f;
    float y = t * float(img.height) - 0.5f;
    float fx0 = floorf(x), fy0 = floorf(y);
    float fx = x - fx0, fy = y - fy0;
    if (g_texture_filter & 1) {
        fx = floorf(fx * 256.0f + 0.5f) * (1.0f / 256.0f);
        fy = floorf(fy * 256.0f + 0.5f) * (1.0f / 256.0f);
    }
    const bool decode = CH == 3 && img.srgb && (g_texture_filter & 2);
    // coordinates beyond the int range wrap to texel 0 deterministically
    int i0 = (fabsf(fx0) < 1.0e9f) ? wrapi(int(fx0), img.width) : 0;
    int j0 = (fabsf(fy0) < 1.0e9f) ? wrapi(int(fy0), img.height) : 0;
    int i1 = (i0 + 1 == img.width) ? 0 : i0 + 1;
    int j1 = (j0 + 1 == img.height) ? 0 : j0 + 1;
    const float *d = img.data;
    for (int c = 0; c < CH; ++c) {
        float t00 = d[(size_t(j0) * img.width + i0) * CH + c];
        float t10 = d[(size_t(j0) * img.width + i1) * CH + c];
        float t01 = d[(size_t(j1) * img.width + i0) * CH + c];
        float t11 = d[(size_t(j1) * img.width + i1) * CH + c];
        if (decode) {
            t00 = SrgbToLinear(t00); t10 = SrgbToLinear(t10); t01 = SrgbToLinear(t01); t11 = SrgbToLinear(t11);
        }
        float a = t00 * (1.0f - fx) + t10 * fx;
        float b = t01 * (1.0f - fx) + t11 * fx;
        out[c] = a * (1.0f - fy) + b * fy;
    }
}
static vec3 SampleTexture3(const pt_scene_desc &d, int texture_id, vec2 uv) {
    if (texture_id < 0 || uint32_t(texture_id) >= d.texture3_count) return V3(0.0f);
    const pt_texture3 &t = d.textures3[texture_id];
    switch (t.type) {
        case PT_TEX_CONSTANT:
            return {t.u.constant.value[0], t.u.constant.value[1], t.u.constant.value[2]};
        case PT_TEX_CHECKERBOARD: {  // kernel/texture.h:10-18
            float s = uv.x * t.u.checkerboard.scale[0] + t.u.checkerboard.offset[0];
            float tt = uv.y * t.u.checkerboard.scale[1] + t.u.checkerboard.offset[1];
            int i = static_cast<int>(floorf(s));
            int j = static_cast<int>(floorf(tt));
            const float *c = ((i + j) & 1) ? t.u.checkerboard.color1 : t.u.checkerboard.color0;
            return {c[0], c[1], c[2]};
        }
        case PT_TEX_IMAGE: {
            int id = t.u.image.image_id;
            if (id < 0 || uint32_t(id) >= d.image3_count) return V3(0.0f);
            float s = uv.x * t.u.image.scale[0] + t.u.image.offset[0];
            float tt = uv.y * t.u.image.scale[1] + t.u.image.offset[1];
            float o[3];
            SampleImage<3>(d.images3[id], s, tt, o);
            return {o[0], o[1], o[2]};
        }
    }
    return V3(0.0f);
}
static float SampleTexture1(const pt_scene_desc &d, int texture_id, vec2 uv) {
    if (texture_id < 0 || uint32_t(texture_id) >= d.texture1_count) return 0.0f;
    const pt_texture1 &t = d.textures1[texture_id];
    switch (t.type) {
        case PT_TEX_CONSTANT:
            return t.u.constant.value;
        case PT_TEX_CHECKERBOARD: {
            float s = uv.x * t.u.checkerboard.scale[0] + t.u.checkerboard.offset[0];
            float tt = uv.y * t.u.checkerboard.scale[1] + t.u.checkerboard.offset[1];
            int i = static_cast<int>(floorf(s));
            int j = static_cast<int>(floorf(tt));
            return ((i + j) & 1) ? t.u.checkerboard.color1 : t.u.checkerboard.color0;
        }
        case PT_TEX_IMAGE: {
            int id = t.u.image.image_id;
            if (id < 0 || uint32_t(id) >= d.image1_count) return 0.0f;
            float s = uv.x * t.u.image.scale[0] + t.u.image.offset[0];
            float tt = uv.y * t.u.image.scale[1] + t.u.image.offset[1];
            float o[1];
            SampleImage<1>(d.images1[id], s, tt, o);
            return o[0];
        }
    }
    return 0.0f;
}

// ---------------------------------------------------------------- sampling
// math/sampling.h:7-20
static inline vec3 ToWorldSpace(vec3 local, vec3 normal) {
    vec3 up;
    if (fabsf(normal.x) < INV_SQRT3) up = {1, 0, 0};
    else if (fabsf(normal.y) < INV_SQRT3) up = {0, 1, 0};
    else up = {0, 0, 1};
    vec3 tangent = normalize(cross(normal, up));
    vec3 bitangent = normalize(cross(normal, tangent));
    return tangent * local.x + bitangent * local.y + normal * local.z;
}
// sinf / cosf of the azimuth (math/sampling.h:33-35, :44-56).  No two libraries round these alike in the last bit
// (CUDA's, glibc's, the device library's), so the build under test and this restatement both use the same fixed
// sequence of correctly rounded fp32 operations, restated here from its definition: quadrant k = int(phi * 2/pi +
// 0.5); r = (phi - k * 1.5703125) - k * 4.8382679e-4; z = r * r; sin r = r + r z (S0 + z (S1 + z S2));
// cos r = (1 - z / 2) + z z (C0 + z (C1 + z C2)); then the rotation by k quadrants.  tests/test_oracle.py holds it
// against double-precision sin / cos (error below 1.5e-7).
static inline void SinCosAzimuth(float phi, float &sin_phi, float &cos_phi) {
    static const float S[3] = {-1.6666654611e-1f, 8.3321608736e-3f, -1.9515295891e-4f};
    static const float C[3] = {4.166664568298827e-2f, -1.388731625493765e-3f, 2.443315711809948e-5f};
    int k = int(phi * 0.63661977236758134f + 0.5f);
    float kf = float(k);
    float r = (phi - kf * 1.5703125f) - kf * 4.83826794897e-4f;
    float z = r * r;
    float sin_r = r + (r * z) * (S[0] + z * (S[1] + z * S[2]));
    float cos_r = (1.0f - 0.5f * z) + (z * z) * (C[0] + z * (C[1] + z * C[2]));
    switch (k & 3) {
        case 0: sin_phi = sin_r; cos_phi = cos_r; break;
        case 1: sin_phi = cos_r; cos_phi = -sin_r; break;
        case 2: sin_phi = -sin_r; cos_phi = -cos_r; break;
        default: sin_phi = -cos_r; cos_phi = sin_r; break;
    }
}
// math/sampling.h:30-38
static inline vec3 SampleCosineHemisphere(vec3 normal, vec2 sample) {
    float z = sqrtf(sample.x);
    float phi = 2 * PI * sample.y;
    float r = sqrtf(std::max(0.0f, 1.0f - z * z));
    float sin_phi, cos_phi;
    SinCosAzimuth(phi, sin_phi, cos_phi);
    vec3 local = {r * cos_phi, r * sin_phi, z};
    return ToWorldSpace(local, normal);
}
// math/sampling.h:40-58
static inline vec3 SampleGGX(vec3 normal, float roughness, vec2 sample) {
    float a2 = roughness * roughness;
    float phi = 2.0f * PI * sample.x;
    float cos_theta = sqrtf((1.0f - sample.y) / (1.0f + (a2 - 1.0f) * sample.y));
    float sin_theta = sqrtf(std::max(0.0f, 1.0f - cos_theta * cos_theta));
    float sin_phi, cos_phi;
    SinCosAzimuth(phi, sin_phi, cos_phi);
    vec3 local = {sin_theta * cos_phi, sin_theta * sin_phi, cos_theta};
    return ToWorldSpace(local, normal);
}
// math/sampling.h:61-68
static inline vec3 SampleTriangle(vec2 sample) {
    float sqr = sqrtf(sample.x);
    float u = 1.0f - sqr;
    float v = sample.y * sqr;
    float w = 1.0f - u - v;
    return {u, v, w};
}
static inline float PowerHeuristic(float f, float g) { return (f * f) / (f * f + g * g); }  // sampling.h:70-72
static inline float Luminance(vec3 c) { return c.x * 0.212671f + c.y * 0.715160f + c.z * 0.072169f; }  // color.h:7-9
static inline vec3 Reflect(vec3 v, vec3 n) { return 2 * dot(v, n) * n - v; }  // geometry.h:7-9
static inline bool Refract(vec3 vec, vec3 normal, float eta, vec3 &result) {  // geometry.h:11-24
    float cos_in = dot(normal, vec);
    float sin2_in = fmaxf(0.f, 1.f - cos_in * cos_in);
    float sin2_t = sin2_in * eta * eta;
    if (sin2_t >= 1.f) return false;
    float cos_t = fmaxf(0.0f, sqrtf(1.0f - sin2_t));
    result = -vec * eta + (cos_in * eta - cos_t) * normal;
    return true;
}
static inline float TriangleArea(vec3 v0, vec3 v1, vec3 v2) {  // geometry.h:26-30
    return length(cross(v1 - v0, v2 - v0)) / 2.0f;
}

// ------------------------------------------------------------------ BSDFs
enum : uint8_t { S_Invalid = 0, S_Diffuse = 1, S_Glossy = 2, S_Specular = 4, S_Reflection = 16, S_Transmission = 32 };
struct BSDF_Sample {  // kernel/bsdf/common.h:23-28
    uint8_t type = S_Invalid;
    float pdf = 0;
    vec3 bsdf{0, 0, 0};
    vec3 w_in{0, 0, 0};
};
static inline float Pow5(float b) { float b2 = b * b; return b2 * b2 * b; }  // math/math.h:7-10
static inline float TrowbridgeReitz(vec3 normal, vec3 micro, float alpha) {  // bsdf/common.h:37-50
    float cos_in = dot(normal, micro);
    if (cos_in <= 0.0f) return 0.0f;
    float a2 = alpha * alpha;
    float denom_term = fmaf((a2 - 1.0f), cos_in * cos_in, 1.0f);
    float denom = denom_term * denom_term + 1e-16f;
    float result = (a2 * INV_PI) / denom;
    return std::isfinite(result) ? result : 0.0f;
}
static inline float SmithG1(vec3 normal, vec3 v, float roughness) {  // bsdf/common.h:56-64
    float cos_in = dot(normal, v);
    if (cos_in <= 0.0f) return 0.0f;
    float a2 = roughness * roughness;
    float denom = cos_in + sqrtf(a2 + (1.0f - a2) * cos_in * cos_in);
    return 2 * cos_in / denom;
}
static inline float SmithG(vec3 n, vec3 w_in, vec3 w_out, float r) { return SmithG1(n, w_in, r) * SmithG1(n, w_out, r); }
static inline float Fresnel(float cos_in, float eta) {  // bsdf/common.h:80-92
    float sin2_in = 1 - cos_in * cos_in;
    float sin2_t = sin2_in * eta * eta;
    if (sin2_t >= 1.0f) return 1.0f;
    float cos_t = fmaxf(0.0f, sqrtf(1 - sin2_t));
    float parallel = (cos_in - eta * cos_t) / (cos_in + eta * cos_t);
    float perp = (eta * cos_in - cos_t) / (eta * cos_in + cos_t);
    return (parallel * parallel + perp * perp) / 2.0f;
}
static inline vec3 FresnelSchlick(float cos_in, vec3 f0) {  // bsdf/common.h:100-102
    return f0 + (1.0f - f0) * Pow5(1.0f - cos_in);
}

// bsdf/lambertian.h:5-42
static BSDF_Sample SampleLambertian(const pt_scene_desc &d, const pt_material &m, const Intersection &is, Sampler &s) {
    vec3 w_in = SampleCosineHemisphere(is.normal, Sample2D(s));
    vec3 albedo = SampleTexture3(d, m.tex0, is.uv);
    BSDF_Sample r;
    r.bsdf = albedo * INV_PI;
    r.pdf = fmaxf(0.0f, dot(is.normal, w_in)) * INV_PI;
    r.type = S_Diffuse;
    r.w_in = w_in;
    return r;
}
// bsdf/perfect_mirror.h:5-23
static BSDF_Sample SampleMirror(const pt_scene_desc &d, const pt_material &m, const Intersection &is, vec3 w_out) {
    vec3 w_in = Reflect(w_out, is.normal);
    float cos_in = fabsf(dot(is.normal, w_in));
    BSDF_Sample r;
    r.pdf = 1.0f;
    r.w_in = w_in;
    r.type = S_Specular | S_Reflection;
    r.bsdf = SampleTexture3(d, m.tex0, is.uv) / cos_in;
    return r;
}
// bsdf/perfect_dielectric.h:5-66
static BSDF_Sample SampleDielectric(const pt_scene_desc &d, const pt_material &m, const Intersection &is, vec3 w_out, Sampler &s) {
    BSDF_Sample result;
    vec3 normal = is.normal;
    float ior = m.w1.f;
    float eta = 1.0f / ior;
    float cos_out = dot(is.normal, w_out);
    if (cos_out < 0.0f) {
        normal = -normal;
        cos_out = -cos_out;
        eta = ior;
    }
    float reflectance = Fresnel(cos_out, eta);
    float transmittance = 1.0f - reflectance;
    vec3 color = SampleTexture3(d, m.tex0, is.uv);
    if (Sample1D(s) < reflectance) {
        vec3 w_in = Reflect(w_out, normal);
        float cos_in = dot(normal, normalize(w_in));
        if (cos_in < 0.0f) { result.type = S_Invalid; return result; }
        result.type = S_Specular | S_Reflection;
        result.w_in = w_in;
        result.bsdf = color * reflectance / cos_in;
        result.pdf = reflectance;
        return result;
    } else {
        vec3 w_in;
        if (!Refract(w_out, normal, eta, w_in)) { result.type = S_Invalid; return result; }
        float cos_in = dot(w_in, normal);
        if (cos_in > 0.0f) { result.type = S_Invalid; return result; }
        cos_in = fabsf(cos_in);
        result.type = S_Specular | S_Transmission;
        result.w_in = w_in;
        result.bsdf = color * transmittance * eta * eta / cos_in;
        result.pdf = transmittance;
        return result;
    }
}
struct MRParams {
    vec3 base_color;
    float metallic, alpha;
};
static inline MRParams MRFetch(const pt_scene_desc &d, const pt_material &m, const Intersection &is) {
    MRParams p;
    p.base_color = SampleTexture3(d, m.tex0, is.uv);
    p.metallic = SampleTexture1(d, m.w1.i, is.uv);
    float alpha = SampleTexture1(d, m.w2, is.uv);
    alpha = std::min(std::max(alpha, 0.02f), 1.0f);  // glm::clamp
    p.alpha = alpha * alpha;
    return p;
}
// bsdf/metallic_roughness.h:8-90
static BSDF_Sample SampleMetallicRoughness(const pt_scene_desc &d, const pt_material &m, const Intersection &is, vec3 w_out, Sampler &s) {
    BSDF_Sample result;
    vec3 normal = is.normal;
    float cos_out = dot(normal, w_out);
    if (cos_out < 0.0f) {
        normal = -normal;
        cos_out = -cos_out;
    }
    MRParams p = MRFetch(d, m, is);
    vec3 F0 = mix(V3(0.04f), p.base_color, p.metallic);
    float w_diff = Luminance(p.base_color) * (1.0f - p.metallic);
    float w_spec = Luminance(F0);
    float w_sum = std::max(w_spec + w_diff, EPSILON);
    w_diff /= w_sum;
    w_spec /= w_sum;
    if (Sample1D(s) < w_spec) {
        vec3 half_vec = SampleGGX(normal, p.alpha, Sample2D(s));
        vec3 w_in = Reflect(w_out, half_vec);
        float cos_in = dot(normal, w_in);
        if (cos_in <= 0.0f) { result.type = S_Invalid; return result; }
        float D = TrowbridgeReitz(normal, half_vec, p.alpha);
        float G = SmithG(normal, w_out, w_in, p.alpha);
        vec3 refl_in = FresnelSchlick(fmaxf(dot(w_out, half_vec), 0.0f), F0);
        float pdf_specular = (D * fmaxf(dot(normal, half_vec), 0.0f)) / (4.0f * fmaxf(fabsf(dot(w_out, half_vec)), EPSILON));
        result.type = S_Glossy | S_Reflection;
        result.bsdf = refl_in * (D * G) / fmaxf(4.0f * cos_in * cos_out, 1e-6f);
        result.pdf = w_spec * pdf_specular;
        result.w_in = w_in;
        return result;
    } else {
        vec3 w_in = SampleCosineHemisphere(normal, Sample2D(s));
        float cos_in = fmaxf(dot(normal, w_in), 0.0f);
        if (cos_in <= 0.0f) { result.type = S_Invalid; return result; }
        vec3 sum_dir = w_in + w_out;
        vec3 half_vec = (length2(sum_dir) >= EPSILON) ? normalize(sum_dir) : normal;
        float cos_f = fabsf(dot(w_out, half_vec));
        vec3 refl_out = FresnelSchlick(cos_f, F0);
        result.type = S_Diffuse | S_Reflection;
        result.w_in = w_in;
        result.bsdf = p.base_color * (1.0f - p.metallic) * (1.0f - refl_out) * INV_PI;
        result.pdf = w_diff * cos_in * INV_PI;
        return result;
    }
}
// bsdf/metallic_roughness.h:92-127
static vec3 EvalMetallicRoughness(const pt_scene_desc &d, const pt_material &m, const Intersection &is, vec3 w_in, vec3 w_out) {
    vec3 normal = is.normal;
    float cos_out = dot(normal, w_out);
    float cos_in = dot(normal, w_in);
    if (dot(normal, w_out) <= 0.0f || dot(normal, w_in) <= 0.0f) return V3(0.0f);
    MRParams p = MRFetch(d, m, is);
    vec3 half_vec = normalize(w_in + w_out);
    vec3 F0 = mix(V3(0.04f), p.base_color, p.metallic);
    float D = TrowbridgeReitz(normal, half_vec, p.alpha);
    float G = SmithG(normal, w_out, w_in, p.alpha);
    vec3 refl_in = FresnelSchlick(dot(w_in, half_vec), F0);
    vec3 specular = refl_in * (D * G) / std::max(4.0f * cos_in * cos_out, EPSILON);
    vec3 refl_out = FresnelSchlick(dot(w_out, half_vec), F0);
    vec3 diffuse = p.base_color * (1.0f - p.metallic) * (1.0f - refl_out) * (1.0f - refl_in) * INV_PI;
    return diffuse + specular;
}
// bsdf/metallic_roughness.h:129-177
static float MetallicRoughnessPDF(const pt_scene_desc &d, const pt_material &m, const Intersection &is, vec3 w_in, vec3 w_out) {
    vec3 normal = is.normal;
    float cos_out = dot(normal, w_out);
    float cos_in = dot(normal, w_in);
    if (cos_out <= 0.0f || cos_in <= 0.0f) return 0.0f;
    MRParams p = MRFetch(d, m, is);
    vec3 F0 = mix(V3(0.04f), p.base_color, p.metallic);
    float w_diff = Luminance(p.base_color) * (1.0f - p.metallic);
    float w_spec = Luminance(F0);
    float w_sum = std::max(w_spec + w_diff, EPSILON);
    w_diff /= w_sum;
    w_spec /= w_sum;
    float pdf_diffuse = cos_in * INV_PI;
    float pdf_specular = 0.0f;
    vec3 half_vec = w_in + w_out;
    if (length2(half_vec) >= EPSILON) {
        half_vec = normalize(half_vec);
        float n_dot_h = std::max(dot(normal, half_vec), 0.0f);
        float out_dot_h = std::max(dot(w_out, half_vec), 0.0f);
        if (n_dot_h > 0.0f && out_dot_h > 0.0f) {
            float D = TrowbridgeReitz(normal, half_vec, p.alpha);
            pdf_specular = (D * n_dot_h) / std::max(4.0f * out_dot_h, EPSILON);
        }
    }
    return w_diff * pdf_diffuse + w_spec * pdf_specular;
}
// kernel/bsdf.h:9-66
static BSDF_Sample SampleMaterial(const pt_scene_desc &d, const pt_material &m, const Intersection &is, vec3 w_out, Sampler &s) {
    switch (m.type) {
        case PT_MAT_LAMBERTIAN: return SampleLambertian(d, m, is, s);
        case PT_MAT_DIELECTRIC: return SampleDielectric(d, m, is, w_out, s);
        case PT_MAT_MIRROR: return SampleMirror(d, m, is, w_out);
        case PT_MAT_METALLIC_ROUGHNESS: return SampleMetallicRoughness(d, m, is, w_out, s);
        default: return {};
    }
}
static vec3 EvalMaterialBSDF(const pt_scene_desc &d, const pt_material &m, const Intersection &is, vec3 w_in, vec3 w_out) {
    switch (m.type) {
        case PT_MAT_LAMBERTIAN: return SampleTexture3(d, m.tex0, is.uv) * INV_PI;
        case PT_MAT_METALLIC_ROUGHNESS: return EvalMetallicRoughness(d, m, is, w_in, w_out);
        default: return V3(0.0f);
    }
}
static float MaterialPDF(const pt_scene_desc &d, const pt_material &m, const Intersection &is, vec3 w_in, vec3 w_out) {
    switch (m.type) {
        case PT_MAT_LAMBERTIAN: return fmaxf(0.0f, dot(is.normal, w_in)) * INV_PI;
        case PT_MAT_METALLIC_ROUGHNESS: return MetallicRoughnessPDF(d, m, is, w_in, w_out);
        default: return 0.0f;
    }
}

// ------------------------------------------------------------------ lights
struct LightSample {  // kernel/light.h:8-13
    vec3 position{0, 0, 0}, normal{0, 0, 0}, radiance{0, 0, 0};
    float pdf = 0;
};
// kernel/light.h:15-59
static LightSample SampleSceneLight(const Scene &scene, Sampler &s) {
    const pt_scene_desc &d = *scene.d;
    uint32_t light_id = DiscreteSample1D(*d.light_sampler, s);
    const pt_area_light &light = d.lights[light_id];
    if (light.geometry_instance_index == -1) return {};
    float p_pick = light.power / d.total_light_power;
    const pt_geometry_instance &g = scene.geometries[light.geometry_instance_index];
    uint32_t tri = DiscreteSample1D(d.triangle_samplers[g.triangle_sampler_index], s);
    uint32_t i0 = scene.indices[g.first_index + tri * 3 + 0];
    uint32_t i1 = scene.indices[g.first_index + tri * 3 + 1];
    uint32_t i2 = scene.indices[g.first_index + tri * 3 + 2];
    const mat4 &T = M(g.transform), &IT = M(g.inv_transpose);
    vec3 p0 = xform_point(T, scene.positions[i0]);
    vec3 p1 = xform_point(T, scene.positions[i1]);
    vec3 p2 = xform_point(T, scene.positions[i2]);
    vec3 n0 = xform_dir(IT, scene.normals[i0]);
    vec3 n1 = xform_dir(IT, scene.normals[i1]);
    vec3 n2 = xform_dir(IT, scene.normals[i2]);
    vec2 t0 = scene.uvs[i0], t1 = scene.uvs[i1], t2 = scene.uvs[i2];
    vec3 uvw = SampleTriangle(Sample2D(s));
    float u = uvw.x, v = uvw.y, w = uvw.z;
    LightSample ls;
    ls.position = u * p0 + v * p1 + w * p2;
    ls.normal = u * n0 + v * n1 + w * n2;  // deliberately un-normalised (light.h:47,54)
    vec2 texcoord = u * t0 + v * t1 + w * t2;
    const pt_material &m = d.materials[g.material_id];
    ls.radiance = SampleTexture3(d, m.tex0, texcoord) * m.w1.f;
    ls.pdf = p_pick * (1.0f / g.total_area);
    return ls;
}

// scene.cpp:580-647
struct LightData {
    std::vector<pt_area_light> lights;
    std::vector<std::vector<float>> tri_prob;
    std::vector<std::vector<uint32_t>> tri_alias;
    std::vector<float> light_prob;
    std::vector<uint32_t> light_alias;
    float total_power = 0;
};
static int ComputeLightData(const vec3 *positions, const uint32_t *indices, pt_geometry_instance *geoms, size_t n,
                            const pt_material *materials, LightData &out) {
    std::vector<float> powers;
    uint32_t light_index = 0;
    float total = 0.0f;
    for (size_t i = 0; i < n; ++i) {
        if (materials[geoms[i].material_id].type != PT_MAT_EMISSIVE) continue;
        float total_area = 0.0f;
        std::vector<float> areas;
        for (uint32_t j = 0; j < geoms[i].index_count; j += 3) {
            uint32_t i0 = indices[geoms[i].first_index + j], i1 = indices[geoms[i].first_index + j + 1], i2 = indices[geoms[i].first_index + j + 2];
            vec3 v0 = xform_point(M(geoms[i].transform), positions[i0]);
            vec3 v1 = xform_point(M(geoms[i].transform), positions[i1]);
            vec3 v2 = xform_point(M(geoms[i].transform), positions[i2]);
            float area = TriangleArea(v0, v1, v2);
            total_area += area;
            areas.push_back(area);
        }
        geoms[i].total_area = total_area;
        geoms[i].area_light_id = int(light_index);
        geoms[i].triangle_sampler_index = int(light_index);
        out.tri_prob.emplace_back(areas.size());
        out.tri_alias.emplace_back(areas.size());
        if (AliasBuild(areas.data(), uint32_t(areas.size()), out.tri_prob.back().data(), out.tri_alias.back().data())) return 1;
        float power = total_area * materials[geoms[i].material_id].w1.f * PI;
        out.lights.push_back({power, int(i)});
        total += power;
        powers.push_back(power);
        light_index++;
    }
    out.total_power = total;
    out.light_prob.resize(powers.size());
    out.light_alias.resize(powers.size());
    if (!powers.empty() && AliasBuild(powers.data(), uint32_t(powers.size()), out.light_prob.data(), out.light_alias.data())) return 1;
    return 0;
}

// -------------------------------------------------------------- transforms
// math/transform.cpp:5-14 with glm::scale / rotate / translate spelled out.
static mat4 Rotate(const mat4 &m, float angle, vec3 v) {
    float c = cosf(angle), s = sinf(angle);
    vec3 axis = normalize(v);
    vec3 temp = (1.0f - c) * axis;
    float R[3][3];
    R[0][0] = c + temp.x * axis.x;
    R[0][1] = temp.x * axis.y + s * axis.z;
    R[0][2] = temp.x * axis.z - s * axis.y;
    R[1][0] = temp.y * axis.x - s * axis.z;
    R[1][1] = c + temp.y * axis.y;
    R[1][2] = temp.y * axis.z + s * axis.x;
    R[2][0] = temp.z * axis.x + s * axis.y;
    R[2][1] = temp.z * axis.y - s * axis.x;
    R[2][2] = c + temp.z * axis.z;
    mat4 r;
    r.c[0] = m.c[0] * R[0][0] + m.c[1] * R[0][1] + m.c[2] * R[0][2];
    r.c[1] = m.c[0] * R[1][0] + m.c[1] * R[1][1] + m.c[2] * R[1][2];
    r.c[2] = m.c[0] * R[2][0] + m.c[1] * R[2][1] + m.c[2] * R[2][2];
    r.c[3] = m.c[3];
    return r;
}
static mat4 GetTransformMatrix(vec3 translate, vec3 rotate_deg, vec3 scale) {
    const float rad = 0.01745329251994329576923690768489f;
    mat4 I = identity();
    mat4 S = I;
    S.c[0] = I.c[0] * scale.x;
    S.c[1] = I.c[1] * scale.y;
    S.c[2] = I.c[2] * scale.z;
    mat4 R = Rotate(I, rotate_deg.x * rad, {1, 0, 0});
    R = Rotate(R, rotate_deg.y * rad, {0, 1, 0});
    R = Rotate(R, rotate_deg.z * rad, {0, 0, 1});
    mat4 T = I;
    T.c[3] = I.c[0] * translate.x + I.c[1] * translate.y + I.c[2] * translate.z + I.c[3];
    return mul(mul(T, R), S);
}
// glm::inverse(mat4) (cofactor expansion, glm/detail/func_matrix.inl compute_inverse<4,4>)
static mat4 Inverse(const mat4 &mm) {
    const float(*m)[4] = reinterpret_cast<const float(*)[4]>(&mm);
    float Coef00 = m[2][2] * m[3][3] - m[3][2] * m[2][3];
    float Coef02 = m[1][2] * m[3][3] - m[3][2] * m[1][3];
    float Coef03 = m[1][2] * m[2][3] - m[2][2] * m[1][3];
    float Coef04 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
    float Coef06 = m[1][1] * m[3][3] - m[3][1] * m[1][3];
    float Coef07 = m[1][1] * m[2][3] - m[2][1] * m[1][3];
    float Coef08 = m[2][1] * m[3][2] - m[3][1] * m[2][2];
    float Coef10 = m[1][1] * m[3][2] - m[3][1] * m[1][2];
    float Coef11 = m[1][1] * m[2][2] - m[2][1] * m[1][2];
    float Coef12 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
    float Coef14 = m[1][0] * m[3][3] - m[3][0] * m[1][3];
    float Coef15 = m[1][0] * m[2][3] - m[2][0] * m[1][3];
    float Coef16 = m[2][0] * m[3][2] - m[3][0] * m[2][2];
    float Coef18 = m[1][0] * m[3][2] - m[3][0] * m[1][2];
    float Coef19 = m[1][0] * m[2][2] - m[2][0] * m[1][2];
    float Coef20 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
    float Coef22 = m[1][0] * m[3][1] - m[3][0] * m[1][1];
    float Coef23 = m[1][0] * m[2][1] - m[2][0] * m[1][1];
    auto V = [](float a, float b, float c, float d) { return vec4{a, b, c, d}; };
    auto vmul = [](vec4 a, vec4 b) { return vec4{a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; };
    auto vsub = [](vec4 a, vec4 b) { return vec4{a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; };
    vec4 Fac0 = V(Coef00, Coef00, Coef02, Coef03), Fac1 = V(Coef04, Coef04, Coef06, Coef07);
    vec4 Fac2 = V(Coef08, Coef08, Coef10, Coef11), Fac3 = V(Coef12, Coef12, Coef14, Coef15);
    vec4 Fac4 = V(Coef16, Coef16, Coef18, Coef19), Fac5 = V(Coef20, Coef20, Coef22, Coef23);
    vec4 Vec0 = V(m[1][0], m[0][0], m[0][0], m[0][0]), Vec1 = V(m[1][1], m[0][1], m[0][1], m[0][1]);
    vec4 Vec2 = V(m[1][2], m[0][2], m[0][2], m[0][2]), Vec3 = V(m[1][3], m[0][3], m[0][3], m[0][3]);
    vec4 Inv0 = vsub(vmul(Vec1, Fac0), vmul(Vec2, Fac1)) + vmul(Vec3, Fac2);
    vec4 Inv1 = vsub(vmul(Vec0, Fac0), vmul(Vec2, Fac3)) + vmul(Vec3, Fac4);
    vec4 Inv2 = vsub(vmul(Vec0, Fac1), vmul(Vec1, Fac3)) + vmul(Vec3, Fac5);
    vec4 Inv3 = vsub(vmul(Vec0, Fac2), vmul(Vec1, Fac4)) + vmul(Vec2, Fac5);
    vec4 SignA = V(+1, -1, +1, -1), SignB = V(-1, +1, -1, +1);
    mat4 inv;
    inv.c[0] = vmul(Inv0, SignA);
    inv.c[1] = vmul(Inv1, SignB);
    inv.c[2] = vmul(Inv2, SignA);
    inv.c[3] = vmul(Inv3, SignB);
    vec4 Row0 = V(inv.c[0].x, inv.c[1].x, inv.c[2].x, inv.c[3].x);
    vec4 Dot0 = vmul(mm.c[0], Row0);
    float Dot1 = (Dot0.x + Dot0.y) + (Dot0.z + Dot0.w);
    float OneOverDeterminant = 1.0f / Dot1;
    for (int j = 0; j < 4; ++j) inv.c[j] = inv.c[j] * OneOverDeterminant;
    return inv;
}
// glm::inverseTranspose(mat4) (glm/gtc/matrix_inverse.inl)
static mat4 InverseTranspose(const mat4 &mm) {
    const float(*m)[4] = reinterpret_cast<const float(*)[4]>(&mm);
    float SubFactor00 = m[2][2] * m[3][3] - m[3][2] * m[2][3];
    float SubFactor01 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
    float SubFactor02 = m[2][1] * m[3][2] - m[3][1] * m[2][2];
    float SubFactor03 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
    float SubFactor04 = m[2][0] * m[3][2] - m[3][0] * m[2][2];
    float SubFactor05 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
    float SubFactor06 = m[1][2] * m[3][3] - m[3][2] * m[1][3];
    float SubFactor07 = m[1][1] * m[3][3] - m[3][1] * m[1][3];
    float SubFactor08 = m[1][1] * m[3][2] - m[3][1] * m[1][2];
    float SubFactor09 = m[1][0] * m[3][3] - m[3][0] * m[1][3];
    float SubFactor10 = m[1][0] * m[3][2] - m[3][0] * m[1][2];
    float SubFactor11 = m[1][0] * m[3][1] - m[3][0] * m[1][1];
    float SubFactor12 = m[1][2] * m[2][3] - m[2][2] * m[1][3];
    float SubFactor13 = m[1][1] * m[2][3] - m[2][1] * m[1][3];
    float SubFactor14 = m[1][1] * m[2][2] - m[2][1] * m[1][2];
    float SubFactor15 = m[1][0] * m[2][3] - m[2][0] * m[1][3];
    float SubFactor16 = m[1][0] * m[2][2] - m[2][0] * m[1][2];
    float SubFactor17 = m[1][0] * m[2][1] - m[2][0] * m[1][1];
    float I[4][4];
    I[0][0] = +(m[1][1] * SubFactor00 - m[1][2] * SubFactor01 + m[1][3] * SubFactor02);
    I[0][1] = -(m[1][0] * SubFactor00 - m[1][2] * SubFactor03 + m[1][3] * SubFactor04);
    I[0][2] = +(m[1][0] * SubFactor01 - m[1][1] * SubFactor03 + m[1][3] * SubFactor05);
    I[0][3] = -(m[1][0] * SubFactor02 - m[1][1] * SubFactor04 + m[1][2] * SubFactor05);
    I[1][0] = -(m[0][1] * SubFactor00 - m[0][2] * SubFactor01 + m[0][3] * SubFactor02);
    I[1][1] = +(m[0][0] * SubFactor00 - m[0][2] * SubFactor03 + m[0][3] * SubFactor04);
    I[1][2] = -(m[0][0] * SubFactor01 - m[0][1] * SubFactor03 + m[0][3] * SubFactor05);
    I[1][3] = +(m[0][0] * SubFactor02 - m[0][1] * SubFactor04 + m[0][2] * SubFactor05);
    I[2][0] = +(m[0][1] * SubFactor06 - m[0][2] * SubFactor07 + m[0][3] * SubFactor08);
    I[2][1] = -(m[0][0] * SubFactor06 - m[0][2] * SubFactor09 + m[0][3] * SubFactor10);
    I[2][2] = +(m[0][0] * SubFactor07 - m[0][1] * SubFactor09 + m[0][3] * SubFactor11);
    I[2][3] = -(m[0][0] * SubFactor08 - m[0][1] * SubFactor10 + m[0][2] * SubFactor11);
    I[3][0] = -(m[0][1] * SubFactor12 - m[0][2] * SubFactor13 + m[0][3] * SubFactor14);
    I[3][1] = +(m[0][0] * SubFactor12 - m[0][2] * SubFactor15 + m[0][3] * SubFactor16);
    I[3][2] = -(m[0][0] * SubFactor13 - m[0][1] * SubFactor15 + m[0][3] * SubFactor17);
    I[3][3] = +(m[0][0] * SubFactor14 - m[0][1] * SubFactor16 + m[0][2] * SubFactor17);
    float Determinant = +m[0][0] * I[0][0] + m[0][1] * I[0][1] + m[0][2] * I[0][2] + m[0][3] * I[0][3];
    mat4 out;
    float *o = reinterpret_cast<float *>(&out);
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) o[c * 4 + r] = I[c][r] / Determinant;
    return out;
}

// -------------------------------------------------------------- integrator
struct PathStats {
    uint64_t closest_rays = 0, shadow_rays = 0;
};
// One path of kernel/integrators/path_trace.cu: GenerateRaysKernel (:76-108), then per
// bounce IntersectClosestKernel (:110-179), MaterialShadeKernel (:181-290) or
// EmissiveHitKernel (:337-389).  Paths only interact through the film, so running one
// path to completion equals the reference's kernel-by-kernel order for that path.
// If capture_bounce >= 0 the extension ray of that bounce is written to *captured.
static vec3 TracePath(const Scene &scene, int px, int py, int iteration, PathStats *st, int capture_bounce,
                      pt_ray *captured, bool *captured_valid) {
    const pt_scene_desc &d = *scene.d;
    Sampler sampler;
    SeedSampler(sampler, px, py, iteration);
    float s0 = Sample1D(sampler), s1 = Sample1D(sampler);
    Sample1D(sampler);  // Sample4D draws four values; the lens pair is unused by the pinhole
    Sample1D(sampler);
    Ray ray = SpawnCameraRay(d.camera, px, py, s0, s1);
    int remaining_bounces = int(d.max_depth);
    vec3 throughput = V3(1.0f), accum = V3(0.0f);
    float last_pdf = -1.0f;
    vec3 last_pos = V3(0.0f);

    for (uint32_t bounce = 0; bounce < d.max_depth; ++bounce) {
        if (capture_bounce == int(bounce) && captured) {
            captured->origin[0] = ray.origin.x; captured->origin[1] = ray.origin.y; captured->origin[2] = ray.origin.z;
            captured->direction[0] = ray.direction.x; captured->direction[1] = ray.direction.y; captured->direction[2] = ray.direction.z;
            captured->tmin = 0.0f;
            captured->tmax = FLT_MAX;
            *captured_valid = true;
        }
        Intersection is;
        bool hit = IntersectScene<CLOSEST>(scene, ray, is, FLT_MAX, nullptr);
        if (st) st->closest_rays++;
        if (!hit) {
            accum += throughput * V3(0.0f);  // black environment, path_trace.cu:351-356
            break;
        }
        const pt_material &material = d.materials[is.material_id];
        if (material.type == PT_MAT_EMISSIVE) {  // path_trace.cu:357-388
            const pt_geometry_instance &geometry = scene.geometries[is.geometry_id];
            vec3 color = SampleTexture3(d, material.tex0, is.uv);
            vec3 radiance = material.w1.f * color;
            float weight_bsdf = 1.0f;
            if (last_pdf > 0.0f) {
                const pt_area_light &light = d.lights[geometry.area_light_id];
                float dist2 = distance2(last_pos, is.pos);
                vec3 w_in = normalize(last_pos - is.pos);
                float cos_light = dot(is.normal, w_in);
                if (cos_light > 0.0f) {
                    float light_pdf = (light.power / d.total_light_power) * (1.0f / geometry.total_area) * dist2 / cos_light;
                    weight_bsdf = PowerHeuristic(last_pdf, light_pdf);
                }
            }
            accum += throughput * radiance * weight_bsdf;
            break;
        }
        // MaterialShadeKernel, path_trace.cu:181-290
        vec3 w_out = -normalize(ray.direction);
        if (!(material.type == PT_MAT_MIRROR || material.type == PT_MAT_DIELECTRIC)) {
            LightSample ls = SampleSceneLight(scene, sampler);
            vec3 w_in = normalize(ls.position - is.pos);
            Ray shadow_ray;
            shadow_ray.origin = is.pos + EPSILON * w_in;
            shadow_ray.direction = w_in;
            float dist = distance(ls.position, is.pos);
            Intersection temp;
            bool occluded = IntersectScene<ANY>(scene, shadow_ray, temp, dist - 1e-4, nullptr);
            if (st) st->shadow_rays++;
            if (!occluded) {
                float cos_surf = dot(is.normal, w_in);
                float cos_light = dot(ls.normal, -w_in);
                if (cos_surf > 0.0f && cos_light > 0.0f) {
                    vec3 bsdf = EvalMaterialBSDF(d, material, is, w_in, w_out);
                    float bsdf_pdf = MaterialPDF(d, material, is, w_in, w_out);
                    float light_pdf = ls.pdf * dist * dist / cos_light;
                    float light_weight = PowerHeuristic(light_pdf, bsdf_pdf);
                    accum += throughput * light_weight * bsdf * cos_surf * ls.radiance / light_pdf;
                }
            }
        }
        BSDF_Sample sample = SampleMaterial(d, material, is, w_out, sampler);
        float cos_surf = fabsf(dot(is.normal, sample.w_in));
        if (sample.pdf < EPSILON || cos_surf < EPSILON || sample.type == S_Invalid) break;
        throughput *= sample.bsdf * cos_surf / sample.pdf;
        if (material.type == PT_MAT_MIRROR || material.type == PT_MAT_DIELECTRIC) last_pdf = -1.0f;
        else last_pdf = sample.pdf;
        last_pos = is.pos;
        float bounces = float(d.max_depth - uint32_t(remaining_bounces));
        if (bounces >= 3) {
            const float survival_prob = std::min(std::max(Luminance(throughput), 0.05f), 0.95f);
            if (Sample1D(sampler) >= survival_prob) break;
            throughput /= survival_prob;
        }
        ray.origin = is.pos + EPSILON * sample.w_in;
        ray.direction = sample.w_in;
        remaining_bounces--;
        if (!(remaining_bounces > 0)) break;
    }
    return accum;
}

// math/tonemap.h:10-14 + path_trace.cu:409-434
static inline float ACES(float x) {
    const float a = 2.51f, b = 0.03f, c = 2.43f, dd = 0.59f, e = 0.14f;
    return clampf((x * (a * x + b)) / (x * (c * x + dd) + e), 0.0f, 1.0f);
}

template <typename F>
static void parallel_for(size_t n, int nthreads, F f) {
    if (nthreads <= 1) {
        f(0, n, 0);
        return;
    }
    std::vector<std::thread> th;
    std::atomic<size_t> next{0};
    const size_t chunk = std::max<size_t>(64, n / (size_t(nthreads) * 64));
    for (int t = 0; t < nthreads; ++t)
        th.emplace_back([&, t]() {
            for (;;) {
                size_t b = next.fetch_add(chunk);
                if (b >= n) break;
                f(b, std::min(n, b + chunk), t);
            }
        });
    for (auto &x : th) x.join();
}

}  // namespace ora

// =================================================================== C API
using namespace ora;
// ---- DebugIntegrator (kernel/integrators/debug.cu:42-239) ----
// math/color.h:19-48
static vec3 WavelengthToXYZ(float wavelength) {
    vec3 result;
    {
        float t1 = (wavelength - 442.0f) * ((wavelength < 442.0f) ? 0.0624f : 0.0374f);
        float t2 = (wavelength - 599.8f) * ((wavelength < 599.8f) ? 0.0264f : 0.0323f);
        float t3 = (wavelength - 501.1f) * ((wavelength < 501.1f) ? 0.0490f : 0.0382f);
        result.x = 0.362f * expf(-0.5f * t1 * t1) + 1.056f * expf(-0.5f * t2 * t2) - 0.065f * expf(-0.5f * t3 * t3);
    }
    {
        float t1 = (wavelength - 568.8f) * ((wavelength < 568.8f) ? 0.0213f : 0.0247f);
        float t2 = (wavelength - 530.9f) * ((wavelength < 530.9f) ? 0.0613f : 0.0322f);
        result.y = 0.821f * expf(-0.5f * t1 * t1) + 0.286f * expf(-0.5f * t2 * t2);
    }
    {
        float t1 = (wavelength - 437.0f) * ((wavelength < 437.0f) ? 0.0845f : 0.0278f);
        float t2 = (wavelength - 459.0f) * ((wavelength < 459.0f) ? 0.0385f : 0.0725f);
        result.z = 1.217f * expf(-0.5f * t1 * t1) + 0.681f * expf(-0.5f * t2 * t2);
    }
    return result;
}
// math/color.h:50-56
static vec3 XYZ_To_LinearRGB(vec3 xyz) {
    return {3.240479f * xyz.x - 1.537150f * xyz.y - 0.498535f * xyz.z,
            -0.969256f * xyz.x + 1.875991f * xyz.y + 0.041556f * xyz.z,
            0.055648f * xyz.x - 0.204043f * xyz.y + 1.057311f * xyz.z};
}
static inline float glm_clamp(float x, float lo, float hi) {  // glm::clamp = min(max(x, lo), hi)
    float m = (x < lo) ? lo : x;
    return (hi < m) ? hi : m;
}
// math/color.h:73-79
static float Gamma_sRGB(float c) {
    c = glm_clamp(c, 0.0f, 1.0f);
    if (c <= 0.0031308f) return 12.92f * c;
    return 1.055f * powf(c, 1.0f / 2.4f) - 0.055f;
}
// debug.cu:42-49
static vec3 HeatMapColor(int depth, int max_depth) {
    float t = static_cast<float>(depth) / max_depth;
    float lambda = 380.0f * (1.0f - t) + 780.0f * t;  // glm::mix(VISIBLE_SPECTRUM_BEGIN, VISIBLE_SPECTRUM_END, t)
    vec3 rgb = XYZ_To_LinearRGB(WavelengthToXYZ(lambda));
    return {Gamma_sRGB(rgb.x), Gamma_sRGB(rgb.y), Gamma_sRGB(rgb.z)};
}
// VisualizeBVH<Mode>, debug.cu:119-178.  The traversal callback runs once per node that
// survives the pop-time cull (kernel/intersection.h:61,181): every expanded interior node and
// every entered leaf of either level.
static vec3 VisualizeBVH(const Scene &scene, const Ray &ray, int mode, int bvh_visualize_depth) {
    Counters c;
    Intersection is;
    IntersectScene<CLOSEST>(scene, ray, is, FLT_MAX, &c);
    if (mode == PT_BVH_BOUNDS_HIT) {
        int depth = int(c.interior + c.tlas_leaves + c.blas_leaves);
        return HeatMapColor(depth, bvh_visualize_depth);
    }
    if (mode == PT_BVH_TLAS) {
        int depth = int(c.tlas_interior + c.tlas_leaves);
        return HeatMapColor(depth, bvh_visualize_depth);
    }
    int tri_count = int(c.tris);  // CLOSEST tests every triangle of a visited leaf: sum of primitive_count
    const float MAX_TRIS = 50.0f;
    float t = logf(tri_count + 1.0f) / logf(MAX_TRIS + 1.0f);
    t = glm_clamp(t, 0.0f, 1.0f);
    vec3 green{0.0f, 0.3f, 0.0f}, yellow{1.0f, 1.0f, 0.0f};
    return green * (1.0f - t) + yellow * t;
}
// DebugView<Mode>, debug.cu:51-117.
static vec3 DebugView(const Scene &scene, const Ray &ray, int mode) {
    const pt_scene_desc &d = *scene.d;
    const vec3 background = V3(0.4f);
    Intersection is;
    if (!IntersectScene<CLOSEST>(scene, ray, is, FLT_MAX, nullptr)) return background;
    const pt_material &material = d.materials[is.material_id];
    switch (mode) {
    case PT_DEBUG_GEOMETRIC_NORMALS: return is.normal;
    case PT_DEBUG_ALBEDO:
        if (material.type == PT_MAT_LAMBERTIAN || material.type == PT_MAT_METALLIC_ROUGHNESS) return SampleTexture3(d, material.tex0, is.uv);
        return V3(0.0f);
    case PT_DEBUG_METALLIC:
        if (material.type == PT_MAT_METALLIC_ROUGHNESS) return V3(SampleTexture1(d, material.w1.i, is.uv));
        return V3(0.0f);
    case PT_DEBUG_ROUGHNESS:
        if (material.type == PT_MAT_METALLIC_ROUGHNESS) return V3(SampleTexture1(d, material.w2, is.uv));
        return V3(0.0f);
    case PT_DEBUG_UV: return vec3{is.uv.x, is.uv.y, 0.0f};
    case PT_DEBUG_DEPTH: {
        float near_clip = d.camera.near_plane, far_clip = d.camera.far_plane, t = is.time;
        if (t < near_clip || t > far_clip) return background;
        float depth = (t - near_clip) / (far_clip - near_clip);
        return V3(glm_clamp(depth, 0.0f, 1.0f));
    }
    }
    return V3(0.0f);
}

extern "C" {

int ora_intersect_aabb(const float *o, const float *dir, const float *bmin, const float *bmax, float t_max, float *t_out) {
    Ray r{{o[0], o[1], o[2]}, {dir[0], dir[1], dir[2]}};
    float t = *t_out;
    bool h = IntersectAABB(r, box_of(bmin, bmax), t, t_max);
    *t_out = t;
    return h;
}
int ora_intersect_triangle(const float *o, const float *dir, const float *v0, const float *v1, const float *v2,
                           float t_max, float *uv, float *t_out) {
    Ray r{{o[0], o[1], o[2]}, {dir[0], dir[1], dir[2]}};
    vec2 b{0, 0};
    bool h = IntersectTriangle(r, {v0[0], v0[1], v0[2]}, {v1[0], v1[1], v1[2]}, {v2[0], v2[1], v2[2]}, b, *t_out, t_max);
    uv[0] = b.x;
    uv[1] = b.y;
    return h;
}

// ---- BVH build: handle-based so the caller can size its buffers
void *ora_tlas_build(const float *positions, const uint32_t *indices, const pt_geometry_instance *instances, uint32_t n) {
    if (n == 0) return nullptr;
    auto *b = new TLASBuilder();
    b->Build(reinterpret_cast<const vec3 *>(positions), indices, instances, n);
    return b;
}
void ora_tlas_sizes(void *h, uint32_t *tlas_nodes, uint32_t *blases, uint32_t *blas_nodes, uint32_t *tri_indices) {
    auto *b = static_cast<TLASBuilder *>(h);
    *tlas_nodes = uint32_t(b->tlas_nodes.size());
    *blases = uint32_t(b->device_blases.size());
    *blas_nodes = uint32_t(b->blas_nodes.size());
    *tri_indices = uint32_t(b->tri_indices.size());
}
void ora_tlas_get(void *h, pt_tlas_node *tlas_nodes, pt_blas *blases, pt_bvh_node *blas_nodes, uint32_t *tri_indices) {
    auto *b = static_cast<TLASBuilder *>(h);
    std::memcpy(tlas_nodes, b->tlas_nodes.data(), b->tlas_nodes.size() * sizeof(pt_tlas_node));
    std::memcpy(blases, b->device_blases.data(), b->device_blases.size() * sizeof(pt_blas));
    std::memcpy(blas_nodes, b->blas_nodes.data(), b->blas_nodes.size() * sizeof(pt_bvh_node));
    std::memcpy(tri_indices, b->tri_indices.data(), b->tri_indices.size() * sizeof(uint32_t));
}
void ora_tlas_free(void *h) { delete static_cast<TLASBuilder *>(h); }

int ora_alias_build(const float *weights, uint32_t n, float *prob, uint32_t *alias) { return AliasBuild(weights, n, prob, alias); }

// ---- light tables (fills total_area / sampler / light ids of the instances in place)
void *ora_light_data(const float *positions, const uint32_t *indices, pt_geometry_instance *geoms, uint32_t n,
                     const pt_material *materials) {
    auto *ld = new LightData();
    if (ComputeLightData(reinterpret_cast<const vec3 *>(positions), indices, geoms, n, materials, *ld)) {
        delete ld;
        return nullptr;
    }
    return ld;
}
uint32_t ora_light_count(void *h) { return uint32_t(static_cast<LightData *>(h)->lights.size()); }
float ora_light_total_power(void *h) { return static_cast<LightData *>(h)->total_power; }
void ora_light_get(void *h, pt_area_light *lights, float *light_prob, uint32_t *light_alias) {
    auto *ld = static_cast<LightData *>(h);
    std::memcpy(lights, ld->lights.data(), ld->lights.size() * sizeof(pt_area_light));
    std::memcpy(light_prob, ld->light_prob.data(), ld->light_prob.size() * sizeof(float));
    std::memcpy(light_alias, ld->light_alias.data(), ld->light_alias.size() * sizeof(uint32_t));
}
uint32_t ora_light_tri_count(void *h, uint32_t light) { return uint32_t(static_cast<LightData *>(h)->tri_prob[light].size()); }
void ora_light_tri_get(void *h, uint32_t light, float *prob, uint32_t *alias) {
    auto *ld = static_cast<LightData *>(h);
    std::memcpy(prob, ld->tri_prob[light].data(), ld->tri_prob[light].size() * sizeof(float));
    std::memcpy(alias, ld->tri_alias[light].data(), ld->tri_alias[light].size() * sizeof(uint32_t));
}
void ora_light_free(void *h) { delete static_cast<LightData *>(h); }

void ora_camera_update(const float *eye, const float *look_at, const float *up, float fovy_deg, int w, int h, pt_camera *out) {
    std::memset(out, 0, sizeof(*out));
    CameraUpdate(eye, look_at, up, fovy_deg, w, h, out);
}
void ora_transform_matrices(const float *translate, const float *rotate_deg, const float *scale, float *transform,
                            float *inv_transform, float *inv_transpose) {
    mat4 T = GetTransformMatrix({translate[0], translate[1], translate[2]}, {rotate_deg[0], rotate_deg[1], rotate_deg[2]},
                                {scale[0], scale[1], scale[2]});
    mat4 I = Inverse(T), IT = InverseTranspose(T);
    std::memcpy(transform, &T, 64);
    std::memcpy(inv_transform, &I, 64);
    std::memcpy(inv_transpose, &IT, 64);
}
void ora_matrix_derive(const float *transform, float *inv_transform, float *inv_transpose) {
    mat4 T;
    std::memcpy(&T, transform, 64);
    mat4 I = Inverse(T), IT = InverseTranspose(T);
    std::memcpy(inv_transform, &I, 64);
    std::memcpy(inv_transpose, &IT, 64);
}
void ora_matrix_mul(const float *a, const float *b, float *out) {
    mat4 A, B;
    std::memcpy(&A, a, 64);
    std::memcpy(&B, b, 64);
    mat4 C = mul(A, B);
    std::memcpy(out, &C, 64);
}

// ---- traversal over ray batches. counters[6] = interior, tlas_leaves, tris, blas_leaves, drops, max stack depth
int ora_trace_closest(const pt_scene_desc *desc, const pt_ray *rays, size_t n, pt_hit *hits, pt_intersection *full,
                      uint64_t *counters, int nthreads) {
    Scene scene = make_scene(desc);
    std::vector<Counters> ctrs(std::max(1, nthreads));
    parallel_for(n, nthreads, [&](size_t b, size_t e, int t) {
        for (size_t i = b; i < e; ++i) {
            Ray r{{rays[i].origin[0], rays[i].origin[1], rays[i].origin[2]},
                  {rays[i].direction[0], rays[i].direction[1], rays[i].direction[2]}};
            Intersection is;
            bool hit = IntersectScene<CLOSEST>(scene, r, is, rays[i].tmax, counters ? &ctrs[t] : nullptr);
            pt_hit h;
            std::memset(&h, 0, sizeof(h));
            if (hit) {
                h.t = is.time; h.u = is.bary.x; h.v = is.bary.y;
                h.prim = is.prim; h.inst = is.geometry_id; h.material_id = is.material_id;
            } else {
                h.t = -1.0f; h.prim = -1; h.inst = -1; h.material_id = -1;
            }
            hits[i] = h;
            if (full) {
                pt_intersection &f = full[i];
                std::memset(&f, 0, sizeof(f));
                f.time = hit ? is.time : -1.0f;
                f.material_id = hit ? is.material_id : -1;
                f.geometry_id = hit ? is.geometry_id : -1;
                if (hit) {
                    f.pos[0] = is.pos.x; f.pos[1] = is.pos.y; f.pos[2] = is.pos.z;
                    f.normal[0] = is.normal.x; f.normal[1] = is.normal.y; f.normal[2] = is.normal.z;
                    f.uv[0] = is.uv.x; f.uv[1] = is.uv.y;
                }
            }
        }
    });
    if (counters) {
        std::memset(counters, 0, 6 * sizeof(uint64_t));
        for (auto &c : ctrs) {
            counters[0] += c.interior; counters[1] += c.tlas_leaves; counters[2] += c.tris;
            counters[3] += c.blas_leaves; counters[4] += c.drops;
            counters[5] = std::max<uint64_t>(counters[5], c.max_depth);
        }
    }
    return 0;
}
// The steps of each ray's closest-hit traversal (Counters::ops), for scheduling studies of the device kernel:
// ray i's steps are out[offsets[i] .. offsets[i + 1]); every ray ends with an 'L' (a ray that misses the scene's
// box is that 'L' alone).  Returns the number of bytes needed; nothing past `capacity` is written.
size_t ora_trace_ops(const pt_scene_desc *desc, const pt_ray *rays, size_t n, char *out, size_t capacity, uint64_t *offsets) {
    Scene scene = make_scene(desc);
    std::string ops;
    size_t total = 0;
    for (size_t i = 0; i < n; ++i) {
        Ray r{{rays[i].origin[0], rays[i].origin[1], rays[i].origin[2]}, {rays[i].direction[0], rays[i].direction[1], rays[i].direction[2]}};
        Counters c;
        ops.clear();
        c.ops = &ops;
        Intersection is;
        IntersectScene<CLOSEST>(scene, r, is, rays[i].tmax, &c);
        ops.push_back('L');
        offsets[i] = total;
        for (size_t k = 0; k < ops.size(); ++k)
            if (total + k < capacity) out[total + k] = ops[k];
        total += ops.size();
    }
    offsets[n] = total;
    return total;
}
// The constants this file uses in place of src/math/constants.h, in that header's order (those it has a use for; 0 otherwise).
void ora_constants(float out[9]) {
    const float v[9] = {SMALL_EPSILON, EPSILON, 0.0f, PI, 0.0f, 0.0f, INV_PI, 0.0f, INV_SQRT3};
    for (int i = 0; i < 9; ++i) out[i] = v[i];
}
void ora_sincos_azimuth(const float *phi, size_t n, float *sin_phi, float *cos_phi) {
    for (size_t i = 0; i < n; ++i) SinCosAzimuth(phi[i], sin_phi[i], cos_phi[i]);
}
int ora_trace_any(const pt_scene_desc *desc, const pt_ray *rays, size_t n, uint8_t *occluded, int nthreads) {
    Scene scene = make_scene(desc);
    parallel_for(n, nthreads, [&](size_t b, size_t e, int) {
        for (size_t i = b; i < e; ++i) {
            Ray r{{rays[i].origin[0], rays[i].origin[1], rays[i].origin[2]},
                  {rays[i].direction[0], rays[i].direction[1], rays[i].direction[2]}};
            Intersection is;
            occluded[i] = IntersectScene<ANY>(scene, r, is, rays[i].tmax, nullptr) ? 1 : 0;
        }
    });
    return 0;
}
int ora_trace_naive(const pt_scene_desc *desc, const pt_ray *rays, size_t n, pt_hit *hits, int nthreads) {
    Scene scene = make_scene(desc);
    parallel_for(n, nthreads, [&](size_t b, size_t e, int) {
        for (size_t i = b; i < e; ++i) {
            Ray r{{rays[i].origin[0], rays[i].origin[1], rays[i].origin[2]},
                  {rays[i].direction[0], rays[i].direction[1], rays[i].direction[2]}};
            Intersection is;
            bool hit = IntersectSceneNaive(scene, r, is, rays[i].tmax);
            pt_hit h;
            std::memset(&h, 0, sizeof(h));
            if (hit) {
                h.t = is.time; h.u = is.bary.x; h.v = is.bary.y;
                h.prim = is.prim; h.inst = is.geometry_id; h.material_id = is.material_id;
            } else {
                h.t = -1.0f; h.prim = -1; h.inst = -1; h.material_id = -1;
            }
            hits[i] = h;
        }
    });
    return 0;
}

// Incoherent ray set: origin uniform in the TLAS root box scaled 1.1x about its centre,
// direction uniform on the sphere (z = 1-2*xi1, phi = 2*pi*xi2), drawn from the PCG
// stream of kernel/sampler/independent.h:25-32 with state = (i << 32) | 0xdeadbeef after two
// discarded draws.
void ora_generate_incoherent_rays(const pt_scene_desc *desc, pt_ray *rays, size_t n, uint64_t first_index) {
    const pt_tlas_node &root = desc->tlas_nodes[0];
    for (size_t k = 0; k < n; ++k) {
        Sampler s;
        s.state = ((first_index + k) << 32) | 0xdeadbeefull;
        Sample1D(s);  // the outputs of an unmixed counter state are ~0: two LCG steps mix it first
        Sample1D(s);
        float o[3];
        for (int a = 0; a < 3; ++a) {
            float c = 0.5f * (root.min[a] + root.max[a]);
            float hw = 0.55f * (root.max[a] - root.min[a]);
            float xi = Sample1D(s);
            o[a] = c + hw * (2.0f * xi - 1.0f);
        }
        float z = 1.0f - 2.0f * Sample1D(s);
        float phi = 2.0f * PI * Sample1D(s);
        float r = sqrtf(fmaxf(0.0f, 1.0f - z * z));
        rays[k].origin[0] = o[0]; rays[k].origin[1] = o[1]; rays[k].origin[2] = o[2];
        rays[k].tmin = 0.0f;
        float sin_phi, cos_phi;
        SinCosAzimuth(phi, sin_phi, cos_phi);
        rays[k].direction[0] = r * cos_phi; rays[k].direction[1] = r * sin_phi; rays[k].direction[2] = z;
        rays[k].tmax = FLT_MAX;
    }
}

// One iteration of PathIntegrator::Render (path_trace.cu:436-483) without the display pass:
// film[p] += accum(p) with the NaN scrub of FinalGather (:391-407).  stats[2] = closest, shadow rays.
int ora_render(const pt_scene_desc *desc, int width, int height, int iteration, float *film, uint64_t *stats, int nthreads) {
    Scene scene = make_scene(desc);
    std::vector<PathStats> st(std::max(1, nthreads));
    parallel_for(size_t(width) * height, nthreads, [&](size_t b, size_t e, int t) {
        for (size_t idx = b; idx < e; ++idx) {
            int x = int(idx % width), y = int(idx / width);
            vec3 a = TracePath(scene, x, y, iteration, &st[t], -1, nullptr, nullptr);
            if (std::isnan(a.x) || std::isnan(a.y) || std::isnan(a.z)) a = V3(0.0f);
            film[idx * 3 + 0] += a.x;
            film[idx * 3 + 1] += a.y;
            film[idx * 3 + 2] += a.z;
        }
    });
    if (stats) {
        stats[0] = stats[1] = 0;
        for (auto &s : st) { stats[0] += s.closest_rays; stats[1] += s.shadow_rays; }
    }
    return 0;
}
// Extension rays entering bounce `bounce` of `iteration`, in pixel order; returns the count.
size_t ora_capture_bounce_rays(const pt_scene_desc *desc, int width, int height, int iteration, int bounce, pt_ray *out, size_t cap) {
    Scene scene = make_scene(desc);
    size_t n = 0;
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x) {
            pt_ray r;
            bool valid = false;
            TracePath(scene, x, y, iteration, nullptr, bounce, &r, &valid);
            if (valid && n < cap) out[n++] = r;
        }
    return n;
}
// PostprocessAndUpload, path_trace.cu:409-434: film/(iteration+1) -> ACES -> (int)(p*255.0) clamped.
void ora_tonemap(const float *film, int width, int height, int iteration, uint8_t *rgba) {
    for (size_t i = 0; i < size_t(width) * height; ++i) {
        for (int c = 0; c < 3; ++c) {
            float radiance = film[i * 3 + c] / static_cast<float>(iteration + 1);
            float p = ACES(radiance);
            int v = (int)(p * 255.0);
            rgba[i * 4 + c] = uint8_t(std::min(std::max(v, 0), 255));
        }
        rgba[i * 4 + 3] = 255;
    }
}
// DebugIntegrator::Render (debug.cu:198-239) for one frame: the BVH view when visualize_bvh is
// set, replaced by the debug view when debug_mode is not None; neither = nothing written.
// image = W*H*3 fp32 (the reference's m_image) and rgba = W*H*4 (SendImageToBuffer, :180-196);
// either may be null.
int ora_debug_render(const pt_scene_desc *desc, int width, int height, const pt_integrator_state *state, float *image,
                     uint8_t *rgba, int nthreads) {
    const bool view = state->debug_mode != PT_DEBUG_NONE;
    if (!view && !state->visualize_bvh) return 0;
    Scene scene = make_scene(desc);
    parallel_for(size_t(width) * height, nthreads, [&](size_t b, size_t e, int) {
        for (size_t idx = b; idx < e; ++idx) {
            int x = int(idx % width), y = int(idx / width);
            Ray ray = SpawnCameraRay(desc->camera, x, y, 0.5f, 0.5f);
            vec3 c = view ? DebugView(scene, ray, state->debug_mode)
                          : VisualizeBVH(scene, ray, state->bvh_visualize_type, state->bvh_visualize_depth);
            if (image) { image[idx * 3 + 0] = c.x; image[idx * 3 + 1] = c.y; image[idx * 3 + 2] = c.z; }
            if (rgba) {
                const float p[3] = {c.x, c.y, c.z};
                for (int k = 0; k < 3; ++k) {
                    int v = (int)(p[k] * 255.0);
                    rgba[idx * 4 + k] = uint8_t(std::min(std::max(v, 0), 255));
                }
                rgba[idx * 4 + 3] = 255;
            }
        }
    });
    return 0;
}
// Direct access to the pieces the shading tests compare one by one.
void ora_sample_texture3(const pt_scene_desc *desc, int texture_id, float u, float v, float *rgb) {
    vec3 c = SampleTexture3(*desc, texture_id, {u, v});
    rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
}
float ora_sample_texture1(const pt_scene_desc *desc, int texture_id, float u, float v) { return SampleTexture1(*desc, texture_id, {u, v}); }
void ora_set_texture_filter(int options) { g_texture_filter = options; }

// Math::PCG_Hash (math/hash.h:5-9) as the oracle restates it, for the known-answer vectors of the reference's own header.
uint32_t ora_pcg_hash(uint32_t input) { return PCG_Hash(input); }

// The oracle's Stack (utils/stack.h:3-22) driven by a script: v > 0 pushes node v, 0 pops.  out receives, per
// pop, the popped node and whether the stack is empty afterwards; returns the final top.
uint32_t ora_stack_script(const uint32_t *script, int n, uint32_t *out) {
    Stack s;
    s.drops = nullptr;
    s.maxdepth = nullptr;
    int k = 0;
    for (int i = 0; i < n; ++i) {
        if (script[i] > 0) {
            s.Push(StackItem{script[i], 0.0f});
        } else {
            out[k++] = s.Pop().node_index;
            out[k++] = s.Empty() ? 1u : 0u;
        }
    }
    return s.top;
}

void ora_pcg_stream(int px, int py, int iteration, int n, float *out) {
    Sampler s;
    SeedSampler(s, px, py, iteration);
    for (int i = 0; i < n; ++i) out[i] = Sample1D(s);
}
void ora_camera_ray(const pt_camera *cam, int i, int j, float sx, float sy, float *o, float *dir) {
    Ray r = SpawnCameraRay(*cam, i, j, sx, sy);
    o[0] = r.origin.x; o[1] = r.origin.y; o[2] = r.origin.z;
    dir[0] = r.direction.x; dir[1] = r.direction.y; dir[2] = r.direction.z;
}
}  // extern "C"
