// Host side of trace mode 4: every instance's triangles carried into world space and one 4-wide BVH
// built over all of them.
//
// The reference keeps a two-level structure (TLAS over instances, one BLAS per mesh, bvh.cpp:266-331 and
// :36-113) and moves every ray into an instance's object space and back (kernel/intersection.h:37-50,
// :139-149).  For a scene whose meshes are placed once or a few times that bookkeeping is a large share of
// a ray's steps, and the instances' boxes overlap, so a ray often searches several of them.  A single tree
// over world-space triangles has neither cost.  It is built when the flattened scene is not much larger
// than the instanced one (FlatPolicy); scenes made of thousands of copies of a mesh keep two levels.
//
// The tree is this library's own (top-down, binned surface-area splits, leaves of up to kFlatMaxLeaf
// triangles when splitting would not pay), not the reference's: the traversal only has to find the same
// triangle, and what it reports for it is recomputed from the instance's own records (FlatTraversal).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "wide_bvh.h"

namespace ptw {

using ptd::DTri;
using ptd::kLeafBit;

#ifndef PT_FLAT_MAX_LEAF
#define PT_FLAT_MAX_LEAF 4
#endif
constexpr uint32_t kFlatMaxLeaf = PT_FLAT_MAX_LEAF;  // <= 7: a leaf reference holds count - 1 in three bits
static_assert(kFlatMaxLeaf >= 1 && kFlatMaxLeaf <= 7, "leaf references hold up to 7 triangles");
#ifndef PT_FLAT_TRAVERSAL_COST
#define PT_FLAT_TRAVERSAL_COST 1.0f  // cost of opening a node, in triangle tests
#endif

// Flatten when the world-space copy stays within a small multiple of the instanced scene and of a fixed
// budget: the tree and its records should stay cache resident like the instanced scene does.
inline bool FlatPolicy(uint64_t flat_triangles, uint64_t unique_triangles) {
    if (flat_triangles == 0 || flat_triangles > (1ull << 25)) return false;
    return flat_triangles <= 2 * unique_triangles + 65536;
}

struct FlatTree {
    struct Node {
        Box box;
        uint32_t left, right;   // children; left == 0xFFFFFFFF: leaf
        uint32_t first, count;  // leaf: its items are order[first .. first + count)
    };
    std::vector<Node> nodes;      // node 0 is the root
    std::vector<uint32_t> order;  // item indices in leaf order
};

// Binned surface-area build over `n` boxes (16 bins per axis, centroids).  A range of at most `max_leaf`
// items becomes a leaf when no split is cheaper than testing the items one by one.
inline FlatTree BuildFlatTree(const Box *boxes, uint32_t n, uint32_t max_leaf, float traversal_cost = PT_FLAT_TRAVERSAL_COST) {
    FlatTree t;
    if (n == 0) return t;
    t.order.resize(n);
    for (uint32_t i = 0; i < n; ++i) t.order[i] = i;
    std::vector<float> cen(size_t(n) * 3);
    for (uint32_t i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a) cen[size_t(i) * 3 + a] = 0.5f * (boxes[i].mn[a] + boxes[i].mx[a]);
    struct Job {
        uint32_t node, first, count;
    };
    std::vector<Job> work;
    t.nodes.reserve(2 * size_t(n));
    t.nodes.push_back({});
    work.push_back({0u, 0u, n});
    constexpr int kBins = 16;
    std::vector<uint32_t> &order = t.order;
    while (!work.empty()) {
        const Job j = work.back();
        work.pop_back();
        Box bounds = boxes[order[j.first]];
        float cmin[3], cmax[3];
        for (int a = 0; a < 3; ++a) cmin[a] = cmax[a] = cen[size_t(order[j.first]) * 3 + a];
        for (uint32_t i = 1; i < j.count; ++i) {
            const uint32_t id = order[j.first + i];
            bounds = Union(bounds, boxes[id]);
            for (int a = 0; a < 3; ++a) {
                const float c = cen[size_t(id) * 3 + a];
                cmin[a] = std::fmin(cmin[a], c);
                cmax[a] = std::fmax(cmax[a], c);
            }
        }
        FlatTree::Node &node = t.nodes[j.node];
        node.box = bounds;
        node.first = j.first;
        node.count = j.count;
        node.left = node.right = 0xFFFFFFFFu;
        if (j.count == 1) continue;
        int best_axis = -1, best_bin = 0;
        float best_cost = 0.0f;
        for (int a = 0; a < 3; ++a) {
            const float ext = cmax[a] - cmin[a];
            if (!(ext > 0.0f)) continue;
            Box bin_box[kBins];
            uint32_t bin_count[kBins] = {};
            const float scale = float(kBins) / ext;
            for (uint32_t i = 0; i < j.count; ++i) {
                const uint32_t id = order[j.first + i];
                int b = int((cen[size_t(id) * 3 + a] - cmin[a]) * scale);
                b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                bin_box[b] = bin_count[b] ? Union(bin_box[b], boxes[id]) : boxes[id];
                bin_count[b]++;
            }
            float right_area[kBins];
            uint32_t right_count[kBins];
            Box acc{};
            uint32_t cnt = 0;
            for (int b = kBins - 1; b > 0; --b) {
                if (bin_count[b]) acc = cnt ? Union(acc, bin_box[b]) : bin_box[b];
                cnt += bin_count[b];
                right_area[b] = cnt ? HalfArea(acc) : 0.0f;
                right_count[b] = cnt;
            }
            cnt = 0;
            for (int b = 0; b < kBins - 1; ++b) {
                if (bin_count[b]) acc = cnt ? Union(acc, bin_box[b]) : bin_box[b];
                cnt += bin_count[b];
                if (cnt == 0 || right_count[b + 1] == 0) continue;
                const float cost = HalfArea(acc) * float(cnt) + right_area[b + 1] * float(right_count[b + 1]);
                if (best_axis < 0 || cost < best_cost) {
                    best_axis = a;
                    best_bin = b;
                    best_cost = cost;
                }
            }
        }
        if (j.count <= max_leaf) {
            const float area = HalfArea(bounds);
            // no plane separates the centroids, or opening two children costs more than testing the items here
            if (best_axis < 0 || !(traversal_cost * area + best_cost < area * float(j.count))) continue;
        }
        uint32_t mid = j.first + j.count / 2;
        uint32_t *first = order.data() + j.first, *last = first + j.count;
        if (best_axis >= 0) {
            const int a = best_axis;
            const float scale = float(kBins) / (cmax[a] - cmin[a]);
            uint32_t *split = std::partition(first, last, [&](uint32_t id) {
                int b = int((cen[size_t(id) * 3 + a] - cmin[a]) * scale);
                b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                return b <= best_bin;
            });
            mid = uint32_t(split - order.data());
        }
        if (best_axis < 0 || mid == j.first || mid == j.first + j.count) {  // coincident centroids: split by count
            int a = 0;
            for (int k = 1; k < 3; ++k)
                if (bounds.mx[k] - bounds.mn[k] > bounds.mx[a] - bounds.mn[a]) a = k;
            mid = j.first + j.count / 2;
            std::nth_element(first, order.data() + mid, last, [&](uint32_t x, uint32_t y) { return cen[size_t(x) * 3 + a] < cen[size_t(y) * 3 + a]; });
        }
        const uint32_t l = uint32_t(t.nodes.size());
        t.nodes.push_back({});
        t.nodes.push_back({});
        t.nodes[j.node].left = l;
        t.nodes[j.node].right = l + 1;
        work.push_back({l, j.first, mid - j.first});
        work.push_back({l + 1, mid, j.first + j.count - mid});
    }
    return t;
}

struct FlatScene {
    std::vector<DNode4> nodes4;
    std::vector<DTri> tris;  // world-space records in leaf order
    uint32_t root = kNoRef;
};

// `world` holds one record per (instance, triangle) with c.y = the triangle's record in the instanced
// scene and c.z = the instance; the result's records are the same ones in leaf order.
inline FlatScene BuildFlatScene(const std::vector<DTri> &world) {
    FlatScene f;
    const uint32_t n = uint32_t(world.size());
    if (n == 0) return f;
    std::vector<Box> boxes(n);
    for (uint32_t i = 0; i < n; ++i) {
        const DTri &r = world[i];
        const float p[3][3] = {{r.a.x, r.a.y, r.a.z}, {r.a.w, r.b.x, r.b.y}, {r.b.z, r.b.w, r.c.x}};
        for (int a = 0; a < 3; ++a) {
            boxes[i].mn[a] = std::fmin(p[0][a], std::fmin(p[1][a], p[2][a]));
            boxes[i].mx[a] = std::fmax(p[0][a], std::fmax(p[1][a], p[2][a]));
        }
    }
    uint32_t max_leaf = kFlatMaxLeaf;
    float traversal_cost = PT_FLAT_TRAVERSAL_COST;
    if (const char *e = getenv("PTAMD_FLAT_MAX_LEAF")) max_leaf = uint32_t(std::min(7, std::max(1, atoi(e))));  // tuning aids
    if (const char *e = getenv("PTAMD_FLAT_TRAVERSAL_COST")) traversal_cost = float(atof(e));
    const FlatTree t = BuildFlatTree(boxes.data(), n, max_leaf, traversal_cost);
    f.tris.resize(n);
    for (uint32_t i = 0; i < n; ++i) f.tris[i] = world[t.order[i]];
    struct Tree {
        const FlatTree *t;
        bool is_leaf(uint32_t i) const { return t->nodes[i].left == 0xFFFFFFFFu; }
        uint32_t left(uint32_t i) const { return t->nodes[i].left; }
        uint32_t right(uint32_t i) const { return t->nodes[i].right; }
        Box box(uint32_t i) const { return t->nodes[i].box; }
        uint32_t leaf_ref(uint32_t i) const { return kLeafBit | ((t->nodes[i].count - 1u) << 28) | t->nodes[i].first; }
    };
    if (t.nodes[0].left == 0xFFFFFFFFu) {
        // a lone leaf: give it a wide node of its own so that the root is always an interior reference
        f.nodes4.emplace_back();
        const Box b = t.nodes[0].box;
        const uint32_t ref = Tree{&t}.leaf_ref(0);
        Encode(f.nodes4[0], &b, &ref, 1);
        f.root = 0;
    } else {
        f.root = Collapse(Tree{&t}, 0u, f.nodes4);
    }
    return f;
}

}  // namespace ptw
