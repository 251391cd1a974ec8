// Host side of the 4-wide traversal (trace modes 2 and 3): collapses binary BVHs into 4-wide nodes
// and builds a top-level tree of its own over the instance boxes.
//
// A binary interior node becomes a wide node whose children are found by repeatedly opening the
// interior child with the largest surface area until there are four children or only leaves are
// left (the usual surface-area collapse).  Leaves are kept as the binary builder made them
// (BLAS::Subdivide, bvh.cpp:67-113), so the triangle records, their order and the leaf references
// are shared with the binary layout, and every child box is the reference's own fp32 box: a box
// test in the wide tree decides exactly as the same box's test in the binary tree does.
//
// The reference pairs instances bottom-up (TLAS::Build, bvh.cpp:266-331), which gives deep, lopsided
// trees for thousands of overlapping instances.  The wide layout does not have to reproduce that
// tree, only its leaves (the instances' world boxes, bit for bit), so its top level is built top
// down with a binned surface-area split (BuildTopLevel) and comes out log-depth.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "dscene.h"

namespace ptw {

using ptd::DNode4;
using ptd::kNoRef;

struct Box {
    float mn[3], mx[3];
};

inline float HalfArea(const Box &b) {
    const float ex = b.mx[0] - b.mn[0], ey = b.mx[1] - b.mn[1], ez = b.mx[2] - b.mn[2];
    return ex * ey + ey * ez + ez * ex;
}

inline void Encode(DNode4 &out, const Box *boxes, const uint32_t *refs, int n) {
    const float nan = std::nanf("");
    float v[32];
    for (int k = 0; k < 4; ++k) {
        for (int a = 0; a < 3; ++a) {
            v[8 * a + k] = k < n ? boxes[k].mn[a] : nan;
            v[8 * a + 4 + k] = k < n ? boxes[k].mx[a] : nan;
        }
        const uint32_t r = k < n ? refs[k] : kNoRef;
        std::memcpy(&v[24 + k], &r, 4);
        v[28 + k] = 0.0f;
    }
    std::memcpy(&out, v, sizeof(out));
}

// Tree: is_leaf(i), left(i), right(i), box(i) -> Box, leaf_ref(i) -> reference word of a leaf.
// Appends the wide nodes of the tree under `root` to `out` and returns the root's reference.
template <class Tree>
uint32_t Collapse(const Tree &tree, uint32_t root, std::vector<DNode4> &out) {
    if (tree.is_leaf(root)) return tree.leaf_ref(root);
    struct Item {
        uint32_t node, slot;
    };
    std::vector<Item> work;
    const uint32_t root_slot = uint32_t(out.size());
    out.emplace_back();
    work.push_back({root, root_slot});
    while (!work.empty()) {
        const Item it = work.back();
        work.pop_back();
        uint32_t kids[4] = {tree.left(it.node), tree.right(it.node), 0, 0};
        int n = 2;
        while (n < 4) {
            int pick = -1;
            float best = -1.0f;
            for (int k = 0; k < n; ++k) {
                if (tree.is_leaf(kids[k])) continue;
                const float area = HalfArea(tree.box(kids[k]));
                if (pick < 0 || area > best) {
                    pick = k;
                    best = area;
                }
            }
            if (pick < 0) break;
            const uint32_t open = kids[pick];
            for (int k = n; k > pick + 1; --k) kids[k] = kids[k - 1];
            kids[pick] = tree.left(open);
            kids[pick + 1] = tree.right(open);
            ++n;
        }
        Box boxes[4];
        uint32_t refs[4];
        for (int k = 0; k < n; ++k) {
            boxes[k] = tree.box(kids[k]);
            if (tree.is_leaf(kids[k])) {
                refs[k] = tree.leaf_ref(kids[k]);
            } else {
                refs[k] = uint32_t(out.size());
                out.emplace_back();
                work.push_back({kids[k], refs[k]});
            }
        }
        Encode(out[it.slot], boxes, refs, n);
    }
    return root_slot;
}

// A binary tree over `n` boxes (leaf i = box i), built top down: centroids are binned along each axis
// (16 bins), the cheapest surface-area split is taken, a median split along the widest axis when no
// plane separates the centroids.  One box per leaf.  Returned as arrays a Tree adapter can walk.
struct TopLevel {
    struct Node {
        Box box;
        uint32_t left, right;  // children; left == right == 0xFFFFFFFF: leaf
        uint32_t item;         // leaf: index of the box
    };
    std::vector<Node> nodes;   // node 0 is the root (when there is at least one box)
};

inline Box Union(const Box &a, const Box &b) {
    Box r;
    for (int k = 0; k < 3; ++k) {
        r.mn[k] = std::fmin(a.mn[k], b.mn[k]);
        r.mx[k] = std::fmax(a.mx[k], b.mx[k]);
    }
    return r;
}

inline TopLevel BuildTopLevel(const Box *boxes, uint32_t n) {
    TopLevel t;
    if (n == 0) return t;
    std::vector<uint32_t> order(n);
    for (uint32_t i = 0; i < n; ++i) order[i] = i;
    auto centroid = [&](uint32_t i, int a) { return 0.5f * (boxes[i].mn[a] + boxes[i].mx[a]); };
    struct Job {
        uint32_t node, first, count;
    };
    std::vector<Job> work;
    t.nodes.reserve(2 * size_t(n));
    t.nodes.push_back({});
    work.push_back({0u, 0u, n});
    constexpr int kBins = 16;
    while (!work.empty()) {
        const Job j = work.back();
        work.pop_back();
        Box bounds = boxes[order[j.first]];
        for (uint32_t i = 1; i < j.count; ++i) bounds = Union(bounds, boxes[order[j.first + i]]);
        t.nodes[j.node].box = bounds;
        if (j.count == 1) {
            t.nodes[j.node].left = t.nodes[j.node].right = 0xFFFFFFFFu;
            t.nodes[j.node].item = order[j.first];
            continue;
        }
        float cmin[3], cmax[3];
        for (int a = 0; a < 3; ++a) {
            cmin[a] = cmax[a] = centroid(order[j.first], a);
            for (uint32_t i = 1; i < j.count; ++i) {
                const float c = centroid(order[j.first + i], a);
                cmin[a] = std::fmin(cmin[a], c);
                cmax[a] = std::fmax(cmax[a], c);
            }
        }
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
                int b = int((centroid(id, a) - cmin[a]) * scale);
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
        uint32_t mid = j.first + j.count / 2;
        uint32_t *first = order.data() + j.first, *last = first + j.count;
        if (best_axis >= 0) {
            const int a = best_axis;
            const float scale = float(kBins) / (cmax[a] - cmin[a]);
            uint32_t *split = std::partition(first, last, [&](uint32_t id) {
                int b = int((centroid(id, a) - cmin[a]) * scale);
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
            std::nth_element(first, order.data() + mid, last, [&](uint32_t x, uint32_t y) { return centroid(x, a) < centroid(y, a); });
        }
        const uint32_t l = uint32_t(t.nodes.size());
        t.nodes.push_back({});
        t.nodes.push_back({});
        t.nodes[j.node].left = l;
        t.nodes[j.node].right = l + 1;
        t.nodes[j.node].item = 0;
        work.push_back({l, j.first, mid - j.first});
        work.push_back({l + 1, mid, j.first + j.count - mid});
    }
    return t;
}


// A tighter world-space box for an instance than the reference's (the world box of the transformed corners of
// the mesh's object-space box, bvh.cpp:273-280, which for a rotated mesh has up to three times the surface,
// so that most rays crossing it miss the mesh).  The extent of the transformed mesh along world axis a is the
// mesh's support function in the direction of row a of the transform, and the mesh's own BVH finds that by
// branch and bound: a box is opened only while its own support can still beat the best vertex found (which
// relies on every node's box containing the triangles under it, as any BVH builder's do).
//
// Tree: is_leaf(i), left(i), right(i), box(i) as for Collapse, plus for_each_vertex(i, f) calling f(x, y, z)
// for the corners of the triangles of leaf i.  Returns max over the mesh's vertices of dir . v; when the search is
// cut short after `budget` boxes the unopened boxes' own supports are folded in, so the result never falls short.
template <class Tree>
double MeshSupport(const Tree &tree, uint32_t root, const double dir[3], int budget = 4096) {
    auto box_support = [&](uint32_t n) {
        const Box b = tree.box(n);
        double s = 0.0;
        for (int a = 0; a < 3; ++a) s += std::fmax(dir[a] * double(b.mn[a]), dir[a] * double(b.mx[a]));
        return s;
    };
    double best = -INFINITY;
    std::vector<uint32_t> stack{root};
    while (!stack.empty()) {
        const uint32_t n = stack.back();
        stack.pop_back();
        const double ub = box_support(n);
        if (!(ub > best)) continue;
        if (--budget < 0) {
            best = std::fmax(best, ub);  // not opened: its own support stands in for its vertices
            continue;
        }
        if (tree.is_leaf(n)) {
            tree.for_each_vertex(n, [&](float x, float y, float z) { best = std::fmax(best, dir[0] * double(x) + dir[1] * double(y) + dir[2] * double(z)); });
        } else {
            const uint32_t l = tree.left(n), r = tree.right(n);
            const bool left_first = box_support(l) >= box_support(r);
            stack.push_back(left_first ? r : l);
            stack.push_back(left_first ? l : r);  // the more promising child is opened first
        }
    }
    return best;
}

// The world box of the transformed mesh (column-major 4x4 transform), widened by a rounding margin and clipped
// to the reference's box: it contains every transformed triangle and never exceeds what the reference tests.
// A ray that misses it cannot hit the instance, so skipping the instance changes no result.
// `scene_extent` is the size of the scene's root box (its diagonal): the rounding of a world-space box test
// grows with the distance the ray has travelled, not with the mesh's own coordinates, so for a small mesh near
// the origin that is hit from across the scene the margin must follow the scene's scale.
template <class Tree>
Box TightInstanceBox(const Tree &tree, uint32_t root, const float *transform, const Box &reference_box, double scene_extent = 0.0) {
    Box r;
    for (int a = 0; a < 3; ++a) {
        const double row[3] = {double(transform[a]), double(transform[4 + a]), double(transform[8 + a])};
        const double neg[3] = {-row[0], -row[1], -row[2]};
        const double hi = MeshSupport(tree, root, row) + double(transform[12 + a]);
        const double lo = -MeshSupport(tree, root, neg) + double(transform[12 + a]);
        // a few units in the last place of the largest coordinate involved: the reference moves the ray into
        // object space instead, and the two roundings must not disagree about a triangle on the box's face
        const double margin = std::fmax(1e-5 * std::fmax(std::fmax(std::fabs(lo), std::fabs(hi)), hi - lo), 1e-6 * scene_extent);
        r.mn[a] = std::fmax(float(lo - margin), reference_box.mn[a]);
        r.mx[a] = std::fmin(float(hi + margin), reference_box.mx[a]);
        if (!(r.mn[a] <= r.mx[a])) return reference_box;  // not finite, or clipped away: keep the reference's box
    }
    return r;
}

}  // namespace ptw
