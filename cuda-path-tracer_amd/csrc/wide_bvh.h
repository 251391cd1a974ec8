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

}  // namespace ptw
