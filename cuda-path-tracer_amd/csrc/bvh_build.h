// Acceleration-structure build on the device: every BLAS (binned SAH, 8 bins) and the TLAS
// (agglomerative pairing) of a scene, producing the arrays of the reference's builder
// (bvh.cpp:5-332) bit for bit: same tree, same node numbering, same triangle permutation.
//
// The reference builds one node at a time, depth first.  Here all nodes of one depth, over all
// meshes at once, are processed per pass, with one thread per triangle position:
//   * bounds and bins are reductions; min / max do not depend on the order of their operands
//     except for the sign of a zero, so node boxes are reduced as (value, position) keys and
//     read back from the winning triangle: the box holds the value the reference's sequential
//     union would have kept (the first one met among equals);
//   * the plane sweep of a node (21 candidates, first strictly smaller cost wins) is sequential
//     code run by one thread per node;
//   * the reference partitions a node's triangle list in place with two cursors
//     (bvh.cpp:80-86), which is not a stable partition.  Where each entry ends up has a closed
//     form in terms of ranks (see PartitionDest), so the same permutation is produced by a scan
//     and a scatter;
//   * nodes are numbered breadth first while building and renumbered at the end: the reference
//     allocates a node's two children when it subdivides the node, in depth-first order, so
//     the left child of the k-th subdivided node (preorder) is node 1 + 2k.
// Nodes of more than kSmall triangles are processed that way.  A node of kSmall triangles or fewer
// is handed, with its whole subtree, to one wave that keeps the triangles in LDS and walks the
// subtree depth first like the reference (reductions across the lanes, the plane sweep and the
// two-cursor partition on one lane): thousands of such subtrees run side by side, and the many
// tiny nodes at the bottom of a tree cost no passes over the whole array.
// The TLAS pairing loop (bvh.cpp:283-331) is a chain of arg-min searches, each depending on
// the previous merge; one workgroup runs the chain with every search spread over its lanes.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/ptamd.h"

namespace ptb {

constexpr int kBins = 8;
constexpr uint32_t kSmall = 256;  // subtrees of this many triangles or fewer are built by one wave
constexpr uint32_t kScanBlock = 256, kScanItems = 8, kScanTile = kScanBlock * kScanItems;
constexpr uint32_t kMaxPrims = kScanTile * 8192u;  // one level of block sums is scanned by a single block

#define PTB_D __device__ __forceinline__

PTB_D float fmin_ref(float a, float b) { return b < a ? b : a; }  // glm::min
PTB_D float fmax_ref(float a, float b) { return a < b ? b : a; }  // glm::max

// Monotone float -> unsigned map (negative values reversed below the positive ones).
PTB_D uint32_t enc(float f) {
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
PTB_D float dec(uint32_t e) { return __uint_as_float((e & 0x80000000u) ? (e & 0x7FFFFFFFu) : ~e); }
// Same, with both zeros on one key: equal values then compete on the position alone.
PTB_D uint32_t enc_eq(float f) { return enc(f == 0.0f ? 0.0f : f); }

struct BNode {  // build-time node, breadth-first numbering; roots are nodes 0..blas_count-1
    uint32_t first, count;  // positions [first, first + count) of the order array
    uint32_t left;          // breadth-first index of the left child (right = left + 1); 0 = leaf
    uint32_t blas;
    float lo[3], hi[3];
    uint32_t cb_lo[3], cb_hi[3];  // centroid bounds of this pass, enc()
    int32_t axis;                 // split axis of this pass, -1 = none
    uint32_t bin_slot;            // this pass's Bins record (only nodes that may split own one)
    float pos;
    uint32_t n_right;             // entries not left of the plane
    uint32_t pairs, front_rights; // K and K' of PartitionDest
    uint32_t internal;            // subdivided nodes in the subtree (itself included)
    uint32_t rank;                // preorder rank among its mesh's subdivided nodes
    uint32_t index;               // final index inside its mesh's node array
    uint32_t small;               // 1: root of a subtree built by one wave
    uint32_t owner, local;        // nodes inside such a subtree: its root, and the node's number in the
                                  // subtree's own depth-first numbering (root 0)
};

struct Bins {  // 8 bins on each axis: enc()-coded boxes and counts
    uint32_t lo[3][kBins][3], hi[3][kBins][3];
    uint32_t count[3][kBins];
};

struct Prims {  // per triangle, mesh-major order (the mesh's triangles keep their own order)
    float *lo[3], *hi[3], *centroid[3];
    uint32_t *blas;  // mesh of the triangle
};

struct MeshInfo {
    uint32_t first_index;  // into the scene's index array
    uint32_t tri_first;    // first triangle position of the mesh
    uint32_t tri_count;
    uint32_t node_offset;  // first node of the mesh in the flattened node array (set at the end)
};

// ------------------------------------------------------------------ setup
__global__ void kb_prims(const float *positions, const uint32_t *indices, const MeshInfo *meshes, uint32_t mesh_count, Prims P, uint32_t n,
                         uint32_t *order, uint32_t *node_of) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    uint32_t lo = 0, hi = mesh_count;  // last mesh with tri_first <= g among the non-empty ones
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (meshes[mid].tri_first <= g) lo = mid; else hi = mid;
    }
    while (meshes[lo].tri_count == 0 || g - meshes[lo].tri_first >= meshes[lo].tri_count) --lo;  // empty meshes share a tri_first
    const MeshInfo m = meshes[lo];
    const uint32_t t = g - m.tri_first;
    const float *v[3];
    for (int k = 0; k < 3; ++k) v[k] = positions + 3 * size_t(indices[m.first_index + 3 * t + k]);
    for (int a = 0; a < 3; ++a) {  // AddPoint x3 (bvh.cpp:38-41), centroid (bvh.cpp:36)
        float mn = fmin_ref(FLT_MAX, v[0][a]), mx = fmax_ref(-FLT_MAX, v[0][a]);
        mn = fmin_ref(mn, v[1][a]);
        mx = fmax_ref(mx, v[1][a]);
        mn = fmin_ref(mn, v[2][a]);
        mx = fmax_ref(mx, v[2][a]);
        P.lo[a][g] = mn;
        P.hi[a][g] = mx;
        P.centroid[a][g] = ((v[0][a] + v[1][a]) + v[2][a]) * (1.0f / 3.0f);
    }
    P.blas[g] = lo;
    order[g] = g;
    node_of[g] = lo;
}

__global__ void kb_roots(BNode *nodes, const MeshInfo *meshes, uint32_t mesh_count, unsigned long long *keys, uint32_t *small_list, uint32_t *small_count) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= mesh_count) return;
    BNode n = {};
    n.first = meshes[b].tri_first;
    n.count = meshes[b].tri_count;
    n.blas = b;
    n.axis = -1;
    if (n.count > 0 && n.count <= kSmall) {
        n.small = 1;
        small_list[atomicAdd(small_count, 1u)] = b;
    }
    nodes[b] = n;
    for (int k = 0; k < 3; ++k) {
        keys[6 * size_t(b) + k] = ~0ull;
        keys[6 * size_t(b) + 3 + k] = 0ull;
    }
}

// True when all 64 lanes of the wave are live and hold the same value (then one lane can speak
// for the wave after a reduction across the lanes).
PTB_D bool wave_uniform(bool live, uint32_t v) {
    const uint32_t v0 = __builtin_amdgcn_readfirstlane(v);
    return __ballot(live && v == v0) == ~0ull;
}
PTB_D uint32_t wave_min(uint32_t x) {
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t o = __shfl_xor(x, d, 64);
        x = o < x ? o : x;
    }
    return x;
}
PTB_D uint32_t wave_max(uint32_t x) {
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t o = __shfl_xor(x, d, 64);
        x = o > x ? o : x;
    }
    return x;
}
PTB_D unsigned long long wave_min64(unsigned long long x) {
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long o = __shfl_xor(x, d, 64);
        x = o < x ? o : x;
    }
    return x;
}
PTB_D unsigned long long wave_max64(unsigned long long x) {
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long o = __shfl_xor(x, d, 64);
        x = o > x ? o : x;
    }
    return x;
}

constexpr uint32_t kAggBlock = 1024, kAggWaves = kAggBlock / 64;

// True when every wave of the block lies inside the same node (s_v = per-wave node, or
// 0xFFFFFFFF for a wave that does not).  Contains a barrier: all threads must call it.
PTB_D bool block_uniform(uint32_t *s_v, bool wave_is_uniform, uint32_t v) {
    if ((threadIdx.x & 63) == 0) s_v[threadIdx.x >> 6] = wave_is_uniform ? v : 0xFFFFFFFFu;
    __syncthreads();
    bool same = s_v[0] != 0xFFFFFFFFu;
    for (uint32_t w = 1; w < kAggWaves; ++w) same = same && s_v[w] == s_v[0];
    return same;
}

// Box of the nodes [begin, end): union of the boxes of their entries, as (value, position) keys.
// A node's entries are contiguous, so most waves (and, for large nodes, whole blocks) lie inside
// one node: they reduce across lanes and through LDS first, and send one atomic per value where
// the lanes alone would send a thousand to the same address (those serialise at ~12 ns each).
__global__ __launch_bounds__(kAggBlock) void kb_bounds_keys(const BNode *nodes, uint32_t begin, uint32_t end, const uint32_t *node_of, const uint32_t *order,
                                                            Prims P, uint32_t n, unsigned long long *keys) {
    __shared__ uint32_t s_v[kAggWaves];
    __shared__ unsigned long long s_k[kAggWaves][6];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t v = i < n ? node_of[i] : 0xFFFFFFFFu;
    const bool live = i < n && v >= begin && v < end;
    const uint32_t prim = live ? order[i] : 0u;
    const bool uniform = wave_uniform(live, v);
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long k[6];
    for (int a = 0; a < 3; ++a) {
        k[a] = ((unsigned long long)enc_eq(P.lo[a][prim]) << 32) | i;
        k[3 + a] = ((unsigned long long)enc_eq(P.hi[a][prim]) << 32) | (0xFFFFFFFFu - i);
        if (uniform) {
            k[a] = wave_min64(k[a]);
            k[3 + a] = wave_max64(k[3 + a]);
        }
    }
    if (uniform && lane == 0)
        for (int q = 0; q < 6; ++q) s_k[wave][q] = k[q];
    const bool whole_block = block_uniform(s_v, uniform, v);
    if (whole_block) {
        if (threadIdx.x < 6) {
            unsigned long long r = s_k[0][threadIdx.x];
            for (uint32_t w = 1; w < kAggWaves; ++w) {
                const unsigned long long o = s_k[w][threadIdx.x];
                r = (threadIdx.x < 3) ? (o < r ? o : r) : (o > r ? o : r);
            }
            if (threadIdx.x < 3) atomicMin(&keys[6 * size_t(s_v[0]) + threadIdx.x], r);
            else atomicMax(&keys[6 * size_t(s_v[0]) + threadIdx.x], r);
        }
        return;
    }
    if (!live || (uniform && lane != 0)) return;
    for (int a = 0; a < 3; ++a) {
        atomicMin(&keys[6 * size_t(v) + a], k[a]);
        atomicMax(&keys[6 * size_t(v) + 3 + a], k[3 + a]);
    }
}
__global__ void kb_bounds_decode(BNode *nodes, uint32_t begin, uint32_t end, const unsigned long long *keys, const uint32_t *order, Prims P) {
    const uint32_t v = begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= end) return;
    BNode &n = nodes[v];
    if (n.count == 0) return;
    for (int a = 0; a < 3; ++a) {
        n.lo[a] = P.lo[a][order[uint32_t(keys[6 * size_t(v) + a])]];
        n.hi[a] = P.hi[a][order[0xFFFFFFFFu - uint32_t(keys[6 * size_t(v) + 3 + a])]];
    }
}

// ------------------------------------------------------------------ one pass over a level
PTB_D bool node_active(const BNode &n) { return n.count > kSmall; }  // smaller nodes belong to a single-wave subtree

__global__ void kb_level_init(BNode *nodes, uint32_t begin, uint32_t end, Bins *bins, uint32_t *slot_counter) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (begin + k >= end) return;
    BNode &n = nodes[begin + k];
    n.axis = -1;
    if (!node_active(n)) return;
    for (int a = 0; a < 3; ++a) {
        n.cb_lo[a] = 0xFFFFFFFFu;
        n.cb_hi[a] = 0u;
    }
    n.bin_slot = atomicAdd(slot_counter, 1u);  // at most (triangles / kSmall) nodes of a level are this large
    Bins &b = bins[n.bin_slot];
    for (int a = 0; a < 3; ++a)
        for (int j = 0; j < kBins; ++j) {
            b.count[a][j] = 0;
            for (int c = 0; c < 3; ++c) {
                b.lo[a][j][c] = 0xFFFFFFFFu;
                b.hi[a][j][c] = 0u;
            }
        }
}

__global__ __launch_bounds__(kAggBlock) void kb_centroid_bounds(BNode *nodes, uint32_t begin, uint32_t end, const uint32_t *node_of, const uint32_t *order,
                                                                Prims P, uint32_t n) {
    __shared__ uint32_t s_v[kAggWaves];
    __shared__ uint32_t s_e[kAggWaves][6];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t v = i < n ? node_of[i] : 0xFFFFFFFFu;
    const bool live = i < n && v >= begin && v < end && node_active(nodes[v]);
    const uint32_t prim = live ? order[i] : 0u;
    const bool uniform = wave_uniform(live, v);
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t e[6];
    for (int a = 0; a < 3; ++a) {
        e[a] = e[3 + a] = enc(P.centroid[a][prim]);
        if (uniform) {
            e[a] = wave_min(e[a]);
            e[3 + a] = wave_max(e[3 + a]);
        }
    }
    if (uniform && lane == 0)
        for (int q = 0; q < 6; ++q) s_e[wave][q] = e[q];
    const bool whole_block = block_uniform(s_v, uniform, v);
    if (whole_block) {
        if (threadIdx.x < 6) {
            uint32_t r = s_e[0][threadIdx.x];
            for (uint32_t w = 1; w < kAggWaves; ++w) {
                const uint32_t o = s_e[w][threadIdx.x];
                r = (threadIdx.x < 3) ? (o < r ? o : r) : (o > r ? o : r);
            }
            if (threadIdx.x < 3) atomicMin(&nodes[s_v[0]].cb_lo[threadIdx.x], r);
            else atomicMax(&nodes[s_v[0]].cb_hi[threadIdx.x - 3], r);
        }
        return;
    }
    if (!live || (uniform && lane != 0)) return;
    for (int a = 0; a < 3; ++a) {
        atomicMin(&nodes[v].cb_lo[a], e[a]);
        atomicMax(&nodes[v].cb_hi[a], e[3 + a]);
    }
}

// bvh.cpp:151-165: bin index of a centroid on every axis whose centroid extent is positive.
// A block (or, failing that, a wave) that lies inside one node bins into LDS first and sends
// each touched bin once.
__global__ __launch_bounds__(kAggBlock) void kb_bin(const BNode *nodes, uint32_t begin, uint32_t end, const uint32_t *node_of, const uint32_t *order, Prims P,
                                                    uint32_t n, Bins *bins) {
    __shared__ uint32_t s_v[kAggWaves];
    __shared__ Bins s_bins[kAggWaves];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t v = i < n ? node_of[i] : 0xFFFFFFFFu;
    const bool live = i < n && v >= begin && v < end && node_active(nodes[v]);
    const bool uniform = wave_uniform(live, v);
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr uint32_t kLoWords = 3 * kBins * 3, kWords = sizeof(Bins) / 4;  // lo[...] | hi[...] | count[...]
    uint32_t *wave_words = reinterpret_cast<uint32_t *>(&s_bins[wave]);
    for (uint32_t w = lane; w < kWords; w += 64) wave_words[w] = (w < kLoWords) ? 0xFFFFFFFFu : 0u;
    const bool whole_block = block_uniform(s_v, uniform, v);  // (barrier: the LDS bins are initialised)
    if (live) {
        const BNode &nd = nodes[v];
        const uint32_t prim = order[i];
        Bins &dst = whole_block ? s_bins[0] : (uniform ? s_bins[wave] : bins[nd.bin_slot]);
        for (int a = 0; a < 3; ++a) {
            const float cmin = dec(nd.cb_lo[a]), cmax = dec(nd.cb_hi[a]);
            const float extent = cmax - cmin;
            if (extent <= 0.0f) continue;
            const float inv_width = 1.0f / fmax_ref(extent, 1e-30f);
            const float scaled = (P.centroid[a][prim] - cmin) * inv_width;
            int j = int(scaled * kBins);
            j = j < 0 ? 0 : (j > kBins - 1 ? kBins - 1 : j);
            atomicAdd(&dst.count[a][j], 1u);
            for (int c = 0; c < 3; ++c) {
                atomicMin(&dst.lo[a][j][c], enc(P.lo[c][prim]));
                atomicMax(&dst.hi[a][j][c], enc(P.hi[c][prim]));
            }
        }
    }
    __syncthreads();
    const uint32_t *src = nullptr;
    uint32_t w0 = 0, step = 0;
    if (whole_block) {
        src = reinterpret_cast<const uint32_t *>(&s_bins[0]);
        w0 = threadIdx.x;
        step = kAggBlock;
    } else if (uniform) {
        src = wave_words;
        w0 = lane;
        step = 64;
    }
    if (!src) return;
    uint32_t *out_words = reinterpret_cast<uint32_t *>(&bins[nodes[v].bin_slot]);
    for (uint32_t w = w0; w < kWords; w += step) {
        const uint32_t x = src[w];
        if (w < kLoWords) {
            if (x != 0xFFFFFFFFu) atomicMin(&out_words[w], x);
        } else if (w < 2 * kLoWords) {
            if (x != 0u) atomicMax(&out_words[w], x);
        } else if (x != 0u) {
            atomicAdd(&out_words[w], x);
        }
    }
}

struct RBox {
    float lo[3], hi[3];
    PTB_D void reset() {
        for (int a = 0; a < 3; ++a) {
            lo[a] = FLT_MAX;
            hi[a] = -FLT_MAX;
        }
    }
    PTB_D void grow(const float *l, const float *h) {
        for (int a = 0; a < 3; ++a) {
            lo[a] = fmin_ref(lo[a], l[a]);
            hi[a] = fmax_ref(hi[a], h[a]);
        }
    }
    PTB_D float area() const {  // SurfaceArea, math/aabb.h
        const float dx = fmax_ref(hi[0] - lo[0], 0.0f), dy = fmax_ref(hi[1] - lo[1], 0.0f), dz = fmax_ref(hi[2] - lo[2], 0.0f);
        return 2.0f * (dx * dy + dy * dz + dz * dx);
    }
};

// FindSplitPlane's sweep and Subdivide's cost test (bvh.cpp:72-77,167-198) for one node, given its
// bins and centroid bounds (enc() coded) and its own box.  False: the node stays a leaf.
PTB_D bool find_split(const Bins &b, const uint32_t *cb_lo, const uint32_t *cb_hi, uint32_t count, const float *node_lo, const float *node_hi, int &axis,
                      float &pos) {
    float best_cost = FLT_MAX, best_pos = 0.0f;
    int best_axis = -1;
    bool empty = false;
    for (int a = 0; a < 3; ++a) empty = empty || (dec(cb_lo[a]) > dec(cb_hi[a]));
    if (!empty) {
        for (int a = 0; a < 3; ++a) {
            const float cmin = dec(cb_lo[a]), cmax = dec(cb_hi[a]);
            const float extent = cmax - cmin;
            if (extent <= 0.0f) continue;
            float left_area[kBins - 1], right_area[kBins - 1];
            uint32_t left_count[kBins - 1], right_count[kBins - 1];
            RBox lbox, rbox;
            lbox.reset();
            rbox.reset();
            uint32_t lsum = 0, rsum = 0;
            for (int i = 0; i < kBins - 1; ++i) {
                float l[3], h[3];
                lsum += b.count[a][i];
                left_count[i] = lsum;
                for (int c = 0; c < 3; ++c) {  // an untouched bin is the empty box
                    l[c] = b.count[a][i] ? dec(b.lo[a][i][c]) : FLT_MAX;
                    h[c] = b.count[a][i] ? dec(b.hi[a][i][c]) : -FLT_MAX;
                }
                lbox.grow(l, h);
                left_area[i] = lbox.area();
                const int rb = kBins - 1 - i;
                rsum += b.count[a][rb];
                right_count[rb - 1] = rsum;
                for (int c = 0; c < 3; ++c) {
                    l[c] = b.count[a][rb] ? dec(b.lo[a][rb][c]) : FLT_MAX;
                    h[c] = b.count[a][rb] ? dec(b.hi[a][rb][c]) : -FLT_MAX;
                }
                rbox.grow(l, h);
                right_area[rb - 1] = rbox.area();
            }
            for (int i = 0; i < kBins - 1; ++i) {
                const float cost = left_count[i] * left_area[i] + right_count[i] * right_area[i];
                if (cost < best_cost) {
                    best_cost = cost;
                    best_axis = a;
                    best_pos = cmin + extent * (float(i + 1) / kBins);
                }
            }
        }
    }
    RBox nb;
    for (int a = 0; a < 3; ++a) {
        nb.lo[a] = node_lo[a];
        nb.hi[a] = node_hi[a];
    }
    const float parent_cost = count * nb.area();
    if (best_axis < 0 || best_cost >= parent_cost) return false;
    axis = best_axis;
    pos = best_pos;
    return true;
}

__global__ void kb_split(BNode *nodes, uint32_t begin, uint32_t end, const Bins *bins) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (begin + k >= end) return;
    BNode &n = nodes[begin + k];
    if (!node_active(n)) return;
    int axis;
    float pos;
    if (!find_split(bins[n.bin_slot], n.cb_lo, n.cb_hi, n.count, n.lo, n.hi, axis, pos)) return;
    n.axis = axis;
    n.pos = pos;
}

__global__ void kb_flags(const BNode *nodes, uint32_t begin, uint32_t end, const uint32_t *node_of, const uint32_t *order, Prims P, uint32_t n, uint32_t *flag) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t v = node_of[i];
    uint32_t f = 0;
    if (v >= begin && v < end && nodes[v].axis >= 0) f = (P.centroid[nodes[v].axis][order[i]] < nodes[v].pos) ? 0u : 1u;
    flag[i] = f;
}

// ---- exclusive prefix sum of n flags into out[0..n] (out[n] = total)
__global__ __launch_bounds__(kScanBlock) void kb_scan_tiles(const uint32_t *in, uint32_t n, uint32_t *out, uint32_t *tile_sums) {
    __shared__ uint32_t wave_sum[kScanBlock / 64];
    const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
    uint32_t v[kScanItems], sum = 0;
    for (uint32_t k = 0; k < kScanItems; ++k) {
        v[k] = (base + k < n) ? in[base + k] : 0u;
        sum += v[k];
    }
    uint32_t incl = sum;  // inclusive scan of the per-thread sums across the wave
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d, 64);
        if (int(threadIdx.x & 63) >= d) incl += up;
    }
    if ((threadIdx.x & 63) == 63) wave_sum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t offset = 0;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) offset += wave_sum[w];
    uint32_t run = offset + incl - sum;
    for (uint32_t k = 0; k < kScanItems; ++k) {
        if (base + k < n) out[base + k] = run;
        run += v[k];
    }
    if (threadIdx.x == kScanBlock - 1) tile_sums[blockIdx.x] = run;
}
__global__ __launch_bounds__(1024) void kb_scan_sums(uint32_t *tile_sums, uint32_t tiles, uint32_t *total) {
    __shared__ uint32_t wave_sum[16];
    uint32_t v[8], sum = 0;
    const uint32_t base = threadIdx.x * 8;
    for (uint32_t k = 0; k < 8; ++k) {
        v[k] = (base + k < tiles) ? tile_sums[base + k] : 0u;
        sum += v[k];
    }
    uint32_t incl = sum;
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d, 64);
        if (int(threadIdx.x & 63) >= d) incl += up;
    }
    if ((threadIdx.x & 63) == 63) wave_sum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t offset = 0;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) offset += wave_sum[w];
    uint32_t run = offset + incl - sum;
    for (uint32_t k = 0; k < 8; ++k) {
        if (base + k < tiles) tile_sums[base + k] = run;
        run += v[k];
    }
    if (threadIdx.x == 1023) *total = run;
}
__global__ void kb_scan_add(uint32_t *out, uint32_t n, const uint32_t *tile_sums, const uint32_t *total) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] += tile_sums[i / kScanTile];
    if (i == 0) out[n] = *total;
}

// Tables of the partition: pfront[first + r] = position of the r-th right-goer from the front,
// qback[first + l] = position of the l-th left-goer from the back.
__global__ void kb_rank_tables(const BNode *nodes, uint32_t begin, uint32_t end, const uint32_t *node_of, const uint32_t *flag, const uint32_t *scan, uint32_t n,
                               uint32_t *pfront, uint32_t *qback) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t v = node_of[i];
    if (v < begin || v >= end || nodes[v].axis < 0) return;
    const uint32_t first = nodes[v].first, count = nodes[v].count;
    const uint32_t base = scan[first];
    const uint32_t rights_before = scan[i] - base;
    if (flag[i]) {
        pfront[first + rights_before] = i;
    } else {
        const uint32_t n_left = count - (scan[first + count] - base);
        const uint32_t lefts_before = (i - first) - rights_before;
        qback[first + (n_left - 1 - lefts_before)] = i;
    }
}

// The two-cursor partition (bvh.cpp:80-86) examines entries alternately from the front (until
// it meets a right-goer, which it throws to the back cursor) and from the back (until it meets
// a left-goer, which lands in the slot just vacated).  With p_r / q_l the tables above:
//   K  = number of r with p_r < q_r (front and back cursors have not crossed);
//   K' = K + 1 when one more right-goer is met from the front before the cursors cross;
//   a left-goer met from the back (back rank l < K) moves to p_l, every other one stays;
//   a right-goer met from the front (front rank r < K') moves to q_{r-1} - 1 (the last slot for
//   r = 0), every other one moves one slot towards the front.
// This kernel finds K and K' per node, allocates the children and names the entries' new owner.
__global__ void kb_pairs(BNode *nodes, uint32_t begin, uint32_t end, const uint32_t *scan, const uint32_t *pfront, const uint32_t *qback, uint32_t *node_count,
                         unsigned long long *keys, uint32_t *small_list, uint32_t *small_count) {
    const uint32_t v = begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= end) return;
    BNode &n = nodes[v];
    if (n.axis < 0) return;
    const uint32_t first = n.first, count = n.count;
    const uint32_t n_right = scan[first + count] - scan[first], n_left = count - n_right;
    const uint32_t m = n_right < n_left ? n_right : n_left;
    uint32_t lo = 0, hi = m;  // smallest r in [0, m] with !(p_r < q_r); the predicate is monotone
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (pfront[first + mid] < qback[first + mid]) lo = mid + 1; else hi = mid;
    }
    const uint32_t K = lo;
    uint32_t K2 = K;
    if (K < n_right && pfront[first + K] < (K > 0 ? qback[first + K - 1] : first + count)) K2 = K + 1;
    n.n_right = n_right;
    n.pairs = K;
    n.front_rights = K2;
    if (n_left == 0 || n_left == count) return;  // bvh.cpp:88-89: the list stays permuted, the node stays a leaf
    const uint32_t left = atomicAdd(node_count, 2u);
    BNode c = {};
    c.blas = n.blas;
    c.axis = -1;
    c.first = first;
    c.count = n_left;
    c.small = n_left <= kSmall;
    nodes[left] = c;
    if (c.small) small_list[atomicAdd(small_count, 1u)] = left;
    c.first = first + n_left;
    c.count = n_right;
    c.small = n_right <= kSmall;
    nodes[left + 1] = c;
    if (c.small) small_list[atomicAdd(small_count, 1u)] = left + 1;
    for (uint32_t s = 0; s < 2; ++s)
        for (int k = 0; k < 3; ++k) {
            keys[6 * size_t(left + s) + k] = ~0ull;
            keys[6 * size_t(left + s) + 3 + k] = 0ull;
        }
    n.left = left;
}

__global__ void kb_scatter(const BNode *nodes, uint32_t begin, uint32_t end, const uint32_t *node_of, const uint32_t *order, const uint32_t *flag, const uint32_t *scan,
                           const uint32_t *pfront, const uint32_t *qback, uint32_t n, uint32_t *order_out, uint32_t *node_of_out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t v = node_of[i];
    if (v < begin || v >= end || nodes[v].axis < 0) {
        order_out[i] = order[i];
        node_of_out[i] = v;
        return;
    }
    const BNode &nd = nodes[v];
    const uint32_t first = nd.first, count = nd.count, last = first + count - 1;
    const uint32_t rights_before = scan[i] - scan[first];
    const uint32_t n_left = count - nd.n_right;
    uint32_t dest;
    if (flag[i]) {
        const uint32_t r = rights_before;
        dest = (r < nd.front_rights) ? (r == 0 ? last : qback[first + r - 1] - 1) : i - 1;
    } else {
        const uint32_t l = n_left - 1 - ((i - first) - rights_before);
        dest = (l < nd.pairs) ? pfront[first + l] : i;
    }
    order_out[dest] = order[i];
    node_of_out[dest] = nd.left ? (dest < first + n_left ? nd.left : nd.left + 1) : v;
}

// ------------------------------------------------------------------ small subtrees, one wave each
struct SmallLds {
    uint32_t prim[kSmall];                                  // triangle of each slot
    float c[3][kSmall], lo[3][kSmall], hi[3][kSmall];       // centroid and box of each slot
    uint16_t perm[kSmall];                                  // entry (position inside the subtree) -> slot
    uint16_t perm_next[kSmall], pfront[kSmall], qback[kSmall];  // partition scratch (see kb_pairs)
    uint16_t nfirst[2 * kSmall], ncount[2 * kSmall], nleft[2 * kSmall];  // the subtree's nodes, numbered as the reference
    float nlo[2 * kSmall][3], nhi[2 * kSmall][3];                        // numbers them: children allocated on subdivision
    uint16_t stack[2 * kSmall];
    Bins bins;
    unsigned long long keys[12];
};

// Subdivide (bvh.cpp:67-113) over the whole subtree of one node, depth first, by one wave.
// Reductions (centroid bounds, bins, child boxes) run across the lanes; the 21 candidate planes
// are costed by 21 lanes (each unions its bins in the sweep's order) and the cheapest with the
// lowest number wins, which is what "first strictly smaller" picks; the partition uses the same
// closed form as the level passes, with ballots for the ranks.
__global__ __launch_bounds__(64) void kb_small_subtrees(BNode *nodes, const uint32_t *small_list, uint32_t small_count, uint32_t *order, Prims P,
                                                        uint32_t *node_count) {
    __shared__ SmallLds L;
    if (blockIdx.x >= small_count) return;
    const uint32_t lane = threadIdx.x;
    const uint32_t root = small_list[blockIdx.x];
    const uint32_t first = nodes[root].first, cnt = nodes[root].count, blas = nodes[root].blas;
    for (uint32_t e = lane; e < cnt; e += 64) {
        const uint32_t prim = order[first + e];
        L.prim[e] = prim;
        L.perm[e] = uint16_t(e);
        for (int a = 0; a < 3; ++a) {
            L.c[a][e] = P.centroid[a][prim];
            L.lo[a][e] = P.lo[a][prim];
            L.hi[a][e] = P.hi[a][prim];
        }
    }
    if (lane == 0) {
        L.nfirst[0] = 0;
        L.ncount[0] = uint16_t(cnt);
        L.nleft[0] = 0;
        for (int a = 0; a < 3; ++a) {
            L.nlo[0][a] = nodes[root].lo[a];
            L.nhi[0][a] = nodes[root].hi[a];
        }
        L.stack[0] = 0;
    }
    __syncthreads();
    uint32_t sp = 1, next_local = 1, splits = 0;  // the same in every lane
    uint32_t *bin_words = reinterpret_cast<uint32_t *>(&L.bins);
    constexpr uint32_t kLoWords = 3 * kBins * 3, kWords = sizeof(Bins) / 4;
    while (sp > 0) {
        const uint32_t l = L.stack[--sp];
        const uint32_t f = L.nfirst[l], c = L.ncount[l];
        if (c <= 2) continue;  // bvh.cpp:69
        uint32_t cb_lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, cb_hi[3] = {0u, 0u, 0u};
        for (uint32_t e = f + lane; e < f + c; e += 64) {
            const uint32_t slot = L.perm[e];
            for (int a = 0; a < 3; ++a) {
                const uint32_t x = enc(L.c[a][slot]);
                cb_lo[a] = x < cb_lo[a] ? x : cb_lo[a];
                cb_hi[a] = x > cb_hi[a] ? x : cb_hi[a];
            }
        }
        for (int a = 0; a < 3; ++a) {
            cb_lo[a] = wave_min(cb_lo[a]);
            cb_hi[a] = wave_max(cb_hi[a]);
        }
        for (uint32_t w = lane; w < kWords; w += 64) bin_words[w] = (w < kLoWords) ? 0xFFFFFFFFu : 0u;
        __syncthreads();
        for (uint32_t e = f + lane; e < f + c; e += 64) {
            const uint32_t slot = L.perm[e];
            for (int a = 0; a < 3; ++a) {
                const float cmin = dec(cb_lo[a]), cmax = dec(cb_hi[a]);
                const float extent = cmax - cmin;
                if (extent <= 0.0f) continue;
                const float inv_width = 1.0f / fmax_ref(extent, 1e-30f);
                const float scaled = (L.c[a][slot] - cmin) * inv_width;
                int j = int(scaled * kBins);
                j = j < 0 ? 0 : (j > kBins - 1 ? kBins - 1 : j);
                atomicAdd(&L.bins.count[a][j], 1u);
                for (int k = 0; k < 3; ++k) {
                    atomicMin(&L.bins.lo[a][j][k], enc(L.lo[k][slot]));
                    atomicMax(&L.bins.hi[a][j][k], enc(L.hi[k][slot]));
                }
            }
        }
        __syncthreads();
        // ---- plane choice (bvh.cpp:167-198, :72-77)
        float cost = FLT_MAX, plane = 0.0f;
        bool candidate = false;
        if (lane < 3 * (kBins - 1)) {
            const int a = int(lane) / (kBins - 1), i = int(lane) % (kBins - 1);
            const float cmin = dec(cb_lo[a]), cmax = dec(cb_hi[a]);
            const float extent = cmax - cmin;
            const bool empty = dec(cb_lo[0]) > dec(cb_hi[0]) || dec(cb_lo[1]) > dec(cb_hi[1]) || dec(cb_lo[2]) > dec(cb_hi[2]);
            if (!empty && !(extent <= 0.0f)) {
                RBox lbox, rbox;
                lbox.reset();
                rbox.reset();
                uint32_t lsum = 0, rsum = 0;
                float lo3[3], hi3[3];
                for (int k = 0; k <= i; ++k) {
                    const uint32_t cnt_k = L.bins.count[a][k];
                    lsum += cnt_k;
                    for (int q = 0; q < 3; ++q) {
                        lo3[q] = cnt_k ? dec(L.bins.lo[a][k][q]) : FLT_MAX;
                        hi3[q] = cnt_k ? dec(L.bins.hi[a][k][q]) : -FLT_MAX;
                    }
                    lbox.grow(lo3, hi3);
                }
                for (int k = kBins - 1; k > i; --k) {
                    const uint32_t cnt_k = L.bins.count[a][k];
                    rsum += cnt_k;
                    for (int q = 0; q < 3; ++q) {
                        lo3[q] = cnt_k ? dec(L.bins.lo[a][k][q]) : FLT_MAX;
                        hi3[q] = cnt_k ? dec(L.bins.hi[a][k][q]) : -FLT_MAX;
                    }
                    rbox.grow(lo3, hi3);
                }
                cost = lsum * lbox.area() + rsum * rbox.area();
                candidate = cost < FLT_MAX;
                plane = cmin + extent * (float(i + 1) / kBins);
            }
        }
        float best_cost = candidate ? cost : FLT_MAX;
        int best_lane = candidate ? int(lane) : 64;
        for (int d = 32; d >= 1; d >>= 1) {
            const float oc = __shfl_xor(best_cost, d, 64);
            const int ol = __shfl_xor(best_lane, d, 64);
            if (ol < 64 && (best_lane == 64 || oc < best_cost || (oc == best_cost && ol < best_lane))) {
                best_cost = oc;
                best_lane = ol;
            }
        }
        RBox nb;
        for (int q = 0; q < 3; ++q) {
            nb.lo[q] = L.nlo[l][q];
            nb.hi[q] = L.nhi[l][q];
        }
        const float parent_cost = c * nb.area();
        if (best_lane == 64 || best_cost >= parent_cost) continue;
        const int axis = best_lane / (kBins - 1);
        const float pos = __shfl(plane, best_lane, 64);

        // ---- partition (bvh.cpp:80-86) in its closed form
        unsigned long long right_mask[kSmall / 64], left_mask[kSmall / 64];
        uint32_t n_right = 0;
        for (uint32_t k = 0; k < kSmall / 64; ++k) {
            const uint32_t e = f + lane + 64 * k;
            const bool valid = e < f + c;
            const bool right = valid && !(L.c[axis][L.perm[valid ? e : f]] < pos);
            right_mask[k] = __ballot(right);
            left_mask[k] = __ballot(valid && !right);
            n_right += __popcll(right_mask[k]);
        }
        const uint32_t n_left = c - n_right;
        const unsigned long long below = (1ull << lane) - 1ull;
        uint32_t rights_seen = 0, lefts_seen = 0;
        for (uint32_t k = 0; k < kSmall / 64; ++k) {
            const uint32_t e = f + lane + 64 * k;
            if ((right_mask[k] >> lane) & 1ull) L.pfront[rights_seen + __popcll(right_mask[k] & below)] = uint16_t(e);
            if ((left_mask[k] >> lane) & 1ull) L.qback[n_left - 1 - (lefts_seen + __popcll(left_mask[k] & below))] = uint16_t(e);
            rights_seen += __popcll(right_mask[k]);
            lefts_seen += __popcll(left_mask[k]);
        }
        __syncthreads();
        const uint32_t m = n_right < n_left ? n_right : n_left;
        uint32_t K = 0;
        for (uint32_t k = 0; k < kSmall / 64; ++k) {
            const uint32_t r = lane + 64 * k;
            K += __popcll(__ballot(r < m && L.pfront[r < m ? r : 0] < L.qback[r < m ? r : 0]));
        }
        uint32_t K2 = K;
        if (K < n_right && uint32_t(L.pfront[K]) < (K > 0 ? uint32_t(L.qback[K - 1]) : f + c)) K2 = K + 1;
        rights_seen = lefts_seen = 0;
        for (uint32_t k = 0; k < kSmall / 64; ++k) {
            const uint32_t e = f + lane + 64 * k;
            const bool is_right = (right_mask[k] >> lane) & 1ull, is_left = (left_mask[k] >> lane) & 1ull;
            if (is_right) {
                const uint32_t r = rights_seen + __popcll(right_mask[k] & below);
                const uint32_t dest = (r < K2) ? (r == 0 ? f + c - 1 : uint32_t(L.qback[r - 1]) - 1) : e - 1;
                L.perm_next[dest] = L.perm[e];
            } else if (is_left) {
                const uint32_t lb = n_left - 1 - (lefts_seen + __popcll(left_mask[k] & below));
                const uint32_t dest = (lb < K) ? uint32_t(L.pfront[lb]) : e;
                L.perm_next[dest] = L.perm[e];
            }
            rights_seen += __popcll(right_mask[k]);
            lefts_seen += __popcll(left_mask[k]);
        }
        __syncthreads();
        for (uint32_t e = f + lane; e < f + c; e += 64) L.perm[e] = L.perm_next[e];
        __syncthreads();
        const uint32_t lc = n_left;
        if (lc == 0 || lc == c) continue;  // bvh.cpp:88-89
        const uint32_t ll = next_local;
        next_local += 2;
        ++splits;
        if (lane < 12) L.keys[lane] = (lane % 6 < 3) ? ~0ull : 0ull;
        __syncthreads();
        for (uint32_t e = f + lane; e < f + c; e += 64) {  // UpdateNodeBounds of both children (bvh.cpp:53-65)
            const uint32_t slot = L.perm[e];
            const uint32_t child = (e < f + lc) ? 0u : 1u;
            for (int a = 0; a < 3; ++a) {
                atomicMin(&L.keys[child * 6 + a], ((unsigned long long)enc_eq(L.lo[a][slot]) << 32) | e);
                atomicMax(&L.keys[child * 6 + 3 + a], ((unsigned long long)enc_eq(L.hi[a][slot]) << 32) | (0xFFFFFFFFu - e));
            }
        }
        __syncthreads();
        if (lane < 12) {
            const uint32_t child = lane / 6, k = lane % 6;
            if (k < 3) L.nlo[ll + child][k] = L.lo[k][L.perm[uint32_t(L.keys[lane])]];
            else L.nhi[ll + child][k - 3] = L.hi[k - 3][L.perm[0xFFFFFFFFu - uint32_t(L.keys[lane])]];
        }
        if (lane == 0) {
            L.nfirst[ll] = uint16_t(f);
            L.ncount[ll] = uint16_t(lc);
            L.nleft[ll] = 0;
            L.nfirst[ll + 1] = uint16_t(f + lc);
            L.ncount[ll + 1] = uint16_t(c - lc);
            L.nleft[ll + 1] = 0;
            L.nleft[l] = uint16_t(ll);
            L.stack[sp] = uint16_t(ll + 1);  // the left child is subdivided first (bvh.cpp:111-112)
            L.stack[sp + 1] = uint16_t(ll);
        }
        sp += 2;
        __syncthreads();
    }
    uint32_t base = 0;
    if (lane == 0 && next_local > 1) base = atomicAdd(node_count, next_local - 1);
    base = __builtin_amdgcn_readfirstlane(base);
    for (uint32_t e = lane; e < cnt; e += 64) order[first + e] = L.prim[L.perm[e]];
    for (uint32_t l = 1 + lane; l < next_local; l += 64) {
        BNode o = {};
        o.first = first + L.nfirst[l];
        o.count = L.ncount[l];
        o.left = L.nleft[l] ? base + L.nleft[l] - 1 : 0u;
        o.blas = blas;
        for (int a = 0; a < 3; ++a) {
            o.lo[a] = L.nlo[l][a];
            o.hi[a] = L.nhi[l][a];
        }
        o.axis = -1;
        o.owner = root;
        o.local = l;
        nodes[base + l - 1] = o;
    }
    if (lane == 0) {
        nodes[root].left = L.nleft[0] ? base + L.nleft[0] - 1 : 0u;
        nodes[root].internal = splits;
    }
}

// Final index of the nodes inside single-wave subtrees: the subtree's k-th subdivided node is the
// mesh's (rank(root) + k)-th, so local number l (>= 1) becomes 2 * rank(root) + l.
__global__ void kb_small_indices(BNode *nodes, uint32_t begin, uint32_t end) {
    const uint32_t v = begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= end) return;
    BNode &n = nodes[v];
    n.index = 2u * nodes[n.owner].rank + n.local;
}

// ------------------------------------------------------------------ renumbering and output
__global__ void kb_subtree_sizes(BNode *nodes, uint32_t begin, uint32_t end) {
    const uint32_t v = begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= end) return;
    BNode &n = nodes[v];
    if (n.small) return;  // counted by the wave that built the subtree
    n.internal = n.left ? 1u + nodes[n.left].internal + nodes[n.left + 1].internal : 0u;
}
__global__ void kb_preorder(BNode *nodes, uint32_t begin, uint32_t end) {
    const uint32_t v = begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= end) return;
    const BNode &n = nodes[v];
    if (!n.left || n.small) return;
    BNode &l = nodes[n.left], &r = nodes[n.left + 1];
    l.index = 1u + 2u * n.rank;
    r.index = 2u + 2u * n.rank;
    l.rank = n.rank + 1u;
    r.rank = n.rank + 1u + l.internal;
}
__global__ void kb_emit_nodes(const BNode *nodes, uint32_t node_count, const MeshInfo *meshes, pt_bvh_node *out) {
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= node_count) return;
    const BNode &n = nodes[v];
    const MeshInfo &m = meshes[n.blas];
    if (m.tri_count == 0) return;
    pt_bvh_node o;
    for (int a = 0; a < 3; ++a) {
        o.min[a] = n.lo[a];
        o.max[a] = n.hi[a];
    }
    if (n.left) {  // child indices are offset into the flattened array (bvh.cpp:245-252)
        o.first_primitive = m.node_offset + nodes[n.left].index;
        o.primitive_count = 0;
    } else {
        o.first_primitive = n.first - m.tri_first;
        o.primitive_count = n.count;
    }
    out[m.node_offset + n.index] = o;
}
__global__ void kb_emit_tris(const uint32_t *order, Prims P, const MeshInfo *meshes, uint32_t n, uint32_t *tri_indices) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t prim = order[i];
    tri_indices[i] = prim - meshes[P.blas[prim]].tri_first;
}

// ------------------------------------------------------------------ TLAS
// World box of an instance: the 8 transformed corners of its mesh's root box (math/aabb.h:50-63).
__global__ void kb_instance_bounds(const pt_geometry_instance *geoms, uint32_t n, const BNode *nodes, pt_tlas_node *tlas, float *abox, uint32_t *active) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const BNode &root = nodes[geoms[i].blas_index];
    const float *T = geoms[i].transform;  // column-major
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int c = 0; c < 8; ++c) {
        const float p[4] = {(c & 1) ? root.hi[0] : root.lo[0], (c & 2) ? root.hi[1] : root.lo[1], (c & 4) ? root.hi[2] : root.lo[2], 1.0f};
        for (int r = 0; r < 3; ++r) {  // mat4 * vec4 in glm's order: (c0*x + c1*y) + (c2*z + c3*w)
            const float a = T[0 * 4 + r] * p[0] + T[1 * 4 + r] * p[1];
            const float b = T[2 * 4 + r] * p[2] + T[3 * 4 + r] * p[3];
            const float w = a + b;
            lo[r] = fmin_ref(lo[r], w);
            hi[r] = fmax_ref(hi[r], w);
        }
    }
    pt_tlas_node t;
    for (int a = 0; a < 3; ++a) {
        t.min[a] = lo[a];
        t.max[a] = hi[a];
        abox[a * n + i] = lo[a];
        abox[(3 + a) * n + i] = hi[a];
    }
    t.instance = i;
    t.left_right = 0;
    tlas[1 + i] = t;  // slot 0 is the root's copy, leaves sit at 1..n in instance order (bvh.cpp:266-281)
    active[i] = 1 + i;
}

// bvh.cpp:283-331.  abox = boxes of the live clusters, structure of arrays with stride n.
// status[0] = 0 on success, 1 when the iteration guard tripped.
// Every "which cluster gives the smallest merged area with cluster x?" is a scan over the live
// clusters spread over the workgroup's lanes; the searches depend on one another (each
// merge or step of the walk decides the next query), so they run one after the other on one CU.
// (Caching every slot's answer across merges was tried: exact, but the per-merge upkeep cost more
// than the scans it saved.)
// Up to kTlasLdsInstances instances the boxes are copied into LDS first and a smaller workgroup
// runs the chain: a search then costs LDS latency and a 4-wave barrier instead of a round trip to
// L2 and a 16-wave barrier.
constexpr uint32_t kTlasThreads = 1024;
#ifndef PT_TLAS_SMALL_THREADS
#define PT_TLAS_SMALL_THREADS 256
#endif
constexpr uint32_t kTlasSmallThreads = PT_TLAS_SMALL_THREADS;
constexpr uint32_t kTlasLdsInstances = 2048;

template <uint32_t kThreads, bool kBoxesInLds>
__global__ __launch_bounds__(kThreads) void kb_tlas_pairing(pt_tlas_node *tlas, float *abox_global, uint32_t *active_global, uint32_t n, uint32_t *status) {
    __shared__ float s_area[2][kThreads / 64];
    __shared__ int s_index[2][kThreads / 64];
    __shared__ float s_box[kBoxesInLds ? 6 * kTlasLdsInstances : 1];
    __shared__ uint32_t s_active[kBoxesInLds ? kTlasLdsInstances : 1];
    const uint32_t tid = threadIdx.x;
    if (kBoxesInLds) {
        for (uint32_t i = tid; i < 6 * n; i += kThreads) s_box[i] = abox_global[i];
        for (uint32_t i = tid; i < n; i += kThreads) s_active[i] = active_global[i];
        __syncthreads();
    }
    // the tables the chain works on: boxes (structure of arrays, stride n) and the node of each live cluster
    auto abox = [&](uint32_t i) -> float & { return kBoxesInLds ? s_box[i] : abox_global[i]; };
    auto active = [&](uint32_t i) -> uint32_t & { return kBoxesInLds ? s_active[i] : active_global[i]; };
    int live = int(n);
    uint32_t next_node = 1 + n;
    uint32_t parity = 0;

    // Cluster of smallest merged area with the box in slot `who`; the lowest slot wins ties
    // (strict < in a forward scan, bvh.cpp:286-297).  Every thread returns the same value.
    // `who` may lie one past the live range right after a merge whose first cluster sat in the
    // last slot; the reference reads that stale slot too.
    auto best_partner = [&](int who) -> int {
        float mine_lo[3], mine_hi[3];
        for (int a = 0; a < 3; ++a) {
            mine_lo[a] = abox(a * n + who);
            mine_hi[a] = abox((3 + a) * n + who);
        }
        float best_area = FLT_MAX;
        int best = -1;
        for (int j = int(tid); j < live; j += int(kThreads)) {
            if (j == who) continue;
            RBox m;
            for (int a = 0; a < 3; ++a) {
                m.lo[a] = fmin_ref(mine_lo[a], abox(a * n + j));
                m.hi[a] = fmax_ref(mine_hi[a], abox((3 + a) * n + j));
            }
            const float area = m.area();
            if (area < best_area) {
                best_area = area;
                best = j;
            }
        }
        // a candidate beats another when its area is smaller, or equal with a lower slot;
        // "no candidate" (-1) never beats a candidate
        for (int d = 32; d >= 1; d >>= 1) {
            const float oa = __shfl_xor(best_area, d, 64);
            const int ob = __shfl_xor(best, d, 64);
            if (ob >= 0 && (best < 0 || oa < best_area || (oa == best_area && ob < best))) {
                best_area = oa;
                best = ob;
            }
        }
        if ((tid & 63) == 0) {
            s_area[parity][tid >> 6] = best_area;
            s_index[parity][tid >> 6] = best;
        }
        __syncthreads();
        float ba = s_area[parity][0];
        int bi = s_index[parity][0];
        for (uint32_t w = 1; w < kThreads / 64; ++w) {
            const float oa = s_area[parity][w];
            const int ob = s_index[parity][w];
            if (ob >= 0 && (bi < 0 || oa < ba || (oa == ba && ob < bi))) {
                ba = oa;
                bi = ob;
            }
        }
        parity ^= 1u;  // the next search writes the other buffer while slow waves still read this one
        return bi;
    };

    int a = 0;
    int b = (live > 1) ? best_partner(a) : -1;
    // every step either merges (live shrinks) or walks the chain to a pair of smaller merged
    // area; the guard only stops a walk that would never end on malformed (NaN) boxes
    unsigned long long guard = 64ull * n + 1024ull;
    bool failed = false;
    while (live > 1) {
        if (guard-- == 0 || b < 0) {
            failed = true;
            break;
        }
        const int c = best_partner(b);
        if (a == c) {
            __syncthreads();  // every lane has finished reading the boxes that are about to change
            if (tid == 0) {
                pt_tlas_node t;
                for (int k = 0; k < 3; ++k) {
                    t.min[k] = fmin_ref(abox(k * n + a), abox(k * n + b));
                    t.max[k] = fmax_ref(abox((3 + k) * n + a), abox((3 + k) * n + b));
                }
                t.instance = 0;
                t.left_right = active(a) | (active(b) << 16);
                tlas[next_node] = t;
                active(a) = next_node;
                active(b) = active(live - 1);
                for (int k = 0; k < 3; ++k) {
                    abox(k * n + a) = t.min[k];
                    abox((3 + k) * n + a) = t.max[k];
                }
                for (int k = 0; k < 6; ++k) abox(k * n + b) = abox(k * n + (live - 1));
            }
            ++next_node;
            --live;
            __syncthreads();
            b = (live > 1) ? best_partner(a) : -1;
        } else {
            a = b;
            b = c;
        }
    }
    if (tid == 0) {
        tlas[0] = tlas[active(a)];
        status[0] = failed ? 1u : 0u;
    }
}

// ------------------------------------------------------------------ host driver
struct BuildResult {
    std::vector<pt_tlas_node> tlas_nodes;
    std::vector<pt_blas> blases;
    std::vector<pt_bvh_node> blas_nodes;
    std::vector<uint32_t> tri_indices;
    uint32_t levels = 0;
    float ms_blas = 0.0f, ms_tlas = 0.0f;  // device time of the two parts
};

struct DeviceArena {  // frees everything it handed out
    std::vector<void *> blocks;
    ~DeviceArena() {
        for (void *p : blocks) (void)hipFree(p);
    }
    template <typename T>
    hipError_t get(T **out, size_t count) {
        void *p = nullptr;
        const hipError_t e = hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T));
        if (e == hipSuccess) blocks.push_back(p);
        *out = static_cast<T *>(p);
        return e;
    }
};

#define PTB_TRY(expr)                                                                                   \
    do {                                                                                                \
        const hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                         \
            err = std::string(#expr) + ": " + hipGetErrorString(e_);                                    \
            return PT_ERR_DEVICE;                                                                       \
        }                                                                                               \
    } while (0)

inline uint32_t Blocks(uint32_t n, uint32_t per = 256) { return (n + per - 1) / per; }

// TLAS::Build (bvh.cpp:204-332) for host arrays in the reference's layouts.
inline int BuildOnDevice(hipStream_t stream, const float *positions, uint32_t vertex_count, const uint32_t *indices, uint32_t index_count,
                         const pt_geometry_instance *geoms, uint32_t geom_count, BuildResult &out, std::string &err) {
    if (geom_count == 0) {
        err = "a scene needs at least one geometry instance";
        return PT_ERR_INVALID;
    }
    if (2ull * geom_count - 1 > 65535) {  // kernel/device_types.h:153-154
        err = "more than 32767 instances do not fit the TLAS's 16-bit child indices";
        return PT_ERR_LIMIT;
    }
    // ---- one mesh per distinct blas_index, defined by the first instance that names it (bvh.cpp:216-236)
    int max_mesh = 0;
    for (uint32_t i = 0; i < geom_count; ++i) {
        if (geoms[i].blas_index < 0) {
            err = "instance without blas_index";
            return PT_ERR_INVALID;
        }
        max_mesh = std::max(max_mesh, geoms[i].blas_index);
    }
    const uint32_t mesh_count = uint32_t(max_mesh) + 1;
    std::vector<MeshInfo> meshes(mesh_count, MeshInfo{0, 0, 0, 0});
    std::vector<char> seen(mesh_count, 0);
    for (uint32_t i = 0; i < geom_count; ++i) {
        const pt_geometry_instance &g = geoms[i];
        if (seen[g.blas_index]) continue;
        seen[g.blas_index] = 1;
        if (g.index_count / 3u == 0) {
            err = "instance with an empty mesh";
            return PT_ERR_INVALID;
        }
        if (uint64_t(g.first_index) + (g.index_count / 3u) * 3ull > index_count) {
            err = "mesh index range exceeds the index array";
            return PT_ERR_INVALID;
        }
        meshes[g.blas_index].first_index = g.first_index;
        meshes[g.blas_index].tri_count = g.index_count / 3u;
    }
    uint64_t total = 0;
    for (MeshInfo &m : meshes) {
        m.tri_first = uint32_t(total);
        total += m.tri_count;
    }
    if (total > kMaxPrims) {
        err = "more than 2^24 triangles in one build";
        return PT_ERR_LIMIT;
    }
    const uint32_t n = uint32_t(total);
    for (uint32_t i = 0; i < index_count; ++i)
        if (indices[i] >= vertex_count) {
            err = "vertex index out of range";
            return PT_ERR_INVALID;
        }

    DeviceArena mem;
    float *d_positions;
    uint32_t *d_indices;
    pt_geometry_instance *d_geoms;
    MeshInfo *d_meshes;
    PTB_TRY(mem.get(&d_positions, size_t(vertex_count) * 3));
    PTB_TRY(mem.get(&d_indices, index_count));
    PTB_TRY(mem.get(&d_geoms, geom_count));
    PTB_TRY(mem.get(&d_meshes, mesh_count));
    PTB_TRY(hipMemcpyAsync(d_positions, positions, size_t(vertex_count) * 12, hipMemcpyHostToDevice, stream));
    PTB_TRY(hipMemcpyAsync(d_indices, indices, size_t(index_count) * 4, hipMemcpyHostToDevice, stream));
    PTB_TRY(hipMemcpyAsync(d_geoms, geoms, size_t(geom_count) * sizeof(pt_geometry_instance), hipMemcpyHostToDevice, stream));
    PTB_TRY(hipMemcpyAsync(d_meshes, meshes.data(), size_t(mesh_count) * sizeof(MeshInfo), hipMemcpyHostToDevice, stream));

    Prims P;
    for (int a = 0; a < 3; ++a) {
        PTB_TRY(mem.get(&P.lo[a], n));
        PTB_TRY(mem.get(&P.hi[a], n));
        PTB_TRY(mem.get(&P.centroid[a], n));
    }
    PTB_TRY(mem.get(&P.blas, n));
    // nodes made by the level passes (every one holds a triangle, every split makes two) and, behind
    // them, the nodes made inside single-wave subtrees (fewer than two per triangle)
    const size_t max_level_nodes = 2 * size_t(n) + mesh_count, max_nodes = max_level_nodes + 2 * size_t(n);
    const size_t max_active = size_t(n) / kSmall + mesh_count;
    uint32_t *order[2], *node_of[2], *flag, *scan, *pfront, *qback, *tile_sums, *counters, *small_list;
    BNode *nodes;
    Bins *bins;
    unsigned long long *keys;
    for (int k = 0; k < 2; ++k) {
        PTB_TRY(mem.get(&order[k], n));
        PTB_TRY(mem.get(&node_of[k], n));
    }
    PTB_TRY(mem.get(&flag, n));
    PTB_TRY(mem.get(&scan, size_t(n) + 1));
    PTB_TRY(mem.get(&pfront, n));
    PTB_TRY(mem.get(&qback, n));
    PTB_TRY(mem.get(&tile_sums, 8192));
    PTB_TRY(mem.get(&counters, 8));  // [0] node count, [1] scan total, [2] TLAS status, [3] bin slots of the pass, [4] small subtrees
    PTB_TRY(mem.get(&small_list, size_t(n) + mesh_count));
    PTB_TRY(mem.get(&nodes, max_nodes));
    PTB_TRY(mem.get(&bins, max_active));
    PTB_TRY(mem.get(&keys, max_level_nodes * 6));
    PTB_TRY(hipMemsetAsync(counters, 0, 8 * 4, stream));

    hipEvent_t ev[3];
    for (hipEvent_t &e : ev) PTB_TRY(hipEventCreate(&e));
    struct EventGuard {
        hipEvent_t *e;
        ~EventGuard() {
            for (int k = 0; k < 3; ++k) (void)hipEventDestroy(e[k]);
        }
    } event_guard{ev};
    PTB_TRY(hipEventRecord(ev[0], stream));

    hipLaunchKernelGGL(kb_prims, dim3(Blocks(n)), dim3(256), 0, stream, d_positions, d_indices, d_meshes, mesh_count, P, n, order[0], node_of[0]);
    hipLaunchKernelGGL(kb_roots, dim3(Blocks(mesh_count)), dim3(256), 0, stream, nodes, d_meshes, mesh_count, keys, small_list, counters + 4);
    hipLaunchKernelGGL(kb_bounds_keys, dim3(Blocks(n, kAggBlock)), dim3(kAggBlock), 0, stream, nodes, 0u, mesh_count, node_of[0], order[0], P, n, keys);
    hipLaunchKernelGGL(kb_bounds_decode, dim3(Blocks(mesh_count)), dim3(256), 0, stream, nodes, 0u, mesh_count, keys, order[0], P);
    uint32_t node_count = mesh_count;
    PTB_TRY(hipMemcpyAsync(counters, &node_count, 4, hipMemcpyHostToDevice, stream));

    std::vector<std::pair<uint32_t, uint32_t>> levels;
    uint32_t begin = 0, end = mesh_count;
    int cur = 0;
    const uint32_t tiles = Blocks(n, kScanTile);
    while (begin < end) {
        levels.emplace_back(begin, end);
        const uint32_t width = end - begin;
        PTB_TRY(hipMemsetAsync(counters + 3, 0, 4, stream));
        hipLaunchKernelGGL(kb_level_init, dim3(Blocks(width)), dim3(256), 0, stream, nodes, begin, end, bins, counters + 3);
        hipLaunchKernelGGL(kb_centroid_bounds, dim3(Blocks(n, kAggBlock)), dim3(kAggBlock), 0, stream, nodes, begin, end, node_of[cur], order[cur], P, n);
        hipLaunchKernelGGL(kb_bin, dim3(Blocks(n, kAggBlock)), dim3(kAggBlock), 0, stream, nodes, begin, end, node_of[cur], order[cur], P, n, bins);
        hipLaunchKernelGGL(kb_split, dim3(Blocks(width, 64)), dim3(64), 0, stream, nodes, begin, end, bins);
        hipLaunchKernelGGL(kb_flags, dim3(Blocks(n)), dim3(256), 0, stream, nodes, begin, end, node_of[cur], order[cur], P, n, flag);
        hipLaunchKernelGGL(kb_scan_tiles, dim3(tiles), dim3(kScanBlock), 0, stream, flag, n, scan, tile_sums);
        hipLaunchKernelGGL(kb_scan_sums, dim3(1), dim3(1024), 0, stream, tile_sums, tiles, counters + 1);
        hipLaunchKernelGGL(kb_scan_add, dim3(Blocks(n)), dim3(256), 0, stream, scan, n, tile_sums, counters + 1);
        hipLaunchKernelGGL(kb_rank_tables, dim3(Blocks(n)), dim3(256), 0, stream, nodes, begin, end, node_of[cur], flag, scan, n, pfront, qback);
        hipLaunchKernelGGL(kb_pairs, dim3(Blocks(width, 64)), dim3(64), 0, stream, nodes, begin, end, scan, pfront, qback, counters, keys, small_list, counters + 4);
        hipLaunchKernelGGL(kb_scatter, dim3(Blocks(n)), dim3(256), 0, stream, nodes, begin, end, node_of[cur], order[cur], flag, scan, pfront, qback, n,
                           order[cur ^ 1], node_of[cur ^ 1]);
        cur ^= 1;
        PTB_TRY(hipMemcpyAsync(&node_count, counters, 4, hipMemcpyDeviceToHost, stream));
        PTB_TRY(hipStreamSynchronize(stream));
        if (node_count > max_level_nodes) {
            err = "internal: node storage exceeded";
            return PT_ERR_DEVICE;
        }
        begin = end;
        end = node_count;
        if (begin < end) {
            hipLaunchKernelGGL(kb_bounds_keys, dim3(Blocks(n, kAggBlock)), dim3(kAggBlock), 0, stream, nodes, begin, end, node_of[cur], order[cur], P, n, keys);
            hipLaunchKernelGGL(kb_bounds_decode, dim3(Blocks(end - begin)), dim3(256), 0, stream, nodes, begin, end, keys, order[cur], P);
        }
    }
    out.levels = uint32_t(levels.size());

    // ---- subtrees of kSmall triangles or fewer, one wave each
    const uint32_t level_nodes = node_count;
    uint32_t small_count = 0;
    PTB_TRY(hipMemcpyAsync(&small_count, counters + 4, 4, hipMemcpyDeviceToHost, stream));
    PTB_TRY(hipStreamSynchronize(stream));
    if (small_count > 0) {
        hipLaunchKernelGGL(kb_small_subtrees, dim3(small_count), dim3(64), 0, stream, nodes, small_list, small_count, order[cur], P, counters);
        PTB_TRY(hipMemcpyAsync(&node_count, counters, 4, hipMemcpyDeviceToHost, stream));
        PTB_TRY(hipStreamSynchronize(stream));
        if (node_count > max_nodes) {
            err = "internal: node storage exceeded";
            return PT_ERR_DEVICE;
        }
    }

    // ---- renumber: subtree sizes bottom-up, preorder ranks top-down
    for (size_t l = levels.size(); l-- > 0;)
        hipLaunchKernelGGL(kb_subtree_sizes, dim3(Blocks(levels[l].second - levels[l].first)), dim3(256), 0, stream, nodes, levels[l].first, levels[l].second);
    for (size_t l = 0; l < levels.size(); ++l)
        hipLaunchKernelGGL(kb_preorder, dim3(Blocks(levels[l].second - levels[l].first)), dim3(256), 0, stream, nodes, levels[l].first, levels[l].second);
    if (node_count > level_nodes)
        hipLaunchKernelGGL(kb_small_indices, dim3(Blocks(node_count - level_nodes)), dim3(256), 0, stream, nodes, level_nodes, node_count);
    std::vector<BNode> roots(mesh_count);
    PTB_TRY(hipMemcpyAsync(roots.data(), nodes, size_t(mesh_count) * sizeof(BNode), hipMemcpyDeviceToHost, stream));
    PTB_TRY(hipStreamSynchronize(stream));
    out.blases.assign(mesh_count, pt_blas{0, 0, 0, 0});
    uint32_t node_total = 0;
    for (uint32_t b = 0; b < mesh_count; ++b) {
        if (meshes[b].tri_count == 0) continue;  // unused blas_index: an all-zero DeviceBLAS (bvh.cpp:238-264)
        const uint32_t mesh_nodes = 1 + 2 * roots[b].internal;
        meshes[b].node_offset = node_total;
        out.blases[b] = pt_blas{node_total, mesh_nodes, meshes[b].tri_first, meshes[b].tri_count};
        node_total += mesh_nodes;
    }
    PTB_TRY(hipMemcpyAsync(d_meshes, meshes.data(), size_t(mesh_count) * sizeof(MeshInfo), hipMemcpyHostToDevice, stream));
    pt_bvh_node *d_out_nodes;
    uint32_t *d_out_tris;
    PTB_TRY(mem.get(&d_out_nodes, node_total));
    PTB_TRY(mem.get(&d_out_tris, n));
    hipLaunchKernelGGL(kb_emit_nodes, dim3(Blocks(node_count)), dim3(256), 0, stream, nodes, node_count, d_meshes, d_out_nodes);
    hipLaunchKernelGGL(kb_emit_tris, dim3(Blocks(n)), dim3(256), 0, stream, order[cur], P, d_meshes, n, d_out_tris);
    PTB_TRY(hipEventRecord(ev[1], stream));

    // ---- TLAS
    pt_tlas_node *d_tlas;
    float *d_abox;
    uint32_t *d_active;
    PTB_TRY(mem.get(&d_tlas, 2 * size_t(geom_count)));
    PTB_TRY(mem.get(&d_abox, 6 * size_t(geom_count)));
    PTB_TRY(mem.get(&d_active, geom_count));
    hipLaunchKernelGGL(kb_instance_bounds, dim3(Blocks(geom_count)), dim3(256), 0, stream, d_geoms, geom_count, nodes, d_tlas, d_abox, d_active);
    if (geom_count <= kTlasLdsInstances)
        hipLaunchKernelGGL((kb_tlas_pairing<kTlasSmallThreads, true>), dim3(1), dim3(kTlasSmallThreads), 0, stream, d_tlas, d_abox, d_active, geom_count, counters + 2);
    else
        hipLaunchKernelGGL((kb_tlas_pairing<kTlasThreads, false>), dim3(1), dim3(kTlasThreads), 0, stream, d_tlas, d_abox, d_active, geom_count, counters + 2);
    PTB_TRY(hipEventRecord(ev[2], stream));
    PTB_TRY(hipGetLastError());

    out.blas_nodes.resize(node_total);
    out.tri_indices.resize(n);
    out.tlas_nodes.resize(2 * size_t(geom_count));
    uint32_t tlas_status = 0;
    PTB_TRY(hipMemcpyAsync(out.blas_nodes.data(), d_out_nodes, size_t(node_total) * sizeof(pt_bvh_node), hipMemcpyDeviceToHost, stream));
    PTB_TRY(hipMemcpyAsync(out.tri_indices.data(), d_out_tris, size_t(n) * 4, hipMemcpyDeviceToHost, stream));
    PTB_TRY(hipMemcpyAsync(out.tlas_nodes.data(), d_tlas, out.tlas_nodes.size() * sizeof(pt_tlas_node), hipMemcpyDeviceToHost, stream));
    PTB_TRY(hipMemcpyAsync(&tlas_status, counters + 2, 4, hipMemcpyDeviceToHost, stream));
    PTB_TRY(hipStreamSynchronize(stream));
    if (tlas_status != 0) {
        err = "TLAS pairing did not converge (non-finite instance bounds?)";
        return PT_ERR_INVALID;
    }
    (void)hipEventElapsedTime(&out.ms_blas, ev[0], ev[1]);
    (void)hipEventElapsedTime(&out.ms_tlas, ev[1], ev[2]);
    return PT_OK;
}

}  // namespace ptb
