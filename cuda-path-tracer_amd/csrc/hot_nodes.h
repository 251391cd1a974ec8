// Host side of the node cache of the closest-hit query kernel: puts the node records that rays visit most at
// the front of their array, so that "is this node cached?" is one comparison of its reference.
//
// Node visits are concentrated: on the helmet's bounce rays 128 of 27 000 4-wide records take 54 % of all
// visits, 256 of 55 000 binary records 51 % (tools/node_hist.py).  The query kernel keeps the first kHot*
// records of the array in LDS, one copy per compute unit shared by its 16 waves, and fetches only the others
// through the vector memory pipeline, which is what limits it (tools/sensitivity.sh).  Which records are
// hot is estimated from the geometry alone: a convex box is crossed by random lines in proportion to its
// surface area, so records are ranked by the world-space area of the box they cover (object-space area times
// the squared scale of every instance that uses the tree).  References are only names: the traversal
// order, the boxes and every result are unchanged by the renumbering.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>
#include <vector>

#include "dscene.h"

namespace pth_hot {

using ptd::DNode;
using ptd::DNode4;
using ptd::kNoRef;

constexpr uint32_t kHotBytes = 16384;          // LDS set aside for cached node records
constexpr uint32_t kHot2 = kHotBytes / 64;     // binary records (DNode)
constexpr uint32_t kHot4 = kHotBytes / 128;    // 4-wide records (DNode4)

inline float HalfArea(const float *mn, const float *mx) {
    const float ex = mx[0] - mn[0], ey = mx[1] - mn[1], ez = mx[2] - mn[2];
    const float a = ex * ey + ey * ez + ez * ex;
    return a == a ? a : 0.0f;
}

// Half the surface area of the box a record covers (the union of its children's boxes).
inline float CoveredArea(const DNode &n) {
    const float mn[3] = {std::fmin(n.a.x, n.c.x), std::fmin(n.a.y, n.c.y), std::fmin(n.a.z, n.c.z)};
    const float mx[3] = {std::fmax(n.b.x, n.d.x), std::fmax(n.b.y, n.d.y), std::fmax(n.b.z, n.d.z)};
    return HalfArea(mn, mx);
}
inline float CoveredArea(const DNode4 &n) {
    const float *v = reinterpret_cast<const float *>(&n);  // lo_x[4], hi_x[4], lo_y[4], hi_y[4], lo_z[4], hi_z[4]
    float mn[3], mx[3];
    for (int a = 0; a < 3; ++a) {
        mn[a] = INFINITY;
        mx[a] = -INFINITY;
        for (int k = 0; k < 4; ++k) {  // fmin / fmax skip the NaN bounds of absent children
            mn[a] = std::fmin(mn[a], v[8 * a + k]);
            mx[a] = std::fmax(mx[a], v[8 * a + 4 + k]);
        }
    }
    return HalfArea(mn, mx);
}

// old index -> new index: the `hot` heaviest records first (heaviest at 0), the others behind them in their
// old order.
inline std::vector<uint32_t> HotFirst(const std::vector<float> &weight, uint32_t hot) {
    const uint32_t n = uint32_t(weight.size());
    hot = std::min(hot, n);
    std::vector<uint32_t> by_weight(n);
    std::iota(by_weight.begin(), by_weight.end(), 0u);
    std::partial_sort(by_weight.begin(), by_weight.begin() + hot, by_weight.end(), [&](uint32_t a, uint32_t b) { return weight[a] != weight[b] ? weight[a] > weight[b] : a < b; });
    std::vector<uint32_t> map(n, 0xFFFFFFFFu);
    for (uint32_t k = 0; k < hot; ++k) map[by_weight[k]] = k;
    uint32_t next = hot;
    for (uint32_t i = 0; i < n; ++i)
        if (map[i] == 0xFFFFFFFFu) map[i] = next++;
    return map;
}

inline uint32_t Renamed(uint32_t ref, const std::vector<uint32_t> &map) { return ref < kNoRef && ref < map.size() ? map[ref] : ref; }

template <class T>
void Reorder(std::vector<T> &nodes, const std::vector<uint32_t> &map) {
    std::vector<T> out(nodes.size());
    for (size_t i = 0; i < nodes.size(); ++i) out[map[i]] = nodes[i];
    nodes.swap(out);
}
inline void RenameChildren(DNode &n, const std::vector<uint32_t> &map) {
    n.a.w = __builtin_bit_cast(float, Renamed(__builtin_bit_cast(uint32_t, n.a.w), map));
    n.b.w = __builtin_bit_cast(float, Renamed(__builtin_bit_cast(uint32_t, n.b.w), map));
}
inline void RenameChildren(DNode4 &n, const std::vector<uint32_t> &map) {
    float *r = reinterpret_cast<float *>(&n.refs);
    for (int k = 0; k < 4; ++k) r[k] = __builtin_bit_cast(float, Renamed(__builtin_bit_cast(uint32_t, r[k]), map));
}

// Renumbers `nodes` hot-first.  scale[i] multiplies record i's covered area (1 for the top level, the summed
// squared scale of a tree's instances below it).  Returns the map for the references held outside the array.
template <class T>
std::vector<uint32_t> PutHotFirst(std::vector<T> &nodes, const std::vector<float> &scale, uint32_t hot) {
    std::vector<float> weight(nodes.size());
    for (size_t i = 0; i < nodes.size(); ++i) weight[i] = CoveredArea(nodes[i]) * scale[i];
    const std::vector<uint32_t> map = HotFirst(weight, hot);
    for (T &n : nodes) RenameChildren(n, map);
    Reorder(nodes, map);
    return map;
}

}  // namespace pth_hot
