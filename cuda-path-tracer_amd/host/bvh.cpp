// Host acceleration-structure build of the facade: BLAS = binned SAH (8 bins per axis),
// TLAS = agglomerative pairing by smallest merged surface area.  The node numbering,
// triangle permutation and every float produced match the algorithm of the
// reference's bvh.cpp:5-332 so that arrays built here are interchangeable with
// arrays the reference's own builder would hand to pt_scene_upload.
//
// Unlike the reference (recursive, one axis at a time) the BLAS build below is
// iterative with an explicit work list and bins all three axes in one sweep.
#include <algorithm>
#include <cstring>

#include "pt_host.h"

namespace pth {

namespace {

constexpr int kBins = 8;

struct PrimInfo {
    float lo[3], hi[3];
    float centroid[3];
};

struct BinSet {
    Box box[3][kBins];
    uint32_t count[3][kBins];
};

inline Box NodeBox(const pt_bvh_node &n) {
    Box b;
    std::memcpy(b.lo, n.min, sizeof(b.lo));
    std::memcpy(b.hi, n.max, sizeof(b.hi));
    return b;
}
inline void SetNodeBox(pt_bvh_node &n, const Box &b) {
    std::memcpy(n.min, b.lo, sizeof(b.lo));
    std::memcpy(n.max, b.hi, sizeof(b.hi));
}
inline Box PrimBox(const PrimInfo &p) {
    Box b;
    std::memcpy(b.lo, p.lo, sizeof(b.lo));
    std::memcpy(b.hi, p.hi, sizeof(b.hi));
    return b;
}

struct Split {
    bool valid = false;
    int axis = 0;
    float pos = 0.0f;
    float cost = FLT_MAX;
};

// Candidate plane of one node (bvh.cpp:131-199): centroid bounds, 8 bins per axis with
// non-zero centroid extent, 7 planes per axis, first strictly smaller cost wins (axis-major order).
Split FindSplitPlane(const std::vector<PrimInfo> &prims, const std::vector<uint32_t> &order, uint32_t first,
                     uint32_t count) {
    Split best;
    Box cb;
    for (uint32_t i = 0; i < count; ++i) cb.Grow(prims[order[first + i]].centroid);
    if (cb.Empty()) return best;
    float extent[3], inv_width[3];
    for (int a = 0; a < 3; ++a) {
        extent[a] = cb.hi[a] - cb.lo[a];
        inv_width[a] = 1.0f / Max(extent[a], 1e-30f);
    }
    BinSet bins;
    std::memset(bins.count, 0, sizeof(bins.count));
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < kBins; ++b) bins.box[a][b] = Box();
    for (uint32_t i = 0; i < count; ++i) {
        const PrimInfo &p = prims[order[first + i]];
        const Box pb = PrimBox(p);
        for (int a = 0; a < 3; ++a) {
            if (extent[a] <= 0.0f) continue;
            const float scaled = (p.centroid[a] - cb.lo[a]) * inv_width[a];
            int b = static_cast<int>(scaled * kBins);
            b = b < 0 ? 0 : (b > kBins - 1 ? kBins - 1 : b);
            bins.count[a][b]++;
            bins.box[a][b].Grow(pb);
        }
    }
    for (int a = 0; a < 3; ++a) {
        if (extent[a] <= 0.0f) continue;
        float left_area[kBins - 1], right_area[kBins - 1];
        uint32_t left_count[kBins - 1], right_count[kBins - 1];
        Box lbox, rbox;
        uint32_t lsum = 0, rsum = 0;
        for (int i = 0; i < kBins - 1; ++i) {
            lsum += bins.count[a][i];
            left_count[i] = lsum;
            lbox.Grow(bins.box[a][i]);
            left_area[i] = lbox.Area();
            const int rb = kBins - 1 - i;
            rsum += bins.count[a][rb];
            right_count[rb - 1] = rsum;
            rbox.Grow(bins.box[a][rb]);
            right_area[rb - 1] = rbox.Area();
        }
        for (int i = 0; i < kBins - 1; ++i) {
            const float cost = left_count[i] * left_area[i] + right_count[i] * right_area[i];
            if (cost < best.cost) {
                best.cost = cost;
                best.axis = a;
                best.pos = cb.lo[a] + extent[a] * (float(i + 1) / kBins);
                best.valid = true;
            }
        }
    }
    return best;
}

}  // namespace

void BLAS::Build(const Scene &scene, const TriangleMesh &triangle_mesh, SplitStrategy split_type) {
    Build(scene.Positions().data(), scene.Indices().data(), triangle_mesh, split_type);
}

void BLAS::Build(const float *positions, const uint32_t *indices, const TriangleMesh &mesh, SplitStrategy split_type) {
    // Only SAH is reachable in the reference (bvh.h:51 default); its Midpoint plane is wrong
    // (bvh.cpp:127) and Median falls off the end of FindSplitPlane, so they are rejected here.
    if (split_type != SplitStrategy::SAH) throw Exception("BLAS::Build: only SplitStrategy::SAH is supported");
    m_nodes.clear();
    m_tri_indices.clear();
    const uint32_t prim_count = mesh.index_count / 3u;
    if (prim_count == 0) return;

    std::vector<PrimInfo> prims(prim_count);
    for (uint32_t t = 0; t < prim_count; ++t) {
        const float *v[3];
        for (int k = 0; k < 3; ++k) v[k] = positions + 3 * size_t(indices[mesh.first_index + t * 3 + k]);
        PrimInfo &p = prims[t];
        Box b;
        b.Grow(v[0]);
        b.Grow(v[1]);
        b.Grow(v[2]);
        std::memcpy(p.lo, b.lo, sizeof(p.lo));
        std::memcpy(p.hi, b.hi, sizeof(p.hi));
        for (int a = 0; a < 3; ++a) p.centroid[a] = ((v[0][a] + v[1][a]) + v[2][a]) * (1.0f / 3.0f);
    }
    m_tri_indices.resize(prim_count);
    for (uint32_t i = 0; i < prim_count; ++i) m_tri_indices[i] = i;
    m_nodes.reserve(2 * size_t(prim_count) - 1);

    auto leaf_bounds = [&](uint32_t first, uint32_t count) {
        Box b;
        for (uint32_t i = 0; i < count; ++i) b.Grow(PrimBox(prims[m_tri_indices[first + i]]));
        return b;
    };

    pt_bvh_node root;
    std::memset(&root, 0, sizeof(root));
    root.first_primitive = 0;
    root.primitive_count = prim_count;
    SetNodeBox(root, leaf_bounds(0, prim_count));
    m_nodes.push_back(root);

    // Depth-first work list; pushing the right child below the left one reproduces the
    // allocation order of the reference's recursion (left subtree completes first).
    std::vector<uint32_t> work{0};
    while (!work.empty()) {
        const uint32_t index = work.back();
        work.pop_back();
        const uint32_t first = m_nodes[index].first_primitive;
        const uint32_t count = m_nodes[index].primitive_count;
        if (count <= 2) continue;
        const Split split = FindSplitPlane(prims, m_tri_indices, first, count);
        const float parent_cost = count * NodeBox(m_nodes[index]).Area();
        if (!split.valid || split.cost >= parent_cost) continue;
        // in-place partition by centroid (bvh.cpp:80-86)
        uint32_t i = first, j = first + count - 1;
        while (i <= j) {
            if (prims[m_tri_indices[i]].centroid[split.axis] < split.pos) {
                ++i;
            } else {
                std::swap(m_tri_indices[i], m_tri_indices[j]);
                if (j == 0) break;  // cannot happen with a valid plane; guards the unsigned wrap
                --j;
            }
        }
        const uint32_t left_count = i - first;
        if (left_count == 0 || left_count == count) continue;
        const uint32_t left_index = uint32_t(m_nodes.size());
        pt_bvh_node l, r;
        std::memset(&l, 0, sizeof(l));
        std::memset(&r, 0, sizeof(r));
        l.first_primitive = first;
        l.primitive_count = left_count;
        SetNodeBox(l, leaf_bounds(first, left_count));
        r.first_primitive = first + left_count;
        r.primitive_count = count - left_count;
        SetNodeBox(r, leaf_bounds(first + left_count, count - left_count));
        m_nodes.push_back(l);
        m_nodes.push_back(r);
        m_nodes[index].first_primitive = left_index;
        m_nodes[index].primitive_count = 0;
        work.push_back(left_index + 1);
        work.push_back(left_index);
    }
}

Box BLAS::Bounds() const {
    if (m_nodes.empty()) return Box();
    return NodeBox(m_nodes[0]);
}

void TLAS::Build(const Scene &scene, const std::vector<pt_geometry_instance> &instances, SplitStrategy split_type) {
    Build(scene.Positions().data(), scene.Indices().data(), instances, split_type);
}

void TLAS::Build(const float *positions, const uint32_t *indices, const std::vector<pt_geometry_instance> &instances,
                 SplitStrategy split_type) {
    m_dirty = true;
    m_tlas_nodes.clear();
    m_device_blases.clear();
    m_blas_nodes.clear();
    m_tri_indices.clear();
    if (instances.empty()) throw Exception("TLAS::Build: a scene needs at least one geometry instance");
    // TLASNode packs child indices into 16 bits each (kernel/device_types.h:153-154).
    if (2 * instances.size() - 1 > 65535) throw Exception("TLAS::Build: more than 32767 instances do not fit 16-bit child indices");

    // ---- one BLAS per distinct blas_index (bvh.cpp:216-236)
    int max_mesh_id = 0;
    for (const auto &inst : instances) {
        if (inst.blas_index < 0) throw Exception("TLAS::Build: instance without blas_index");
        max_mesh_id = std::max(max_mesh_id, inst.blas_index);
    }
    std::vector<BLAS> blases(size_t(max_mesh_id) + 1);
    std::vector<Box> instance_bounds(instances.size());
    for (size_t i = 0; i < instances.size(); ++i) {
        const auto &inst = instances[i];
        BLAS &blas = blases[inst.blas_index];
        if (blas.Nodes().empty()) {
            TriangleMesh mesh{inst.first_index, inst.index_count, inst.vertex_count};
            blas.Build(positions, indices, mesh, split_type);
            if (blas.Nodes().empty()) throw Exception("TLAS::Build: instance with an empty mesh");
        }
        // world bounds = box of the 8 transformed corners (math/aabb.h:50-63)
        const Box ob = blas.Bounds();
        Mat4 T;
        std::memcpy(T.m, inst.transform, sizeof(T.m));
        Box wb;
        for (int c = 0; c < 8; ++c) {
            const Vec3 corner{(c & 1) ? ob.hi[0] : ob.lo[0], (c & 2) ? ob.hi[1] : ob.lo[1], (c & 4) ? ob.hi[2] : ob.lo[2]};
            const Vec3 w = TransformPoint(T, corner);
            const float p[3] = {w.x, w.y, w.z};
            wb.Grow(p);
        }
        instance_bounds[i] = wb;
    }

    // ---- flatten with node-offset fix-up (bvh.cpp:238-264)
    m_device_blases.assign(blases.size(), pt_blas{0, 0, 0, 0});
    for (size_t id = 0; id < blases.size(); ++id) {
        const BLAS &blas = blases[id];
        if (blas.Nodes().empty()) continue;
        const uint32_t node_offset = uint32_t(m_blas_nodes.size());
        const uint32_t tri_offset = uint32_t(m_tri_indices.size());
        for (pt_bvh_node n : blas.Nodes()) {
            if (n.primitive_count == 0) n.first_primitive += node_offset;
            m_blas_nodes.push_back(n);
        }
        m_tri_indices.insert(m_tri_indices.end(), blas.TriIndices().begin(), blas.TriIndices().end());
        m_device_blases[id] = pt_blas{node_offset, uint32_t(blas.Nodes().size()), tri_offset, uint32_t(blas.TriIndices().size())};
    }

    // ---- agglomerative TLAS (bvh.cpp:266-331): slot 0 is reserved for the root copy,
    // leaves sit at 1..n in instance order, merged nodes are appended.
    const size_t n = instances.size();
    m_tlas_nodes.reserve(2 * n);
    auto make_node = [](const Box &b, uint32_t instance, uint32_t left_right) {
        pt_tlas_node t;
        std::memcpy(t.min, b.lo, sizeof(t.min));
        std::memcpy(t.max, b.hi, sizeof(t.max));
        t.instance = instance;
        t.left_right = left_right;
        return t;
    };
    auto box_of = [](const pt_tlas_node &t) {
        Box b;
        std::memcpy(b.lo, t.min, sizeof(b.lo));
        std::memcpy(b.hi, t.max, sizeof(b.hi));
        return b;
    };
    m_tlas_nodes.push_back(make_node(Box(), 0, 0));
    std::vector<uint32_t> active(n);  // TLAS node index of each still-unmerged cluster
    for (uint32_t i = 0; i < n; ++i) {
        active[i] = uint32_t(m_tlas_nodes.size());
        m_tlas_nodes.push_back(make_node(instance_bounds[i], i, 0));
    }
    int live = int(n);
    auto best_partner = [&](int who) {
        float best_area = FLT_MAX;
        int best = -1;
        const Box mine = box_of(m_tlas_nodes[active[who]]);
        for (int j = 0; j < live; ++j) {
            if (j == who) continue;
            Box merged = mine;
            merged.Grow(box_of(m_tlas_nodes[active[j]]));
            const float area = merged.Area();
            if (area < best_area) {
                best_area = area;
                best = j;
            }
        }
        return best;
    };
    int a = 0;
    int b = best_partner(a);
    while (live > 1) {
        const int c = best_partner(b);
        if (a == c) {
            const uint32_t ia = active[a], ib = active[b];
            Box merged = box_of(m_tlas_nodes[ia]);
            merged.Grow(box_of(m_tlas_nodes[ib]));
            m_tlas_nodes.push_back(make_node(merged, 0, ia | (ib << 16)));
            active[a] = uint32_t(m_tlas_nodes.size() - 1);
            active[b] = active[live - 1];
            --live;
            b = best_partner(a);
        } else {
            a = b;
            b = c;
        }
    }
    m_tlas_nodes[0] = m_tlas_nodes[active[a]];
}

void TLAS::BuildOnDevice(pt_ctx *ctx, const float *positions, uint32_t vertex_count, const uint32_t *indices, uint32_t index_count,
                         const std::vector<pt_geometry_instance> &instances) {
    m_dirty = true;
    pt_bvh *h = nullptr;
    if (pt_bvh_build(ctx, positions, vertex_count, indices, index_count, instances.data(), uint32_t(instances.size()), &h) != PT_OK)
        throw Exception(std::string("TLAS::BuildOnDevice: ") + pt_last_error(ctx));
    uint32_t n_tlas = 0, n_blas = 0, n_nodes = 0, n_tris = 0;
    pt_bvh_sizes(h, &n_tlas, &n_blas, &n_nodes, &n_tris);
    m_tlas_nodes.resize(n_tlas);
    m_device_blases.resize(n_blas);
    m_blas_nodes.resize(n_nodes);
    m_tri_indices.resize(n_tris);
    pt_bvh_get(h, m_tlas_nodes.data(), m_device_blases.data(), m_blas_nodes.data(), m_tri_indices.data());
    pt_bvh_free(h);
}

void TLAS::UpdateDevice(pt_scene_desc &scene) {
    scene.tlas_nodes = m_tlas_nodes.data();
    scene.tlas_node_count = uint32_t(m_tlas_nodes.size());
    scene.blases = m_device_blases.data();
    scene.blas_count = uint32_t(m_device_blases.size());
    scene.blas_nodes = m_blas_nodes.data();
    scene.blas_node_count = uint32_t(m_blas_nodes.size());
    scene.tri_indices = m_tri_indices.data();
    scene.tri_index_count = uint32_t(m_tri_indices.size());
    m_dirty = false;
}

void TLAS::FreeDevice(pt_scene_desc &scene) {
    scene.tlas_nodes = nullptr;
    scene.blases = nullptr;
    scene.blas_nodes = nullptr;
    scene.tri_indices = nullptr;
    scene.tlas_node_count = scene.blas_count = scene.blas_node_count = scene.tri_index_count = 0;
}

}  // namespace pth
