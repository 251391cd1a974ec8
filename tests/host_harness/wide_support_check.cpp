// Host check of the instance-box clipping of the 4-wide layout (csrc/wide_bvh.h: MeshSupport, TightInstanceBox)
// against brute force over the vertices, on random triangle soups under a median-split tree of its own.
// Built by tests/test_wide_support.py with the host sanitizers; needs no GPU.  Exit status 0 = every check held.
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <random>

#include "../../cuda-path-tracer_amd/csrc/wide_bvh.h"

namespace {

struct SoupTree {
    struct Node {
        ptw::Box box;
        uint32_t left = 0, right = 0, first = 0, count = 0;  // count > 0: leaf over tris[first, first + count)
    };
    std::vector<float> verts;  // 9 floats per triangle
    std::vector<uint32_t> order;
    std::vector<Node> nodes;
    bool is_leaf(uint32_t i) const { return nodes[i].count > 0; }
    uint32_t left(uint32_t i) const { return nodes[i].left; }
    uint32_t right(uint32_t i) const { return nodes[i].right; }
    ptw::Box box(uint32_t i) const { return nodes[i].box; }
    void for_each_vertex(uint32_t i, const std::function<void(float, float, float)> &f) const {
        for (uint32_t k = 0; k < nodes[i].count; ++k) {
            const float *t = &verts[9 * order[nodes[i].first + k]];
            for (int c = 0; c < 3; ++c) f(t[3 * c], t[3 * c + 1], t[3 * c + 2]);
        }
    }
    ptw::Box bounds(uint32_t first, uint32_t count) const {
        ptw::Box b{{INFINITY, INFINITY, INFINITY}, {-INFINITY, -INFINITY, -INFINITY}};
        for (uint32_t k = 0; k < count; ++k)
            for (int c = 0; c < 3; ++c)
                for (int a = 0; a < 3; ++a) {
                    const float v = verts[9 * order[first + k] + 3 * c + a];
                    b.mn[a] = std::fmin(b.mn[a], v);
                    b.mx[a] = std::fmax(b.mx[a], v);
                }
        return b;
    }
    uint32_t build(uint32_t first, uint32_t count, uint32_t leaf_size) {
        const uint32_t me = uint32_t(nodes.size());
        nodes.emplace_back();
        nodes[me].box = bounds(first, count);
        if (count <= leaf_size) {
            nodes[me].first = first;
            nodes[me].count = count;
            return me;
        }
        int axis = 0;
        for (int a = 1; a < 3; ++a)
            if (nodes[me].box.mx[a] - nodes[me].box.mn[a] > nodes[me].box.mx[axis] - nodes[me].box.mn[axis]) axis = a;
        auto key = [&](uint32_t t) { return verts[9 * t + axis] + verts[9 * t + 3 + axis] + verts[9 * t + 6 + axis]; };
        std::nth_element(order.begin() + first, order.begin() + first + count / 2, order.begin() + first + count,
                         [&](uint32_t a, uint32_t b) { return key(a) < key(b); });
        const uint32_t l = build(first, count / 2, leaf_size);
        const uint32_t r = build(first + count / 2, count - count / 2, leaf_size);
        nodes[me].left = l;
        nodes[me].right = r;
        return me;
    }
};

int failures = 0;
#define CHECK(cond, ...)                          \
    do {                                          \
        if (!(cond)) {                            \
            ++failures;                           \
            std::fprintf(stderr, "FAILED %s:%d: %s: ", __FILE__, __LINE__, #cond); \
            std::fprintf(stderr, __VA_ARGS__);    \
            std::fprintf(stderr, "\n");           \
        }                                         \
    } while (0)

}  // namespace

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? std::atoi(argv[1]) : 200;
    std::mt19937 gen(12345);
    std::uniform_real_distribution<float> uni(-1.0f, 1.0f);
    long searches = 0;
    for (int round = 0; round < rounds; ++round) {
        SoupTree tree;
        const uint32_t tris = round == 0 ? 1u : 1u + uint32_t(gen() % 3000u);
        // a flat, elongated cloud along a random diagonal: its own box is mostly empty
        const float sx = 0.05f + std::fabs(uni(gen)) * 3.0f, sy = 0.05f + std::fabs(uni(gen)), sz = 0.02f + 0.2f * std::fabs(uni(gen));
        const float tilt = uni(gen);
        for (uint32_t t = 0; t < tris; ++t) {
            const float cx = uni(gen) * sx, cy = uni(gen) * sy, cz = uni(gen) * sz;
            for (int c = 0; c < 3; ++c) {
                const float x = cx + 0.05f * uni(gen), y = cy + 0.05f * uni(gen), z = cz + 0.05f * uni(gen);
                tree.verts.push_back(x);
                tree.verts.push_back(y + tilt * x);
                tree.verts.push_back(z - tilt * x);
            }
            tree.order.push_back(t);
        }
        const uint32_t root = tree.build(0, tris, 1u + uint32_t(gen() % 4u));

        // support function against brute force, with an ample and with a starved budget
        for (int k = 0; k < 8; ++k) {
            const double dir[3] = {uni(gen), uni(gen), uni(gen)};
            double brute = -INFINITY;
            for (size_t v = 0; v < tree.verts.size(); v += 3) brute = std::fmax(brute, dir[0] * double(tree.verts[v]) + dir[1] * double(tree.verts[v + 1]) + dir[2] * double(tree.verts[v + 2]));
            const double full = ptw::MeshSupport(tree, root, dir, 1 << 30);
            CHECK(full == brute, "round %d: support %.17g, brute force %.17g", round, full, brute);
            for (int budget : {0, 1, 7, 64}) {
                const double cut = ptw::MeshSupport(tree, root, dir, budget);
                CHECK(cut >= brute, "round %d budget %d: support %.17g falls short of %.17g", round, budget, cut, brute);
            }
            searches += 5;
        }

        // instance box: rotation about a random axis, non-uniform scale, translation (column-major 4x4)
        float m[16] = {};
        {
            double ax[3] = {uni(gen), uni(gen), uni(gen) + 1e-3};
            const double len = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
            for (double &a : ax) a /= len;
            const double ang = 3.14159265358979 * uni(gen), c = std::cos(ang), s = std::sin(ang);
            const double R[3][3] = {{c + ax[0] * ax[0] * (1 - c), ax[0] * ax[1] * (1 - c) - ax[2] * s, ax[0] * ax[2] * (1 - c) + ax[1] * s},
                                    {ax[1] * ax[0] * (1 - c) + ax[2] * s, c + ax[1] * ax[1] * (1 - c), ax[1] * ax[2] * (1 - c) - ax[0] * s},
                                    {ax[2] * ax[0] * (1 - c) - ax[1] * s, ax[2] * ax[1] * (1 - c) + ax[0] * s, c + ax[2] * ax[2] * (1 - c)}};
            const double scale[3] = {0.2 + 2.0 * std::fabs(uni(gen)), 0.2 + 2.0 * std::fabs(uni(gen)), 0.2 + 2.0 * std::fabs(uni(gen))};
            for (int col = 0; col < 3; ++col)
                for (int row = 0; row < 3; ++row) m[4 * col + row] = float(R[row][col] * scale[col]);
            for (int row = 0; row < 3; ++row) m[12 + row] = 20.0f * uni(gen);
            m[15] = 1.0f;
        }
        // the reference's box: the world box of the transformed corners of the object-space box (bvh.cpp:273-280)
        const ptw::Box ob = tree.box(root);
        ptw::Box ref{{INFINITY, INFINITY, INFINITY}, {-INFINITY, -INFINITY, -INFINITY}};
        for (int corner = 0; corner < 8; ++corner) {
            const float p[3] = {corner & 1 ? ob.mx[0] : ob.mn[0], corner & 2 ? ob.mx[1] : ob.mn[1], corner & 4 ? ob.mx[2] : ob.mn[2]};
            for (int a = 0; a < 3; ++a) {
                const float w = m[a] * p[0] + m[4 + a] * p[1] + m[8 + a] * p[2] + m[12 + a];
                ref.mn[a] = std::fmin(ref.mn[a], w);
                ref.mx[a] = std::fmax(ref.mx[a], w);
            }
        }
        const ptw::Box tight = ptw::TightInstanceBox(tree, root, m, ref);
        double exact_lo[3] = {INFINITY, INFINITY, INFINITY}, exact_hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (size_t v = 0; v < tree.verts.size(); v += 3)
            for (int a = 0; a < 3; ++a) {
                const double w = double(m[a]) * tree.verts[v] + double(m[4 + a]) * tree.verts[v + 1] + double(m[8 + a]) * tree.verts[v + 2] + double(m[12 + a]);
                exact_lo[a] = std::fmin(exact_lo[a], w);
                exact_hi[a] = std::fmax(exact_hi[a], w);
            }
        for (int a = 0; a < 3; ++a) {
            CHECK(tight.mn[a] >= ref.mn[a] && tight.mx[a] <= ref.mx[a], "round %d axis %d: box leaves the reference's", round, a);
            // every vertex inside, unless the reference's own (fp32) box already cuts that close
            CHECK(double(tight.mn[a]) <= exact_lo[a] || tight.mn[a] == ref.mn[a], "round %d axis %d: low face %.9g above the mesh's %.17g", round, a, tight.mn[a], exact_lo[a]);
            CHECK(double(tight.mx[a]) >= exact_hi[a] || tight.mx[a] == ref.mx[a], "round %d axis %d: high face %.9g below the mesh's %.17g", round, a, tight.mx[a], exact_hi[a]);
            const double slack = 3e-5 * std::fmax(std::fmax(std::fabs(exact_lo[a]), std::fabs(exact_hi[a])), exact_hi[a] - exact_lo[a]);
            if (tris <= 1000) {  // within the default budget the faces sit on the mesh, up to the rounding margin
                CHECK(double(tight.mn[a]) >= exact_lo[a] - slack, "round %d axis %d: low face %.9g far below the mesh's %.17g", round, a, tight.mn[a], exact_lo[a]);
                CHECK(double(tight.mx[a]) <= exact_hi[a] + slack, "round %d axis %d: high face %.9g far above the mesh's %.17g", round, a, tight.mx[a], exact_hi[a]);
            }
        }
        // a transform row that is not finite: that axis keeps the reference's faces, the box stays inside the reference's
        float bad[16];
        std::memcpy(bad, m, sizeof(bad));
        const int bad_axis = int(gen() % 3u);
        bad[4 * int(gen() % 3u) + bad_axis] = (gen() & 1u) ? NAN : INFINITY;
        const ptw::Box kept = ptw::TightInstanceBox(tree, root, bad, ref);
        CHECK(kept.mn[bad_axis] == ref.mn[bad_axis] && kept.mx[bad_axis] == ref.mx[bad_axis], "round %d: axis %d of a transform that is not finite left the reference's faces", round, bad_axis);
        for (int a = 0; a < 3; ++a) CHECK(kept.mn[a] >= ref.mn[a] && kept.mx[a] <= ref.mx[a] && kept.mn[a] <= kept.mx[a], "round %d axis %d: box of a damaged transform leaves the reference's", round, a);
    }
    std::printf("%d meshes, %ld support searches, %d failures\n", rounds, searches, failures);
    return failures == 0 ? 0 : 1;
}
