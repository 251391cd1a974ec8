// Test infrastructure (never linked into the product): a driver over the few files of the reference that
// compile on their own in the build container - src/math/hash.h, src/utils/stack.h (both only need
// src/utils/cl_happy.h), third_party/tinyobjloader and third_party/stb - used UNMODIFIED from where
// they lie under /root/reference (see oracle/Makefile, target `ref`).  It writes known-answer vectors
// that tests/golden/make_ref_fixtures.py packs into tests/golden/ref_pins.npz, against which the
// oracle's restatements (and the loaders) are pinned.  Everything that needs glm, CUDA or the absent
// submodules stays unpinned.
//
//   ref_probe hash <n>                 n PCG_Hash(input) pairs, inputs from a fixed LCG plus edge values
//   ref_probe stack <script...>        Stack<uint32_t, 64>: v > 0 pushes v, 0 pops; prints pops and the final top
//   ref_probe obj <file.obj>           tinyobj::ObjReader on the file: attrib arrays and per-corner indices
//   ref_probe image <file> <out.raw>   stbi_load(file, 3 channels): "w h channels_in_file", pixels to out.raw
//   ref_probe png <w> <h> <out.png>    stbi_write_png of a fixed test pattern (RGB)
//   ref_probe constants                bit patterns of the fp32 constants of src/math/constants.h
#include <cstdint>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "utils/cl_happy.h"
#include "math/hash.h"
#include "math/constants.h"
#include "utils/stack.h"
#include "tiny_obj_loader.h"
#include "stb_image.h"
#include "stb_image_write.h"

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    const std::string cmd = argv[1];
    if (cmd == "constants") {
        const float v[] = {Math::SMALL_EPSILON, Math::EPSILON, Math::ONE_MINUS_EPSILON, Math::PI, Math::HALF_PI, Math::TWO_PI, Math::INV_PI,
                           Math::SQRT3, Math::INV_SQRT3};
        for (float f : v) {
            uint32_t bits;
            memcpy(&bits, &f, 4);
            printf("%u\n", bits);
        }
        return 0;
    }
    if (cmd == "hash") {
        const int n = atoi(argv[2]);
        uint32_t x = 0x12345u;
        const uint32_t edge[] = {0u, 1u, 0xFFFFFFFFu, 0x80000000u, 0x7FFFFFFFu, 0xdeadbeefu, 0x9E3779B9u};
        for (int i = 0; i < n; ++i) {
            const uint32_t in = i < 7 ? edge[i] : x;
            x = x * 1664525u + 1013904223u;
            printf("%u %u\n", in, Math::PCG_Hash(in));
        }
        return 0;
    }
    if (cmd == "stack") {
        Stack<uint32_t, 64> s;
        for (int i = 2; i < argc; ++i) {
            const uint32_t v = uint32_t(strtoul(argv[i], nullptr, 10));
            if (v > 0) s.Push(v);
            else {
                const uint32_t popped = s.Pop();
                printf("%u %d\n", popped, int(s.Empty()));
            }
        }
        printf("top %u\n", s.top);
        return 0;
    }
    if (cmd == "obj") {
        tinyobj::ObjReaderConfig config;
        tinyobj::ObjReader reader;
        if (!reader.ParseFromFile(argv[2], config)) return 3;
        const auto &a = reader.GetAttrib();
        printf("vertices %zu", a.vertices.size());
        for (float v : a.vertices) printf(" %.9g", v);
        printf("\nnormals %zu", a.normals.size());
        for (float v : a.normals) printf(" %.9g", v);
        printf("\ntexcoords %zu", a.texcoords.size());
        for (float v : a.texcoords) printf(" %.9g", v);
        printf("\n");
        for (const auto &shape : reader.GetShapes()) {
            printf("shape %zu", shape.mesh.num_face_vertices.size());
            for (auto c : shape.mesh.num_face_vertices) printf(" %d", int(c));
            printf("\nindices %zu", shape.mesh.indices.size());
            for (const auto &i : shape.mesh.indices) printf(" %d %d %d", i.vertex_index, i.normal_index, i.texcoord_index);
            printf("\n");
        }
        return 0;
    }
    if (cmd == "image") {
        int w = 0, h = 0, c = 0;
        unsigned char *px = stbi_load(argv[2], &w, &h, &c, 3);  // image.cpp:79
        if (!px) return 4;
        FILE *f = fopen(argv[3], "wb");
        fwrite(px, 1, size_t(w) * h * 3, f);
        fclose(f);
        stbi_image_free(px);
        printf("%d %d %d\n", w, h, c);
        return 0;
    }
    if (cmd == "png") {
        const int w = atoi(argv[2]), h = atoi(argv[3]);
        std::vector<unsigned char> px(size_t(w) * h * 3);
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x)
                for (int k = 0; k < 3; ++k) px[(size_t(y) * w + x) * 3 + k] = (unsigned char)((x * 7 + y * 13 + k * 101 + (x ^ y)) & 0xFF);
        return stbi_write_png(argv[4], w, h, 3, px.data(), w * 3) ? 0 : 5;  // image.cpp:123
    }
    return 2;
}
