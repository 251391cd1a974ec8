// extern "C" wrappers over the host facade (declared in include/ptamd_host.h) plus the
// headless image writers.  Exceptions stop here and become status codes.
#include <algorithm>
#include <cstdio>
#include <vector>
#include <cstring>
#include <string>

#include "../../include/ptamd_host.h"
#include "pt_host.h"

using namespace pth;

struct pth_scene {
    Scene scene;
};
struct pth_integrator {
    explicit pth_integrator(pt_ctx *ctx) : impl(ctx) {}
    PathIntegrator impl;
};
struct pth_bvh {
    TLAS tlas;
};

namespace {
thread_local std::string g_error;
template <typename F>
int Guard(F &&f) {
    try {
        f();
        return PT_OK;
    } catch (const Exception &e) {
        g_error = e.what();
        return PT_ERR_INVALID;
    } catch (const std::exception &e) {
        g_error = e.what();
        return PT_ERR_DEVICE;
    }
}
int Fail(const char *msg) {
    g_error = msg;
    return PT_ERR_INVALID;
}
Mat4 Load(const float m[16]) {
    Mat4 r;
    std::memcpy(r.m, m, sizeof(r.m));
    return r;
}
Vec3 V(const float v[3]) { return {v[0], v[1], v[2]}; }
}  // namespace

extern "C" {

const char *pth_last_error(void) { return g_error.c_str(); }

int pth_transform_matrix(const float t[3], const float r[3], const float s[3], float out[16]) {
    if (!t || !r || !s || !out) return Fail("pth_transform_matrix: null argument");
    const Mat4 M = GetTransformMatrix(V(t), V(r), V(s));
    std::memcpy(out, M.m, sizeof(M.m));
    return PT_OK;
}
int pth_matrix_derive(const float transform[16], float inv_transform[16], float inv_transpose[16]) {
    if (!transform || !inv_transform || !inv_transpose) return Fail("pth_matrix_derive: null argument");
    const Mat4 T = Load(transform), I = Inverse(T), IT = InverseTranspose(T);
    std::memcpy(inv_transform, I.m, sizeof(I.m));
    std::memcpy(inv_transpose, IT.m, sizeof(IT.m));
    return PT_OK;
}
int pth_matrix_mul(const float a[16], const float b[16], float out[16]) {
    if (!a || !b || !out) return Fail("pth_matrix_mul: null argument");
    const Mat4 C = Mul(Load(a), Load(b));
    std::memcpy(out, C.m, sizeof(C.m));
    return PT_OK;
}
int pth_gltf_local_matrix(const float t[3], const float q[4], const float s[3], const float matrix[16], float out[16]) {
    if (!t || !q || !s || !matrix || !out) return Fail("pth_gltf_local_matrix: null argument");
    const Mat4 I = Mat4::Identity();
    const Mat4 R = Mul(Mul(Mul(Translate(I, V(t)), FromQuat(q[0], q[1], q[2], q[3])), Scale(I, V(s))), Load(matrix));
    std::memcpy(out, R.m, sizeof(R.m));
    return PT_OK;
}

int pth_alias_build(const float *weights, uint32_t n, float *prob, uint32_t *alias) {
    if (!weights || !prob || !alias || n == 0) return Fail("pth_alias_build: empty or null input");
    return Guard([&] {
        DiscreteSampler1D s;
        s.UpdateWeights(std::vector<float>(weights, weights + n));
        std::memcpy(prob, s.ProbabilityTable().data(), n * sizeof(float));
        std::memcpy(alias, s.AliasTable().data(), n * sizeof(uint32_t));
    });
}

int pth_camera_make(const float eye[3], const float look_at[3], const float up[3], float fovy_deg, int width, int height,
                    pt_camera *out) {
    if (!eye || !look_at || !up || !out || width <= 0 || height <= 0) return Fail("pth_camera_make: bad argument");
    Camera c;
    c.GetFilm().width = width;
    c.GetFilm().height = height;
    c.SetDefaults(V(eye), V(look_at), V(up), fovy_deg);
    c.ResetDefaults();
    std::memset(out, 0, sizeof(*out));
    c.UpdateDevice(*out);
    return PT_OK;
}

int pth_bvh_build(const float *positions, const uint32_t *indices, const pt_geometry_instance *instances, uint32_t n, pth_bvh **out) {
    if (!positions || !indices || !instances || !out) return Fail("pth_bvh_build: null argument");
    auto *b = new pth_bvh();
    const int st = Guard([&] { b->tlas.Build(positions, indices, std::vector<pt_geometry_instance>(instances, instances + n)); });
    if (st != PT_OK) {
        delete b;
        return st;
    }
    *out = b;
    return PT_OK;
}
int pth_bvh_sizes(const pth_bvh *b, uint32_t *tn, uint32_t *bl, uint32_t *bn, uint32_t *ti) {
    if (!b) return Fail("pth_bvh_sizes: null handle");
    *tn = uint32_t(b->tlas.TlasNodes().size());
    *bl = uint32_t(b->tlas.DeviceBlases().size());
    *bn = uint32_t(b->tlas.BlasNodes().size());
    *ti = uint32_t(b->tlas.TriIndices().size());
    return PT_OK;
}
int pth_bvh_get(const pth_bvh *b, pt_tlas_node *tn, pt_blas *bl, pt_bvh_node *bn, uint32_t *ti) {
    if (!b) return Fail("pth_bvh_get: null handle");
    std::memcpy(tn, b->tlas.TlasNodes().data(), b->tlas.TlasNodes().size() * sizeof(pt_tlas_node));
    std::memcpy(bl, b->tlas.DeviceBlases().data(), b->tlas.DeviceBlases().size() * sizeof(pt_blas));
    std::memcpy(bn, b->tlas.BlasNodes().data(), b->tlas.BlasNodes().size() * sizeof(pt_bvh_node));
    std::memcpy(ti, b->tlas.TriIndices().data(), b->tlas.TriIndices().size() * sizeof(uint32_t));
    return PT_OK;
}
void pth_bvh_free(pth_bvh *b) { delete b; }

int pth_scene_create(pth_scene **out) {
    if (!out) return Fail("pth_scene_create: null argument");
    return Guard([&] {
        auto *s = new pth_scene();
        s->scene.InitDefaults();
        *out = s;
    });
}
void pth_scene_destroy(pth_scene *s) { delete s; }
int pth_scene_set_integrator(pth_scene *s, int max_depth, int max_iterations) {
    if (!s || max_depth < 1) return Fail("pth_scene_set_integrator: bad argument");
    s->scene.SetIntegrator(max_depth, max_iterations);
    return PT_OK;
}
int pth_scene_load(const char *path, pth_image_decoder decode, void *user, int width, int height, int max_depth, pth_scene **out) {
    if (!path || !out) return Fail("pth_scene_load: null argument");
    *out = nullptr;
    pth_scene *s = new pth_scene();
    const int st = Guard([&] {
        s->scene.InitDefaults();
        LoadSceneFile(s->scene, path, decode, user, width, height, max_depth);
    });
    if (st != PT_OK) {
        delete s;
        return st;
    }
    *out = s;
    return PT_OK;
}
int pth_scene_set_camera(pth_scene *s, const float eye[3], const float look_at[3], const float up[3], float fovy_deg, int w, int h) {
    if (!s || !eye || !look_at || !up || w <= 0 || h <= 0) return Fail("pth_scene_set_camera: bad argument");
    Camera &c = s->scene.GetCamera();
    c.GetFilm().width = w;
    c.GetFilm().height = h;
    c.SetDefaults(V(eye), V(look_at), V(up), fovy_deg);
    c.ResetDefaults();
    return PT_OK;
}
int pth_scene_add_mesh(pth_scene *s, const float *p, const float *n, const float *uv, uint32_t vc, const uint32_t *idx,
                       uint32_t ic, uint32_t *first_index) {
    if (!s || !first_index) return Fail("pth_scene_add_mesh: null argument");
    return Guard([&] { *first_index = s->scene.AddMesh(p, n, uv, vc, idx, ic).first_index; });
}
int pth_scene_add_material(pth_scene *s, const pt_material *m, int *id) {
    if (!s || !m || !id) return Fail("pth_scene_add_material: null argument");
    return Guard([&] { *id = s->scene.AddMaterial(*m); });
}
int pth_scene_add_texture1(pth_scene *s, const pt_texture1 *t, int *id) {
    if (!s || !t || !id) return Fail("pth_scene_add_texture1: null argument");
    return Guard([&] { *id = s->scene.AddTexture1(*t); });
}
int pth_scene_add_texture3(pth_scene *s, const pt_texture3 *t, int *id) {
    if (!s || !t || !id) return Fail("pth_scene_add_texture3: null argument");
    return Guard([&] { *id = s->scene.AddTexture3(*t); });
}
int pth_scene_add_image(pth_scene *s, const float *texels, int w, int h, int ch, int srgb, int *id) {
    if (!s || !id) return Fail("pth_scene_add_image: null argument");
    return Guard([&] { *id = s->scene.AddImage(texels, w, h, ch, srgb); });
}
int pth_scene_new_blas_index(pth_scene *s, int *blas_index) {
    if (!s || !blas_index) return Fail("pth_scene_new_blas_index: null argument");
    *blas_index = int(s->scene.NewBlasIndex());
    return PT_OK;
}
int pth_scene_add_instance(pth_scene *s, uint32_t first_index, uint32_t index_count, uint32_t vertex_count, int material_id,
                           const float transform[16], int blas_index, int *instance_id) {
    if (!s || !transform || !instance_id) return Fail("pth_scene_add_instance: null argument");
    return Guard([&] {
        TriangleMesh mesh{first_index, index_count, vertex_count};
        *instance_id = s->scene.AddInstance(mesh, material_id, Load(transform), blas_index);
    });
}
int pth_scene_finalize(pth_scene *s) {
    if (!s) return Fail("pth_scene_finalize: null scene");
    return Guard([&] { s->scene.Finalize(); });
}
int pth_scene_finalize_device(pth_scene *s, pt_ctx *ctx) {
    if (!s || !ctx) return Fail("pth_scene_finalize_device: null argument");
    return Guard([&] { s->scene.Finalize(ctx); });
}
int pth_scene_get_desc(pth_scene *s, pt_scene_desc *out) {
    if (!s || !out) return Fail("pth_scene_get_desc: null argument");
    return Guard([&] { *out = s->scene.GetDeviceScene(); });
}
int pth_scene_update_device(pth_scene *s, pt_ctx *ctx) {
    if (!s || !ctx) return Fail("pth_scene_update_device: null argument");
    return Guard([&] { s->scene.UpdateDevice(ctx); });
}

int pth_integrator_create(pt_ctx *ctx, pth_integrator **out) {
    if (!ctx || !out) return Fail("pth_integrator_create: null argument");
    *out = new pth_integrator(ctx);
    return PT_OK;
}
void pth_integrator_destroy(pth_integrator *i) { delete i; }
int pth_integrator_init(pth_integrator *i, int w, int h) {
    if (!i) return Fail("pth_integrator_init: null integrator");
    return Guard([&] { i->impl.Init(w, h); });
}
int pth_integrator_resize(pth_integrator *i, int w, int h) {
    if (!i) return Fail("pth_integrator_resize: null integrator");
    return Guard([&] { i->impl.Resize(w, h); });
}
int pth_integrator_reset(pth_integrator *i, int w, int h) {
    if (!i) return Fail("pth_integrator_reset: null integrator");
    return Guard([&] { i->impl.Reset(w, h); });
}
int pth_integrator_render(pth_integrator *i, int w, int h, pth_scene *s, uint8_t *rgba_device, int iteration) {
    if (!i || !s) return Fail("pth_integrator_render: null argument");
    return Guard([&] { i->impl.Render(w, h, s->scene, rgba_device, iteration, IntegratorState{}); });
}
int pth_debug_render(pt_ctx *ctx, int w, int h, pth_scene *s, const pt_integrator_state *state, uint8_t *rgba_device) {
    if (!ctx || !s || !state) return Fail("pth_debug_render: null argument");
    return Guard([&] {
        IntegratorState st;
        st.visualize_bvh = state->visualize_bvh != 0;
        st.bvh_visualize_type = BVH_VisualizeMode(state->bvh_visualize_type);
        st.bvh_visualize_depth = state->bvh_visualize_depth;
        st.debug_mode = DebugVisualizeMode(state->debug_mode);
        DebugIntegrator(ctx).Render(w, h, s->scene, rgba_device, 0, st);
    });
}
int pth_render_offscreen(pt_ctx *ctx, pth_scene *s, const pth_offscreen_options *options, pth_offscreen_result *result) {
    if (!ctx || !s || !options || !result) return Fail("pth_render_offscreen: null argument");
    return Guard([&] { RenderOffscreen(ctx, s->scene, *options, *result); });
}
int pth_checkpoint_read(const char *path, int *width, int *height, uint32_t *iterations, float *film_rgb) {
    if (!path || !width || !height || !iterations) return Fail("pth_checkpoint_read: null argument");
    return Guard([&] {
        std::vector<float> film;
        if (!ReadCheckpoint(path, *width, *height, *iterations, film_rgb ? &film : nullptr)) throw Exception(std::string("no checkpoint at ") + path);
        if (film_rgb) std::memcpy(film_rgb, film.data(), film.size() * sizeof(float));
    });
}

// ------------------------------------------------------------ headless output
static uint32_t Crc32(const uint8_t *data, size_t n, uint32_t crc) {
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1);
            table[i] = c;
        }
        init = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ data[i]) & 0xFF] ^ (crc >> 8);
    return ~crc;
}
static void PutBE32(std::vector<uint8_t> &v, uint32_t x) {
    v.push_back(uint8_t(x >> 24));
    v.push_back(uint8_t(x >> 16));
    v.push_back(uint8_t(x >> 8));
    v.push_back(uint8_t(x));
}
static void Chunk(std::vector<uint8_t> &png, const char tag[4], const std::vector<uint8_t> &payload) {
    PutBE32(png, uint32_t(payload.size()));
    const size_t start = png.size();
    png.insert(png.end(), tag, tag + 4);
    png.insert(png.end(), payload.begin(), payload.end());
    PutBE32(png, Crc32(png.data() + start, png.size() - start, 0));
}

int pth_save_png(const char *path, const uint8_t *rgba, int width, int height) {
    if (!path || !rgba || width <= 0 || height <= 0) return Fail("pth_save_png: bad argument");
    // scanlines: filter byte 0 + RGB
    std::vector<uint8_t> raw;
    raw.reserve(size_t(height) * (1 + 3 * size_t(width)));
    for (int y = 0; y < height; ++y) {
        raw.push_back(0);
        for (int x = 0; x < width; ++x) {
            const uint8_t *p = rgba + 4 * (size_t(y) * width + x);
            raw.push_back(p[0]);
            raw.push_back(p[1]);
            raw.push_back(p[2]);
        }
    }
    // zlib stream of stored (uncompressed) deflate blocks
    std::vector<uint8_t> z{0x78, 0x01};
    uint32_t a = 1, b = 0;
    for (uint8_t c : raw) {
        a = (a + c) % 65521u;
        b = (b + a) % 65521u;
    }
    size_t pos = 0;
    while (pos < raw.size() || raw.empty()) {
        const size_t n = std::min<size_t>(65535, raw.size() - pos);
        const bool last = pos + n >= raw.size();
        z.push_back(last ? 1 : 0);
        z.push_back(uint8_t(n & 0xFF));
        z.push_back(uint8_t(n >> 8));
        z.push_back(uint8_t(~n & 0xFF));
        z.push_back(uint8_t((~n >> 8) & 0xFF));
        z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
        pos += n;
        if (last) break;
    }
    PutBE32(z, (b << 16) | a);
    std::vector<uint8_t> png{0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    PutBE32(ihdr, uint32_t(width));
    PutBE32(ihdr, uint32_t(height));
    const uint8_t tail[5] = {8, 2, 0, 0, 0};  // 8-bit, truecolour
    ihdr.insert(ihdr.end(), tail, tail + 5);
    Chunk(png, "IHDR", ihdr);
    Chunk(png, "IDAT", z);
    Chunk(png, "IEND", {});
    FILE *f = std::fopen(path, "wb");
    if (!f) return Fail("pth_save_png: cannot open output file");
    const size_t w = std::fwrite(png.data(), 1, png.size(), f);
    std::fclose(f);
    return w == png.size() ? PT_OK : Fail("pth_save_png: short write");
}

int pth_save_hdr(const char *path, const float *film, int width, int height, int iterations) {
    if (!path || !film || width <= 0 || height <= 0 || iterations <= 0) return Fail("pth_save_hdr: bad argument");
    FILE *f = std::fopen(path, "wb");
    if (!f) return Fail("pth_save_hdr: cannot open output file");
    std::fprintf(f, "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n", height, width);
    std::vector<uint8_t> row(4 * size_t(width));
    const float inv = 1.0f / float(iterations);
    for (int y = 0; y < height; ++y) {
        for (int x = 0; x < width; ++x) {
            const float *p = film + 3 * (size_t(y) * width + x);
            const float r = p[0] * inv, g = p[1] * inv, b = p[2] * inv;
            const float m = std::max(r, std::max(g, b));
            uint8_t *o = &row[4 * size_t(x)];
            if (!(m > 1e-32f)) {
                o[0] = o[1] = o[2] = o[3] = 0;
            } else {
                int e;
                const float scale = std::frexp(m, &e) * 256.0f / m;
                o[0] = uint8_t(std::max(0.0f, r * scale));
                o[1] = uint8_t(std::max(0.0f, g * scale));
                o[2] = uint8_t(std::max(0.0f, b * scale));
                o[3] = uint8_t(e + 128);
            }
        }
        std::fwrite(row.data(), 1, row.size(), f);
    }
    std::fclose(f);
    return PT_OK;
}

}  // extern "C"
