// Host scene model of the facade: alias tables, camera basis, light tables and the
// assembly of the host-pointer scene description handed to pt_scene_upload.
#include <algorithm>
#include <climits>
#include <cstring>
#include <numeric>

#include "pt_host.h"

namespace pth {

// ------------------------------------------------------------ DiscreteSampler1D
// Alias method as discrete_sampler.cpp:14-76: scaled probabilities n*w/sum; repeatedly pair
// the largest over-full entry with the smallest under-full one; entries within 1e-5 of 1
// are snapped to 1; leftovers are clamped to 1.  The two heaps are kept with
// std::push_heap / std::pop_heap, the same primitive std::priority_queue is built on, so
// ties resolve like the reference's queues under the same standard library.
void DiscreteSampler1D::UpdateWeights(const std::vector<float> &weights) {
    const uint32_t n = uint32_t(weights.size());
    const float sum = std::accumulate(weights.begin(), weights.end(), 0.0f);
    if (sum == 0.0f) throw Exception("Cannot use discrete sampler if its CDF is 0!");
    constexpr float kEps = 1e-5f;
    m_probability_table.assign(n, 0.0f);
    m_alias_table.assign(n, UINT_MAX);
    std::vector<float> &p = m_probability_table;
    for (uint32_t i = 0; i < n; ++i) p[i] = n * weights[i] / sum;

    std::vector<uint32_t> over, under;
    auto over_less = [&p](uint32_t a, uint32_t b) { return p[a] < p[b]; };    // max-heap on p
    auto under_less = [&p](uint32_t a, uint32_t b) { return p[a] > p[b]; };   // min-heap on p
    auto classify = [&](uint32_t i) {
        if (p[i] > 1.0f + kEps) {
            over.push_back(i);
            std::push_heap(over.begin(), over.end(), over_less);
        } else if (p[i] < 1.0f - kEps) {
            under.push_back(i);
            std::push_heap(under.begin(), under.end(), under_less);
        } else {
            p[i] = 1.0f;
        }
    };
    for (uint32_t i = 0; i < n; ++i) classify(i);
    while (!over.empty() && !under.empty()) {
        std::pop_heap(over.begin(), over.end(), over_less);
        const uint32_t big = over.back();
        over.pop_back();
        std::pop_heap(under.begin(), under.end(), under_less);
        const uint32_t small = under.back();
        under.pop_back();
        m_alias_table[small] = big;
        p[big] -= 1.0f - p[small];
        classify(big);
    }
    for (uint32_t i : over) p[i] = 1.0f;
    for (uint32_t i : under) p[i] = 1.0f;
}

void DiscreteSampler1D::UpdateDevice(pt_alias_table &sampler) const {
    sampler.alias_table = m_alias_table.data();
    sampler.probability_table = m_probability_table.data();
    sampler.size = uint32_t(m_alias_table.size());
    sampler._pad = 0;
}

// ----------------------------------------------------------------------- Camera
void Camera::SetDefaults(Vec3 position, Vec3 look_at, Vec3 up, float fovy_deg) {
    m_default_params.position = position;
    m_default_params.look_at = look_at;
    m_default_params.up = up;
    m_default_params.fovy = fovy_deg;
}

void Camera::Update() {  // camera.cpp:6-20
    m_front = Normalize(m_look_at - m_position);
    m_right = Normalize(Cross(m_front, m_up));
    m_up = Normalize(Cross(m_right, m_front));
    const float theta = Radians(m_fovy);
    const float half_height = std::tan(theta * 0.5f);
    const float aspect = float(m_film.width) / float(m_film.height);
    const float half_width = aspect * half_height;
    m_pixel_length[0] = (2.0f * half_width) / float(m_film.width);
    m_pixel_length[1] = (2.0f * half_height) / float(m_film.height);
    m_dirty = true;
}

void Camera::ResetDefaults() {  // camera.cpp:22-29
    m_look_at = m_default_params.look_at;
    m_fovy = m_default_params.fovy;
    m_position = m_default_params.position;
    m_up = m_default_params.up;
    Update();
}

void Camera::SetResolution(int width, int height) {  // camera.cpp:31-34
    m_film.width = width;
    m_film.height = height;
    Update();
}

void Camera::UpdateDevice(pt_camera &c) {  // camera.cpp:36-51
    c.resolution[0] = m_film.width;
    c.resolution[1] = m_film.height;
    const Vec3 *src[4] = {&m_position, &m_front, &m_up, &m_right};
    float *dst[4] = {c.position, c.front, c.up, c.right};
    for (int k = 0; k < 4; ++k) {
        dst[k][0] = src[k]->x;
        dst[k][1] = src[k]->y;
        dst[k][2] = src[k]->z;
    }
    c.fovy = m_fovy;
    c.pixel_length[0] = m_pixel_length[0];
    c.pixel_length[1] = m_pixel_length[1];
    c.focal_length = 1.0f;
    c.near_plane = near_plane;
    c.far_plane = far_plane;
    m_dirty = false;
}

// ------------------------------------------------------------------------ Scene
Scene::Scene() {
    std::memset(&m_desc, 0, sizeof(m_desc));
    std::memset(&m_desc_light_sampler, 0, sizeof(m_desc_light_sampler));
    m_bvh = std::make_unique<TLAS>();
}
Scene::~Scene() = default;

void Scene::InitDefaults() {
    pt_texture1 t1;
    std::memset(&t1, 0, sizeof(t1));
    t1.type = PT_TEX_CONSTANT;
    t1.u.constant.value = 0.98f;
    m_texture_pool.textures1.push_back(t1);
    pt_texture3 t3;
    std::memset(&t3, 0, sizeof(t3));
    t3.type = PT_TEX_CONSTANT;
    t3.u.constant.value[0] = t3.u.constant.value[1] = t3.u.constant.value[2] = 0.98f;
    m_texture_pool.textures3.push_back(t3);
    pt_material m;
    std::memset(&m, 0, sizeof(m));
    m.type = PT_MAT_LAMBERTIAN;
    m.tex0 = 0;
    m_materials.push_back(m);
}

void Scene::SetIntegrator(int max_depth, int max_iterations) {
    m_max_depth = max_depth;
    m_max_iterations = max_iterations;
    m_dirty = true;
}

TriangleMesh Scene::AddMesh(const float *positions, const float *normals, const float *uvs, uint32_t vertex_count,
                            const uint32_t *indices, uint32_t index_count) {
    if (!positions || !normals || !uvs || !indices) throw Exception("Scene::AddMesh: null array");
    if (index_count % 3 != 0) throw Exception("Scene::AddMesh: index count is not a multiple of 3");
    const uint32_t base_vertex = uint32_t(m_positions.size() / 3);
    TriangleMesh mesh;
    mesh.first_index = uint32_t(m_indices.size());
    mesh.index_count = index_count;
    mesh.vertex_count = vertex_count;
    m_indices.reserve(m_indices.size() + index_count);
    for (uint32_t i = 0; i < index_count; ++i) {
        if (indices[i] >= vertex_count) throw Exception("Scene::AddMesh: index out of range");
        m_indices.push_back(base_vertex + indices[i]);
    }
    m_positions.insert(m_positions.end(), positions, positions + 3 * size_t(vertex_count));
    m_normals.insert(m_normals.end(), normals, normals + 3 * size_t(vertex_count));
    m_texcoords.insert(m_texcoords.end(), uvs, uvs + 2 * size_t(vertex_count));
    m_dirty = true;
    return mesh;
}

int Scene::AddMaterial(const pt_material &material) {
    m_materials.push_back(material);
    m_dirty = true;
    return int(m_materials.size()) - 1;
}
int Scene::AddTexture1(const pt_texture1 &t) {
    m_texture_pool.textures1.push_back(t);
    m_texture_pool.dirty = true;
    return int(m_texture_pool.textures1.size()) - 1;
}
int Scene::AddTexture3(const pt_texture3 &t) {
    m_texture_pool.textures3.push_back(t);
    m_texture_pool.dirty = true;
    return int(m_texture_pool.textures3.size()) - 1;
}
int Scene::AddImage(const float *texels, int width, int height, int channels, int srgb) {
    if (!texels || width <= 0 || height <= 0 || (channels != 1 && channels != 3))
        throw Exception("Scene::AddImage: images are fp32 with 1 or 3 channels");
    TexturePool::HostImage img;
    img.width = width;
    img.height = height;
    img.channels = channels;
    img.srgb = srgb;
    img.texels.assign(texels, texels + size_t(width) * height * channels);
    auto &pool = channels == 1 ? m_texture_pool.images1 : m_texture_pool.images3;
    pool.push_back(std::move(img));
    m_texture_pool.dirty = true;
    return int(pool.size()) - 1;
}

int Scene::AddInstance(const TriangleMesh &mesh, int material_id, const Mat4 &transform, int blas_index) {
    if (material_id < 0 || size_t(material_id) >= m_materials.size()) throw Exception("Scene::AddInstance: unknown material");
    pt_geometry_instance g;
    std::memset(&g, 0, sizeof(g));
    g.material_id = material_id;
    g.first_index = mesh.first_index;
    g.index_count = mesh.index_count;
    g.vertex_count = mesh.vertex_count;
    const Mat4 inv = Inverse(transform), inv_t = InverseTranspose(transform);
    std::memcpy(g.transform, transform.m, sizeof(g.transform));
    std::memcpy(g.inv_transform, inv.m, sizeof(g.inv_transform));
    std::memcpy(g.inv_transpose, inv_t.m, sizeof(g.inv_transpose));
    g.blas_index = blas_index;
    g.total_area = 0.0f;
    g.triangle_sampler_index = -1;
    g.area_light_id = -1;
    m_geometries.push_back(g);
    m_dirty = true;
    return int(m_geometries.size()) - 1;
}

void Scene::ComputeLightData() {
    m_lights.clear();
    m_triangle_samplers.clear();
    std::vector<float> light_powers;
    float total_light_power = 0.0f;
    for (size_t i = 0; i < m_geometries.size(); ++i) {
        pt_geometry_instance &g = m_geometries[i];
        const pt_material &material = m_materials[g.material_id];
        if (material.type != PT_MAT_EMISSIVE) continue;
        Mat4 T;
        std::memcpy(T.m, g.transform, sizeof(T.m));
        float total_area = 0.0f;
        std::vector<float> tri_areas;
        tri_areas.reserve(g.index_count / 3);
        for (uint32_t j = 0; j < g.index_count; j += 3) {
            Vec3 v[3];
            for (int k = 0; k < 3; ++k) {
                const float *p = &m_positions[3 * size_t(m_indices[g.first_index + j + k])];
                v[k] = TransformPoint(T, {p[0], p[1], p[2]});
            }
            const float area = Length(Cross(v[1] - v[0], v[2] - v[0])) / 2.0f;  // math/geometry.h:26-30
            total_area += area;
            tri_areas.push_back(area);
        }
        const int light_index = int(m_lights.size());
        g.total_area = total_area;
        g.area_light_id = light_index;
        g.triangle_sampler_index = light_index;
        m_triangle_samplers.emplace_back();
        m_triangle_samplers.back().UpdateWeights(tri_areas);
        const float power = total_area * material.w1.f * 3.14159265358979323846f;
        m_lights.push_back(pt_area_light{power, int(i)});
        total_light_power += power;
        light_powers.push_back(power);
    }
    m_total_light_power = total_light_power;
    if (!light_powers.empty()) m_light_sampler.UpdateWeights(light_powers);
}

void Scene::Finalize() {
    if (m_geometries.empty()) throw Exception("Scene::Finalize: the scene has no objects");
    m_bvh->Build(*this, m_geometries);
    ComputeLightData();
    m_dirty = true;
    m_texture_pool.dirty = true;
    m_bvh->Dirty() = true;
    m_camera.Dirty() = true;
}

void Scene::Finalize(pt_ctx *ctx) {
    if (!ctx) throw Exception("Scene::Finalize: null context");
    if (m_geometries.empty()) throw Exception("Scene::Finalize: the scene has no objects");
    m_bvh->BuildOnDevice(ctx, m_positions.data(), uint32_t(m_positions.size() / 3), m_indices.data(), uint32_t(m_indices.size()), m_geometries);
    ComputeLightData();
    m_dirty = true;
    m_texture_pool.dirty = true;
    m_bvh->Dirty() = true;
    m_camera.Dirty() = true;
}

void Scene::FillDesc() {
    pt_scene_desc &d = m_desc;
    d.max_depth = uint32_t(m_max_depth);
    d.vertex_count = uint32_t(m_positions.size() / 3);
    d.positions = m_positions.data();
    d.normals = m_normals.data();
    d.uvs = m_texcoords.data();
    d.indices = m_indices.data();
    d.index_count = uint32_t(m_indices.size());
    d.geometries = m_geometries.data();
    d.geometry_count = uint32_t(m_geometries.size());
    m_desc_tri_samplers.resize(m_triangle_samplers.size());
    for (size_t i = 0; i < m_triangle_samplers.size(); ++i) m_triangle_samplers[i].UpdateDevice(m_desc_tri_samplers[i]);
    d.triangle_samplers = m_desc_tri_samplers.data();
    d.triangle_sampler_count = uint32_t(m_desc_tri_samplers.size());
    d.materials = m_materials.data();
    d.material_count = uint32_t(m_materials.size());
    d.lights = m_lights.data();
    d.light_count = uint32_t(m_lights.size());
    d.total_light_power = m_total_light_power;
    if (!m_lights.empty()) {
        m_light_sampler.UpdateDevice(m_desc_light_sampler);
        d.light_sampler = &m_desc_light_sampler;
    } else {
        d.light_sampler = nullptr;
    }
    d.textures1 = m_texture_pool.textures1.data();
    d.texture1_count = uint32_t(m_texture_pool.textures1.size());
    d.textures3 = m_texture_pool.textures3.data();
    d.texture3_count = uint32_t(m_texture_pool.textures3.size());
    auto fill_images = [](const std::vector<TexturePool::HostImage> &src, std::vector<pt_image> &dst) {
        dst.resize(src.size());
        for (size_t i = 0; i < src.size(); ++i)
            dst[i] = pt_image{src[i].texels.data(), src[i].width, src[i].height, src[i].channels, src[i].srgb};
    };
    fill_images(m_texture_pool.images1, m_desc_images1);
    fill_images(m_texture_pool.images3, m_desc_images3);
    d.images1 = m_desc_images1.data();
    d.image1_count = uint32_t(m_desc_images1.size());
    d.images3 = m_desc_images3.data();
    d.image3_count = uint32_t(m_desc_images3.size());
    m_bvh->UpdateDevice(d);
    m_camera.UpdateDevice(d.camera);
}

const pt_scene_desc &Scene::GetDeviceScene() {
    FillDesc();
    return m_desc;
}

bool Scene::UpdateDevice(pt_ctx *ctx) {
    const bool geometry_dirty = m_dirty || m_texture_pool.dirty || m_bvh->Dirty();
    const bool camera_dirty = m_camera.Dirty();
    if (geometry_dirty) {
        FillDesc();
        if (pt_scene_upload(ctx, &m_desc) != PT_OK) throw Exception(std::string("pt_scene_upload: ") + pt_last_error(ctx));
        m_dirty = false;
        m_texture_pool.dirty = false;
    } else if (camera_dirty) {
        m_camera.UpdateDevice(m_desc.camera);
        if (pt_camera_update(ctx, &m_desc.camera) != PT_OK) throw Exception(std::string("pt_camera_update: ") + pt_last_error(ctx));
    }
    return geometry_dirty || camera_dirty;
}

// --------------------------------------------------------------- PathIntegrator
static void Check(pt_ctx *ctx, int status, const char *what) {
    if (status != PT_OK) throw Exception(std::string(what) + ": " + pt_last_error(ctx));
}
void PathIntegrator::Init(int w, int h) { Check(m_ctx, pt_integrator_init(m_ctx, w, h), "pt_integrator_init"); }
void PathIntegrator::Destroy() { Check(m_ctx, pt_integrator_destroy(m_ctx), "pt_integrator_destroy"); }
void PathIntegrator::Resize(int w, int h) { Check(m_ctx, pt_integrator_resize(m_ctx, w, h), "pt_integrator_resize"); }
void PathIntegrator::Reset(int w, int h) { Check(m_ctx, pt_integrator_reset(m_ctx, w, h), "pt_integrator_reset"); }
void PathIntegrator::Render(int w, int h, Scene &scene, uint8_t *data, int iteration, const IntegratorState &) {
    scene.UpdateDevice(m_ctx);
    Check(m_ctx, pt_render(m_ctx, w, h, iteration, data), "pt_render");
}

// -------------------------------------------------------------- DebugIntegrator
// kernel/integrators/debug.cu:198-239.
void DebugIntegrator::Render(int w, int h, Scene &scene, uint8_t *data, int, const IntegratorState &state) {
    scene.UpdateDevice(m_ctx);
    pt_integrator_state st;
    st.visualize_bvh = state.visualize_bvh ? 1 : 0;
    st.bvh_visualize_type = int(state.bvh_visualize_type);
    st.bvh_visualize_depth = state.bvh_visualize_depth;
    st.debug_mode = int(state.debug_mode);
    Check(m_ctx, pt_debug_render(m_ctx, w, h, &st, nullptr, data), "pt_debug_render");
}

}  // namespace pth
