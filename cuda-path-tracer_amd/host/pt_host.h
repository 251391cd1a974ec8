// Host C++ facade of the MI355X ray-scene path: the classes a user of the CUDA
// path tracer already knows (Scene, Camera, Film, BLAS, TLAS, DiscreteSampler1D,
// TexturePool, Integrator / PathIntegrator), re-implemented on top of the C ABI
// in include/ptamd.h.  Method names and meaning follow the reference headers
// cited at each class; glm types are replaced by plain float arrays.
#pragma once
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/ptamd.h"
#include "hmath.h"

namespace pth {

// utils/exception.h:6-29 - errors on the host side are exceptions, as in the reference.
class Exception : public std::runtime_error {
public:
    explicit Exception(const std::string &what) : std::runtime_error(what) {}
};

enum class SplitStrategy { SAH, Midpoint, Median };  // bvh.h:9
enum class IntegratorType { Debug, Path };           // kernel/integrators/integrator.h:6-9

struct TriangleMesh {  // kernel/device_types.h:181-185
    uint32_t first_index = 0, index_count = 0, vertex_count = 0;
};

class Scene;

// bvh.h:21-47. Binned-SAH bottom-level hierarchy over one triangle range.
class BLAS {
public:
    void Build(const Scene &scene, const TriangleMesh &triangle_mesh, SplitStrategy split_type);
    // Same build from raw arrays (positions: xyz triples, indices: the scene's index buffer).
    void Build(const float *positions, const uint32_t *indices, const TriangleMesh &triangle_mesh,
               SplitStrategy split_type);
    Box Bounds() const;  // object-space bounds of the root
    const std::vector<pt_bvh_node> &Nodes() const { return m_nodes; }
    const std::vector<uint32_t> &TriIndices() const { return m_tri_indices; }

private:
    std::vector<pt_bvh_node> m_nodes;
    std::vector<uint32_t> m_tri_indices;
};

// bvh.h:49-66. Top-level hierarchy over instances plus the flattened BLAS arrays.
class TLAS {
public:
    void Build(const Scene &scene, const std::vector<pt_geometry_instance> &instances,
               SplitStrategy split_type = SplitStrategy::SAH);
    void Build(const float *positions, const uint32_t *indices,
               const std::vector<pt_geometry_instance> &instances,
               SplitStrategy split_type = SplitStrategy::SAH);
    // The same arrays, bit for bit, built on the GPU behind `ctx` (pt_bvh_build).
    void BuildOnDevice(pt_ctx *ctx, const float *positions, uint32_t vertex_count, const uint32_t *indices, uint32_t index_count,
                       const std::vector<pt_geometry_instance> &instances);
    // Points the description at this object's arrays (the reference uploads here, bvh.cpp:334-352;
    // the upload itself now happens inside pt_scene_upload).
    void UpdateDevice(pt_scene_desc &scene);
    void FreeDevice(pt_scene_desc &scene);
    bool &Dirty() { return m_dirty; }

    const std::vector<pt_tlas_node> &TlasNodes() const { return m_tlas_nodes; }
    const std::vector<pt_blas> &DeviceBlases() const { return m_device_blases; }
    const std::vector<pt_bvh_node> &BlasNodes() const { return m_blas_nodes; }
    const std::vector<uint32_t> &TriIndices() const { return m_tri_indices; }

private:
    bool m_dirty = false;
    std::vector<pt_tlas_node> m_tlas_nodes;
    std::vector<pt_blas> m_device_blases;
    std::vector<pt_bvh_node> m_blas_nodes;
    std::vector<uint32_t> m_tri_indices;
};

// discrete_sampler.h:5-15. Alias-method tables.
class DiscreteSampler1D {
public:
    void UpdateWeights(const std::vector<float> &weights);
    void UpdateDevice(pt_alias_table &sampler) const;
    const std::vector<uint32_t> &AliasTable() const { return m_alias_table; }
    const std::vector<float> &ProbabilityTable() const { return m_probability_table; }

private:
    std::vector<uint32_t> m_alias_table;
    std::vector<float> m_probability_table;
};

// film.h:10-20.
class Film {
public:
    int width = 0, height = 0;
    int color_space = 0;            // 0 = rgb, 1 = xyz (parsed, unused by the kernels)
    int reconstruction_filter = 0;  // 0 = box, 1 = gaussian (parsed, unused)
};

// camera.h:9-63 / camera.cpp:6-51.
class Camera {
public:
    void SetDefaults(Vec3 position, Vec3 look_at, Vec3 up, float fovy_deg);
    void SetResolution(int width, int height);
    void Update();
    void ResetDefaults();
    bool &Dirty() { return m_dirty; }
    void UpdateDevice(pt_camera &camera);
    int Width() const { return m_film.width; }
    int Height() const { return m_film.height; }
    Vec3 &Position() { return m_position; }
    Vec3 &LookAt() { return m_look_at; }
    const Vec3 &Front() const { return m_front; }
    const Vec3 &Up() const { return m_up; }
    const Vec3 &Right() const { return m_right; }
    Film &GetFilm() { return m_film; }
    float near_plane = 0.1f, far_plane = 100.0f;

private:
    bool m_dirty = false;
    struct {
        float fovy = 45.0f;
        Vec3 up{0, 1, 0}, look_at{0, 0, -1}, position{0, 0, 0};
    } m_default_params;
    Vec3 m_position, m_look_at, m_front, m_up, m_right;
    float m_fovy = 45.0f;
    float m_pixel_length[2] = {0, 0};
    Film m_film;
};

// Texture.h:8-23. Host texture descriptors and decoded fp32 images.
struct TexturePool {
    std::vector<pt_texture1> textures1;
    std::vector<pt_texture3> textures3;
    struct HostImage {
        std::vector<float> texels;
        int width = 0, height = 0, channels = 0, srgb = 0;
    };
    std::vector<HostImage> images1, images3;
    bool dirty = false;
};

// scene.h:23-186 without the JSON front end: the same host-side model (global vertex /
// index arrays, instances, materials, textures, lights, camera) filled programmatically.
class Scene {
public:
    Scene();
    ~Scene();

    // InitDefaults, scene.cpp:235-260: texture 0 of each kind = constant 0.98, material 0 = Lambertian.
    void InitDefaults();
    void SetIntegrator(int max_depth, int max_iterations);
    // Appends a mesh like ParseObject does (scene.cpp:386-401, :448-459): indices are rebased.
    TriangleMesh AddMesh(const float *positions, const float *normals, const float *uvs, uint32_t vertex_count,
                         const uint32_t *indices, uint32_t index_count);
    int AddMaterial(const pt_material &material);
    int AddTexture1(const pt_texture1 &texture);
    int AddTexture3(const pt_texture3 &texture);
    int AddImage(const float *texels, int width, int height, int channels, int srgb);
    uint32_t NewBlasIndex() { return m_current_blas_index++; }
    // transform = object -> world; inverse and inverse-transpose are derived as scene.cpp:415-417.
    int AddInstance(const TriangleMesh &mesh, int material_id, const Mat4 &transform, int blas_index);
    // Tail of LoadScene, scene.cpp:676-686: builds the TLAS and the light tables.
    void Finalize();
    void Finalize(pt_ctx *ctx);  // same, with the acceleration structures built on the GPU behind ctx

    int Width() const { return m_camera.Width(); }
    int Height() const { return m_camera.Height(); }
    int MaxIterations() const { return m_max_iterations; }
    int MaxDepth() const { return m_max_depth; }
    Camera &GetCamera() { return m_camera; }
    const std::vector<float> &Positions() const { return m_positions; }
    const std::vector<uint32_t> &Indices() const { return m_indices; }
    const std::vector<pt_geometry_instance> &Geometries() const { return m_geometries; }
    const TLAS &Bvh() const { return *m_bvh; }

    // Scene::UpdateDevice, scene.cpp:52-96: returns whether anything was dirty.
    bool UpdateDevice(pt_ctx *ctx);
    // The host-pointer description handed to pt_scene_upload (valid while this Scene is unchanged).
    const pt_scene_desc &GetDeviceScene();

private:
    void ComputeLightData();  // scene.cpp:580-647
    void FillDesc();

    bool m_dirty = false;
    int m_max_depth = 4, m_max_iterations = 1000;
    std::vector<float> m_positions, m_normals, m_texcoords;
    std::vector<uint32_t> m_indices;
    std::vector<pt_geometry_instance> m_geometries;
    std::vector<DiscreteSampler1D> m_triangle_samplers;
    std::unique_ptr<TLAS> m_bvh;
    std::vector<pt_material> m_materials;
    TexturePool m_texture_pool;
    std::vector<pt_area_light> m_lights;
    DiscreteSampler1D m_light_sampler;
    float m_total_light_power = 0.0f;
    Camera m_camera;
    uint32_t m_current_blas_index = 0;

    pt_scene_desc m_desc;
    std::vector<pt_alias_table> m_desc_tri_samplers;
    pt_alias_table m_desc_light_sampler;
    std::vector<pt_image> m_desc_images1, m_desc_images3;
};

enum class DebugVisualizeMode { None, GeometricNormals, Albedo, UV, Depth, Metallic, Roughness };  // kernel/device_types.h:10-18
enum class BVH_VisualizeMode { BoundsHit, TLAS, TrianglesHit };                                      // kernel/device_types.h:20-24

struct IntegratorState {  // kernel/device_types.h:26-34
    bool dirty = false;

    bool visualize_bvh = false;
    BVH_VisualizeMode bvh_visualize_type = BVH_VisualizeMode::BoundsHit;
    int bvh_visualize_depth = 500;

    DebugVisualizeMode debug_mode = DebugVisualizeMode::None;
};

// kernel/integrators/integrator.h:11-24.
class Integrator {
public:
    virtual ~Integrator() = default;
    virtual void Init(int width, int height) = 0;
    virtual void Destroy() = 0;
    virtual void Resize(int width, int height) = 0;
    virtual void Reset(int width, int height) = 0;
    virtual IntegratorType Type() = 0;
    // `data` is a device pointer to width*height RGBA8 (may be null: film only).
    virtual void Render(int width, int height, Scene &scene, uint8_t *data, int iteration,
                        const IntegratorState &state) = 0;
};

// kernel/integrators/path_trace.h:82-116.
class PathIntegrator : public Integrator {
public:
    explicit PathIntegrator(pt_ctx *ctx) : m_ctx(ctx) {}
    void Init(int width, int height) override;
    void Destroy() override;
    void Resize(int width, int height) override;
    void Reset(int width, int height) override;
    IntegratorType Type() override { return IntegratorType::Path; }
    void Render(int width, int height, Scene &scene, uint8_t *data, int iteration,
                const IntegratorState &state) override;

private:
    pt_ctx *m_ctx;
};

// kernel/integrators/debug.h:6-24.  The views are stateless on the device, so Init / Destroy /
// Resize / Reset have nothing to allocate; `iteration` is unused, as in the reference.
class DebugIntegrator : public Integrator {
public:
    explicit DebugIntegrator(pt_ctx *ctx) : m_ctx(ctx) {}
    void Init(int, int) override {}
    void Destroy() override {}
    void Resize(int, int) override {}
    void Reset(int, int) override {}
    IntegratorType Type() override { return IntegratorType::Debug; }
    void Render(int width, int height, Scene &scene, uint8_t *data, int iteration,
                const IntegratorState &state) override;

private:
    pt_ctx *m_ctx;
};

// Scene::LoadScene (scene.cpp:649-687) without the acceleration structures (host/scene_loader.cpp); the
// decoder type is pth_image_decoder of include/ptamd_host.h.
void LoadSceneFile(Scene &scene, const std::string &path,
                   int (*decode)(void *, const uint8_t *, size_t, int, int *, int *, int *, uint8_t *, size_t), void *user,
                   int width, int height, int max_depth);

// Application::Run's offscreen branch (application.cpp:58-76) with progressive images and a film
// checkpoint; see include/ptamd_host.h for the options.
void RenderOffscreen(pt_ctx *ctx, Scene &scene, const struct pth_offscreen_options &options, struct pth_offscreen_result &result);
bool ReadCheckpoint(const std::string &path, int &width, int &height, uint32_t &iterations, std::vector<float> *film);

}  // namespace pth
