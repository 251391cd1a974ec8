/*
 * ptamd.h - C ABI of the MI355X ray-scene hot path (libptamd.so).
 *
 * This is the boundary a maintainer of the CUDA path tracer binds in place of
 * its CUDA integrator: the scene is handed over as HOST arrays laid out exactly
 * as the reference builds them (its BVHNode / TLASNode / GeometryInstance /
 * Material / Texture<T> PODs), the library re-packs them for gfx950 and owns
 * all device memory.  Every entry point returns an int status (PT_OK == 0);
 * no exception crosses the boundary and no glm / torch type appears in a
 * signature.  Matrices are float[16], column-major (glm::mat4 memory order).
 *
 * Citations are file:line in the reference tree (src/...).
 */
#ifndef PTAMD_H
#define PTAMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ status */
enum {
    PT_OK = 0,
    PT_ERR_INVALID = 1,  /* bad argument / inconsistent scene description   */
    PT_ERR_DEVICE = 2,   /* a HIP call failed (message in pt_last_error)     */
    PT_ERR_STATE = 3,    /* call order violated (e.g. render before upload)  */
    PT_ERR_LIMIT = 4     /* a documented size limit was exceeded             */
};

/* --------------------------------------------- reference-layout scene PODs */

/* BVHNode, kernel/device_types.h:132-144 (32 B). primitive_count > 0 => leaf,
 * else first_primitive is the left child and the right child is left+1.      */
typedef struct pt_bvh_node {
    float min[3], max[3];
    uint32_t first_primitive, primitive_count;
} pt_bvh_node;

/* TLASNode, kernel/device_types.h:146-155 (32 B). left_right == 0 => leaf;
 * low 16 bits = left child, high 16 bits = right child.                       */
typedef struct pt_tlas_node {
    float min[3], max[3];
    uint32_t instance, left_right;
} pt_tlas_node;

/* DeviceBLAS, kernel/device_types.h:157-163 (16 B). */
typedef struct pt_blas {
    uint32_t first_node, node_count, first_index, index_count;
} pt_blas;

/* GeometryInstance + TriangleMesh, kernel/device_types.h:181-204 (224 B). */
typedef struct pt_geometry_instance {
    int32_t material_id;
    uint32_t first_index, index_count, vertex_count; /* TriangleMesh */
    float transform[16];     /* object -> world               */
    float inv_transform[16]; /* world  -> object              */
    float inv_transpose[16]; /* covectors object -> world     */
    int32_t blas_index;
    float total_area;
    int32_t triangle_sampler_index;
    int32_t area_light_id;
} pt_geometry_instance;

/* Material::Type, kernel/device_types.h:62-72. */
enum {
    PT_MAT_LAMBERTIAN = 0,
    PT_MAT_DIELECTRIC = 1,
    PT_MAT_MIRROR = 2,
    PT_MAT_METALLIC_ROUGHNESS = 3,
    PT_MAT_EMISSIVE = 4
};

/* Material, kernel/device_types.h:62-82 (16 B tagged union).
 *   emissive:            tex0 = color_texture,  w1.f = emittance
 *   dielectric:          tex0 = albedo_texture, w1.f = ior
 *   lambertian / mirror: tex0 = albedo_texture
 *   metallic_roughness:  tex0 = albedo, w1.i = metallic_texture, w2 = roughness */
typedef struct pt_material {
    uint8_t type;
    uint8_t _pad[3];
    int32_t tex0;
    union { float f; int32_t i; } w1;
    int32_t w2;
} pt_material;

/* TextureType, kernel/device_types.h:227-231. */
enum { PT_TEX_CONSTANT = 0, PT_TEX_IMAGE = 1, PT_TEX_CHECKERBOARD = 2 };

/* Texture<float>, kernel/device_types.h:208-242 (32 B). */
typedef struct pt_texture1 {
    uint8_t type;
    uint8_t _pad[3];
    union {
        struct { float value; } constant;
        struct { int32_t width, height; float scale[2], offset[2]; int32_t image_id; } image;
        struct { float color0, color1; float scale[2], offset[2]; } checkerboard;
    } u;
} pt_texture1;

/* Texture<glm::vec3>, kernel/device_types.h:208-242 (44 B). */
typedef struct pt_texture3 {
    uint8_t type;
    uint8_t _pad[3];
    union {
        struct { float value[3]; } constant;
        struct { int32_t width, height; float scale[2], offset[2]; int32_t image_id; } image;
        struct { float color0[3], color1[3]; float scale[2], offset[2]; } checkerboard;
    } u;
} pt_texture3;

/* A decoded host image as the reference keeps it before upload
 * (Image<float> / Image<glm::vec3>, texture.cpp:9-74): tightly packed fp32,
 * row-major, channels == 1 or 3.                                            */
typedef struct pt_image {
    const float *data;
    int32_t width, height, channels;
    int32_t srgb; /* ColorSpace::sRGB tag; applied only with PT_TEX_SRGB_DECODE (pt_set_texture_filter) */
} pt_image;

/* DeviceDiscreteSampler1D, kernel/device_types.h:122-127 (host pointers here). */
typedef struct pt_alias_table {
    const uint32_t *alias_table;
    const float *probability_table;
    uint32_t size;
    uint32_t _pad;
} pt_alias_table;

/* AreaLight, kernel/device_types.h:260-263. */
typedef struct pt_area_light {
    float power;
    int32_t geometry_instance_index;
} pt_area_light;

/* DeviceCamera, kernel/device_types.h:102-118 (without the dead film pointer). */
typedef struct pt_camera {
    int32_t resolution[2];
    float position[3];
    float front[3];
    float up[3];
    float right[3];
    float near_plane, far_plane;
    float focal_length;
    float fovy;
    float pixel_length[2];
} pt_camera;

/* DeviceScene, kernel/device_scene.h:5-35, with DeviceTLAS
 * (kernel/device_types.h:165-177) and DeviceTexturePool (:244-256) inlined.
 * All pointers are HOST pointers and are only read during pt_scene_upload.   */
typedef struct pt_scene_desc {
    uint32_t max_depth;
    uint32_t vertex_count;
    const float *positions; /* vertex_count * 3 */
    const float *normals;   /* vertex_count * 3 */
    const float *uvs;       /* vertex_count * 2 */
    const uint32_t *indices;
    uint32_t index_count;
    uint32_t geometry_count;
    const pt_geometry_instance *geometries;
    const pt_alias_table *triangle_samplers;
    uint32_t triangle_sampler_count;
    uint32_t tlas_node_count;
    const pt_tlas_node *tlas_nodes;
    const pt_blas *blases;
    uint32_t blas_count;
    uint32_t blas_node_count;
    const pt_bvh_node *blas_nodes;
    const uint32_t *tri_indices;
    uint32_t tri_index_count;
    uint32_t material_count;
    const pt_material *materials;
    const pt_area_light *lights;
    uint32_t light_count;
    float total_light_power;
    const pt_alias_table *light_sampler; /* may be NULL when light_count == 0 */
    const pt_texture1 *textures1;
    const pt_texture3 *textures3;
    uint32_t texture1_count, texture3_count;
    const pt_image *images1;
    const pt_image *images3;
    uint32_t image1_count, image3_count;
    pt_camera camera;
} pt_scene_desc;

/* ------------------------------------------------------------- ray queries */

/* One ray, 32 B = two 16-B words. tmin is reserved (the reference uses a fixed
 * 1e-9 lower bound, math/intersection.h:63) and must be 0. tmax is the
 * t_max argument of IntersectScene (kernel/intersection.h:155).              */
typedef struct pt_ray {
    float origin[3];
    float tmin;
    float direction[3];
    float tmax;
} pt_ray;

/* Closest-hit record, 32 B. prim = triangle index inside the instance's mesh
 * (tri_indices value), inst = GeometryInstance index (Intersection::geometry_id),
 * u,v = barycentrics of vertices 1 and 2. Miss: t = -1, prim = inst = -1.     */
typedef struct pt_hit {
    float t, u, v;
    int32_t prim;
    int32_t inst;
    int32_t material_id;
    uint32_t _pad[2];
} pt_hit;

/* Intersection, kernel/intersection.h:10-17 (44 B). */
typedef struct pt_intersection {
    float time;
    float pos[3];
    float normal[3];
    float uv[2];
    int32_t material_id;
    int32_t geometry_id;
} pt_intersection;

/* Counters of the last pt_render / pt_trace_* call plus accumulated kernel
 * timings (valid when profiling was switched on with pt_set_profiling).       */
#define PT_MAX_TRACKED_BOUNCES 16
typedef struct pt_stats {
    uint64_t closest_rays;   /* extension rays traced by the last pt_render   */
    uint64_t shadow_rays;    /* any-hit rays traced by the last pt_render     */
    uint64_t paths;          /* camera paths started                          */
    uint32_t bounces_run;    /* bounce iterations that had work               */
    uint32_t stack_spills;   /* traversal-stack entries that left LDS         */
    uint32_t stack_drops;    /* pushes dropped at depth 64 (utils/stack.h:8-11)*/
    uint32_t nan_pixels;     /* samples scrubbed by the film accumulate       */
    uint32_t queue_sizes[PT_MAX_TRACKED_BOUNCES]; /* extension queue per bounce */
    double ms_closest, ms_any, ms_generate, ms_shade, ms_film; /* summed ms   */
    uint32_t n_closest, n_any, n_generate, n_shade, n_film;    /* launches    */
    uint32_t _reserved;
    uint64_t shaded_paths;   /* closest hits of the last pt_render: paths that entered a material / emissive queue, all bounces */
} pt_stats;

/* The record sizes the reference's host vectors have; a reinterpret_cast of
 * those vectors is only valid while these hold.                              */
#ifdef __cplusplus
#define PT_SIZE_CHECK(type, bytes) static_assert(sizeof(type) == (bytes), #type " layout")
#else
#define PT_SIZE_CHECK(type, bytes) _Static_assert(sizeof(type) == (bytes), #type " layout")
#endif
PT_SIZE_CHECK(pt_bvh_node, 32);
PT_SIZE_CHECK(pt_tlas_node, 32);
PT_SIZE_CHECK(pt_blas, 16);
PT_SIZE_CHECK(pt_geometry_instance, 224);
PT_SIZE_CHECK(pt_material, 16);
PT_SIZE_CHECK(pt_texture1, 32);
PT_SIZE_CHECK(pt_texture3, 44);
PT_SIZE_CHECK(pt_area_light, 8);
PT_SIZE_CHECK(pt_ray, 32);
PT_SIZE_CHECK(pt_hit, 32);
PT_SIZE_CHECK(pt_intersection, 44);
#undef PT_SIZE_CHECK

typedef struct pt_ctx pt_ctx;

/* ---------------------------------------------------------------- lifecycle */

/* Creates a context on HIP device `device` with its own stream. */
int pt_ctx_create(int device, pt_ctx **out);
void pt_ctx_destroy(pt_ctx *ctx);
/* Message of the last failing call on ctx (ctx == NULL: last failing create). */
const char *pt_last_error(const pt_ctx *ctx);
/* Library version string. */
const char *pt_version(void);
/* Use a caller-owned hipStream_t for all later work (NULL = the ctx's own). */
int pt_set_stream(pt_ctx *ctx, void *hip_stream);
int pt_synchronize(pt_ctx *ctx);
/* Bracket each kernel class with HIP events on the stream and sum into pt_stats. */
int pt_set_profiling(pt_ctx *ctx, int enabled);
int pt_get_stats(pt_ctx *ctx, pt_stats *out);
int pt_reset_stats(pt_ctx *ctx);

/* ---------------------------------------------------- scene upload (re-pack) */

/* Replaces Scene::UpdateDevice (scene.cpp:52-96), TLAS::UpdateDevice
 * (bvh.cpp:334-352), TexturePool::UpdateDevice (texture.cpp:76-109),
 * DiscreteSampler1D::UpdateDevice (discrete_sampler.cpp:78-81) and
 * Camera::UpdateDevice (camera.cpp:36-51): validates the description, re-packs
 * BVH / triangles / instances for gfx950 and uploads everything.             */
int pt_scene_upload(pt_ctx *ctx, const pt_scene_desc *desc);
/* Options of the software bilinear filter that replaces the CUDA texture object of the reference
 * (texture.cpp:60-69: linear filter, wrap addressing, normalised coordinates, fp32 texels); they
 * take effect at the next pt_scene_upload.  0 (default): fp32 weights from x = u*W - 0.5, texels as
 * stored.  PT_TEX_FIXED_POINT_WEIGHTS: the two weights are rounded to 8 fractional bits, the
 * precision the CUDA texture unit filters with.  PT_TEX_SRGB_DECODE: texels of images tagged srgb
 * (the reference sets texture_desc.sRGB for albedo images, texture.cpp:67-69) are linearised before
 * filtering; whether CUDA does that for float4 texels cannot be settled without a CUDA device, so it
 * is off unless asked for.                                                                      */
enum { PT_TEX_FIXED_POINT_WEIGHTS = 1, PT_TEX_SRGB_DECODE = 2 };
int pt_set_texture_filter(pt_ctx *ctx, int options);
/* Replaces Camera::UpdateDevice (camera.cpp:36-51) for a camera-only change. */
int pt_camera_update(pt_ctx *ctx, const pt_camera *camera);
/* Replaces Scene::~Scene / TLAS::FreeDevice / TexturePool::FreeDevice
 * (scene.cpp:22-50, bvh.cpp:354-365, texture.cpp:111-129).                   */
int pt_scene_free(pt_ctx *ctx);

/* -------------------------------------------------------------- integrator */

/* Replace PathIntegrator::Init / Destroy / Resize / Reset
 * (kernel/integrators/path_trace.cu:16-74).                                  */
int pt_integrator_init(pt_ctx *ctx, int width, int height);
int pt_integrator_destroy(pt_ctx *ctx);
int pt_integrator_resize(pt_ctx *ctx, int width, int height);
int pt_integrator_reset(pt_ctx *ctx, int width, int height);

/* Replaces PathIntegrator::Render (path_trace.cu:436-483): one sample per
 * pixel for `iteration`, film accumulate, then display = film/(iteration+1)
 * -> ACES -> 8-bit into rgba_device (W*H uchar4 DEVICE pointer, caller-owned
 * like the reference's `data`; NULL skips the display pass). Synchronous.     */
int pt_render(pt_ctx *ctx, int width, int height, int iteration, uint8_t *rgba_device);
/* Same, without the trailing host synchronisation (for back-to-back samples). */
int pt_render_async(pt_ctx *ctx, int width, int height, int iteration, uint8_t *rgba_device);

/* `samples` consecutive iterations (iteration .. iteration+samples-1) in ONE wavefront pass:
 * all their paths are in flight together, so every launch carries `samples` times the work
 * (path state grows to samples * W*H slots; MI355X's HBM is sized for that).  The film receives
 * the samples of a pixel in iteration order, so the result is bit-identical to `samples` calls
 * of pt_render.  Asynchronous like pt_render_async.  Consecutive passes rotate over three streams of
 * the context's own, each starting after the work already submitted to the context's stream;
 * every other entry point makes the context's stream wait for the passes in flight, and
 * pt_synchronize waits for all of them.  No counterpart in the reference, whose Render traces
 * one sample per call.                                                                       */
int pt_render_batch(pt_ctx *ctx, int width, int height, int iteration, int samples, uint8_t *rgba_device);

/* Restricts the path pipeline to the pixels [x0, x0+w) x [y0, y0+h) of the frame: the render calls
 * above then trace, and add to the film, only those pixels, with path state and launches sized for
 * the window.  A pixel keeps its frame coordinates, hence its sampler seed (kernel/sampler.h: the
 * seed is a hash of x, y and the iteration), so a window's pixels are bit-identical to the same
 * pixels of a full-frame render, and films of disjoint windows add up to the full film.  This is
 * what image-tile sharding across GPUs rests on (every rank renders all iterations of its own
 * tile; pt_film_reduce sums the films).  x0 = y0 = w = h = 0 selects the whole frame again, as
 * pt_integrator_init does.  Waits for the passes in flight.  No counterpart in the reference.   */
int pt_set_render_window(pt_ctx *ctx, int x0, int y0, int w, int h);

/* Film access (the reference keeps m_image private; the headless path and the
 * multi-GPU reduce need it). rgb = W*H*3 fp32, row-major.                     */
int pt_film_download(pt_ctx *ctx, float *rgb_host);
int pt_film_copy_to_device(pt_ctx *ctx, float *rgb_device);
int pt_film_copy_from_device(pt_ctx *ctx, const float *rgb_device);
/* Restores a film saved with pt_film_download (checkpoint / resume).               */
int pt_film_upload(pt_ctx *ctx, const float *rgb_host);
/* The display image of the context's film after `iterations` samples, W*H*4 bytes in HOST memory:
 * what Application::SaveImage (application.cpp:157-179) copies back before writing the PNG.   */
int pt_display_download(pt_ctx *ctx, int iterations, uint8_t *rgba_host);
/* PostprocessAndUpload (path_trace.cu:409-434) on an arbitrary device film:
 * out = quantise(ACES(film / iterations)). film_device NULL = the ctx's film. */
int pt_tonemap(pt_ctx *ctx, const float *film_device, int width, int height, int iterations,
               uint8_t *rgba_device);

/* ------------------------------------------------------- multi-GPU film reduce */

/* One process per GPU renders its share of the iterations into its own film (the film is the only
 * shared mutable state of the reference's integrator and it is additive, path_trace.cu:391-407); the
 * job ends with film(root) = sum of all films.  No counterpart in the reference, which runs on one GPU.
 * pt_film_reduce issues ncclReduce(sum, fp32, W*H*3, in place) on the context's stream, ordered after
 * the render passes in flight like every other film access; `comm` is an ncclComm_t of the caller's
 * (RCCL), whose GPU is the context's.  librccl is bound at run time on first use, so single-GPU users
 * need not have it.  The three helpers wrap ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy for
 * hosts that do not link RCCL themselves; id128 is the 128-byte ncclUniqueId, created on one rank and
 * handed to the others by the caller.                                                            */
int pt_film_reduce(pt_ctx *ctx, void *nccl_comm, int root);
int pt_rccl_unique_id(pt_ctx *ctx, void *id128);
int pt_rccl_comm_create(pt_ctx *ctx, int nranks, const void *id128, int rank, void **nccl_comm_out);
int pt_rccl_comm_destroy(pt_ctx *ctx, void *nccl_comm);

/* ------------------------------------------------------- debug integrator */

/* DebugVisualizeMode / BVH_VisualizeMode / IntegratorState (kernel/device_types.h:10-34;
 * the UI-only `dirty` flag is not part of the boundary).                       */
enum {
    PT_DEBUG_NONE = 0,
    PT_DEBUG_GEOMETRIC_NORMALS = 1,
    PT_DEBUG_ALBEDO = 2,
    PT_DEBUG_UV = 3,
    PT_DEBUG_DEPTH = 4,
    PT_DEBUG_METALLIC = 5,
    PT_DEBUG_ROUGHNESS = 6
};
enum { PT_BVH_BOUNDS_HIT = 0, PT_BVH_TLAS = 1, PT_BVH_TRIANGLES_HIT = 2 };
typedef struct pt_integrator_state {
    int32_t visualize_bvh;       /* bool */
    int32_t bvh_visualize_type;  /* PT_BVH_* */
    int32_t bvh_visualize_depth; /* heat-map scale, 500 in the reference */
    int32_t debug_mode;          /* PT_DEBUG_* */
} pt_integrator_state;

/* Replaces DebugIntegrator::Render (kernel/integrators/debug.cu:198-239): one pixel-centre
 * camera ray per pixel; the BVH heat map (VisualizeBVH, :119-178: nodes visited / TLAS nodes
 * visited / triangles tested) when visualize_bvh is set, replaced by the attribute view
 * (DebugView, :51-117: normals, albedo, uv, depth, metallic, roughness) when debug_mode is not
 * PT_DEBUG_NONE; with neither, nothing is written, as in the reference.  image_device =
 * W*H*3 fp32 (the reference's private m_image), rgba_device = W*H uchar4 after
 * SendImageToBuffer (:180-196); both are DEVICE pointers and either may be NULL.  width and
 * height must equal the camera's resolution.  Needs no pt_integrator_init.  Asynchronous on
 * the context's stream.                                                        */
int pt_debug_render(pt_ctx *ctx, int width, int height, const pt_integrator_state *state,
                    float *image_device, uint8_t *rgba_device);

/* ------------------------------------------------------------ device BVH build */

/* Replaces TLAS::Build and the BLAS::Build calls inside it (bvh.cpp:5-332) with a build on the
 * GPU: one binned-SAH BLAS per distinct blas_index (defined by the first instance naming it),
 * instance world bounds, agglomerative TLAS.  Inputs are HOST arrays in the reference's
 * layouts (Scene::Positions / Indices, GeometryInstance[] with transform, first_index,
 * index_count and blas_index filled); the result is read back into HOST arrays identical, bit
 * for bit, to what the reference's sequential builder produces (node numbering and triangle
 * permutation included), ready for pt_scene_desc.  Limits: 32767 instances (16-bit TLAS child
 * indices, kernel/device_types.h:153-154), 2^24 triangles per call.  Synchronous.            */
typedef struct pt_bvh pt_bvh;
int pt_bvh_build(pt_ctx *ctx, const float *positions, uint32_t vertex_count, const uint32_t *indices,
                 uint32_t index_count, const pt_geometry_instance *geometries, uint32_t geometry_count,
                 pt_bvh **out);
/* Array lengths of the result (TLAS::UpdateDevice, bvh.cpp:334-352). */
int pt_bvh_sizes(const pt_bvh *bvh, uint32_t *tlas_nodes, uint32_t *blases, uint32_t *blas_nodes,
                 uint32_t *tri_indices);
/* Copies the result into caller arrays of those lengths (any may be NULL). */
int pt_bvh_get(const pt_bvh *bvh, pt_tlas_node *tlas_nodes, pt_blas *blases, pt_bvh_node *blas_nodes,
               uint32_t *tri_indices);
/* Device time of the BLAS and TLAS parts (ms) and the depth of the deepest BLAS. */
int pt_bvh_timing(const pt_bvh *bvh, float *ms_blas, float *ms_tlas, uint32_t *levels);
void pt_bvh_free(pt_bvh *bvh);

/* ------------------------------------------------- batch ray queries (metric) */

/* IntersectScene<CLOSEST> (kernel/intersection.h:154-235) over n rays.
 * rays / hits are DEVICE pointers, 16-B aligned.                             */
int pt_trace_closest(pt_ctx *ctx, const pt_ray *rays_device, size_t n, pt_hit *hits_device);
/* IntersectScene<ANY> with t_max = ray.tmax; occluded[i] = 1 on any hit. */
int pt_trace_any(pt_ctx *ctx, const pt_ray *rays_device, size_t n, uint8_t *occluded_device);
/* Expands hit records to the reference's Intersection (pos, shading normal, uv,
 * material, geometry; kernel/intersection.h:96-99,145-149).                  */
int pt_resolve_hits(pt_ctx *ctx, const pt_ray *rays_device, const pt_hit *hits_device, size_t n,
                    pt_intersection *out_device);
/* pt_trace_closest that also writes four visit counters per ray (interior nodes expanded,
 * TLAS leaves entered, triangles tested, BLAS leaves visited): the reference's traversal
 * callback hook (kernel/intersection.h:24-31,61,181) as a query. counters4 = n * 4 uint32. */
int pt_trace_closest_counted(pt_ctx *ctx, const pt_ray *rays_device, size_t n, pt_hit *hits_device,
                             uint32_t *counters4_device);
/* The same four counters for an any-hit query (the search stops at the first occluder found). */
int pt_trace_any_counted(pt_ctx *ctx, const pt_ray *rays_device, size_t n, uint8_t *occluded_device,
                         uint32_t *counters4_device);
/* pt_trace_closest for callers that cannot allocate (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc): no counterpart in the reference.  The call allocates
 * nothing and creates no events: its scratch (work-feed cursors, event counters, traversal-stack
 * spill area for 1024 persistent waves) is a file-scope device array of 60.5 MiB + 16.25 KiB that every
 * GPU the library runs on holds until the process ends.  Results are those of pt_trace_closest, bit
 * for bit.  n > PT_STATIC_MAX_RAYS returns PT_ERR_LIMIT.  All calls on one GPU share that GPU's
 * array: calls that overlap on one GPU (two contexts, two streams) race; calls on one stream do not.
 * Stack spills / drops of these calls are not added to pt_stats.  Asynchronous on the context's
 * stream like pt_trace_closest.  "Allocates nothing" is about the library's calls: in trace modes
 * 2 / 3 the kernel behind it keeps a few spilled registers (under 64 bytes per lane) in the private
 * segment that the HIP runtime provides to any kernel that declares one; trace modes 0 / 1 use none. */
#define PT_STATIC_MAX_RAYS ((size_t)1 << 26)
int pt_trace_closest_static(pt_ctx *ctx, const pt_ray *rays_device, size_t n, pt_hit *hits_device);
/* Host-pointer conveniences (stage through device memory, synchronous). */
int pt_trace_closest_host(pt_ctx *ctx, const pt_ray *rays, size_t n, pt_hit *hits);
int pt_trace_any_host(pt_ctx *ctx, const pt_ray *rays, size_t n, uint8_t *occluded);
int pt_resolve_hits_host(pt_ctx *ctx, const pt_ray *rays, const pt_hit *hits, size_t n,
                         pt_intersection *out);

/* Fills n incoherent rays on the device: origin uniform in the scene's root
 * box scaled 1.1x about its centre, direction uniform on the sphere, tmax
 * FLT_MAX, drawn from the per-ray PCG stream state = (i << 32) | 0xdeadbeef
 * (i = first_index + ray number) after two discarded draws.                  */
int pt_generate_incoherent_rays(pt_ctx *ctx, pt_ray *rays_device, size_t n, uint64_t first_index);
/* Captures the extension-ray queue of `bounce` (0 = camera rays) of `iteration`
 * into rays_device (capacity rays); *count receives the number written.       */
int pt_capture_bounce_rays(pt_ctx *ctx, int width, int height, int iteration, int bounce,
                           pt_ray *rays_device, size_t capacity, size_t *count);

/* Captures the shadow rays (next-event estimation, path_trace.cu:221-243) that the shading of
 * `bounce` emits in `iteration`, with their t_max = distance - 1e-4, in queue order.          */
int pt_capture_shadow_rays(pt_ctx *ctx, int width, int height, int iteration, int bounce,
                           pt_ray *rays_device, size_t capacity, size_t *count);

/* Traversal variant, two independent bits; 0 (default) answers every query exactly like the reference,
 * including what its any-hit search does after the first accepted triangle: it finishes the instance
 * (boxes tested against the latest accepted distance) and reports a hit only if the world-space distance
 * recomputed from the last accepted triangle is not negative (kernel/intersection.h:101-109, :139-149).
 * Bit 0 (modes 1, 3): any-hit searches (pt_trace_any, shadow rays) stop at the first accepted triangle
 *   farther than 1e-2 (nearer ones follow the rule above, so a triangle touching the ray's origin is
 *   judged as in the reference) and skip boxes that start beyond t_max * (1 + 2^-12), which the reference
 *   does not do (kernel/intersection.h:120-121,209-210 test child boxes against FLT_MAX): fewer nodes are
 *   visited (2.5x faster shadow rays in the 20 001-instance scene).  A box test and a triangle test round
 *   differently, so in principle "occluded" can turn into "free" for a triangle within rounding of t_max;
 *   the margin keeps the decision with the triangle test where the reference's own shadow rays produce
 *   that case (a ray grazing the flat light it was aimed at: the ill-conditioned distance to the light's own
 *   triangle can fall before t_max = distance - 1e-4, path_trace.cu:221-227, and the reference then counts
 *   the light as its own occluder).  Measured against the CPU oracle
 *   on 2.1 M captured shadow rays of three scenes and against mode 0 on 17.6 M of five: no flag differs.
 * Bit 1 (modes 2, 3): the trees are walked as 4-wide nodes (128 bytes: the reference's own fp32 boxes
 *   of four children, tested with the reference's arithmetic), the top level is rebuilt top down over
 *   the reference's instance boxes, each clipped to the world box of its transformed mesh (the reference's
 *   box of a rotated mesh is the box of a rotated box, bvh.cpp:273-280; a ray that misses the clipped box
 *   cannot hit the instance), and the search inside an instance is bounded by the best hit of
 *   the instances visited before.  Every box decides as in the reference; the visiting order differs,
 *   so rays with two triangles at the same distance may report the other one (same t to rounding;
 *   all other rays get bit-identical records), and the 64-entry stack limit of the reference
 *   (utils/stack.h:8-11) is not reproduced: deep trees are traversed completely.                  */
int pt_set_trace_mode(pt_ctx *ctx, int mode);

#ifdef __cplusplus
}
#endif
#endif /* PTAMD_H */
