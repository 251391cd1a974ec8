/*
 * ptamd_host.h - C wrappers over the host C++ facade (cuda-path-tracer_amd/host),
 * so that language bindings (the Python test-suite, bench.py) can drive the same
 * Scene / TLAS / Camera / DiscreteSampler1D / PathIntegrator objects a C++ user
 * links against.  None of these calls touches the GPU except pth_scene_update_device
 * and the pth_integrator_* family, which forward to include/ptamd.h.
 *
 * Every function returns PT_OK or an error code; pth_last_error() holds the message
 * of the last failing pth_* call of the calling thread.  Citations are file:line in
 * the reference tree (src/...).
 */
#ifndef PTAMD_HOST_H
#define PTAMD_HOST_H

#include "ptamd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pth_scene pth_scene;
typedef struct pth_integrator pth_integrator;

const char *pth_last_error(void);

/* ---- math helpers with glm's operation order (math/transform.cpp:5-14, scene.cpp:415-417) */
int pth_transform_matrix(const float translate[3], const float rotate_deg[3], const float scale[3], float out[16]);
int pth_matrix_derive(const float transform[16], float inv_transform[16], float inv_transpose[16]);
int pth_matrix_mul(const float a[16], const float b[16], float out[16]);
/* glTF node matrix: translate * mat4(quat xyzw) * scale * matrix (mesh/gltf_model.h:56-64). */
int pth_gltf_local_matrix(const float translation[3], const float rotation_xyzw[4], const float scale[3],
                          const float matrix[16], float out[16]);

/* ---- DiscreteSampler1D::UpdateWeights (discrete_sampler.cpp:14-76) */
int pth_alias_build(const float *weights, uint32_t n, float *probability_table, uint32_t *alias_table);

/* ---- Camera::Update (camera.cpp:6-20) + Camera::UpdateDevice (:36-51) */
int pth_camera_make(const float eye[3], const float look_at[3], const float up[3], float fovy_deg, int width,
                    int height, pt_camera *out);

/* ---- TLAS::Build (bvh.cpp:206-332) on raw arrays; arrays are copied out with pth_bvh_get */
typedef struct pth_bvh pth_bvh;
int pth_bvh_build(const float *positions, const uint32_t *indices, const pt_geometry_instance *instances, uint32_t n,
                  pth_bvh **out);
int pth_bvh_sizes(const pth_bvh *bvh, uint32_t *tlas_nodes, uint32_t *blases, uint32_t *blas_nodes, uint32_t *tri_indices);
int pth_bvh_get(const pth_bvh *bvh, pt_tlas_node *tlas_nodes, pt_blas *blases, pt_bvh_node *blas_nodes, uint32_t *tri_indices);
void pth_bvh_free(pth_bvh *bvh);

/* ---- Scene (scene.h:23-186), filled programmatically */
int pth_scene_create(pth_scene **out); /* runs InitDefaults (scene.cpp:235-260) */
void pth_scene_destroy(pth_scene *scene);
int pth_scene_set_integrator(pth_scene *scene, int max_depth, int max_iterations);
int pth_scene_set_camera(pth_scene *scene, const float eye[3], const float look_at[3], const float up[3], float fovy_deg,
                         int width, int height);
/* returns the mesh's first_index through *first_index (index_count / vertex_count are the inputs) */
int pth_scene_add_mesh(pth_scene *scene, const float *positions, const float *normals, const float *uvs,
                       uint32_t vertex_count, const uint32_t *indices, uint32_t index_count, uint32_t *first_index);
int pth_scene_add_material(pth_scene *scene, const pt_material *material, int *material_id);
int pth_scene_add_texture1(pth_scene *scene, const pt_texture1 *texture, int *texture_id);
int pth_scene_add_texture3(pth_scene *scene, const pt_texture3 *texture, int *texture_id);
int pth_scene_add_image(pth_scene *scene, const float *texels, int width, int height, int channels, int srgb, int *image_id);
int pth_scene_new_blas_index(pth_scene *scene, int *blas_index);
int pth_scene_add_instance(pth_scene *scene, uint32_t first_index, uint32_t index_count, uint32_t vertex_count,
                           int material_id, const float transform[16], int blas_index, int *instance_id);
/* TLAS::Build + ComputeLightData (scene.cpp:676-686, :580-647) */
int pth_scene_finalize(pth_scene *scene);
/* Same, with the BLAS / TLAS build done by pt_bvh_build on the GPU behind ctx (identical arrays). */
int pth_scene_finalize_device(pth_scene *scene, pt_ctx *ctx);
/* host-pointer description, valid until the scene changes or is destroyed */
int pth_scene_get_desc(pth_scene *scene, pt_scene_desc *out);
/* Scene::UpdateDevice (scene.cpp:52-96) */
int pth_scene_update_device(pth_scene *scene, pt_ctx *ctx);

/* ---- Scene::LoadScene (scene.cpp:649-687): scene JSON (current schema, and the older capitalised
 * schema of Scenes/cornell.json etc. through this project's adapter), glTF 2.0 and Wavefront OBJ, read
 * in C++ with the reference's id order (materials by key, one instance per glTF primitive per node,
 * FACE_NORMALS de-indexing, T * Rx * Ry * Rz * S transforms).  The scene comes back filled but not
 * finalised: finish with pth_scene_finalize or pth_scene_finalize_device.  width / height /
 * max_depth > 0 override the file's values.
 * Image files (PATH textures, glTF images) are decoded by a callback: pth_decode_image below (this
 * library's own JPEG / PNG decoder) or the caller's; scenes without image textures may pass NULL.  The callback is called
 * twice per image: first with pixels == NULL to report width, height and channel count, then with a
 * buffer of width*height*channels bytes to fill (8 bits per channel, rows top to bottom).
 * want_channels: 0 = as stored in the file, 3 = RGB, 4 = RGBA.  Non-zero return = failure.        */
typedef int (*pth_image_decoder)(void *user, const uint8_t *file_bytes, size_t n_bytes, int want_channels, int *width,
                                 int *height, int *channels, uint8_t *pixels, size_t capacity);
int pth_scene_load(const char *path, pth_image_decoder decode, void *user, int width, int height, int max_depth,
                   pth_scene **out);
/* A pth_image_decoder of this library's own (host/image_decode.cpp), in place of the stb_image the reference decodes
 * its textures with (image.cpp:58-103, gltf_model.cpp:96-152): baseline, extended-sequential and progressive
 * Huffman-coded JPEG with 1 or 3 components of 8 bits (sampling factors 1 and 2, restart intervals), and PNG of every
 * colour type and bit depth, interlaced or not (16-bit samples keep their high byte, palettes and tRNS are expanded).
 * JPEG output is within 3 / 255 per channel of stb_image's and libjpeg's (inverse DCT in floating point, chroma of
 * subsampled files interpolated with (3, 1) / 4 weights); PNG output is exact.  `user` is ignored.  Returns 0, or 1
 * with the reason in pth_decode_image_error() (thread-local).                                                      */
int pth_decode_image(void *user, const uint8_t *file_bytes, size_t n_bytes, int want_channels, int *width, int *height,
                     int *channels, uint8_t *pixels, size_t capacity);
const char *pth_decode_image_error(void);

/* ---- PathIntegrator (kernel/integrators/path_trace.h:82-116) */
int pth_integrator_create(pt_ctx *ctx, pth_integrator **out);
void pth_integrator_destroy(pth_integrator *integrator);
int pth_integrator_init(pth_integrator *integrator, int width, int height);
int pth_integrator_resize(pth_integrator *integrator, int width, int height);
int pth_integrator_reset(pth_integrator *integrator, int width, int height);
int pth_integrator_render(pth_integrator *integrator, int width, int height, pth_scene *scene, uint8_t *rgba_device,
                          int iteration);
/* ---- DebugIntegrator::Render (kernel/integrators/debug.h:6-24, debug.cu:198-239): scene
 * update, then pt_debug_render into rgba_device.                              */
int pth_debug_render(pt_ctx *ctx, int width, int height, pth_scene *scene, const pt_integrator_state *state,
                     uint8_t *rgba_device);

/* ---- Application::Run, offscreen branch (application.cpp:58-76), as a library call: renders
 * iterations [first, scene max or `iterations`) of `scene` on `ctx`, `batch` samples per pass,
 * and writes <output_prefix>.<n>samp.png every `save_every` iterations (0: only at the end) and at
 * the end (Application::SaveImage, :157-179; no time stamp in the name).  With a checkpoint path
 * the film is saved there ("PTFILM1" header, width, height, iterations done, fp32 RGB) at every
 * save; with `resume` an existing checkpoint of the same size is loaded first and rendering
 * continues after its last iteration - the film ends up bit-identical to an uninterrupted run.  */
typedef struct pth_offscreen_options {
    uint32_t iterations;   /* 0 = the scene's MaxIterations() */
    uint32_t batch;        /* samples traced per pass (1..1024), 0 = a sixteenth of the job within 8..128, at most 160 GB of path state */
    uint32_t save_every;   /* 0 = only at the end */
    int32_t resume;
    const char *output_prefix;    /* NULL = no PNG output */
    const char *checkpoint_path;  /* NULL = no checkpoint */
    int32_t window[4];            /* x0, y0, w, h of the pixels to render (pt_set_render_window); all 0 = the whole frame.
                                     Pixels outside stay zero in the film: one rank's part of an image-tile sharded job. */
} pth_offscreen_options;
typedef struct pth_offscreen_result {
    uint32_t first_iteration;  /* where rendering started (non-zero after a resume) */
    uint32_t iterations_done;  /* total samples in the film */
    uint32_t images_written;
    double seconds;            /* render time, saves excluded */
} pth_offscreen_result;
int pth_render_offscreen(pt_ctx *ctx, pth_scene *scene, const pth_offscreen_options *options,
                         pth_offscreen_result *result);
/* Reads a checkpoint's header (and, when film_rgb is not NULL, its W*H*3 floats). */
int pth_checkpoint_read(const char *path, int *width, int *height, uint32_t *iterations, float *film_rgb);

/* ---- headless output replacing the display path (application.cpp:157-179, image.cpp:105-139):
 * writes an 8-bit RGB PNG (stored-deflate) or a Radiance .hdr of a fp32 film / iterations. */
int pth_save_png(const char *path, const uint8_t *rgba, int width, int height);
int pth_save_hdr(const char *path, const float *film_rgb, int width, int height, int iterations);

#ifdef __cplusplus
}
#endif
#endif /* PTAMD_HOST_H */
