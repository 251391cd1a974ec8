/* A host program in plain C against the C ABI alone (include/ptamd.h, include/ptamd_host.h), the way the
 * reference's Application::Run drives Scene + PathIntegrator in its offscreen branch
 * (application.cpp:58-76): load a scene file, build its acceleration structures on the GPU, render,
 * write the PNG and a film checkpoint.
 *
 *   headless_render <scene.json> <output prefix> [iterations] [samples per pass] [tile_rank tile_world]
 *
 * With the last two arguments only the rank's band of rows is rendered (pt_set_render_window), as one
 * rank of an image-tile sharded job would.  Image textures (the helmet scene's JPEGs) are decoded by
 * the library's own decoder, pth_decode_image, handed to the loader as its callback.                 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ptamd.h"
#include "ptamd_host.h"

static int fail(const char *what, pt_ctx *ctx) {
    fprintf(stderr, "%s: %s | %s\n", what, ctx ? pt_last_error(ctx) : "", pth_last_error());
    return 1;
}

int main(int argc, char **argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: %s scene.json output_prefix [iterations] [samples_per_pass] [tile_rank tile_world]\n", argv[0]);
        return 2;
    }
    const unsigned iterations = argc > 3 ? (unsigned)atoi(argv[3]) : 0u;
    const unsigned batch = argc > 4 ? (unsigned)atoi(argv[4]) : 0u;
    pt_ctx *ctx = NULL;
    if (pt_ctx_create(0, &ctx) != PT_OK) return fail("pt_ctx_create", ctx);
    pth_scene *scene = NULL;
    if (pth_scene_load(argv[1], pth_decode_image, NULL, 0, 0, 0, &scene) != PT_OK) return fail("pth_scene_load", ctx);
    if (pth_scene_finalize_device(scene, ctx) != PT_OK) return fail("pth_scene_finalize_device", ctx);
    pt_scene_desc desc;
    if (pth_scene_get_desc(scene, &desc) != PT_OK) return fail("pth_scene_get_desc", ctx);
    const int w = desc.camera.resolution[0], h = desc.camera.resolution[1];
    if (pth_scene_update_device(scene, ctx) != PT_OK) return fail("pth_scene_update_device", ctx);
    pth_offscreen_options opt;
    memset(&opt, 0, sizeof(opt));
    if (argc > 6) { /* a band of whole 8-pixel rows, the way an image-tile sharded job deals out the frame */
        const int rank = atoi(argv[5]), world = atoi(argv[6]);
        const int strips = (h + 7) / 8, base = strips / world, extra = strips % world;
        int row = 0, rows = 0, r;
        for (r = 0; r <= rank; ++r) {
            row += rows;
            rows = 8 * (base + (r < extra ? 1 : 0));
            if (rows > h - row) rows = h - row;
        }
        if (rows <= 0) {
            printf("rank %d of %d has no rows of a %dx%d frame\n", rank, world, w, h);
            return 0;
        }
        opt.window[0] = 0;
        opt.window[1] = row;
        opt.window[2] = w;
        opt.window[3] = rows;
    }
    char checkpoint[1024];
    snprintf(checkpoint, sizeof(checkpoint), "%s.film", argv[2]);
    opt.iterations = iterations;
    opt.batch = batch;
    opt.output_prefix = argv[2];
    opt.checkpoint_path = checkpoint;
    pth_offscreen_result res;
    if (pth_render_offscreen(ctx, scene, &opt, &res) != PT_OK) return fail("pth_render_offscreen", ctx);
    pt_stats st;
    if (pt_get_stats(ctx, &st) != PT_OK) return fail("pt_get_stats", ctx);
    printf("%dx%d, %u iterations in %.3f s, %u image(s); last pass: %llu paths, %llu closest-hit rays, %llu shadow rays\n", w, h, res.iterations_done,
           res.seconds, res.images_written, (unsigned long long)st.paths, (unsigned long long)st.closest_rays, (unsigned long long)st.shadow_rays);
    pth_scene_destroy(scene);
    pt_ctx_destroy(ctx);
    return 0;
}
