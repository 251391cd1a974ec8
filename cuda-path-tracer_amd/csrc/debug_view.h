// The debug integrator: BVH heat maps and surface-attribute views, one pixel-centre camera
// ray per pixel (kernel/integrators/debug.cu:42-239).  Both of the reference's kernels
// (VisualizeBVH, DebugView) are one pass here: the pixels stream through the same persistent
// waves as every other closest-hit query, in the counting variant, whose per-ray counters are
// exactly what the reference's traversal callback adds up (kernel/intersection.h:61,181: the
// callback runs once per node that survives the pop-time cull, i.e. once per expanded interior
// node and once per entered leaf, at either level).
#pragma once
#include "kernels.h"

namespace ptd {

struct DebugParams {
    int visualize_bvh, bvh_type, bvh_depth, debug_mode;  // IntegratorState, kernel/device_types.h:26-34
    int width, height;
};

PTD float glm_clamp(float x, float lo, float hi) {  // glm::clamp = min(max(x, lo), hi)
    const float m = (x < lo) ? lo : x;
    return (hi < m) ? hi : m;
}

// Wyman et al. fit of the CIE 1931 matching functions, math/color.h:19-48.
PTD V3 wavelength_to_xyz(float wavelength) {
    V3 r;
    {
        const float t1 = (wavelength - 442.0f) * ((wavelength < 442.0f) ? 0.0624f : 0.0374f);
        const float t2 = (wavelength - 599.8f) * ((wavelength < 599.8f) ? 0.0264f : 0.0323f);
        const float t3 = (wavelength - 501.1f) * ((wavelength < 501.1f) ? 0.0490f : 0.0382f);
        r.x = 0.362f * expf(-0.5f * t1 * t1) + 1.056f * expf(-0.5f * t2 * t2) - 0.065f * expf(-0.5f * t3 * t3);
    }
    {
        const float t1 = (wavelength - 568.8f) * ((wavelength < 568.8f) ? 0.0213f : 0.0247f);
        const float t2 = (wavelength - 530.9f) * ((wavelength < 530.9f) ? 0.0613f : 0.0322f);
        r.y = 0.821f * expf(-0.5f * t1 * t1) + 0.286f * expf(-0.5f * t2 * t2);
    }
    {
        const float t1 = (wavelength - 437.0f) * ((wavelength < 437.0f) ? 0.0845f : 0.0278f);
        const float t2 = (wavelength - 459.0f) * ((wavelength < 459.0f) ? 0.0385f : 0.0725f);
        r.z = 1.217f * expf(-0.5f * t1 * t1) + 0.681f * expf(-0.5f * t2 * t2);
    }
    return r;
}
PTD V3 xyz_to_linear_rgb(V3 xyz) {  // math/color.h:50-56
    return mk3(3.240479f * xyz.x - 1.537150f * xyz.y - 0.498535f * xyz.z,
               -0.969256f * xyz.x + 1.875991f * xyz.y + 0.041556f * xyz.z,
               0.055648f * xyz.x - 0.204043f * xyz.y + 1.057311f * xyz.z);
}
PTD float gamma_srgb(float c) {  // math/color.h:73-79
    c = glm_clamp(c, 0.0f, 1.0f);
    if (c <= 0.0031308f) return 12.92f * c;
    return 1.055f * powf(c, 1.0f / 2.4f) - 0.055f;
}
PTD V3 heat_map_color(int depth, int max_depth) {  // debug.cu:42-49
    const float t = float(depth) / float(max_depth);
    const float lambda = 380.0f * (1.0f - t) + 780.0f * t;  // glm::mix over the visible spectrum, math/spectrum.h:5-6
    const V3 rgb = xyz_to_linear_rgb(wavelength_to_xyz(lambda));
    return mk3(gamma_srgb(rgb.x), gamma_srgb(rgb.y), gamma_srgb(rgb.z));
}

// Stream i = path-slot order (8x8 pixel tiles, one per wave), so a wave's rays form a bundle.
struct DebugIO {
    static constexpr bool kExactDistance = true;
    static constexpr bool kStageCull = false;  // a pixel whose ray misses the scene is coloured by done() like any other
    const DScene &S;
    DebugParams p;
    float *image;    // W*H*3, may be null
    uint8_t *rgba;   // W*H*4, may be null

    PTD bool load(uint32_t i, V3 &o, V3 &d, float &tmax, uint32_t &tag) const {
        int x, y;
        if (!slot_to_pixel(i, p.width, p.height, x, y)) return false;
        tag = i;
        const DCamera &c = S.camera;  // SpawnCameraRay with sample (0.5, 0.5), kernel/camera.h:10-22
        const float fx = ((float(x) + 0.5f) - 0.5f * float(c.res_x)) * c.px;
        const float fy = ((float(y) + 0.5f) - 0.5f * float(c.res_y)) * c.py;
        const V3 film_point = c.position + c.front * c.focal_length + c.right * fx - c.up * fy;
        o = c.position;
        d = normalize(film_point - c.position);
        tmax = FLT_MAX;
        return true;
    }

    // VisualizeBVH<Mode>, debug.cu:119-178.
    PTD V3 bvh_view(const Traversal<MODE_CLOSEST, true> &tr) const {
        if (p.bvh_type == PT_BVH_BOUNDS_HIT) return heat_map_color(int(tr.ctr.interior + tr.ctr.tlas_leaves + tr.ctr.blas_leaves), p.bvh_depth);
        if (p.bvh_type == PT_BVH_TLAS) return heat_map_color(int(tr.tlas_interior + tr.ctr.tlas_leaves), p.bvh_depth);
        // TrianglesHit: a closest-hit search tests every triangle of a leaf it visits, so the
        // triangles tested are the sum of the visited leaves' primitive counts
        const int tri_count = int(tr.ctr.tris);
        float t = logf(float(tri_count) + 1.0f) / logf(50.0f + 1.0f);
        t = glm_clamp(t, 0.0f, 1.0f);
        const V3 green = mk3(0.0f, 0.3f, 0.0f), yellow = mk3(1.0f, 1.0f, 0.0f);
        return green * (1.0f - t) + yellow * t;
    }

    // DebugView<Mode>, debug.cu:51-117.
    PTD V3 attribute_view(const Traversal<MODE_CLOSEST, true> &tr, bool hit) const {
        const V3 background = splat(0.4f);
        if (!hit) return background;
        const Isect is = resolve_hit(S, tr.o, tr.d, tr.hit.u, tr.hit.v, tr.hit.rec, tr.hit.inst);
        const DMaterial m = S.materials[is.material_id];
        const uint32_t type = m.type & 0xFFu;
        switch (p.debug_mode) {
            case PT_DEBUG_GEOMETRIC_NORMALS: return is.normal;
            case PT_DEBUG_ALBEDO:
                if (type == PT_MAT_LAMBERTIAN || type == PT_MAT_METALLIC_ROUGHNESS) return sample_tex3(S, m.tex0, is.uv);
                return splat(0.0f);
            case PT_DEBUG_METALLIC:
                if (type == PT_MAT_METALLIC_ROUGHNESS) return splat(sample_tex1(S, int(m.w1), is.uv));
                return splat(0.0f);
            case PT_DEBUG_ROUGHNESS:
                if (type == PT_MAT_METALLIC_ROUGHNESS) return splat(sample_tex1(S, m.w2, is.uv));
                return splat(0.0f);
            case PT_DEBUG_UV: return mk3(is.uv.x, is.uv.y, 0.0f);
            case PT_DEBUG_DEPTH: {
                const float near_clip = S.camera.near_plane, far_clip = S.camera.far_plane, t = is.time;
                if (t < near_clip || t > far_clip) return background;
                const float depth = (t - near_clip) / (far_clip - near_clip);
                return splat(glm_clamp(depth, 0.0f, 1.0f));
            }
            default: break;
        }
        return splat(0.0f);
    }

    PTD void done(uint32_t i, const Traversal<MODE_CLOSEST, true> &tr, bool hit) const {
        int x, y;
        slot_to_pixel(i, p.width, p.height, x, y);
        const size_t index = size_t(x) + size_t(p.width) * size_t(y);
        // Render (debug.cu:198-239) runs the BVH view first and lets the attribute view overwrite it
        const V3 c = (p.debug_mode != PT_DEBUG_NONE) ? attribute_view(tr, hit) : bvh_view(tr);
        if (image) {
            image[3 * index + 0] = c.x;
            image[3 * index + 1] = c.y;
            image[3 * index + 2] = c.z;
        }
        if (rgba) {  // SendImageToBuffer, debug.cu:180-196
            const float v[3] = {c.x, c.y, c.z};
            uchar4 out;
            uint8_t *q = reinterpret_cast<uint8_t *>(&out);
            for (int k = 0; k < 3; ++k) {
                int b = int(double(v[k]) * 255.0);
                b = b < 0 ? 0 : (b > 255 ? 255 : b);
                q[k] = uint8_t(b);
            }
            q[3] = 255;
            reinterpret_cast<uchar4 *>(rgba)[index] = out;
        }
    }
    PTD bool has_pending() const { return false; }
    PTD void flush() const {}
    PTD void finish() const {}
};

__global__ __launch_bounds__(64) void k_debug_view(DScene S, DebugParams p, float *image, uint8_t *rgba, uint32_t n, uint32_t *cursors, uint32_t cursor_count, uint2 *spill,
                                                   uint32_t spill_stride, DevStats *stats) {
    LaneEvents ev = make_events(spill + size_t(blockIdx.x) * kWave, spill_stride);
    DebugIO io{S, p, image, rgba};
    __shared__ unsigned long long lds[kWaveLdsEntries];
    const auto stk = make_lane_stack(lds, spill + size_t(blockIdx.x) * kWave, spill_stride);
    trace_stream<MODE_CLOSEST, true>(S, io, n, make_feed(cursors, cursor_count, n), stk, ev);
    flush_events(stats, ev);
}

}  // namespace ptd
