// Device shading library: texture lookup, BSDFs, light sampling, hit resolution.
// Each function follows the reference routine cited at it, with glm's operation order.
#pragma once
#include "traverse.h"

namespace ptd {

constexpr float kEps = 1e-5f;                       // Math::EPSILON, math/constants.h:5
constexpr float kPi = 3.14159265358979323846f;      // math/constants.h:8
constexpr float kInvPi = 0.31830988618379067154f;   // math/constants.h:11
constexpr float kInvSqrt3 = 0.57735026919f;         // math/constants.h:13

struct Isect {  // Intersection, kernel/intersection.h:10-17
    float time;
    V3 pos, normal;
    V2 uv;
    int material_id, geometry_id;
};

// ------------------------------------------------------------------ textures
// kernel/texture.h:41-93.  The reference samples image textures through a CUDA texture
// object (bilinear, wrap, normalised coordinates; texture.cpp:60-69).  gfx950 has no texture
// unit, so the filter is defined in software, in fp32:
//   x = s*W - 0.5, i0 = floor(x), fx = x - i0, columns i0 mod W and (i0+1) mod W (rows alike),
//   value = lerp_y(lerp_x(T00,T10), lerp_x(T01,T11)), lerp(a,b,f) = a*(1-f) + b*f.
// Non-finite coordinates return 0.  Two properties of the CUDA unit are open questions without a CUDA
// device to ask, and are options (pt_set_texture_filter) rather than guesses: its bilinear weights
// have 8 fractional bits, and it may or may not linearise fp32 texels of a texture flagged sRGB
// (texture.cpp:67-69 sets the flag on float4 albedo images).  Default: fp32 weights, no decode.
PTD int wrapi(int i, int n) {
    int m = i % n;
    return m < 0 ? m + n : m;
}
PTD void bilinear_setup(const DImage &img, float s, float t, int &i0, int &i1, int &j0, int &j1, float &fx, float &fy) {
    const float x = s * float(img.width) - 0.5f;
    const float y = t * float(img.height) - 0.5f;
    const float fx0 = floorf(x), fy0 = floorf(y);
    fx = x - fx0;
    fy = y - fy0;
    if (img.filter & kTexFixedPointWeights) {
        fx = floorf(fx * 256.0f + 0.5f) * (1.0f / 256.0f);
        fy = floorf(fy * 256.0f + 0.5f) * (1.0f / 256.0f);
    }
    i0 = (fabsf(fx0) < 1.0e9f) ? wrapi(int(fx0), img.width) : 0;
    j0 = (fabsf(fy0) < 1.0e9f) ? wrapi(int(fy0), img.height) : 0;
    i1 = (i0 + 1 == img.width) ? 0 : i0 + 1;
    j1 = (j0 + 1 == img.height) ? 0 : j0 + 1;
}
PTD float srgb_to_linear(float c) { return c <= 0.04045f ? c * (1.0f / 12.92f) : powf((c + 0.055f) * (1.0f / 1.055f), 2.4f); }
PTD float4 srgb_decode(float4 t, int32_t filter) {
    if (!(filter & kTexSrgbDecode)) return t;
    return make_float4(srgb_to_linear(t.x), srgb_to_linear(t.y), srgb_to_linear(t.z), t.w);
}
PTD float lerp2(float t00, float t10, float t01, float t11, float fx, float fy) {
    const float a = t00 * (1.0f - fx) + t10 * fx;
    const float b = t01 * (1.0f - fx) + t11 * fx;
    return a * (1.0f - fy) + b * fy;
}
PTD V3 sample_image3(const DImage &img, float s, float t) {
    if (!isfinite(s) || !isfinite(t) || img.width <= 0 || img.height <= 0) return splat(0.0f);
    int i0, i1, j0, j1;
    float fx, fy;
    bilinear_setup(img, s, t, i0, i1, j0, j1, fx, fy);
    const float4 *d = static_cast<const float4 *>(img.data);
    const float4 t00 = srgb_decode(d[size_t(j0) * img.width + i0], img.filter), t10 = srgb_decode(d[size_t(j0) * img.width + i1], img.filter);
    const float4 t01 = srgb_decode(d[size_t(j1) * img.width + i0], img.filter), t11 = srgb_decode(d[size_t(j1) * img.width + i1], img.filter);
    return mk3(lerp2(t00.x, t10.x, t01.x, t11.x, fx, fy), lerp2(t00.y, t10.y, t01.y, t11.y, fx, fy),
               lerp2(t00.z, t10.z, t01.z, t11.z, fx, fy));
}
PTD float sample_image1(const DImage &img, float s, float t) {
    if (!isfinite(s) || !isfinite(t) || img.width <= 0 || img.height <= 0) return 0.0f;
    int i0, i1, j0, j1;
    float fx, fy;
    bilinear_setup(img, s, t, i0, i1, j0, j1, fx, fy);
    const float *d = static_cast<const float *>(img.data);
    return lerp2(d[size_t(j0) * img.width + i0], d[size_t(j0) * img.width + i1], d[size_t(j1) * img.width + i0],
                 d[size_t(j1) * img.width + i1], fx, fy);
}
PTD V3 sample_tex3(const DScene &S, int id, V2 uv) {
    if (id < 0 || uint32_t(id) >= S.tex3_count) return splat(0.0f);
    const DTex &t = S.tex3[id];
    if (t.type == PT_TEX_CONSTANT) return mk3(t.c0[0], t.c0[1], t.c0[2]);
    const float s = uv.x * t.scale[0] + t.offset[0];
    const float tt = uv.y * t.scale[1] + t.offset[1];
    if (t.type == PT_TEX_CHECKERBOARD) {
        const int i = int(floorf(s)), j = int(floorf(tt));
        return ((i + j) & 1) ? mk3(t.c1[0], t.c1[1], t.c1[2]) : mk3(t.c0[0], t.c0[1], t.c0[2]);
    }
    if (t.type == PT_TEX_IMAGE) {
        if (!t.img_data) return splat(0.0f);
        return sample_image3(DImage{t.img_data, t.img_width, t.img_height, t.filter, 0}, s, tt);
    }
    return splat(0.0f);
}
PTD float sample_tex1(const DScene &S, int id, V2 uv) {
    if (id < 0 || uint32_t(id) >= S.tex1_count) return 0.0f;
    const DTex &t = S.tex1[id];
    if (t.type == PT_TEX_CONSTANT) return t.c0[0];
    const float s = uv.x * t.scale[0] + t.offset[0];
    const float tt = uv.y * t.scale[1] + t.offset[1];
    if (t.type == PT_TEX_CHECKERBOARD) {
        const int i = int(floorf(s)), j = int(floorf(tt));
        return ((i + j) & 1) ? t.c1[0] : t.c0[0];
    }
    if (t.type == PT_TEX_IMAGE) {
        if (!t.img_data) return 0.0f;
        return sample_image1(DImage{t.img_data, t.img_width, t.img_height, t.filter, 0}, s, tt);
    }
    return 0.0f;
}

// --------------------------------------------------------- hit resolution
PTD V3 ld3(const float *p, uint32_t i) { return mk3(p[3 * size_t(i)], p[3 * size_t(i) + 1], p[3 * size_t(i) + 2]); }
PTD V2 ld2(const float *p, uint32_t i) { return V2{p[2 * size_t(i)], p[2 * size_t(i) + 1]}; }

// Expands (triangle record, barycentrics, instance) into the reference's Intersection:
// interpolated attributes in object space (kernel/intersection.h:96-99), then world position,
// world distance and normalised inverse-transpose normal (:145-149).
PTD Isect resolve_hit(const DScene &S, V3 ray_o, V3 ray_d, float u, float v, uint32_t rec, int inst) {
    Isect is;
    const float4 *tp = reinterpret_cast<const float4 *>(S.tris + rec);
    const float4 ta = tp[0], tb = tp[1], tc = tp[2];
    const V3 p0 = mk3(ta.x, ta.y, ta.z), p1 = mk3(ta.w, tb.x, tb.y), p2 = mk3(tb.z, tb.w, tc.x);
    const uint32_t tri = as_u(tc.y);
    const float4 *ip = reinterpret_cast<const float4 *>(S.insts + inst);
    const float4 id = ip[3];
    const uint32_t first_index = as_u(id.w);
    const uint32_t i0 = S.indices[first_index + tri * 3u + 0], i1 = S.indices[first_index + tri * 3u + 1],
                   i2 = S.indices[first_index + tri * 3u + 2];
    const V3 n0 = ld3(S.normals, i0), n1 = ld3(S.normals, i1), n2 = ld3(S.normals, i2);
    const V2 uv0 = ld2(S.uvs, i0), uv1 = ld2(S.uvs, i1), uv2 = ld2(S.uvs, i2);
    const float w = 1.0f - u - v;
    const V3 best_normal = normalize(w * n0 + u * n1 + v * n2);
    const V2 best_uv = w * uv0 + u * uv1 + v * uv2;
    const V3 best_pos = w * p0 + u * p1 + v * p2;
    const float4 *fp = reinterpret_cast<const float4 *>(S.fwd + inst);
    const M34 T = load_m34(fp[0], fp[1], fp[2]);
    const M34 N = load_m34(fp[3], fp[4], fp[5]);
    is.pos = xform(T, best_pos, 1.0f);
    is.uv = best_uv;
    is.time = dot(is.pos - ray_o, ray_d);
    is.normal = normalize(xform(N, best_normal, 0.0f));
    is.material_id = int(as_u(id.y));
    is.geometry_id = inst;
    return is;
}

// ---------------------------------------------------------------- sampling
PTD V3 to_world(V3 local, V3 normal) {  // math/sampling.h:7-20
    V3 up;
    if (fabsf(normal.x) < kInvSqrt3) up = mk3(1, 0, 0);
    else if (fabsf(normal.y) < kInvSqrt3) up = mk3(0, 1, 0);
    else up = mk3(0, 0, 1);
    const V3 tangent = normalize(cross(normal, up));
    const V3 bitangent = normalize(cross(normal, tangent));
    return tangent * local.x + bitangent * local.y + normal * local.z;
}
PTD V3 sample_cosine_hemisphere(V3 normal, float sx, float sy) {  // math/sampling.h:30-38
    const float z = sqrtf(sx);
    const float phi = 2 * kPi * sy;
    const float r = sqrtf(gmax(0.0f, 1.0f - z * z));
    float sin_phi, cos_phi;
    sincos_azimuth(phi, sin_phi, cos_phi);
    return to_world(mk3(r * cos_phi, r * sin_phi, z), normal);
}
PTD V3 sample_ggx(V3 normal, float roughness, float sx, float sy) {  // math/sampling.h:40-58
    const float a2 = roughness * roughness;
    const float phi = 2.0f * kPi * sx;
    const float cos_theta = sqrtf((1.0f - sy) / (1.0f + (a2 - 1.0f) * sy));
    const float sin_theta = sqrtf(gmax(0.0f, 1.0f - cos_theta * cos_theta));
    float sin_phi, cos_phi;
    sincos_azimuth(phi, sin_phi, cos_phi);
    return to_world(mk3(sin_theta * cos_phi, sin_theta * sin_phi, cos_theta), normal);
}
PTD float power_heuristic(float f, float g) { return (f * f) / (f * f + g * g); }                 // sampling.h:70-72
PTD float luminance(V3 c) { return c.x * 0.212671f + c.y * 0.715160f + c.z * 0.072169f; }         // color.h:7-9
PTD V3 reflect(V3 v, V3 n) { return 2 * dot(v, n) * n - v; }                                      // geometry.h:7-9
PTD bool refract(V3 vec, V3 normal, float eta, V3 &result) {                                      // geometry.h:11-24
    const float cos_in = dot(normal, vec);
    const float sin2_in = fmaxf(0.f, 1.f - cos_in * cos_in);
    const float sin2_t = sin2_in * eta * eta;
    if (sin2_t >= 1.f) return false;
    const float cos_t = fmaxf(0.0f, sqrtf(1.0f - sin2_t));
    result = -vec * eta + (cos_in * eta - cos_t) * normal;
    return true;
}

// alias-table draw, kernel/sampler.h:57-70
PTD uint32_t discrete_sample(const DAlias &t, uint64_t &rng) {
    const float x = sample1d(rng);
    float scaled = x * float(t.size);
    uint32_t i = uint32_t(scaled);
    if (scaled >= float(t.size)) {
        i = t.size - 1;
        scaled = float(i);
    }
    const float y = scaled - float(i);
    return (y < t.prob[i]) ? i : t.alias[i];
}

// -------------------------------------------------------------------- BSDFs
enum : uint32_t { S_Invalid = 0, S_Diffuse = 1, S_Glossy = 2, S_Specular = 4, S_Reflection = 16, S_Transmission = 32 };
struct BsdfSample {  // kernel/bsdf/common.h:23-28
    uint32_t type;
    float pdf;
    V3 bsdf, w_in;
};
PTD float pow5(float b) {
    const float b2 = b * b;
    return b2 * b2 * b;
}
PTD float trowbridge_reitz(V3 normal, V3 micro, float alpha) {  // bsdf/common.h:37-50
    const float cos_in = dot(normal, micro);
    if (cos_in <= 0.0f) return 0.0f;
    const float a2 = alpha * alpha;
    const float denom_term = fmaf((a2 - 1.0f), cos_in * cos_in, 1.0f);
    const float denom = denom_term * denom_term + 1e-16f;
    const float result = (a2 * kInvPi) / denom;
    return isfinite(result) ? result : 0.0f;
}
PTD float smith_g1(V3 normal, V3 v, float roughness) {  // bsdf/common.h:56-64
    const float cos_in = dot(normal, v);
    if (cos_in <= 0.0f) return 0.0f;
    const float a2 = roughness * roughness;
    const float denom = cos_in + sqrtf(a2 + (1.0f - a2) * cos_in * cos_in);
    return 2 * cos_in / denom;
}
PTD float smith_g(V3 n, V3 a, V3 b, float r) { return smith_g1(n, a, r) * smith_g1(n, b, r); }
PTD float fresnel(float cos_in, float eta) {  // bsdf/common.h:80-92
    const float sin2_in = 1 - cos_in * cos_in;
    const float sin2_t = sin2_in * eta * eta;
    if (sin2_t >= 1.0f) return 1.0f;
    const float cos_t = fmaxf(0.0f, sqrtf(1 - sin2_t));
    const float parallel = (cos_in - eta * cos_t) / (cos_in + eta * cos_t);
    const float perp = (eta * cos_in - cos_t) / (eta * cos_in + cos_t);
    return (parallel * parallel + perp * perp) / 2.0f;
}
PTD V3 fresnel_schlick(float cos_in, V3 f0) { return f0 + (1.0f - f0) * pow5(1.0f - cos_in); }  // bsdf/common.h:100-102

struct MRParams {
    V3 base_color;
    float metallic, alpha;
};
PTD MRParams mr_fetch(const DScene &S, const DMaterial &m, const Isect &is) {
    MRParams p;
    p.base_color = sample_tex3(S, m.tex0, is.uv);
    p.metallic = sample_tex1(S, int(m.w1), is.uv);
    float alpha = sample_tex1(S, m.w2, is.uv);
    alpha = gmin(gmax(alpha, 0.02f), 1.0f);
    p.alpha = alpha * alpha;
    return p;
}

PTD BsdfSample sample_lambertian(V3 albedo, const Isect &is, uint64_t &rng) {  // lambertian.h:5-23
    const float sx = sample1d(rng), sy = sample1d(rng);
    const V3 w_in = sample_cosine_hemisphere(is.normal, sx, sy);
    BsdfSample r;
    r.bsdf = albedo * kInvPi;
    r.pdf = fmaxf(0.0f, dot(is.normal, w_in)) * kInvPi;
    r.type = S_Diffuse;
    r.w_in = w_in;
    return r;
}
PTD BsdfSample sample_mirror(V3 albedo, const Isect &is, V3 w_out) {  // perfect_mirror.h:5-23
    const V3 w_in = reflect(w_out, is.normal);
    const float cos_in = fabsf(dot(is.normal, w_in));
    BsdfSample r;
    r.pdf = 1.0f;
    r.w_in = w_in;
    r.type = S_Specular | S_Reflection;
    r.bsdf = albedo / cos_in;
    return r;
}
PTD BsdfSample sample_dielectric(V3 color, const DMaterial &m, const Isect &is, V3 w_out, uint64_t &rng) {  // perfect_dielectric.h:5-66
    BsdfSample result;
    result.type = S_Invalid;
    result.pdf = 0.0f;
    result.bsdf = splat(0.0f);
    result.w_in = splat(0.0f);
    V3 normal = is.normal;
    const float ior = as_f(m.w1);
    float eta = 1.0f / ior;
    float cos_out = dot(is.normal, w_out);
    if (cos_out < 0.0f) {
        normal = -normal;
        cos_out = -cos_out;
        eta = ior;
    }
    const float reflectance = fresnel(cos_out, eta);
    const float transmittance = 1.0f - reflectance;
    if (sample1d(rng) < reflectance) {
        const V3 w_in = reflect(w_out, normal);
        const float cos_in = dot(normal, normalize(w_in));
        if (cos_in < 0.0f) return result;
        result.type = S_Specular | S_Reflection;
        result.w_in = w_in;
        result.bsdf = color * reflectance / cos_in;
        result.pdf = reflectance;
        return result;
    }
    V3 w_in;
    if (!refract(w_out, normal, eta, w_in)) return result;
    float cos_in = dot(w_in, normal);
    if (cos_in > 0.0f) return result;
    cos_in = fabsf(cos_in);
    result.type = S_Specular | S_Transmission;
    result.w_in = w_in;
    result.bsdf = color * transmittance * eta * eta / cos_in;
    result.pdf = transmittance;
    return result;
}
PTD BsdfSample sample_metallic_roughness(const MRParams &p, const Isect &is, V3 w_out, uint64_t &rng) {  // metallic_roughness.h:8-90
    BsdfSample result;
    result.type = S_Invalid;
    result.pdf = 0.0f;
    result.bsdf = splat(0.0f);
    result.w_in = splat(0.0f);
    V3 normal = is.normal;
    float cos_out = dot(normal, w_out);
    if (cos_out < 0.0f) {
        normal = -normal;
        cos_out = -cos_out;
    }
    const V3 F0 = mix3(splat(0.04f), p.base_color, p.metallic);
    float w_diff = luminance(p.base_color) * (1.0f - p.metallic);
    float w_spec = luminance(F0);
    const float w_sum = gmax(w_spec + w_diff, kEps);
    w_diff /= w_sum;
    w_spec /= w_sum;
    if (sample1d(rng) < w_spec) {
        const float sx = sample1d(rng), sy = sample1d(rng);
        const V3 half_vec = sample_ggx(normal, p.alpha, sx, sy);
        const V3 w_in = reflect(w_out, half_vec);
        const float cos_in = dot(normal, w_in);
        if (cos_in <= 0.0f) return result;
        const float D = trowbridge_reitz(normal, half_vec, p.alpha);
        const float G = smith_g(normal, w_out, w_in, p.alpha);
        const V3 refl_in = fresnel_schlick(fmaxf(dot(w_out, half_vec), 0.0f), F0);
        const float pdf_specular = (D * fmaxf(dot(normal, half_vec), 0.0f)) / (4.0f * fmaxf(fabsf(dot(w_out, half_vec)), kEps));
        result.type = S_Glossy | S_Reflection;
        result.bsdf = refl_in * (D * G) / fmaxf(4.0f * cos_in * cos_out, 1e-6f);
        result.pdf = w_spec * pdf_specular;
        result.w_in = w_in;
        return result;
    }
    const float sx = sample1d(rng), sy = sample1d(rng);
    const V3 w_in = sample_cosine_hemisphere(normal, sx, sy);
    const float cos_in = fmaxf(dot(normal, w_in), 0.0f);
    if (cos_in <= 0.0f) return result;
    const V3 sum_dir = w_in + w_out;
    const V3 half_vec = (length2(sum_dir) >= kEps) ? normalize(sum_dir) : normal;
    const float cos_f = fabsf(dot(w_out, half_vec));
    const V3 refl_out = fresnel_schlick(cos_f, F0);
    result.type = S_Diffuse | S_Reflection;
    result.w_in = w_in;
    result.bsdf = p.base_color * (1.0f - p.metallic) * (1.0f - refl_out) * kInvPi;
    result.pdf = w_diff * cos_in * kInvPi;
    return result;
}
PTD V3 eval_metallic_roughness(const MRParams &p, const Isect &is, V3 w_in, V3 w_out) {  // metallic_roughness.h:92-127
    const V3 normal = is.normal;
    const float cos_out = dot(normal, w_out);
    const float cos_in = dot(normal, w_in);
    if (cos_out <= 0.0f || cos_in <= 0.0f) return splat(0.0f);
    const V3 half_vec = normalize(w_in + w_out);
    const V3 F0 = mix3(splat(0.04f), p.base_color, p.metallic);
    const float D = trowbridge_reitz(normal, half_vec, p.alpha);
    const float G = smith_g(normal, w_out, w_in, p.alpha);
    const V3 refl_in = fresnel_schlick(dot(w_in, half_vec), F0);
    const V3 specular = refl_in * (D * G) / gmax(4.0f * cos_in * cos_out, kEps);
    const V3 refl_out = fresnel_schlick(dot(w_out, half_vec), F0);
    const V3 diffuse = p.base_color * (1.0f - p.metallic) * (1.0f - refl_out) * (1.0f - refl_in) * kInvPi;
    return diffuse + specular;
}
PTD float metallic_roughness_pdf(const MRParams &p, const Isect &is, V3 w_in, V3 w_out) {  // metallic_roughness.h:129-177
    const V3 normal = is.normal;
    const float cos_out = dot(normal, w_out);
    const float cos_in = dot(normal, w_in);
    if (cos_out <= 0.0f || cos_in <= 0.0f) return 0.0f;
    const V3 F0 = mix3(splat(0.04f), p.base_color, p.metallic);
    float w_diff = luminance(p.base_color) * (1.0f - p.metallic);
    float w_spec = luminance(F0);
    const float w_sum = gmax(w_spec + w_diff, kEps);
    w_diff /= w_sum;
    w_spec /= w_sum;
    const float pdf_diffuse = cos_in * kInvPi;
    float pdf_specular = 0.0f;
    V3 half_vec = w_in + w_out;
    if (length2(half_vec) >= kEps) {
        half_vec = normalize(half_vec);
        const float n_dot_h = gmax(dot(normal, half_vec), 0.0f);
        const float out_dot_h = gmax(dot(w_out, half_vec), 0.0f);
        if (n_dot_h > 0.0f && out_dot_h > 0.0f) {
            const float D = trowbridge_reitz(normal, half_vec, p.alpha);
            pdf_specular = (D * n_dot_h) / gmax(4.0f * out_dot_h, kEps);
        }
    }
    return w_diff * pdf_diffuse + w_spec * pdf_specular;
}

// ------------------------------------------------------------------- lights
struct LightSample {  // kernel/light.h:8-13
    V3 position, normal, radiance;
    float pdf;
};
PTD LightSample sample_scene_light(const DScene &S, uint64_t &rng) {  // kernel/light.h:15-59
    LightSample ls;
    ls.position = ls.normal = ls.radiance = splat(0.0f);
    ls.pdf = 0.0f;
    const uint32_t light_id = discrete_sample(S.light_sampler, rng);
    const pt_area_light light = S.lights[light_id];
    if (light.geometry_instance_index == -1) return ls;
    const float p_pick = light.power / S.total_light_power;
    const int gi = light.geometry_instance_index;
    const float4 *fp = reinterpret_cast<const float4 *>(S.fwd + gi);
    const M34 T = load_m34(fp[0], fp[1], fp[2]);
    const M34 N = load_m34(fp[3], fp[4], fp[5]);
    const float4 fm = fp[6];
    const float total_area = fm.x;
    const int sampler_index = int(as_u(fm.z));
    const int material_id = int(as_u(fm.w));
    const uint32_t first_index = as_u(reinterpret_cast<const float4 *>(S.insts + gi)[3].w);
    const uint32_t tri = discrete_sample(S.tri_samplers[sampler_index], rng);
    const uint32_t i0 = S.indices[first_index + tri * 3 + 0], i1 = S.indices[first_index + tri * 3 + 1],
                   i2 = S.indices[first_index + tri * 3 + 2];
    const V3 p0 = xform(T, ld3(S.positions, i0), 1.0f), p1 = xform(T, ld3(S.positions, i1), 1.0f),
             p2 = xform(T, ld3(S.positions, i2), 1.0f);
    const V3 n0 = xform(N, ld3(S.normals, i0), 0.0f), n1 = xform(N, ld3(S.normals, i1), 0.0f),
             n2 = xform(N, ld3(S.normals, i2), 0.0f);
    const V2 t0 = ld2(S.uvs, i0), t1 = ld2(S.uvs, i1), t2 = ld2(S.uvs, i2);
    const float sx = sample1d(rng), sy = sample1d(rng);
    const float sqr = sqrtf(sx);  // Math::SampleTriangle, math/sampling.h:61-68
    const float u = 1.0f - sqr;
    const float v = sy * sqr;
    const float w = 1.0f - u - v;
    ls.position = u * p0 + v * p1 + w * p2;
    ls.normal = u * n0 + v * n1 + w * n2;  // un-normalised on purpose (kernel/light.h:47,54)
    const V2 texcoord = u * t0 + v * t1 + w * t2;
    const DMaterial m = S.materials[material_id];
    ls.radiance = sample_tex3(S, m.tex0, texcoord) * as_f(m.w1);
    ls.pdf = p_pick * (1.0f / total_area);
    return ls;
}

// ------------------------------------------------ lookups split into load levels
// The same arithmetic as sample_tex3 / sample_tex1 / discrete_sample above, cut at every point
// where a load depends on an earlier one, so that a caller can request all loads of one level
// (for several textures and for the light chain) before it waits for any of them.
struct TexPlan {
    int mode;  // 0: value 0, 1: c0, 2: c1, 3: image (four taps at the offsets below)
    uint32_t o00, o10, o01, o11;
    float fx, fy;
    int32_t filter;
};
PTD TexPlan tex_plan(const DTex &t, bool valid, V2 uv) {
    TexPlan p;
    p.mode = 0;
    p.o00 = p.o10 = p.o01 = p.o11 = 0;
    p.fx = p.fy = 0.0f;
    p.filter = 0;
    if (!valid) return p;
    if (t.type == PT_TEX_CONSTANT) {
        p.mode = 1;
        return p;
    }
    const float s = uv.x * t.scale[0] + t.offset[0];
    const float tt = uv.y * t.scale[1] + t.offset[1];
    if (t.type == PT_TEX_CHECKERBOARD) {
        const int i = int(floorf(s)), j = int(floorf(tt));
        p.mode = ((i + j) & 1) ? 2 : 1;
        return p;
    }
    if (t.type != PT_TEX_IMAGE || !t.img_data) return p;
    if (!isfinite(s) || !isfinite(tt) || t.img_width <= 0 || t.img_height <= 0) return p;
    int i0, i1, j0, j1;
    bilinear_setup(DImage{t.img_data, t.img_width, t.img_height, t.filter, 0}, s, tt, i0, i1, j0, j1, p.fx, p.fy);
    p.filter = t.filter;
    p.o00 = uint32_t(j0) * uint32_t(t.img_width) + uint32_t(i0);
    p.o10 = uint32_t(j0) * uint32_t(t.img_width) + uint32_t(i1);
    p.o01 = uint32_t(j1) * uint32_t(t.img_width) + uint32_t(i0);
    p.o11 = uint32_t(j1) * uint32_t(t.img_width) + uint32_t(i1);
    p.mode = 3;
    return p;
}
PTD V3 tex3_value(const TexPlan &p, const DTex &t, float4 t00, float4 t10, float4 t01, float4 t11) {
    if (p.mode == 1) return mk3(t.c0[0], t.c0[1], t.c0[2]);
    if (p.mode == 2) return mk3(t.c1[0], t.c1[1], t.c1[2]);
    if (p.mode == 3) {
        t00 = srgb_decode(t00, p.filter); t10 = srgb_decode(t10, p.filter); t01 = srgb_decode(t01, p.filter); t11 = srgb_decode(t11, p.filter);
        return mk3(lerp2(t00.x, t10.x, t01.x, t11.x, p.fx, p.fy), lerp2(t00.y, t10.y, t01.y, t11.y, p.fx, p.fy),
                   lerp2(t00.z, t10.z, t01.z, t11.z, p.fx, p.fy));
    }
    return splat(0.0f);
}
PTD float tex1_value(const TexPlan &p, const DTex &t, float t00, float t10, float t01, float t11) {
    if (p.mode == 1) return t.c0[0];
    if (p.mode == 2) return t.c1[0];
    if (p.mode == 3) return lerp2(t00, t10, t01, t11, p.fx, p.fy);
    return 0.0f;
}

// A texture lookup up to (not including) its texel reads: the descriptor is read here and
// boiled down to what the second half needs, so that it does not stay in registers.
struct TexReq {
    TexPlan p;
    const void *data;  // texels of an image texture
    V3 cval;           // value of a constant / checkerboard texture (scalar textures: .x)
};
PTD TexReq tex_request(const DTex *table, uint32_t count, int id, V2 uv) {
    TexReq r;
    r.data = nullptr;
    r.cval = splat(0.0f);
    const bool valid = id >= 0 && uint32_t(id) < count;
    const DTex &t = table[valid ? id : 0];
    r.p = tex_plan(t, valid && count > 0, uv);
    if (r.p.mode == 1) r.cval = mk3(t.c0[0], t.c0[1], t.c0[2]);
    else if (r.p.mode == 2) r.cval = mk3(t.c1[0], t.c1[1], t.c1[2]);
    else if (r.p.mode == 3) r.data = t.img_data;
    return r;
}
PTD V3 tex3_finish(const TexReq &r, float4 t00, float4 t10, float4 t01, float4 t11) {
    if (r.p.mode == 3) {
        t00 = srgb_decode(t00, r.p.filter); t10 = srgb_decode(t10, r.p.filter); t01 = srgb_decode(t01, r.p.filter); t11 = srgb_decode(t11, r.p.filter);
        return mk3(lerp2(t00.x, t10.x, t01.x, t11.x, r.p.fx, r.p.fy), lerp2(t00.y, t10.y, t01.y, t11.y, r.p.fx, r.p.fy),
                   lerp2(t00.z, t10.z, t01.z, t11.z, r.p.fx, r.p.fy));
    }
    return r.cval;  // zero when the lookup is invalid
}
PTD float tex1_finish(const TexReq &r, float t00, float t10, float t01, float t11) {
    if (r.p.mode == 3) return lerp2(t00, t10, t01, t11, r.p.fx, r.p.fy);
    return r.cval.x;
}

struct AliasPick {  // kernel/sampler.h:57-70, up to the table reads
    uint32_t i;
    float y;
};
PTD AliasPick alias_begin(uint32_t size, float x) {
    float scaled = x * float(size);
    uint32_t i = uint32_t(scaled);
    if (scaled >= float(size)) {
        i = size - 1;
        scaled = float(i);
    }
    return AliasPick{i, scaled - float(i)};
}
PTD uint32_t alias_end(AliasPick p, float prob_i, uint32_t alias_i) { return (p.y < prob_i) ? p.i : alias_i; }

struct V3x3 {
    V3 a, b, c;
};
PTD V3x3 ld3x3(const float *p, uint32_t i0, uint32_t i1, uint32_t i2) { return V3x3{ld3(p, i0), ld3(p, i1), ld3(p, i2)}; }

// math/tonemap.h:10-14
PTD float aces(float x) {
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    return fminf(fmaxf((x * (a * x + b)) / (x * (c * x + d) + e), 0.0f), 1.0f);
}

}  // namespace ptd
