// Device-resident scene layout, re-packed for 16-byte vector loads on gfx950.
//
//  * DNode (64 B, half a 128-B line): one interior node holding BOTH child boxes and both
//    child references, so an interior visit is one 64-B fetch instead of the reference's three
//    dependent 32-B fetches (parent, then the two children; kernel/intersection.h:54,114-121).
//    TLAS and BLAS interior nodes share this format and one array.
//  * DTri (48 B): the three object-space positions of a triangle, stored in leaf order, plus
//    the triangle's index inside its mesh.  Replaces the tri_indices -> indices -> positions
//    chain and the eager normal/uv fetch of kernel/intersection.h:67-84.
//  * DInst (64 B): the 3x4 world->object matrix, the BLAS root reference and the material tag.
//  * DInstFwd (112 B): object->world and inverse-transpose matrices plus light fields, read
//    only when a hit is resolved.
//
// Child reference encoding (32 bit):
//   bit31 = 0          interior: index into the DNode array
//   bit31 = 1, TLAS    leaf: bits[30:0] = GeometryInstance index
//   bit31 = 1, BLAS    leaf: bits[30:28] = triangle count - 1 (0..6) and bits[27:0] = first DTri,
//                      or bits[30:28] = 7 and bits[27:0] = index into big_leaves {first, count}
//   0x7FFFFFFF         "nothing" (used inside the traversal loop; an absent child of a 4-wide node)
// The 4-wide layout of trace mode 2 (DNode4) uses the same references; interior ones index its own array.
#pragma once
#include "../../include/ptamd.h"
#include "dmath.h"

namespace ptd {

constexpr uint32_t kLeafBit = 0x80000000u;
constexpr uint32_t kNoRef = 0x7FFFFFFFu;
constexpr uint32_t kMaxTris = 1u << 28;

struct DNode {
    float4 a;  // left  min.xyz, w = left  ref bits
    float4 b;  // left  max.xyz, w = right ref bits
    float4 c;  // right min.xyz
    float4 d;  // right max.xyz
};
// 4-wide node (trace modes 2 and 3), 128 B = one cache line: the fp32 boxes of four children, stored by
// axis so that one 16-byte load brings the same bound of all four, and the four child references.
// An absent child has NaN bounds (every comparison of the slab test fails) and the reference kNoRef.
struct DNode4 {
    float4 lo_x, hi_x, lo_y, hi_y, lo_z, hi_z;  // component k belongs to child k
    float4 refs;                                // bits: child references
    float4 _pad;
};
struct DTri {
    float4 a;  // p0.xyz, p1.x
    float4 b;  // p1.yz, p2.xy
    float4 c;  // p2.z, bits(triangle index in mesh), 0, 0
};
struct DInst {
    float4 a;  // inv_transform c0.xyz, c1.x
    float4 b;  // c1.yz, c2.xy
    float4 c;  // c2.z, c3.xyz
    float4 d;  // bits: root ref, material id, material type, mesh first_index
};
struct DInstFwd {
    float4 t0, t1, t2;  // transform columns c0..c3 packed as 12 floats
    float4 n0, n1, n2;  // inv_transpose columns c0..c3 packed as 12 floats
    float4 m;           // total_area, bits(area_light_id), bits(triangle_sampler_index), bits(material_id)
};
// Vertex attributes of one triangle record (same index as DTri): what shading a hit needs.
struct DTriShade {
    float n[9];   // object-space normals of the three corners
    float uv[6];  // their texture coordinates
    float _pad;
};
// One emissive instance as the light sampler sees it.
struct DLight {
    float power;
    int32_t gi;          // GeometryInstance index, -1: no geometry (the sample is black)
    uint32_t tri_base;   // first entry of this light in light_alias / light_tris
    uint32_t tri_count;
    float total_area;
    int32_t tex0;        // the emissive material's colour texture
    float emittance;
    uint32_t _pad;
};
// One triangle of a light in world space.
struct DLightTri {
    float p[9];   // corners, transform * position
    float n[9];   // inv_transpose * normal, not normalised (kernel/light.h:35-37)
    float uv[6];
};
// Unified texture descriptor for Texture<float> and Texture<glm::vec3>.
struct DTex {
    int32_t type;
    int32_t image_id;
    float scale[2], offset[2];
    float c0[3], c1[3];  // constant value / checkerboard colours (scalar textures use [0])
    int32_t width, height;
    // the image behind an image texture, copied here so that a lookup goes descriptor -> texels
    // without a third dependent load (null when image_id is out of range)
    const void *img_data;
    int32_t img_width, img_height;
    int32_t filter, _pad;  // kTexFixedPointWeights | kTexSrgbDecode, as chosen with pt_set_texture_filter before the upload
};
// Options of the software texture filter that stands in for the CUDA texture unit (texture.cpp:60-69).
constexpr int32_t kTexFixedPointWeights = 1;  // bilinear weights rounded to 8 fractional bits, as the CUDA unit stores them
constexpr int32_t kTexSrgbDecode = 2;         // texels of images tagged sRGB are linearised before filtering
struct DImage {
    const void *data;  // float (1 channel) or float4 (3 channels + padding)
    int32_t width, height;
    int32_t filter, _pad;  // as DTex::filter
};
struct DAlias {
    const float *prob;
    const uint32_t *alias;
    uint32_t size;
    uint32_t _pad;
};
struct DMaterial {  // pt_material bit layout
    uint32_t type;   // low byte
    int32_t tex0;
    uint32_t w1;     // float or int bits
    int32_t w2;
};
struct DCamera {
    int32_t res_x, res_y;
    float px, py;  // pixel_length
    float focal_length;
    V3 position, front, up, right;
    float near_plane, far_plane;
};

struct DScene {
    const DNode *nodes;
    const DNode4 *nodes4;         // the same trees collapsed to 4-wide nodes (trace mode 2)
    const uint32_t *inst_root4;   // per instance: reference of its BLAS root in nodes4
    const DTri *tris;
    const DInst *insts;
    const DInstFwd *fwd;
    const uint2 *big_leaves;
    const uint32_t *indices;
    const float *positions, *normals, *uvs;
    const DMaterial *materials;
    const DTex *tex1, *tex3;
    const DImage *images1, *images3;
    const pt_area_light *lights;
    const DAlias *tri_samplers;
    DAlias light_sampler;
    const DTriShade *tri_shade;
    const DLight *dlights;
    const uint2 *light_alias;     // {bits(probability), alias} per light triangle
    const DLightTri *light_tris;
    uint32_t tex1_count, tex3_count, image1_count, image3_count;
    uint32_t light_count, material_count, geometry_count;
    uint32_t root_ref;
    uint32_t root_ref4;           // TLAS root in nodes4
    float root_min[3], root_max[3];
    uint32_t max_depth;
    uint32_t any_hit_culls_with_tmax;  // trace mode 1: any-hit searches skip boxes that start beyond the ray's t_max
    float total_light_power;
    DCamera camera;
};

}  // namespace ptd
