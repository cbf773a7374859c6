// jpt_types.h -- wire formats at the boundary (byte-compatible with the reference)
// and the flattened device layout the native kernels traverse.
#pragma once

#include <stddef.h>
#include <stdint.h>

namespace jpt {

// ---- reference wire formats -------------------------------------------------------------------

struct Vec4 {  // BVH::vec4, src/bvh/vec.h:45-51
    float x, y, z, w;
};

struct RefTriangle {  // BVH::Triangle, src/bvh/bvh.h:22-29 (builder-internal)
    Vec4 vertices[3];
    Vec4 centroid;
    Vec4 normals[3];
    float uvs[3][2];
    uint32_t materialIndex;
    uint32_t _pad;
};

struct RefBvhNode {  // BVH::BVHNode, src/bvh/bvh.h:46-54 == main.glsl:45-52
    Vec4 aabbMin;
    Vec4 aabbMax;
    uint32_t left_child;
    uint32_t right_child;
    uint32_t first_tri_index;
    uint32_t tri_count;
};

struct RefTlasNode {  // BVH::TLASNode, src/bvh/bvh.h:56-62 == main.glsl:54-60
    float aabbMin[3];
    uint32_t leftRight;  // lo16 = left, hi16 = right, 0 => leaf
    float aabbMax[3];
    uint32_t blas;
};

struct RefInstance {  // BVH::BLASInstance, src/bvh/bvh.h:64-72 == main.glsl:84-93
    float transform[16];
    float inverse_transform[16];
    Vec4 aabbMin;
    Vec4 aabbMax;
    uint32_t blas_index;
    uint32_t material[3];
};

struct RefTriGeometry {  // GpuTriangleGeometry, render_parameters.h:59-62 == main.glsl:14-16
    Vec4 vertices[3];
};

struct RefTriData {  // GpuTriangleData, render_parameters.h:64-71 == main.glsl:18-24
    float n0[3];
    uint32_t material_index;
    Vec4 n1;
    Vec4 n2;
    float uvs[3][2];
    uint32_t _pad[2];
};

struct RefMaterial {  // GpuMaterial, render_parameters.h:49-57 == main.glsl:32-43
    Vec4 albedo;
    Vec4 emission;  // rgb colour, w = energy multiplier
    float metallic;
    float roughness;
    int32_t albedo_texture_index;
    float padding[5];
};

struct RefCamera {  // Camera, render_parameters.h:14-21 == main.glsl:110-117
    float vp[16];
    float ivp[16];
    Vec4 position;
    uint32_t frame_index;
    float near_;
    float far_;
    uint32_t _pad;
};

struct RefTemporalParams {  // TemporalReprojection::RenderParameters, temporal_reprojection.h:16-23 == temporal_reprojection.glsl:4-11
    float deltaMatrix[16];
    int32_t width, height;
    uint32_t frame_count;
    float blendFactor;  // not read by the shader (literal 0.75, temporal_reprojection.glsl:64)
    float nearPlane, farPlane;
};

static_assert(sizeof(RefTemporalParams) == 88, "TemporalReprojection::RenderParameters");
static_assert(sizeof(RefTriangle) == 144, "Triangle");
static_assert(sizeof(RefBvhNode) == 48, "BVHNode");
static_assert(sizeof(RefTlasNode) == 32, "TLASNode");
static_assert(sizeof(RefInstance) == 176, "BLASInstance");
static_assert(sizeof(RefTriGeometry) == 48, "GpuTriangleGeometry");
static_assert(sizeof(RefTriData) == 80, "GpuTriangleData");
static_assert(sizeof(RefMaterial) == 64, "GpuMaterial");
static_assert(sizeof(RefCamera) == 160, "Camera");

// ---- flattened device layout (native kernels) -------------------------------------------------
//
// One 64-byte record per internal node carrying BOTH children's boxes and references, so one
// expansion is a single aligned 64-B fetch (4 x dwordx4 per lane) instead of the reference's three
// 48-B node reads (main.glsl:277,286-287).  The same record serves BLAS and TLAS levels.
//
// Child reference encoding (int32):
//   ref >= 0          internal node, index into the node array of its level
//   ref <  0, BLAS    leaf: first triangle = ~ref (bits 0..24), count - 1 in bits 25..30
//   ref <  0, TLAS    leaf: instance id = ~ref
struct alignas(64) WideNode {
    float lmin[3], lmax[3];
    float rmin[3], rmax[3];
    int32_t left, right;
    uint32_t _pad[2];
};
static_assert(sizeof(WideNode) == 64, "WideNode");

// 4-wide record of the native route (128 B, one L1 line): up to four children's boxes, struct-of-arrays so
// the four slab tests read whole float4s, plus four child references (same encoding as WideNode;
// kEmptyChild marks an unused slot).  Built by collapsing the binary tree (flatten4): a ray makes about half
// as many dependent fetches as on the two-child records.
// Byte layout: the three lo planes in the first 64 bytes (with the child references), the three hi planes 64 bytes
// further on, so "the plane of axis a a ray meets first / last" is at a * 16 + (direction negative ? 64 : 0) and that
// offset XOR 64: the walk fetches near and far planes directly and never compares the two planes of an axis.
struct alignas(128) WideNode4 {
    float lo_x[4], lo_y[4], lo_z[4];
    int32_t child[4];
    float hi_x[4], hi_y[4], hi_z[4];
    uint32_t _pad[4];
};
static_assert(offsetof(WideNode4, child) == 48 && offsetof(WideNode4, hi_x) == 64, "WideNode4 plane offsets");
static_assert(sizeof(WideNode4) == 128, "WideNode4");
constexpr int32_t kEmptyChild = (int32_t)0x80000000;  // ~0x7fffffff: never a valid leaf reference

// 48-byte triangle record: v0 and the two Moller-Trumbore edges, precomputed with the same float
// subtractions intersectTriangle performs (main.glsl:231-232), so t,u,v are bit-identical.
// The fourth words hold cross(e1, e2) (main.glsl:253's geometric normal, the same three float expressions), so the
// front/back test of a triangle test is one dot product.  Records are in reference order: triangle i of the
// shading arrays (tri_data) is record i.
struct alignas(16) WideTri {
    float v0[3];
    float nx;
    float e1[3];
    float ny;
    float e2[3];
    float nz;
};
static_assert(sizeof(WideTri) == 48, "WideTri");

// 64-byte shading record of a triangle: the 72 bytes get_shading_data needs of the 80-byte GpuTriangleData (vertex normals,
// uvs, material slot) without its padding -- ONE aligned cache line per shaded hit, where the 80-byte stride always
// straddles two.  Same values, same order of use; made on upload from the reference-layout array, index for index.
struct alignas(64) ShadeTri {
    float n0[3];
    float n1[3];
    float n2[3];
    float uvs[3][2];
    uint32_t material_index;
};
static_assert(sizeof(ShadeTri) == 64, "ShadeTri");

// 64-byte hot instance record (traversal): rows of the 3x4 inverse transform + BLAS root reference.
struct alignas(64) WideInstance {
    float inv[12];   // column-major 3x4: c0.xyz, c1.xyz, c2.xyz, c3.xyz
    int32_t root;    // child reference of the BLAS root (may be a leaf)
    uint32_t _pad[3];
};
static_assert(sizeof(WideInstance) == 64, "WideInstance");

// ---- reach records: which triangles can the REFERENCE's own traversal reach? ---------------------------------
//
// The boxes of the reference's trees are nested (a node's box is the exact min/max over a superset of its child's
// vertices, bvh.cpp:19-37,299-303) and every operation of intersectAABB (main.glsl:259-268) is monotone in the box
// planes, so a ray that passes a leaf's box passes every ancestor's box: ray_trace_tlas / ray_trace_blas test triangle
// T of instance I if and only if the world ray passes I's world box (the TLAS leaf) and the local ray passes the box
// of the BLAS leaf that holds T (oracle/jpt_oracle.h, JPTO_FLAG_REACH_ONLY; up to the `d < hitInfo.t` culls).  Float
// rounding makes those tests fail for a few rays in 10^7 whose triangle test succeeds ("cracks": the reference's ray
// goes through).  The native builder records the two boxes, so that a hit found on ITS tree can be checked against
// what the reference would have reached, and the image equals the reference's, cracks included.
struct alignas(16) ReachTri {   // per triangle (device order): the box of the reference BLAS leaf that holds it
    float lo[3];
    uint32_t always;            // 1: the reference's BLAS root is this leaf -- never box-tested (main.glsl:272-283)
    float hi[3];
    uint32_t _pad;
};
static_assert(sizeof(ReachTri) == 32, "ReachTri");
struct alignas(64) ReachInst {  // per instance
    float lo[3];                // world box the reference's BLASInstance::update_aabb computes (bvh.h:90-115) ...
    uint32_t _p0;
    float hi[3];
    uint32_t _p1;
    float root_lo[3];           // ... from the box of the reference BLAS root (kept for refits on the device)
    uint32_t _p2;
    float root_hi[3];
    uint32_t _p3;
};
static_assert(sizeof(ReachInst) == 64, "ReachInst");

constexpr int kLeafCountShift = 25;
constexpr uint32_t kLeafFirstMask = (1u << kLeafCountShift) - 1u;
constexpr int kMaxLeafTris = 64;

}  // namespace jpt
