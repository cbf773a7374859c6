/*
 * jpt_oracle.h -- CPU restatement (plain C99) of the GDPathTracing hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under gdpathtracing_amd/ may include,
 * link or call this.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / reported CPU baseline.
 *
 * PARITY UNPINNED.  The reference has no tests, golden vectors or fixtures
 * (SURVEY.md section 4), its GLSL cannot be executed here (no Vulkan/glslang/
 * Godot), and its C++ builder (src/bvh/bvh.cpp) includes godot-cpp headers
 * that are an un-vendored, empty submodule -- building it would need stand-in
 * headers, which the build rules forbid, so there is no oracle/_ref.  This
 * restatement follows the reference source line by line (citations on every
 * function) and is cross-checked by an independent numpy restatement
 * (tests/np_restatement.py) and by libstdc++'s own std::nth_element, but it
 * has never been compared with an execution of the reference itself.
 *
 * Reference files restated (paths relative to /root/reference):
 *   src/bvh/vec.h, src/bvh/bvh.h, src/bvh/bvh.cpp          (BLAS/TLAS build)
 *   src/path_tracing/render_parameters.h                   (wire formats)
 *   project/addons/jar_path_tracing/src/shaders/main.glsl  (tracer)
 *   project/addons/jar_path_tracing/src/shaders/brdfs.glsl (BRDF)
 *   .../shaders/progressive_rendering.glsl + post_processing/progressive_rendering.cpp
 *
 * Pinned semantics for what GLSL leaves implementation-defined are listed in
 * oracle_pins.h and DESIGN.md ("Pinned semantics").
 */
#ifndef JPT_ORACLE_H
#define JPT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- wire formats (little-endian f32/u32), SURVEY.md 8(a) T-1..T-11 ---- */

typedef struct { float x, y, z, w; } jpto_vec4;              /* vec.h:45-51  16 B */

typedef struct {                                              /* bvh.h:22-29  144 B */
    jpto_vec4 vertices[3];
    jpto_vec4 centroid;
    jpto_vec4 normals[3];
    float     uvs[3][2];
    uint32_t  materialIndex;
    uint32_t  _pad;
} jpto_triangle;

typedef struct {                                              /* bvh.h:46-54 / main.glsl:45-52  48 B */
    jpto_vec4 aabbMin;
    jpto_vec4 aabbMax;
    uint32_t  left_child;
    uint32_t  right_child;
    uint32_t  first_tri_index;
    uint32_t  tri_count;
} jpto_bvh_node;

typedef struct {                                              /* bvh.h:56-62 / main.glsl:54-60  32 B */
    float    aabbMin[3];
    uint32_t leftRight;
    float    aabbMax[3];
    uint32_t blas;
} jpto_tlas_node;

typedef struct {                                              /* bvh.h:64-72 / main.glsl:84-93  176 B */
    float     transform[16];
    float     inverse_transform[16];
    jpto_vec4 aabbMin;
    jpto_vec4 aabbMax;
    uint32_t  blas_index;
    uint32_t  material[3];
} jpto_blas_instance;

typedef struct { jpto_vec4 vertices[3]; } jpto_tri_geometry;  /* render_parameters.h:59-62  48 B */

typedef struct {                                              /* render_parameters.h:64-71  80 B */
    float    n0[3];
    uint32_t material_index;
    jpto_vec4 n1;
    jpto_vec4 n2;
    float    uvs[3][2];
    uint32_t _pad[2];
} jpto_tri_data;

typedef struct {                                              /* render_parameters.h:49-57  64 B */
    jpto_vec4 albedo;
    jpto_vec4 emission;      /* rgb colour, w energy multiplier */
    float     metallic;
    float     roughness;
    int32_t   albedo_texture_index;
    float     padding[5];
} jpto_material;

typedef struct {                                              /* render_parameters.h:14-21  160 B */
    float     vp[16];
    float     ivp[16];
    jpto_vec4 position;
    uint32_t  frame_index;
    float     near_;
    float     far_;
    uint32_t  _pad;
} jpto_camera;

/* ---- scene in reference layout (what GeometryGroup3D::get_*_buffer emit) ---- */

typedef struct {
    const jpto_tri_geometry  *tri_geom;   uint32_t n_tri;
    const jpto_tri_data      *tri_data;
    const jpto_material      *materials;  uint32_t n_mat;
    const jpto_bvh_node      *bvh_nodes;  uint32_t n_nodes;
    const jpto_blas_instance *instances;  uint32_t n_inst;
    const jpto_tlas_node     *tlas_nodes; uint32_t n_tlas;
    const uint8_t            *tex_rgba8;  /* n_layers * res * res * 4, may be NULL */
    int32_t tex_res;
    int32_t n_layers;
    int32_t sampler_mode;                 /* bit 0: repeat (else clamp-to-edge), bit 1: linear (else nearest); 0 = default */
} jpto_scene_view;

/* exact event counters, SURVEY.md 8(d) */
typedef struct {
    uint64_t rays;            /* ray_trace() invocations             main.glsl:352 */
    uint64_t blas_expand;     /* A: BLAS internal-node expansions    main.glsl:286-287 */
    uint64_t tri_tests;       /* Bt: intersectTriangle calls         main.glsl:224 */
    uint64_t tlas_expand;     /* C: TLAS internal-node expansions    main.glsl:333-334 */
    uint64_t inst_visits;     /* D: TLAS leaf (instance) visits      main.glsl:316-317 */
    uint64_t shaded_hits;     /* E: get_shading_data calls           main.glsl:364 */
    uint64_t stack_overflow;  /* pushes past 64 entries (reference has no check) */
} jpto_counters;

enum { JPTO_ACCUM_REF_LDR8 = 0, JPTO_ACCUM_HDR_F32 = 1 };
enum {
    JPTO_FLAG_NO_CULL = 1,     /* every box test passes: BVH-independent closest hit */
    /* Which triangles can the reference's traversal reach at all?  The boxes of a reference tree are nested (a node's
     * box is the exact min/max over a superset of its child's vertices, bvh.cpp:19-37,299-303) and every operation of
     * intersectAABB (main.glsl:259-268) is monotone in the box planes, so a ray that passes a LEAF's box passes every
     * ancestor's box: ray_trace_tlas / ray_trace_blas test triangle T of instance I if and only if the world ray passes
     * I's world box (the TLAS leaf; no test when the TLAS is a single leaf) and the local ray passes the box of T's BLAS
     * leaf (no test when the BLAS root is the leaf) -- up to the distance culls `d < hitInfo.t` and NaN corner cases.
     * With this flag the walk ignores every internal box and every distance cull and applies exactly those two tests:
     * the image must equal the normal walk's (tests/test_oracle_render.py), which is what lets a DIFFERENT tree
     * reproduce the reference's image, "cracks" included (gdpathtracing_amd/csrc: reach records). */
    JPTO_FLAG_REACH_ONLY = 2,
    /* the shader's DEBUG_STEPS build (main.glsl:4,358-361,423-427; commented out as shipped): the image is
     * clamp(triangle tests of the primary ray / 256) in all three channels, one ray per pixel, depth = far */
    JPTO_FLAG_DEBUG_STEPS = 4
};

/* ---- builder (oracle_bvh.c) ---- */

typedef struct jpto_builder jpto_builder;

typedef struct {               /* one ArrayMesh surface (bvh.cpp:192-198) */
    const float   *vertices;   /* n_vertices * 3 */
    const float   *normals;    /* n_vertices * 3 */
    const float   *uvs;        /* n_vertices * 2 */
    const int32_t *indices;    /* n_indices */
    int32_t n_vertices;
    int32_t n_indices;
} jpto_surface;

jpto_builder *jpto_builder_create(void);
void          jpto_builder_destroy(jpto_builder *b);
/* BVHBuilder::BuildBVH (bvh.cpp:187-223): appends triangles+nodes, returns root node index */
uint32_t      jpto_builder_add_mesh(jpto_builder *b, const jpto_surface *surfaces, int32_t n_surfaces);
/* BLASInstance::{set_materials,set_transform} (bvh.h:73-115). transform12 = Transform3D:
 * basis rows [xx xy xz; yx yy yz; zx zy zz] (row-major, as godot stores it) then origin xyz. */
uint32_t      jpto_builder_add_instance(jpto_builder *b, uint32_t root, const float *transform12,
                                        const int32_t *material_ids, int32_t n_material_ids);
/* TLAS::build (bvh.cpp:264-317) + GpuTriangle split (geometry_group3d.cpp:356-365) */
void          jpto_builder_finish(jpto_builder *b);
uint32_t      jpto_builder_counts(const jpto_builder *b, uint32_t *n_tri, uint32_t *n_nodes,
                                  uint32_t *n_inst, uint32_t *n_tlas);
const jpto_triangle      *jpto_builder_triangles(const jpto_builder *b);
const jpto_tri_geometry  *jpto_builder_tri_geom(const jpto_builder *b);
const jpto_tri_data      *jpto_builder_tri_data(const jpto_builder *b);
const jpto_bvh_node      *jpto_builder_nodes(const jpto_builder *b);
const jpto_blas_instance *jpto_builder_instances(const jpto_builder *b);
const jpto_tlas_node     *jpto_builder_tlas(const jpto_builder *b);

/* libstdc++ std::nth_element restated in C, on whole triangles keyed by centroid[axis]
 * (bvh.cpp:173-176).  Exposed for the KAT against the real std::nth_element. */
void jpto_nth_element_centroid(jpto_triangle *tris, int32_t first, int32_t nth, int32_t last, int32_t axis);
/* godot Transform3D::affine_inverse restated (float), transform12 -> inverse12 */
void jpto_affine_inverse(const float *t12, float *out12);

/* ---- tracer (oracle_trace.c) ---- */

/* One dispatch of main.glsl (main.glsl:404-436) over rows [y0,y1): writes float radiance (rgb,1)
 * and reversed-Z depth. */
void jpto_trace_frame(const jpto_scene_view *scene, const jpto_camera *camera,
                      int32_t width, int32_t height, int32_t max_bounces, uint32_t flags,
                      int32_t y0, int32_t y1, float *radiance_rgba, float *depth,
                      jpto_counters *counters);

/* N frames of render() + ProgressiveRendering::render (path_tracing_camera.cpp:193-232,
 * progressive_rendering.cpp:47-66): frame_index = first_frame_index + f, frame_count = f+1.
 * accum_rgba: W*H*4 float (the rgba32f frameBuffer); ldr_rgba8: W*H*4 (screen after ACES);
 * depth: W*H (last frame).  n_threads <= 0 -> all online cores. Returns threads used. */
int32_t jpto_render(const jpto_scene_view *scene, const jpto_camera *camera,
                    int32_t width, int32_t height, int32_t max_bounces,
                    int32_t n_frames, uint32_t first_frame_index, int32_t accum_mode,
                    uint32_t flags, int32_t n_threads,
                    float *accum_rgba, uint8_t *ldr_rgba8, float *depth,
                    jpto_counters *counters);

/* ---- temporal reprojection (oracle_post.c) ---- */

/* TemporalReprojection::RenderParameters, temporal_reprojection.h:16-23 == temporal_reprojection.glsl:4-11 (88 B) */
typedef struct {
    float    deltaMatrix[16];
    int32_t  width, height;
    uint32_t frame_count;
    float    blendFactor;   /* ignored by the shader (literal 0.75, temporal_reprojection.glsl:64) */
    float    nearPlane, farPlane;
} jpto_temporal_params;

/* imageStore(outputImage rgba8) of main.glsl:434 for n_pixels radiance values */
void jpto_screen_rgba8(const float *radiance_rgba, size_t n_pixels, uint8_t *screen_rgba8);
/* one dispatch of temporal_reprojection.glsl:30-72; screen in/out, fb1/fb2 = the two history images */
void jpto_temporal_reproject(const jpto_temporal_params *p, uint8_t *screen_rgba8, const float *depth, float *fb1, float *fb2);

/* ---- small pieces exposed for known-answer tests ---- */
void  jpto_prng_seed(uint32_t px, uint32_t py, uint32_t frame, uint32_t seed_out[2]);   /* main.glsl:176-181 */
void  jpto_pcg2d(uint32_t seed[2], float out[2]);                                       /* main.glsl:163-174 */
void  jpto_sincos(float x, float *s, float *c);                                         /* pinned sin/cos */
float jpto_intersect_aabb(const float o[3], const float rD[3], const float bmin[3], const float bmax[3]); /* main.glsl:259-268 */
int   jpto_intersect_triangle(const float o[3], const float d[3], const float v0[3], const float v1[3],
                              const float v2[3], float t_max, float out_tuv[3], int *front);  /* main.glsl:224-257 */
void  jpto_primary_ray(const jpto_camera *camera, int32_t width, int32_t height, int32_t px, int32_t py,
                       float o[3], float d[3], uint32_t seed_after[2]);                 /* main.glsl:405-421 */
/* brdf / sample / pdf on an explicit shading record (brdfs.glsl:10-138) */
typedef struct {
    float normal[3], out_dir[3], lambert_out, diffuse_albedo[3], fresnel_0[3], roughness;
} jpto_shading;
void  jpto_brdf(const jpto_shading *s, const float l[3], float out[3]);
void  jpto_sample_brdf(const jpto_shading *s, const float xi[2], float out[3]);
float jpto_brdf_density(const jpto_shading *s, const float l[3]);
uint8_t jpto_unorm8(float x);
/* texture(textureArray, vec3(u, v, layer)) (main.glsl:214) under scene->sampler_mode (bit 0 repeat, bit 1 linear) */
void  jpto_sample_texture(const jpto_scene_view *scene, float u, float v, int32_t layer, float out[3]);
void  jpto_aces(const float in[3], float out[3]);                                       /* progressive_rendering.glsl:19-26 */

#ifdef __cplusplus
}
#endif
#endif
