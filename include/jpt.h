/*
 * jpt.h -- C ABI of libjpt_hip.so, the MI355X (gfx950) back end for the GDPathTracing hot path.
 *
 * This is the drop-in boundary.  In the reference the path sits behind the
 * `gdcs::ComputeShader` object (absent submodule src/gdcs) that PathTracingCamera and
 * ProgressiveRendering drive, plus the CPU builder in src/bvh that GeometryGroup3D::build calls.
 * Each entry point below names the reference call(s) it replaces (paths relative to the reference
 * repository).  Plain pointers and sizes only; caller-owned host memory, library-owned device memory;
 * every function returns 0 on success or a negative JPT_E_* code and records a message readable with
 * jpt_last_error().  One context per GPU and per host thread; calls are blocking unless noted.
 *
 * All matrices are column-major float[16] as src/utils.h:15-49 writes them; all structs are the
 * little-endian wire formats of SURVEY.md 8(a) T-3..T-11 (gdpathtracing_amd/csrc/jpt_types.h).
 */
#ifndef JPT_H
#define JPT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JPT_ABI_VERSION 6

typedef struct jpt_ctx jpt_ctx;

enum {
    JPT_OK = 0,
    JPT_E_INVALID = -1,      /* bad argument / call order */
    JPT_E_DEVICE = -2,       /* HIP error (no GPU, launch failure, out of memory) */
    JPT_E_LIMIT = -3,        /* scene exceeds a format limit (e.g. 16-bit TLAS child index, bvh.h:59) */
    JPT_E_STATE = -4         /* scene / params / camera not set */
};

/* accumulation modes */
enum {
    JPT_ACCUM_REF_LDR8 = 0,  /* per frame: clamp01 + 8-bit quantise, then float sum (what the reference does:
                                main.glsl:98,434 -> progressive_rendering.glsl:33-37) */
    JPT_ACCUM_HDR_F32 = 1    /* pure float sum of radiance */
};

/* Sampler of texture(textureArray, vec3(uv, layer)) (main.glsl:213-214).  The reference sets the sampler state inside
 * the absent gdcs (format set up at path_tracing_camera.cpp:178-184), so it is a parameter here;
 * bit 0 = repeat instead of clamp-to-edge, bit 1 = linear instead of nearest.  What each mode computes is pinned in
 * oracle/oracle_trace.c::sample_texture (Vulkan texel addressing, UNORM8 texels, no sRGB decode, no mipmaps:
 * geometry_group3d.cpp:294-300). */
enum {
    JPT_SAMPLER_NEAREST_CLAMP = 0,   /* default: Godot's RDSamplerState defaults */
    JPT_SAMPLER_NEAREST_REPEAT = 1,
    JPT_SAMPLER_LINEAR_CLAMP = 2,
    JPT_SAMPLER_LINEAR_REPEAT = 3
};

/* post-processing after the path-tracing pass: PathTracingCamera::Denoising (path_tracing_camera.h:30-34,
 * the switch at path_tracing_camera.cpp:207-225) */
enum {
    JPT_DENOISE_PROGRESSIVE = 0, /* ProgressiveRendering: running sum, screen = ACES(mean) (progressive_rendering.glsl) */
    JPT_DENOISE_TEMPORAL = 1,    /* TemporalReprojection: blend with the reprojected history (temporal_reprojection.glsl) */
    JPT_DENOISE_NONE = 2         /* nothing: the screen is main.glsl's own rgba8 store (main.glsl:434) */
};

/* BVH builders */
enum {
    JPT_BUILD_REFERENCE_EXACT = 0, /* reproduces src/bvh/bvh.cpp node-for-node (incl. its default-box quirk);
                                      kernels traverse the same tree in the same order */
    JPT_BUILD_SAH = 1,             /* native binned-SAH builder, flattened wide-fetch layout (fast path).  Also runs the
                                      reference's builder once per mesh to record which leaf box holds each triangle
                                      ("reach records"): a hit found on the native tree is checked against the two box
                                      tests that decide whether the reference's traversal reaches that triangle at all
                                      (main.glsl:259-350), so the image equals the reference's even where float rounding
                                      lets a reference ray slip through its own boxes.  The reference's
                                      trees are kept beside the native ones: a hit that ties with another at exactly the
                                      same distance -- whose winner is a matter of the reference's visiting order,
                                      main.glsl:247 -- is decided by the reference's own walk */
    JPT_BUILD_SAH_WATERTIGHT = 2   /* the native builder alone (half the build time): every triangle hit is found; differs
                                      from the reference's image in the few pixels per 10^7 paths where the reference's
                                      tree has such a crack */
};

/* which reference-layout buffer (GeometryGroup3D::get_*_buffer, geometry_group3d.cpp:40-68) */
enum {
    JPT_BUF_TRI_GEOMETRY = 0, /* GpuTriangleGeometry[]  48 B */
    JPT_BUF_TRI_DATA = 1,     /* GpuTriangleData[]      80 B */
    JPT_BUF_MATERIALS = 2,    /* GpuMaterial[]          64 B */
    JPT_BUF_BVH_NODES = 3,    /* BVH::BVHNode[]         48 B */
    JPT_BUF_INSTANCES = 4,    /* BVH::BLASInstance[]   176 B */
    JPT_BUF_TLAS_NODES = 5,   /* BVH::TLASNode[]        32 B */
    JPT_BUF_TRIANGLES = 6,    /* BVH::Triangle[]       144 B (builder-internal, bvh.h:22-29) */
    /* not reference buffers -- the reach records of a native-tree scene (csrc/jpt_types.h), for tests and audits: */
    JPT_BUF_REACH_TRIANGLES = 7, /* 32 B per triangle (JPT_BUF_TRI_GEOMETRY order): lo.xyz, always, hi.xyz, pad */
    JPT_BUF_REACH_INSTANCES = 8  /* 64 B per instance: world box lo.xyz _ hi.xyz _, reference root box lo.xyz _ hi.xyz _ */
};

/* One ArrayMesh surface as BVHBuilder::BuildBVH reads it (bvh.cpp:192-198). */
typedef struct {
    const float   *vertices;   /* n_vertices * 3   Mesh::ARRAY_VERTEX  */
    const float   *normals;    /* n_vertices * 3   Mesh::ARRAY_NORMAL  */
    const float   *uvs;        /* n_vertices * 2   Mesh::ARRAY_TEX_UV  */
    const int32_t *indices;    /* n_indices        Mesh::ARRAY_INDEX   */
    int32_t n_vertices;
    int32_t n_indices;
} jpt_surface;

typedef struct {
    uint64_t rays;             /* ray segments traced = ray_trace() invocations (main.glsl:352) */
    uint64_t frames;           /* frames rendered since jpt_accum_reset */
    uint64_t blas_expand;      /* node expansions / tests, only filled by counting renders (jpt_render_counted) */
    uint64_t tri_tests;
    uint64_t tlas_expand;
    uint64_t inst_visits;
    uint64_t shaded_hits;
    double   last_render_ms;   /* device time of the last jpt_render (HIP events on the ctx stream) */
    double   last_trace_ms;    /* of which: the path-tracing kernel(s) */
    double   last_build_ms;    /* host time of the last scene commit (builder + flatten + upload) */
    uint64_t phase[8];         /* counting renders: wave-level phase statistics of the tracing kernels (diagnostic) */
    uint64_t sky_culled;       /* counting renders: primary rays finished without a walk (their pixel lies outside the screen
                                  rectangles of the TLAS root's boxes); they ARE counted in `rays` and `tlas_expand`, as the
                                  reference expands the root for them, but no record is fetched */
    double   last_primary_ms;  /* kernel timing on: the bounce-0 launch alone (last_trace_ms = all traversal launches) */
    uint64_t set_aside;        /* blocking renders on a native tree with reach records: path vertices whose hit is undecidable on
                                  the native tree -- the reference's traversal cannot reach it (a crack of its boxes), or
                                  another triangle lies at exactly the same distance --, traced again on the reference's own
                                  trees after the last bounce */
    uint64_t set_aside_dropped;/* ... of which: more than the set-aside buffer holds (1/64 of the paths, at least 65 536) -- those
                                  were shaded as found (the native tree's closest hit: the image differs from the reference's
                                  in those pixels).  0 on every scene measured; a scene that reports more renders exactly with
                                  JPT_KERNEL_REFERENCE_LAYOUT or JPT_UPLOAD_WALK_AS_GIVEN / JPT_BUILD_REFERENCE_EXACT */
    /* ABI 4 -- counting renders with the wavefront kernels: how long the walks are.  A launch cannot end before its longest
     * ray does, one dependent fetch after another (launch tails). */
    uint64_t walk_steps_max;    /* most record steps (internal records + leaf turns + instance entries) one ray took */
    uint64_t walk_steps_hist[8];/* rays that took < 16, < 64, < 256, < 1024, < 4096, < 16384, < 65536, more steps */
    /* ABI 5 -- counting renders: path vertices that go on although their throughput has become exactly (0, 0, 0) (brdf()
     * returns 0 when n.v < 0, brdfs.glsl:14): every later vertex of such a path adds 0 * emission (main.glsl:380), SURVEY.md
     * section 7 lists ending them as a result-preserving freedom; DESIGN.md section 8 says what share of the rays they are */
    uint64_t zero_throughput;   /* ... such vertices (each is the start of one more ray segment) */
} jpt_stats;

/* ---- lifetime --------------------------------------------------------------------------------- */

/* replaces: RenderingServer::create_local_rendering_device() + new ComputeShader(...)
 * (path_tracing_camera.cpp:114,139; progressive_rendering.cpp:25).
 * device_id = JPT_DEVICE_HOST_ONLY makes a builder-only context: the jpt_scene_* calls and
 * jpt_scene_get_reference_buffer work without a GPU; everything that renders returns JPT_E_DEVICE
 * (there is no CPU fallback). */
#define JPT_DEVICE_HOST_ONLY (-1)
int jpt_create(int device_id, jpt_ctx **out);
void jpt_destroy(jpt_ctx *ctx);
const char *jpt_last_error(const jpt_ctx *ctx);   /* ctx may be NULL: error of the last failed jpt_create */
int jpt_abi_version(void);

/* Run all work of this ctx on an existing HIP stream (hipStream_t as void*), e.g. torch's current
 * stream.  NULL restores the context's own stream.  Work already queued on the previous stream is waited for.
 * No reference counterpart. */
int jpt_set_stream(jpt_ctx *ctx, void *hip_stream);
/* The stream this ctx orders its work on (its own one unless jpt_set_stream replaced it), as hipStream_t: lets a
 * framework queue its own work -- a collective on the finished rows -- behind the renders (bench.py wraps it in
 * torch.cuda.ExternalStream).  No reference counterpart. */
int jpt_get_stream(jpt_ctx *ctx, void **hip_stream);
/* Queued renders (jpt_render_async) run their path kernels on four or six internal streams ("pipeline slots").  The HIP
 * runtime deals the streams of a process onto a small pool of hardware queues PER PRIORITY LEVEL, and streams that share
 * a queue run in submission order; so by default the slots are created at the device's HIGHEST stream priority, whose
 * pool they have to themselves as long as the host, torch and RCCL keep their streams at the normal level (DESIGN.md
 * section 4: C3 1.05 ms per queued render against 1.34 with everything at the normal level).  That choice also lets the
 * slots pre-empt the host's own compute work on a shared device.  An embedding application that wants otherwise says so
 * per context, before its first queued render or at any later time (existing slot streams are drained and re-made):
 *   DEFAULT  the library's rule (highest level)
 *   NORMAL   the level everything else uses: no pre-emption of host work; the queued rate then depends on which streams
 *            happen to share a hardware queue
 *   HIGH / LOW   the device's highest / lowest level
 * The pool has GPU_MAX_HW_QUEUES queues per level -- four unless the environment says otherwise at the process's FIRST HIP call.
 * Six renders in flight are worth 4 % on a 1920x1080x8-spp render and 15 % on small ones when the slots' six streams get a queue
 * each, and cost as much when they have to share four.  PROCESS ENVIRONMENT: the library never writes it (a setenv from a
 * shared library races with every getenv of a multi-threaded host and changes the queue count for all other HIP users of the
 * process).  A host that wants six renders in flight exports GPU_MAX_HW_QUEUES=6 itself, before its first HIP call and before it
 * starts threads (bench.py and the Python binding do, with os.environ.setdefault at import; six, not more: with eight a SECOND
 * context's streams pair up on some queues and it renders a third slower).  Every context MEASURES, before its first queued
 * render, whether six of its streams run side by side (~1 ms) -- six slots if they do, four if not: a host that exports nothing
 * gets four, which is also all that one blocking render per displayed frame (the addon's use) can ever use.
 * jpt_renders_in_flight: the slots the last queued render was dealt among (0 before the first).
 * No reference counterpart. */
enum { JPT_STREAM_PRIORITY_DEFAULT = 0, JPT_STREAM_PRIORITY_NORMAL = 1, JPT_STREAM_PRIORITY_HIGH = 2, JPT_STREAM_PRIORITY_LOW = 3 };
int jpt_set_stream_priority(jpt_ctx *ctx, int32_t priority);
int jpt_renders_in_flight(const jpt_ctx *ctx);

/* Device memory a context may spend on renders in flight.  A wavefront render keeps 168 bytes per path (pixel x frame of
 * the render window) in a workspace, and queued renders (jpt_render_async) use four or six workspaces at once -- 11 GB for
 * four 1920x1080x8-spp renders: fine on a 288 GB device the library has to itself, the first thing an application sharing
 * the GPU with its own renderer will want to cap.
 *   renders_in_flight  1..8 workspaces / pipeline slots; 0: the library's rule (4, or 6 where six slot streams run side by side;
 *                      2 when one workspace exceeds 24 GiB).
 *                      1 serialises queued renders (about 1.5 x the time per render at C3's size).
 *   workspace_budget_bytes  most bytes ONE workspace may take; a render with more frames than fit runs as batches of
 *                      frames in frame order, same image (0: the library's rule, 24 GiB).  One frame is the smallest
 *                      batch: when a single frame of the current resolution does not fit a budget set here, the
 *                      render calls return JPT_E_LIMIT and allocate nothing (the audit kernel needs no workspace).
 * Takes effect with the next render; workspaces no longer allowed are freed at once (the call waits for renders in
 * flight).  jpt_get_workspace_bytes reports what is allocated now.  No reference counterpart (the reference's
 * workspace is the two images of one frame).
 * Host side: the blocking read-backs (jpt_read_*) go through one PINNED staging buffer per context, as large as the
 * largest read-back made so far (133 MB after a jpt_read_accum_f32 of a 3840x2160 image); this call and every change of
 * resolution or partition give it back, the next read-back allocates what it needs.  It is not part of the bytes reported. */
int jpt_set_memory_policy(jpt_ctx *ctx, int32_t renders_in_flight, uint64_t workspace_budget_bytes);
int jpt_get_workspace_bytes(jpt_ctx *ctx, uint64_t *bytes_out);

/* ---- scene ingest, route (i): reference layout ------------------------------------------------ */

/* replaces: the six cs->create_storage_buffer_uniform(geometry_group->get_*_buffer(), b, 1) calls and
 * cs->create_layered_image_uniform(textures, ...) (path_tracing_camera.cpp:170-175,178-184).  Buffers are
 * byte-for-byte what GeometryGroup3D emits (geometry_group3d.cpp:40-68).  tex_rgba8 may be NULL.
 * What the kernels then walk is the library's NATIVE tree (four-child quantised records built over the uploaded
 * triangles of every BLAS an instance names), and the uploaded boxes that decide what the reference's traversal can reach
 * -- each triangle's BVHNode leaf (tri_count > 0), each instance's TLAS leaf -- become the reach records that keep the
 * image the reference's (see JPT_BUILD_SAH): the addon keeps GeometryGroup3D::build() and renders at the native route's
 * rate; the uploaded trees themselves are kept on the device too, and decide exact distance ties (see JPT_BUILD_SAH).
 * No builder of the reference is re-run.  Arrays that are not such a tree (a node or triangle reachable twice,
 * boxes that are not nested, an instance in no / several TLAS leaves, transform and inverse_transform that do not belong
 * together) are walked node for node as uploaded, like JPT_UPLOAD_WALK_AS_GIVEN; jpt_scene_upload_note says why.
 * After a native upload jpt_scene_get_reference_buffer returns the native trees in reference layout (triangles in the
 * native order), not the caller's arrays. */
int jpt_scene_upload_reference_layout(jpt_ctx *ctx,
                                      const void *tri_geometry, uint32_t n_triangles,
                                      const void *tri_data,
                                      const void *materials, uint32_t n_materials,
                                      const void *bvh_nodes, uint32_t n_bvh_nodes,
                                      const void *blas_instances, uint32_t n_instances,
                                      const void *tlas_nodes, uint32_t n_tlas_nodes,
                                      const uint8_t *tex_rgba8, int32_t tex_res, int32_t n_layers);

/* How the next jpt_scene_upload_reference_layout treats the trees it is given (no reference counterpart). */
enum {
    JPT_UPLOAD_NATIVE_TREE = 0,   /* default: native tree + reach records from the uploaded leaf / instance boxes */
    JPT_UPLOAD_WALK_AS_GIVEN = 1  /* audit route: the uploaded BVHNode / TLASNode arrays are walked node for node, in the
                                     reference's visit order (the six event counters equal the reference's too);
                                     ~20x slower on the demo scene, whose reference boxes are inflated to the origin */
};
int jpt_set_upload_mode(jpt_ctx *ctx, int32_t mode);
/* Which tree the kernels walk for the scene this context holds (>= 0), or a negative JPT_E_* code. */
enum {
    JPT_TREE_NONE = 0,              /* no scene */
    JPT_TREE_AS_GIVEN = 1,          /* uploaded reference-layout arrays, node for node */
    JPT_TREE_REFERENCE_EXACT = 2,   /* JPT_BUILD_REFERENCE_EXACT commit */
    JPT_TREE_NATIVE_REACH = 3,      /* native tree + reach records (JPT_BUILD_SAH commit, or a native upload) */
    JPT_TREE_NATIVE_WATERTIGHT = 4  /* JPT_BUILD_SAH_WATERTIGHT commit */
};
int jpt_scene_tree_kind(jpt_ctx *ctx);
/* Empty unless the last jpt_scene_upload_reference_layout is walked as given: then the reason. */
const char *jpt_scene_upload_note(const jpt_ctx *ctx);
/* 1: hits at exactly equal distance -- whose winner is a matter of the reference's visiting order, main.glsl:247 -- are
 * decided as the reference decides them for the scene this context holds (by the reference's own walk of the trees kept
 * beside the native ones; trivially on JPT_TREE_AS_GIVEN / JPT_TREE_REFERENCE_EXACT trees).  0: they fall to the native
 * tree's order, and *why_out (may be NULL; valid until the next scene call) says why: a JPT_BUILD_SAH_WATERTIGHT commit, or
 * an upload whose BLAS nodes are not numbered in pre-order (left child = parent + 1, as bvh.cpp:108-185 numbers them), which
 * the restricted walk needs -- such an upload still renders on the native tree with reach records; only the winners of exact
 * ties may differ from the reference's.  jpt_stats.set_aside keeps counting the tied vertices either way.  Negative: JPT_E_*. */
int jpt_scene_ties_exact(jpt_ctx *ctx, const char **why_out);

/* ---- scene ingest, route (ii): native build --------------------------------------------------- */

/* replaces GeometryGroup3D::build's tail (geometry_group3d.cpp:305-365): */
int jpt_scene_begin(jpt_ctx *ctx);
/*   BVHBuilder::BuildBVH(bvh_nodes, triangles, mesh)                 geometry_group3d.cpp:311, bvh.cpp:187-223 */
int jpt_scene_add_mesh(jpt_ctx *ctx, const jpt_surface *surfaces, int32_t n_surfaces, uint32_t *mesh_id_out);
/*   BLASInstance::{set_materials,set_transform}                      geometry_group3d.cpp:328-333, bvh.h:73-115
 *   transform12 = godot Transform3D: basis rows (xx xy xz yx yy yz zx zy zz) then origin (x y z). */
int jpt_scene_add_instance(jpt_ctx *ctx, uint32_t mesh_id, const float *transform12,
                           const int32_t *material_ids, int32_t n_material_ids);
/*   materials.push_back(gpu_material) / textures                     geometry_group3d.cpp:271-304 */
int jpt_scene_set_materials(jpt_ctx *ctx, const void *materials, uint32_t n_materials);
int jpt_scene_set_textures(jpt_ctx *ctx, const uint8_t *tex_rgba8, int32_t tex_res, int32_t n_layers);
/*   TLAS::build + Triangle -> GpuTriangle split + upload             geometry_group3d.cpp:350-365 */
int jpt_scene_commit(jpt_ctx *ctx, int32_t builder);
/* GeometryGroup3D::get_*_buffer() (geometry_group3d.cpp:40-68): valid after a REFERENCE_EXACT commit.
 * out may be NULL to query the size. */
int jpt_scene_get_reference_buffer(jpt_ctx *ctx, int32_t which, void *out, size_t capacity, size_t *size_out);

/* The committed scene of `src` (either route) made the scene of `dst` as well -- another context, normally on another
 * GPU: host arrays are copied and uploaded to dst's device, no builder runs again.  For one scene on several GPUs. */
int jpt_scene_share(jpt_ctx *dst, jpt_ctx *src);

/* ---- moving instances ---------------------------------------------------------
 * The reference has no incremental path: a moved MeshInstance3D means GeometryGroup3D::build() again
 * (geometry_group3d.cpp:78-366) and new ComputeShader buffers; its README lists a runtime TLAS update as wanted
 * (README.md:39-40).  Here the BLASes stay on the device; only the BLASInstance records
 * (BLASInstance::set_transform + update_aabb, bvh.h:81-115) and the TLAS (TLAS::build, bvh.cpp:264-317) are
 * redone and uploaded.  The result equals a fresh commit / upload of the moved scene. */
/*   route (ii): new Transform3D for instance `instance` (index in jpt_scene_add_instance order) ... */
int jpt_scene_set_instance_transform(jpt_ctx *ctx, uint32_t instance, const float *transform12);
/*   ... then one call that rebuilds instance records + TLAS with the builder of the last commit */
int jpt_scene_update_tlas(jpt_ctx *ctx);
/*   route (ii), on the device: ALL instance transforms at once (n_instances x 12 floats, jpt_scene_add_instance
 *   order).  Nothing is rebuilt on the host: one kernel recomputes the BLASInstance records from the transforms (the
 *   host builder's own arithmetic, bit for bit) and one refits the boxes of the TLAS records bottom-up over the
 *   topology of the last build.  They run on a stream of their own and write a COPY of the instance level that no
 *   render in flight reads (four copies: as many as renders in flight), so a queue of "refit, render" steps stays
 *   pipelined and the host never waits (jpt_scene_update_tlas drains the stream); the ctx stream is ordered after
 *   the refit.  Scenes committed with JPT_BUILD_SAH only; renders with the default kernel (the other kernels' arrays
 *   and jpt_scene_get_reference_buffer's host mirrors are refreshed by the next jpt_scene_update_tlas, which also
 *   re-optimises the topology: call it now and then when instances travel far).  The closest hit does not depend on
 *   the topology, so the image equals a fresh commit of the moved scene except at exact distance ties. */
int jpt_scene_refit_tlas(jpt_ctx *ctx, const float *transforms12, uint32_t n_instances);
/*   route (i): the caller re-ran BLASInstance::set_transform / TLAS::build itself; same instance count and
 *   the same blas_index per instance as the uploaded scene (otherwise: JPT_E_INVALID, upload the whole scene) */
int jpt_scene_update_reference_tlas(jpt_ctx *ctx, const void *blas_instances, uint32_t n_instances,
                                    const void *tlas_nodes, uint32_t n_tlas_nodes);

/* ---- per-render state ------------------------------------------------------------------------- */

/* replaces: Params upload (RenderParameters, path_tracing_camera.cpp:129-133,142; only width/height are
 * read by the shader, main.glsl:407,411), the literal 5 of main.glsl:377 (= max_bounces + 1), and the
 * rgba8 / r32f / rgba32f image creation (path_tracing_camera.cpp:148-165, progressive_rendering.cpp:35-39). */
int jpt_set_params(jpt_ctx *ctx, int32_t width, int32_t height, int32_t max_bounces,
                   int32_t accum_mode, int32_t sampler_mode);

/* Which device pipeline renders (no reference counterpart; both give the same image):
 *   WAVEFRONT          queue-based path tracer over the flattened 64-byte-node layout (default, fast);
 *   REFERENCE_LAYOUT   one thread per pixel straight over the six reference-layout buffers, node for node
 *                      as main.glsl:270-350 (audit route). */
enum { JPT_KERNEL_WAVEFRONT = 0, JPT_KERNEL_REFERENCE_LAYOUT = 1 };
int jpt_set_kernel(jpt_ctx *ctx, int32_t variant);

/* replaces: building main.glsl with `#define DEBUG_STEPS` (main.glsl:4, commented out as shipped; the reference's one
 * verification aid): every frame's image is clamp(hitInfo.steps / 256) in all three channels -- the number of
 * intersectTriangle calls the PRIMARY ray made (main.glsl:225,358-361,423-427) --, one ray per pixel, depth = far; the
 * post-processing pass runs on it as on any frame.  Rendered by the audit kernel (whatever jpt_set_kernel says), which
 * walks the scene's tree in reference layout: on the reference's own tree (JPT_UPLOAD_WALK_AS_GIVEN uploads,
 * JPT_BUILD_REFERENCE_EXACT commits) the counts are the reference's, on a native tree they are that tree's. */
int jpt_set_debug_steps(jpt_ctx *ctx, int32_t enable);

/* Per-launch timing of the traversal kernels (HIP events recorded around each launch on the context's stream;
 * jpt_stats.last_trace_ms).  Off by default: each event costs a few microseconds between kernels. */
int jpt_set_kernel_timing(jpt_ctx *ctx, int32_t enable);

/* Multi-GPU screen partition (no reference counterpart): this context renders the
 * 8-row strips s with s % world == rank.  Default rank 0 of 1 = whole image. */
int jpt_set_partition(jpt_ctx *ctx, int32_t rank, int32_t world);

/* replaces: cs->update_storage_buffer_uniform(camera_rid, camera.to_packed_byte_array())
 * (path_tracing_camera.cpp:198-200).  camera160 = struct Camera (render_parameters.h:14-21); its
 * frame_index field is ignored -- jpt_render takes the frame index explicitly. */
int jpt_set_camera(jpt_ctx *ctx, const void *camera160);

/* replaces n_frames iterations of PathTracingCamera::render's GPU work: cs->compute(main.glsl)
 * + ProgressiveRendering::render (path_tracing_camera.cpp:199-214, progressive_rendering.cpp:53-65).
 * Frame f uses camera.frame_index = first_frame_index + f and continues the accumulation
 * (frame_count = frames since jpt_accum_reset).  Blocking; device time in jpt_stats. */
int jpt_render(jpt_ctx *ctx, int32_t n_frames, uint32_t first_frame_index);
/* Same result, kernels compiled with event counters; fills the jpt_stats counter fields. */
int jpt_render_counted(jpt_ctx *ctx, int32_t n_frames, uint32_t first_frame_index);
/* Asynchronous form: enqueue and return; jpt_sync() or a jpt_read_* waits.  Consecutive asynchronous renders are
 * pipelined: their kernels run on alternating internal streams with separate workspaces (one render's launch
 * tails overlap the next render's kernels), the accumulation kernels run in call order (chained by events through
 * the ctx stream, which waits for each of them: work queued on the ctx stream afterwards sees the result), so the
 * framebuffers hold exactly what serial execution would leave (C3: 1.78 -> 1.28 ms per render when queued; up to four
 * renders are in flight, each with its own workspace).  A host that keeps only ONE render in flight (enqueue, own work,
 * jpt_sync or a read, enqueue ...) is recognised after its third such render and served like jpt_render from then on
 * (the pipelined launches are narrow: alone they take half as long again), still without blocking the caller. */
int jpt_render_async(jpt_ctx *ctx, int32_t n_frames, uint32_t first_frame_index);
int jpt_sync(jpt_ctx *ctx);

/* replaces: camera_moved -> frame_count = 1 (progressive_rendering.cpp:53-57).  In temporal mode: the history
 * images start from zero again, as for a newly created TemporalReprojection (temporal_reprojection.cpp:42-43). */
int jpt_accum_reset(jpt_ctx *ctx);
/* replaces: the frame_count word of ProgressiveRendering's Params (progressive_rendering.cpp:53-61, read at
 * progressive_rendering.glsl:34,39), for hosts that keep that counter themselves (the gdcs-shaped adapter hands over
 * what the reference's own ProgressiveRendering::render computed).  The next frame rendered is accumulated as frame
 * number `next_frame_count`: 1 overwrites the float sum (what jpt_accum_reset arranges), n > 1 adds to whatever the
 * buffer holds -- zeros after jpt_set_params, like the reference's freshly created frameBuffer image -- and the screen
 * is ACES(sum / n).  (The reference's very first frame has n = 2 when the camera's transform is the identity:
 * previous_transform starts as the identity, so camera_moved is false.) */
int jpt_set_progressive_frame_count(jpt_ctx *ctx, uint32_t next_frame_count);

/* replaces: the denoising_mode switch of PathTracingCamera::render (path_tracing_camera.cpp:207-225).
 *   JPT_DENOISE_PROGRESSIVE (default)  as described above.
 *   JPT_DENOISE_NONE       the screen (jpt_read_ldr_rgba8) is the rgba8 image main.glsl stored for the LAST frame
 *                          of the call, no tone mapping (path_tracing_camera.cpp:222-224).
 *   JPT_DENOISE_TEMPORAL   jpt_render takes n_frames = 1: trace the frame, then one dispatch of
 *                          temporal_reprojection.glsl with the parameters of jpt_set_temporal_params: the screen is
 *                          ACES(mix(frame, reprojected history, 0.75)); jpt_read_accum_f32 returns the rgba32f
 *                          history image that dispatch wrote.  Whole image on one context (no partition).
 * Changing the mode restarts the accumulation / history. */
int jpt_set_denoising_mode(jpt_ctx *ctx, int32_t mode);
/* replaces: cs->update_storage_buffer_uniform(render_parameters_rid, ...) of TemporalReprojection::render
 * (temporal_reprojection.cpp:67): the 88-byte TemporalReprojection::RenderParameters (temporal_reprojection.h:16-23:
 * deltaMatrix[16] column-major, width, height, frame_count, blendFactor, nearPlane, farPlane).  frame_count picks
 * the history image to read (even: frameBuffer1) and to write; blendFactor is not read by the shader
 * (temporal_reprojection.glsl:64 uses the literal 0.75) and is not read here. */
int jpt_set_temporal_params(jpt_ctx *ctx, const void *render_parameters88);

/* Which of main.glsl's two images a render produces (ABI 5).  The reference stores both every frame (main.glsl:434-435), but its
 * r32f depth image has one reader, TemporalReprojection (temporal_reprojection.glsl:45-58; add_existing_buffer at
 * temporal_reprojection.cpp:33): in the progressive and "none" modes nothing ever looks at it ("result-
 * preserving freedoms").  JPT_OUTPUT_DEPTH off: the kernels neither keep the first-hit distances nor write the depth image
 * (8.3 MB per 1080p render), and jpt_read_depth_f32 answers JPT_E_STATE.  Default: on, as the reference; JPT_DENOISE_TEMPORAL
 * renders produce it whatever this says.  The colour image, the accumulation and the display image are not affected. */
enum { JPT_OUTPUT_DEPTH = 1 };
int jpt_set_outputs(jpt_ctx *ctx, uint32_t outputs);

/* ---- outputs ---------------------------------------------------------------------------------- */

/* replaces: cs->get_image_uniform_buffer(output_texture_rid) (path_tracing_camera.cpp:228-229):
 * W*H*4 bytes, the screen image after ACES(sum / frame_count) (progressive_rendering.glsl:39-45).  Waits for the work
 * queued on the context; the bytes travel through the context's pinned staging buffer and one host copy, so `out` may be
 * ordinary (pageable) memory without the copy running at pageable speed. */
int jpt_read_ldr_rgba8(jpt_ctx *ctx, uint8_t *out);
/* Split form of jpt_read_ldr_rgba8 for double-buffered display (the reference stalls on
 * the read-back every frame, path_tracing_camera.cpp:228-230): `begin` enqueues the device->host copy of the
 * current screen image into a pinned staging buffer behind the work already queued on the context's stream
 * and returns at once; the host may queue the next jpt_render_async; `end` waits for that copy only and hands
 * the bytes out.  One read-back may be in flight per context. */
int jpt_readback_ldr_begin(jpt_ctx *ctx);
int jpt_readback_ldr_end(jpt_ctx *ctx, uint8_t *out);
/* the rgba32f frameBuffer (progressive_rendering.glsl:10,37): W*H*4 floats (like every jpt_read_*: through pinned staging) */
int jpt_read_accum_f32(jpt_ctx *ctx, float *out);
/* the r32f depthBuffer (main.glsl:99,435), last frame: W*H floats */
int jpt_read_depth_f32(jpt_ctx *ctx, float *out);

/* Device pointers of this context's LOCAL framebuffers (its partition's rows, strip-major), for
 * plumbing (RCCL gather through torch): float4 accumulation, and its size in bytes. */
void *jpt_device_accum(jpt_ctx *ctx, size_t *bytes_out);
/* Rank 0 after the gather: scatter `world` rank-major local buffers (device pointer) into this context's
 * full W*H framebuffers so jpt_read_* return the assembled image.  The gathering context's OWN rows are read in place,
 * from its local buffers as its last render left them: its slot of `device_gathered` is not read and need not be filled
 * (no copy of a rank's piece to itself). */
int jpt_assemble_from_ranks(jpt_ctx *ctx, const void *device_gathered, int32_t world);
/* The display image alone.  Every rank holds the complete sums of its own rows, so its rgba8 rows are final:
 * gathering them (4 bytes per pixel instead of 16) is all a displayed frame needs -- what the reference reads back
 * is this image (path_tracing_camera.cpp:228-229).  Afterwards jpt_read_ldr_rgba8 / jpt_readback_ldr_* on the
 * gathering rank return the whole image; the accumulation buffers stay distributed. */
void *jpt_device_ldr(jpt_ctx *ctx, size_t *bytes_out);
int jpt_assemble_ldr_from_ranks(jpt_ctx *ctx, const void *device_gathered_rgba8, int32_t world);
int32_t jpt_local_rows(jpt_ctx *ctx);

int jpt_get_stats(jpt_ctx *ctx, jpt_stats *out);

/* ---- one image on several GPUs from ONE process ------------------------------------------------------
 * The addon's host is a single C++ process (path_tracing_camera.cpp:193-232).  A jpt_multi owns one context per listed
 * device, each rendering its strips of the screen partition (jpt_set_partition); jpt_multi_render fans the render out,
 * every peer pushes its float4 accumulation rows to device 0 with a peer-to-peer copy over its own xGMI link (on a copy
 * stream of its own device, behind an event on the rank's render: the N - 1 transfers are in flight together, and the next
 * renders' path kernels overlap with them) and device 0 assembles them -- its own rows in place --, so the
 * jpt_multi_read_* calls return the whole image -- bit-identical to one GPU's.  Scene set-up: build it once on
 * jpt_multi_ctx(m, 0) with the jpt_scene_* calls, then jpt_multi_share_scene.  (Processes that hold one GPU each --
 * bench.py under torch.distributed -- use jpt_set_partition / jpt_device_accum / jpt_assemble_from_ranks with RCCL
 * send/recv in between instead.)  A device id may be listed more than once (rehearsal on a box with fewer GPUs). */
typedef struct jpt_multi jpt_multi;
int jpt_multi_create(const int *device_ids, int n_devices, jpt_multi **out);
void jpt_multi_destroy(jpt_multi *m);
const char *jpt_multi_last_error(const jpt_multi *m);     /* m may be NULL: error of the last failed jpt_multi_create */
int jpt_multi_world(const jpt_multi *m);
jpt_ctx *jpt_multi_ctx(jpt_multi *m, int rank);            /* the context of one rank (scene calls, statistics) */
int jpt_multi_share_scene(jpt_multi *m);                   /* jpt_scene_share(rank r, rank 0) for every other rank */
/* Moving instances under a jpt_multi: the moving-instance calls above, applied to EVERY rank's replica (a scene call made
 * on jpt_multi_ctx(m, 0) alone would leave the other ranks rendering their strips of the old scene state).  Same
 * arguments and results as jpt_scene_set_instance_transform / jpt_scene_update_tlas / jpt_scene_refit_tlas /
 * jpt_scene_update_reference_tlas; the first failing rank's error is reported. */
int jpt_multi_set_instance_transform(jpt_multi *m, uint32_t instance, const float *transform12);
int jpt_multi_update_tlas(jpt_multi *m);
int jpt_multi_refit_tlas(jpt_multi *m, const float *transforms12, uint32_t n_instances);
int jpt_multi_update_reference_tlas(jpt_multi *m, const void *blas_instances, uint32_t n_instances,
                                    const void *tlas_nodes, uint32_t n_tlas_nodes);
int jpt_multi_set_params(jpt_multi *m, int32_t width, int32_t height, int32_t max_bounces, int32_t accum_mode, int32_t sampler_mode);
int jpt_multi_set_camera(jpt_multi *m, const void *camera160);
int jpt_multi_accum_reset(jpt_multi *m);
/* what crosses the links each render: 0 (default) the float4 accumulation rows (16 B per pixel; BASELINE.json's exchange),
 * 1 only the finished rgba8 display rows (4 B per pixel, what the reference reads back) */
int jpt_multi_set_gather(jpt_multi *m, int32_t ldr_only);
/* asynchronous: queues the render on every rank, the gather and the assembly; jpt_multi_sync or a read waits */
int jpt_multi_render(jpt_multi *m, int32_t n_frames, uint32_t first_frame_index);
int jpt_multi_sync(jpt_multi *m);
/* What the last jpt_multi_render issued for its gather (diagnostic; SURVEY.md 8(e): the N - 1 transfers must be concurrent,
 * one per xGMI link): the number of peer copies, the number of DISTINCT streams they were issued on (each peer pushes on a
 * stream of its own device: = n_peer_copies), and how often rank 0's own piece was copied (0: the assembly reads it in place). */
int jpt_multi_gather_plan(const jpt_multi *m, int32_t *n_peer_copies, int32_t *n_distinct_streams, int32_t *own_piece_copies);
int jpt_multi_read_ldr_rgba8(jpt_multi *m, uint8_t *out);
int jpt_multi_read_accum_f32(jpt_multi *m, float *out);

/* ---- audit entry points (tests; no reference counterpart) ---------------------------------------------------------------
 * The native walk's box tests are conservative tests on quantised planes.  These run that one step
 * on caller-made inputs so its conservativeness can be tested directly (tests/test_quantized_walk.py), not only sampled
 * through images. */
/* n_nodes four-child records in the float form (128 bytes each: lo_x[4] lo_y[4] lo_z[4] child[4] hi_x[4] hi_y[4] hi_z[4]
 * pad[4]; an unused slot has child = INT32_MIN) -> their 64-byte quantised form, with the function uploads use. */
int jpt_debug_quantize_nodes4(const void *nodes4, uint32_t n_nodes, void *nodesq_out);
/* One record step per case.  A case is 32 bytes: ray origin xyz, direction xyz (as the walk holds them: the local ray of the
 * level), the closest distance found so far (hitInfo.t), and the index of the record to expand.  The records' child
 * references must be k + 1 for slot k.  taken_out[i] gets bit k set when the walk keeps child k (descends into it or
 * pushes it).  device_id >= 0: the kernel runs the very function the tracing kernels inline.  JPT_DEVICE_HOST_ONLY: a
 * host restatement of the same arithmetic, with its reciprocals moved host_rcp_ulps ulps away from zero (negative: towards
 * zero) -- v_rcp_f32 is accurate to 1 ulp. */
int jpt_debug_node_step4(int device_id, const void *nodes4, uint32_t n_nodes, const void *cases32, uint32_t n_cases,
                         int32_t host_rcp_ulps, uint8_t *taken_out);
const char *jpt_debug_last_error(void);

/* ---- environment ------------------------------------------------------------------------------------------------------
 * The library READS these once per process (at the first jpt_create; gdpathtracing_amd/csrc/jpt_tuning.h) and never writes the
 * environment.  They are for tests, audits and profiling runs; an embedding application needs none of them -- everything a
 * host decides at run time has a call above (jpt_set_stream_priority, jpt_set_memory_policy, jpt_set_upload_mode, ...).
 *
 *   variable                  default   meaning
 *   JPT_SKY_CULL              1         0: every primary ray is traced (audits; the image is the same)
 *   JPT_WORKSPACE_BUDGET_MB   24576     most MiB one render's workspace may take before it is split into batches of frames
 *                                       (jpt_set_memory_policy overrides per context)
 *   JPT_PIPELINE              1         0: queued renders run one after another (per-kernel profiling: tools/pmc.sh, tools/diag.sh)
 *   JPT_PIPE_SLOTS            0         2..8: renders in flight (0: the library's rule -- 4, or 6 where six slot streams run side by side)
 *   JPT_GROUPS                0         1..4: frame groups of a blocking render (0: the library's rule; 1 for per-kernel profiling)
 *   JPT_UPLOAD_WALK           --        "given": every reference-layout upload is walked node for node as uploaded (JPT_UPLOAD_WALK_AS_GIVEN)
 *   JPT_SET_ASIDE_CAP         -1        records of the set-aside buffer (-1: 1/64 of the paths, at least 65 536; tests force 0)
 *   JPT_TAIL                  -1        a wave walks its last, long rays with all its lanes: -1 on scenes of >= 200 000 triangles, 0 never, 1 always
 *   JPT_TAIL_ROUNDS           128       ... from this many rounds after its block's queue ran dry
 *   JPT_TAIL_LANES            8         ... once it is down to this many rays
 *   JPT_LIB                   --        (Python binding only) path of the library to load instead of gdpathtracing_amd/libjpt_hip.so
 *
 * The HOST may want to export, before its first HIP call (see jpt_set_stream_priority above):
 *   GPU_MAX_HW_QUEUES=6             six renders in flight instead of four for queued renders (the HIP runtime's variable)
 *   HSA_ENABLE_IPC_MODE_LEGACY=0    multi-process runs (RCCL between ranks) on drivers that only support dmabuf IPC
 */

#ifdef __cplusplus
}
#endif
#endif /* JPT_H */
