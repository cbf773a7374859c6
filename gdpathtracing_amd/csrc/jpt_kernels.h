// jpt_kernels.h -- what the host layer (jpt_capi.cpp) sees of the device code.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "jpt_nodeq.h"
#include "jpt_shade.h"
#include "jpt_types.h"

namespace jpt {

constexpr int kStripRows = 8;  // screen partition granule (jpt_set_partition)

struct DevCounters {  // SURVEY.md 8(d) event counters
    unsigned long long rays, blas_expand, tri_tests, tlas_expand, inst_visits, shaded_hits;
    // wave-level phase statistics of the tracing kernels (counting builds only): how many times a phase ran
    // in a wave and how many lanes took part -- [0] rounds, [1] node iterations, [2] lanes in them, [3] leaf
    // phases, [4] lanes in them, [5] instance phases, [6] lanes in them, [7] primary rays finished by the sky cull
    unsigned long long phase[8];
    // ... and the length of the walks: the most record steps (internal records + leaves + instance entries) any one ray took, and
    // how many rays took < 16, < 64, < 256, < 1024, < 4096, < 16384, < 65536, more (a launch cannot end before its longest ray)
    unsigned long long walk_max;
    unsigned long long walk_hist[8];
    // path vertices that go on with a throughput of exactly (0, 0, 0) (jpt_stats.zero_throughput)
    unsigned long long zero_thr;
};

struct FrameParams {
    int32_t width, height;      // full image (Params.width/height, main.glsl:103-104)
    int32_t local_rows;         // rows this context renders
    int32_t rank, world;        // 8-row strips s with s % world == rank
    int32_t max_bounces;        // main.glsl:377 literal 5 == max_bounces + 1
    int32_t accum_mode;         // JPT_ACCUM_*
    uint32_t frame_index;       // camera.frame_index of this frame (main.glsl:409)
    uint32_t frame_count;       // ProgressiveRendering frame_count of this frame (progressive_rendering.cpp:53-60)
    int32_t n_frames;           // frames rendered by one launch (wide kernels)
    int32_t depth_frame;        // launch-local index of the frame whose first-hit distance is the depth image (-1: none)
    int32_t display_mode;       // 0: screen = ACES of the running mean (progressive_rendering.glsl:39-45)
                                // 1: screen = the last frame's own rgba8 store (main.glsl:434), no pass after it
    int32_t debug_steps = 0;    // the shader's DEBUG_STEPS build (main.glsl:358-361,423-427): audit kernel only
};

// device view of ExactShadow (jpt_builder.h)
struct TieShadowDev {
    const RefBvhNode* __restrict__ bvh = nullptr;
    const RefTriGeometry* __restrict__ tri_geom = nullptr;   // reference order
    const RefInstance* __restrict__ instances = nullptr;
    const RefTlasNode* __restrict__ tlas = nullptr;
    const uint32_t* __restrict__ tri_native = nullptr;       // reference triangle -> native triangle
    const uint32_t* __restrict__ native_ref = nullptr;       // native triangle -> reference triangle
    const uint32_t* __restrict__ tri_leaf = nullptr;         // reference triangle -> its leaf node
    const uint32_t* __restrict__ subtree_end = nullptr;      // BVH node -> one past its subtree (pre-order numbering)
    const uint32_t* __restrict__ tlas_parent = nullptr;
    const uint32_t* __restrict__ inst_tlas_leaf = nullptr;
    bool ok = false;
    bool tlas_current = false;   // false after a device refit: `instances` and `tlas` are those of the last HOST update
};

// All device-resident scene data of a context.
struct DeviceScene {
    // reference layout (drop-in route + cold shading data)
    const RefTriGeometry* ref_tri_geom = nullptr;
    const ShadeTri* shade_tris = nullptr;   // (the 80-byte GpuTriangleData array itself stays on the host: jpt_scene_get_reference_buffer)
    const RefMaterial* ref_materials = nullptr;
    const RefBvhNode* ref_bvh = nullptr;
    const RefInstance* ref_instances = nullptr;
    const RefTlasNode* ref_tlas = nullptr;
    const uint8_t* tex = nullptr;
    uint32_t n_tris = 0, n_materials = 0, n_ref_bvh = 0, n_instances = 0, n_ref_tlas = 0;
    int32_t tex_res = 0, n_layers = 0;
    int32_t sampler_mode = 0;  // JPT_SAMPLER_* (jpt_set_params)
    // flattened layout (native route)
    const WideNode* blas_nodes = nullptr;
    const WideTri* wide_tris = nullptr;
    const WideNode* tlas_nodes = nullptr;
    const WideInstance* wide_instances = nullptr;
    int32_t tlas_root = 0;  // child reference of the TLAS root
    uint32_t n_blas_nodes = 0, n_tlas_nodes = 0;
    // four-child collapse of the same trees (native builder only; null otherwise)
    bool use4 = false;
    uint32_t stack_need4 = 0;                // worst-case pending entries of a walk over the four-child records (compute_stack_need)
    const WideNode4* nodes4 = nullptr;       // BLAS records followed by the TLAS records, one index space (float form: the
                                             // device refit of the TLAS works on these) ...
    const WideNodeQ* nodesq = nullptr;       // ... and their 64-byte quantised form, same indices: what the kernels walk
    const WideInstance* wide_instances4 = nullptr;
    int32_t tlas_root4 = 0;
    // reach records (jpt_types.h): null unless the scene was committed with JPT_BUILD_SAH
    const ReachTri* reach_tri = nullptr;
    const ReachInst* reach_inst = nullptr;
    // the reference's own trees beside a native scene (ExactShadow, jpt_builder.h): what wf2_finish decides exact distance
    // ties on (jpt_tie_walk.h).  x.ok false: not available (a watertight scene, uploaded trees that are not numbered in
    // pre-order) -- ties are then left to the order of the native walk.
    TieShadowDev x;

    SceneShading shading() const
    {
        SceneShading s;
        s.tri_data = shade_tris;
        s.instances = ref_instances;
        s.materials = ref_materials;
        s.tex = tex;
        s.n_materials = n_materials;
        s.n_instances = n_instances;
        s.sampler_mode = sampler_mode;
        s.reach_tri = reach_tri;
        s.reach_inst = reach_inst;
        s.retrace_ties = x.ok && reach_tri != nullptr;
        s.tex_res = tex_res;
        s.n_layers = n_layers;
        return s;
    }
};

#if defined(__HIPCC__)
// local row -> image row for the strip-interleaved partition
__device__ __forceinline__ int local_to_global_row(int ly, const FrameParams& fp)
{
    const int strip = ly / kStripRows;
    return (strip * fp.world + fp.rank) * kStripRows + (ly - strip * kStripRows);
}

// progressive_rendering.glsl:28-46 for one pixel, preceded by the rgba8 store of main.glsl:434 in
// REF_LDR8 mode.  accum: rgba32f frameBuffer; ldr: the rgba8 screen after ACES.
__device__ __forceinline__ void accumulate_pixel(const FrameParams& fp, size_t idx, f3 radiance, float4* __restrict__ accum,
                                                 uint32_t* __restrict__ ldr)
{
    f3 cur = radiance;
    if (fp.accum_mode == 0) {
        cur = mk3(from_unorm8(unorm8(radiance.x)), from_unorm8(unorm8(radiance.y)), from_unorm8(unorm8(radiance.z)));
    }
    f3 sum = cur;
    if (fp.frame_count > 1) {
        const float4 prev = accum[idx];
        sum = mk3(cur.x + prev.x, cur.y + prev.y, cur.z + prev.z);
    }
    accum[idx] = make_float4(sum.x, sum.y, sum.z, 1.0f);
    if (fp.display_mode == 1) {
        ldr[idx] = unorm8(radiance.x) | (unorm8(radiance.y) << 8) | (unorm8(radiance.z) << 16) | 0xFF000000u;
        return;
    }
    const float fc = (float)fp.frame_count;
    const f3 col = aces_film(mk3(sum.x / fc, sum.y / fc, sum.z / fc) * 1.0f);
    ldr[idx] = unorm8(col.x) | (unorm8(col.y) << 8) | (unorm8(col.z) << 16) | 0xFF000000u;
}

__device__ __forceinline__ void flush_counters(const DevCounters& c, DevCounters* __restrict__ out)
{
    if (c.rays) atomicAdd(&out->rays, c.rays);
    if (c.blas_expand) atomicAdd(&out->blas_expand, c.blas_expand);
    if (c.tri_tests) atomicAdd(&out->tri_tests, c.tri_tests);
    if (c.tlas_expand) atomicAdd(&out->tlas_expand, c.tlas_expand);
    if (c.inst_visits) atomicAdd(&out->inst_visits, c.inst_visits);
    if (c.shaded_hits) atomicAdd(&out->shaded_hits, c.shaded_hits);
    for (int k = 0; k < 8; k++)
        if (c.phase[k]) atomicAdd(&out->phase[k], c.phase[k]);
    if (c.walk_max) atomicMax(&out->walk_max, c.walk_max);
    for (int k = 0; k < 8; k++)
        if (c.walk_hist[k]) atomicAdd(&out->walk_hist[k], c.walk_hist[k]);
    if (c.zero_thr) atomicAdd(&out->zero_thr, c.zero_thr);
}
__device__ __forceinline__ void count_walk(DevCounters& c, uint32_t steps)
{
    if (steps > c.walk_max) c.walk_max = steps;
    int b = 0;
    for (uint32_t lim = 16u; b < 7 && steps >= lim; lim <<= 2) b++;
    c.walk_hist[b]++;
}
#endif

// one frame over the reference layout (jpt_kernels_ref.hip); counters may be null
void launch_ref_frame(hipStream_t stream, const DeviceScene& ds, const FrameParams& fp, const RefCamera& cam, float4* accum,
                      uint32_t* ldr, float* depth, DevCounters* counters);

// The persistent-block pipeline (jpt_kernels_wf2.hip): one render of fp.n_frames frames over the flattened layout;
// fp.frame_index / fp.frame_count are those of the FIRST frame.  The first (max_bounces + 2) *
// wf2_segments() u32 of `workspace` are per-bounce, per-segment queue sizes afterwards; rows 1..max_bounces
// sum to the secondary ray segments traced.  trace_events: pairs around wf2_primary and each wf2_trace.
// The sky cell of every 8 x 8 tile of the context's share of the image whose four corner rays agree on one (wf2_sky_tiles: the
// tile-level test of wf2_accumulate, one lane per tile).  Depends on the camera, the image size and the partition only -- the host
// runs it when those change, not per render.  wf2_sky_tile_count: words `tile_cell` must hold.
size_t wf2_sky_tile_count(int width, int local_rows);
void launch_sky_tiles(hipStream_t stream, const FrameParams& fp, const RefCamera& cam, uint32_t* tile_cell);
uint32_t wf2_segments();
uint32_t trace_stack_capacity();  // entries a lane's traversal stack can hold (LDS + scratch)
size_t wf2_workspace_bytes(int width, int local_rows, int n_frames, int max_bounces);
// Screen rectangles (pixels, inclusive) of the boxes the TLAS root offers a ray; a primary ray through a pixel
// outside all of them is known to fail all of the root's box tests, i.e. to reach the sky after exactly one TLAS
// expansion, without being traced.  n < 0: unknown, trace everything.  Filled on the host (jpt_capi.cpp).
struct SkyCull {
    int32_t n = -1;
    int32_t x0[4], y0[4], x1[4], y1[4];
};

// helper streams / events for running frame groups concurrently (all null: serial); owned by the context
struct Wf2Async {
    hipStream_t aux_stream[3] = {nullptr, nullptr, nullptr};
    hipEvent_t fork = nullptr, join[3] = {nullptr, nullptr, nullptr};
    SkyCull cull;          // for the primary launch of this render
    int trace_chain = 1;   // wf2_trace: consecutive segments per block (1: lowest latency; 4 when renders are queued)
    hipEvent_t before_acc = nullptr;  // the accumulation kernel waits for this event (whatever its stream)
    const uint32_t* sky_tiles = nullptr;   // per 8 x 8 tile of the context's share of the image: its one rgba8 sky cell, if it has one
                                           // (launch_sky_tiles; null: wf2_accumulate decides every culled pixel by itself)
};
void launch_wf2_render(hipStream_t stream, const DeviceScene& ds, const FrameParams& fp, const RefCamera& cam, void* workspace,
                       float4* accum, uint32_t* ldr, float* depth, DevCounters* counters, hipEvent_t* trace_events,
                       const Wf2Async& async);

// one wave busy for `ticks` of the device's wall clock (hipDeviceAttributeWallClockRate), to see which streams run side by side
void launch_queue_spin(hipStream_t stream, long long ticks);

// Moving instances without the host (jpt_scene_refit_tlas): instance records from new transforms, then the boxes of the
// four-child TLAS records bottom-up over the unchanged topology.  transforms12: n x 12 floats on the device; bvh: the
// reference-layout BLAS nodes (root boxes); order / level_start: tlas4_refit_schedule, on the device.
void launch_tlas4_refit(hipStream_t stream, const float* transforms12, uint32_t n_instances, const RefBvhNode* bvh,
                        RefInstance* ref_instances, WideInstance* wide_instances, WideInstance* wide_instances4, WideNode4* nodes4,
                        uint32_t n_blas_records, const uint32_t* order, const uint32_t* level_start, uint32_t n_levels,
                        ReachInst* reach_instances /* may be null */, WideNodeQ* nodesq, uint32_t n_tlas_records,
                        const float* cut_boxes = nullptr, const uint32_t* cut_range = nullptr /* RefScene::inst_cut_*: may be null */);

// one dispatch of temporal_reprojection.glsl over a whole image (jpt_kernels_post.hip): screen rgba8 in/out, depth
// read-only, hist1 / hist2 the two rgba32f history images
void launch_temporal(hipStream_t stream, const RefTemporalParams& tp, uint32_t* screen, const float* depth, float4* hist1,
                     float4* hist2);

int wf2_wanted_groups(int n_frames, size_t paths);
// pixels of this context's share of the image that lie outside the render's window (the tile-aligned bounding rectangle
// of the sky cull's screen rectangles): the primary launch does not even enumerate them (their rays are sky by the
// cull's argument; the event counters are completed with their number on the host)
uint64_t wf2_pixels_outside_window(const SkyCull& cull, const FrameParams& fp);

// rank-major gathered strips -> full framebuffer (multi-GPU assemble); the rows of `own_rank`, the gathering rank itself, are
// read from `own` (its local buffer), not from the gathered pieces
void launch_assemble(hipStream_t stream, const float4* gathered, const float4* own, int own_rank, int world, int width, int height,
                     int max_local_rows, float4* accum_full, uint32_t* ldr_full, uint32_t frame_count);

// the display image alone (each rank has already tone-mapped its own rows)
void launch_assemble_ldr(hipStream_t stream, const uint32_t* gathered, const uint32_t* own, int own_rank, int world, int width,
                         int height, int max_local_rows, uint32_t* ldr_full);

}  // namespace jpt
