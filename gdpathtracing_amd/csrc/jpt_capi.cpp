// jpt_capi.cpp -- the C ABI of include/jpt.h over the HIP kernels and the host builder.
#include "../../include/jpt.h"

#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "jpt_builder.h"
#include "jpt_kernels.h"
#include "jpt_instance_math.h"
#include "jpt_tuning.h"

using namespace jpt;

namespace {

thread_local std::string g_create_error;

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    ~DevBuf() { release(); }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    hipError_t resize(size_t count)
    {
        if (count == n && (p || count == 0)) return hipSuccess;
        release();
        if (count == 0) return hipSuccess;
        hipError_t e = hipMalloc((void**)&p, count * sizeof(T));
        if (e == hipSuccess) n = count;
        return e;
    }
    hipError_t upload(const std::vector<T>& v, hipStream_t s)
    {
        hipError_t e = resize(v.size());
        if (e != hipSuccess || v.empty()) return e;
        return hipMemcpyAsync(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s);
    }
};

}  // namespace

struct jpt_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    Wf2Async async;  // helper streams / events of the frame groups (launch_wf2_render)
    std::string error;

    // host scene
    SceneBuilder builder;
    RefScene ref;
    WideScene wide;
    bool building = false, scene_ready = false, ref_is_exact = false, native_tree = false, tlas_dirty = false;
    bool host_scene_ready = false;  // c->ref / c->wide hold a complete scene (also true on host-only contexts)
    bool from_commit = false;       // the scene came from jpt_scene_commit: c->builder holds its meshes and transforms
    int32_t upload_mode = JPT_UPLOAD_NATIVE_TREE;  // jpt_set_upload_mode
    int32_t slot_priority = JPT_STREAM_PRIORITY_DEFAULT;  // jpt_set_stream_priority
    int32_t max_slots = 0;              // jpt_set_memory_policy: renders in flight (0: the library's rule)
    uint64_t workspace_budget = 0;      // ... and bytes per workspace (0: tuning().workspace_budget_mb)
    std::string upload_note;        // why the last reference-layout upload is walked as given (empty: it is not)
    std::string ties_note;          // why exact distance ties fall to the native tree's order (empty: they are decided exactly)
    std::vector<RefMaterial> pending_materials;
    std::vector<uint8_t> pending_tex;
    int32_t pending_tex_res = 0, pending_layers = 0;

    // device scene
    DevBuf<RefTriGeometry> d_tri_geom;
    DevBuf<ShadeTri> d_shade_tris;
    DevBuf<RefMaterial> d_materials;
    DevBuf<RefBvhNode> d_bvh;
    DevBuf<RefInstance> d_instances;
    DevBuf<RefTlasNode> d_tlas;
    DevBuf<uint8_t> d_tex;
    DevBuf<WideNode> d_wblas, d_wtlas;
    DevBuf<WideTri> d_wtris;
    DevBuf<WideInstance> d_winst, d_winst4;
    DevBuf<WideNode4> d_nodes4;   // four-child records: BLAS part, then room for the TLAS part (one index space)
    DevBuf<WideNodeQ> d_nodesq;   // their quantised form (jpt_nodeq.h), same indices: what the kernels walk
    // the reference's own trees beside a native scene (ExactShadow): two-child records + the triangle map
    DevBuf<RefBvhNode> d_x_bvh;
    DevBuf<RefTriGeometry> d_x_tri_geom;
    DevBuf<RefInstance> d_x_inst;
    DevBuf<RefTlasNode> d_x_tlas;
    DevBuf<uint32_t> d_x_tri_native, d_x_native_ref, d_x_tri_leaf, d_x_subtree_end, d_x_tlas_parent, d_x_inst_leaf;
    DevBuf<ReachTri> d_reach_tri;   // reach records (JPT_BUILD_SAH): per triangle / per instance (one per copy of the instance level)
    DevBuf<ReachInst> d_reach_inst;
    DevBuf<float> d_cut_boxes;         // RefScene::inst_cut_boxes / inst_cut_range (device refits)
    DevBuf<uint32_t> d_cut_range;
    BuildMode build_mode = BuildMode::ReferenceExact;
    DeviceScene ds;

    // per-render state
    bool params_set = false, camera_set = false;
    int32_t width = 0, height = 0, max_bounces = 4, accum_mode = 0, sampler_mode = 0;
    int32_t rank = 0, world = 1, local_rows = 0;
    RefCamera camera;
    uint32_t frame_count = 0;  // frames accumulated since reset
    int32_t kernel_variant = JPT_KERNEL_WAVEFRONT;
    bool debug_steps = false;  // jpt_set_debug_steps: the shader's DEBUG_STEPS build, on the audit kernel
    uint32_t outputs = JPT_OUTPUT_DEPTH;   // jpt_set_outputs: which of main.glsl's images the renders produce beside the colour
    bool depth_valid = false;              // d_depth holds the last render's depth image
    DevBuf<char> d_workspace;
    // render pipelining (jpt_render_async): consecutive asynchronous renders run their path kernels on two helper
    // streams with two workspaces, so one render's launch tails overlap the next render's kernels; the accumulation
    // kernels stay on the context's stream, in order
    static constexpr int kPipeSlots = 8;   // the most; the rule is 4, or 6 where six of the slots' streams run side by side (six_queues_probe)
    int six_queues = -1;                   // -1: not probed yet; 0 / 1
    int last_pipe_slots = 0;               // jpt_renders_in_flight
    bool aux_borrowed[3] = {};             // async.aux_stream[k] is pipe_stream[k] (ensure_group_streams): not destroyed on its own
    DevBuf<char> d_workspace_more[kPipeSlots - 1];  // slot 0 is d_workspace
    hipStream_t pipe_stream[kPipeSlots] = {};
    uint64_t async_seq = 0;
    hipEvent_t ev_paths_done[kPipeSlots] = {}, ev_acc_done[kPipeSlots] = {};
    bool acc_done_valid[kPipeSlots] = {};
    std::vector<uint32_t> h_qcount;  // per-bounce queue sizes of the last wavefront render
    std::vector<hipEvent_t> trace_events;  // pairs around each wf_trace launch of the last render
    int32_t trace_events_used = 0;
    bool kernel_timing = false;

    // framebuffers (local rows of this partition)
    DevBuf<float4> d_accum;
    DevBuf<uint32_t> d_ldr;
    DevBuf<float> d_depth;
    DevBuf<DevCounters> d_counters;
    // the tiles' sky cells (launch_sky_tiles): a function of the camera, the image size and the partition -- made when one of them
    // changes, read by every accumulation until then
    DevBuf<uint32_t> d_sky_tiles;
    RefCamera sky_tiles_camera;
    int32_t sky_tiles_key[5] = {0, 0, 0, 0, 0};   // width, height, local_rows, rank, world
    bool sky_tiles_valid = false;
    // assembled full image on the gathering rank
    DevBuf<float4> d_full_accum;
    DevBuf<uint32_t> d_full_ldr;
    bool assembled = false;      // d_full_accum + d_full_ldr hold the whole image (jpt_assemble_from_ranks)
    bool assembled_ldr = false;  // d_full_ldr only (jpt_assemble_ldr_from_ranks)

    // post-processing mode (PathTracingCamera::Denoising) and the temporal pass's state
    int32_t denoise = JPT_DENOISE_PROGRESSIVE;
    RefTemporalParams temporal;
    bool temporal_set = false, hist_valid = false;
    DevBuf<float4> d_hist1, d_hist2;   // frameBuffer1 / frameBuffer2 of temporal_reprojection.glsl:16-17
    float4* hist_written = nullptr;    // the one the last temporal pass wrote

    // pinned staging for the split read-back
    // device refit of the instance level (jpt_scene_refit_tlas)
    static constexpr int kRefitStages = 4;   // refits the host may queue before it has to wait for a copy to leave its buffer
    float* h_refit_t12[kRefitStages] = {};   // pinned staging for the transforms
    size_t h_refit_floats[kRefitStages] = {};
    hipEvent_t ev_refit_copied[kRefitStages] = {};
    uint64_t refit_seq = 0;
    // Several copies of the instance level (RefInstance + WideInstance arrays, TLAS tail of d_nodes4): a refit writes
    // a copy the renders in flight do NOT read, so the renders after a refit overlap with the renders before it.
    static constexpr int kInstanceSets = 8;   // as many as renders in flight can be: a queue of animation steps stays pipelined
    DevBuf<RefInstance> d_instances_more[kInstanceSets - 1];
    DevBuf<WideInstance> d_winst4_more[kInstanceSets - 1];
    DevBuf<ReachInst> d_reach_inst_more[kInstanceSets - 1];
    size_t tlas4_cap = 0;              // records reserved per TLAS tail
    int cur_set = 0;                   // which copy new renders read
    bool set_b_ready = false;          // copies 1.. exist and mirror the last host upload
    hipStream_t refit_stream = nullptr;
    hipEvent_t ev_set_retired[kInstanceSets] = {}, ev_refit_done = nullptr;
    bool set_retired_valid[kInstanceSets] = {};
    uint64_t refit_wait_seq = 0, slot_refit_seen[kPipeSlots] = {};
    int idle_streak = 0;   // queued renders in a row that found nothing in flight (do_render_batch)
    DevBuf<uint32_t> d_tlas4_order, d_tlas4_levels;
    uint32_t n_tlas4_levels = 0;
    bool refit_active = false;         // the device's instance level is ahead of the host mirrors (and of the other kernels' arrays)
    bool cull_boxes_current = true;    // c->wide.tlas_nodes4 holds the boxes of the copy new renders read (sky cull)
    std::vector<uint32_t> tlas4_order_h, tlas4_levels_h;   // the refit schedule, host copy
    uint32_t* h_ldr_pinned = nullptr;
    size_t h_ldr_pinned_px = 0;
    void* h_read_pinned = nullptr;   // blocking read-backs (staged_read)
    size_t h_read_bytes = 0;
    hipEvent_t ev_readback = nullptr;
    bool readback_pending = false;
    bool readback_full = false;  // the read-back in flight copies the assembled image (else: this context's rows)

    jpt_stats stats;
};

namespace {

int fail(jpt_ctx* c, int code, const std::string& msg)
{
    if (c) c->error = msg;
    return code;
}

int hip_fail(jpt_ctx* c, hipError_t e, const char* what)
{
    return fail(c, JPT_E_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
}

#define HIP_TRY(c, expr)                                          \
    do {                                                          \
        hipError_t e_ = (expr);                                   \
        if (e_ != hipSuccess) return hip_fail((c), e_, #expr);    \
    } while (0)

int32_t rows_of_rank(int32_t height, int32_t rank, int32_t world)
{
    const int32_t n_strips = (height + kStripRows - 1) / kStripRows;
    int32_t rows = 0;
    for (int32_t s = rank; s < n_strips; s += world) {
        const int32_t y0 = s * kStripRows;
        rows += (y0 + kStripRows <= height) ? kStripRows : (height - y0);
    }
    return rows;
}

int32_t max_rows_of_any_rank(int32_t height, int32_t world)
{
    int32_t m = 0;
    for (int32_t r = 0; r < world; r++) m = std::max(m, rows_of_rank(height, r, world));
    return m;
}

// The blocking read-backs' pinned staging buffer grows to the largest read-back ever made (133 MB after one jpt_read_accum_f32 of a
// 3840 x 2160 image) and would otherwise be held until jpt_destroy: given back whenever the framebuffers are re-made (another
// resolution or partition) and by jpt_set_memory_policy; the next read-back allocates what it needs.
void release_read_staging(jpt_ctx* c)
{
    if (c->h_read_pinned) (void)hipHostFree(c->h_read_pinned);
    c->h_read_pinned = nullptr;
    c->h_read_bytes = 0;
}

int alloc_framebuffers(jpt_ctx* c)
{
    if ((size_t)max_rows_of_any_rank(c->height, c->world) * c->width != c->d_accum.n) release_read_staging(c);
    c->local_rows = rows_of_rank(c->height, c->rank, c->world);
    // every rank allocates the same (maximum) size so a gather sees equal-sized pieces
    const size_t npx = (size_t)max_rows_of_any_rank(c->height, c->world) * c->width;
    HIP_TRY(c, c->d_accum.resize(npx));
    HIP_TRY(c, c->d_ldr.resize(npx));
    HIP_TRY(c, c->d_depth.resize(npx));
    HIP_TRY(c, c->d_counters.resize(1));
    if (npx) {
        HIP_TRY(c, hipMemsetAsync(c->d_accum.p, 0, npx * sizeof(float4), c->stream));
        HIP_TRY(c, hipMemsetAsync(c->d_ldr.p, 0, npx * sizeof(uint32_t), c->stream));
        HIP_TRY(c, hipMemsetAsync(c->d_depth.p, 0, npx * sizeof(float), c->stream));
    }
    c->frame_count = 0;
    c->assembled = c->assembled_ldr = false;
    return JPT_OK;
}

// Four-child records on the device: [ BLAS records | TLAS records ], internal child references of the TLAS part and
// the TLAS root shifted by the BLAS count.  The TLAS part has room for the largest TLAS of this instance count
// (<= one record per instance), so a TLAS update rewrites the tail only.
int upload_nodes4(jpt_ctx* c, bool tlas_only)
{
    const WideScene& w = c->wide;
    const size_t nb = w.blas_nodes4.size(), nt = w.tlas_nodes4.size();
    const size_t cap_t = std::max<size_t>(w.instances4.size(), std::max<size_t>(nt, 1));
    const size_t cap = nb + (size_t)jpt_ctx::kInstanceSets * cap_t;  // one TLAS tail per copy of the instance level (jpt_scene_refit_tlas)
    hipStream_t s = c->stream;
    if (w.instances4.empty() && nb == 0) {
        c->ds.nodes4 = nullptr;
        c->ds.nodesq = nullptr;
        c->ds.tlas_root4 = 0;
        return JPT_OK;
    }
    auto quantised = [](const std::vector<WideNode4>& in) {
        std::vector<WideNodeQ> out(in.size());
        for (size_t i = 0; i < in.size(); i++) quantize_node4(in[i], out[i]);
        return out;
    };
    if (!tlas_only || c->d_nodes4.n < cap) {
        HIP_TRY(c, c->d_nodes4.resize(cap));
        HIP_TRY(c, c->d_nodesq.resize(cap));
        if (nb) {
            const std::vector<WideNodeQ> q = quantised(w.blas_nodes4);
            HIP_TRY(c, hipMemcpyAsync(c->d_nodes4.p, w.blas_nodes4.data(), nb * sizeof(WideNode4), hipMemcpyHostToDevice, s));
            HIP_TRY(c, hipMemcpyAsync(c->d_nodesq.p, q.data(), nb * sizeof(WideNodeQ), hipMemcpyHostToDevice, s));
            HIP_TRY(c, hipStreamSynchronize(s));  // `q` is pageable host memory
        }
    }
    for (int copy = 0; copy < jpt_ctx::kInstanceSets; copy++) {
        const size_t base = nb + (size_t)copy * cap_t;
        std::vector<WideNode4> tail(w.tlas_nodes4);
        for (WideNode4& n : tail)
            for (int k = 0; k < 4; k++)
                if (n.child[k] >= 0) n.child[k] += (int32_t)base;
        const std::vector<WideNodeQ> qtail = quantised(tail);
        if (nt) HIP_TRY(c, hipMemcpyAsync(c->d_nodes4.p + base, tail.data(), nt * sizeof(WideNode4), hipMemcpyHostToDevice, s));
        if (nt) HIP_TRY(c, hipMemcpyAsync(c->d_nodesq.p + base, qtail.data(), nt * sizeof(WideNodeQ), hipMemcpyHostToDevice, s));
        HIP_TRY(c, hipStreamSynchronize(s));  // `tail`, `qtail` are pageable host memory
    }
    c->tlas4_cap = cap_t;
    c->cur_set = 0;
    c->set_b_ready = false;  // the instance arrays of the other copies are made (again) by the next refit
    for (int k = 0; k < jpt_ctx::kInstanceSets; k++) c->set_retired_valid[k] = false;
    c->ds.nodes4 = c->d_nodes4.p;
    c->ds.nodesq = c->d_nodesq.p;
    c->ds.tlas_root4 = w.tlas_root4 >= 0 ? w.tlas_root4 + (int32_t)nb : w.tlas_root4;
    {
        // bottom-up schedule of the TLAS records, for refits on the device
        std::vector<uint32_t>&order = c->tlas4_order_h, &levels = c->tlas4_levels_h;
        tlas4_refit_schedule(w, order, levels);
        HIP_TRY(c, c->d_tlas4_order.upload(order, s));
        HIP_TRY(c, c->d_tlas4_levels.upload(levels, s));
        HIP_TRY(c, hipStreamSynchronize(s));  // pageable host vectors
        c->n_tlas4_levels = (uint32_t)levels.size() - 1u;
        c->cull_boxes_current = true;
    }
    return JPT_OK;
}

// Why exact distance ties fall to the native tree's order, for jpt_scene_ties_exact (this used to be silent).  A
// tree of the reference's own -- uploads walked as given, JPT_BUILD_REFERENCE_EXACT -- needs no shadow: its walk IS the
// reference's.
void note_ties(jpt_ctx* c)
{
    const ExactShadow& x = c->ref.exact;
    c->ties_note.clear();
    if (c->native_tree) {
        if (c->ref.reach_tri.empty()) c->ties_note = "the scene has no reach records (JPT_BUILD_SAH_WATERTIGHT): the native tree's order decides";
        else if (x.instances.empty()) c->ties_note.clear();   // nothing to hit
        else if (!x.valid) c->ties_note = "the reference's TLAS of this scene could not be built (TLAS::build, bvh.cpp:264-317)";
        else if (!x.resolvable)
            c->ties_note = "the uploaded trees are not in the shape the restricted walk indexes: every BLAS numbered in pre-order "
                           "(left child = parent + 1, what build_recursive emits, bvh.cpp:108-185) and every instance in exactly one "
                           "TLAS leaf reachable from slot 0";
    }
}

// The shadow (the reference's own trees beside a native scene, ExactShadow) -> the device, for jpt_tie_walk.h.
// `instances_only`: the BLAS part is there already (a TLAS update).  Not having a shadow is not an error: exact distance
// ties are then decided by the order of the native walk.
int upload_shadow(jpt_ctx* c, bool instances_only)
{
    TieShadowDev& d = c->ds.x;
    d.ok = false;
    const ExactShadow& x = c->ref.exact;
    note_ties(c);
    if (!x.valid || !x.resolvable || !c->native_tree || c->ref.reach_tri.empty() || c->device < 0 || x.instances.empty())
        return JPT_OK;
    hipStream_t s = c->stream;
    if (!instances_only) {
        HIP_TRY(c, c->d_x_bvh.upload(x.bvh_nodes, s));
        HIP_TRY(c, c->d_x_tri_geom.upload(x.tri_geom, s));
        HIP_TRY(c, c->d_x_tri_native.upload(x.tri_native, s));
        HIP_TRY(c, c->d_x_native_ref.upload(x.native_ref, s));
        HIP_TRY(c, c->d_x_tri_leaf.upload(x.tri_leaf, s));
        HIP_TRY(c, c->d_x_subtree_end.upload(x.subtree_end, s));
    }
    HIP_TRY(c, c->d_x_inst.upload(x.instances, s));
    HIP_TRY(c, c->d_x_tlas.upload(x.tlas_nodes, s));
    HIP_TRY(c, c->d_x_tlas_parent.upload(x.tlas_parent, s));
    HIP_TRY(c, c->d_x_inst_leaf.upload(x.inst_tlas_leaf, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    d.bvh = c->d_x_bvh.p;
    d.tri_geom = c->d_x_tri_geom.p;
    d.instances = c->d_x_inst.p;
    d.tlas = c->d_x_tlas.p;
    d.tri_native = c->d_x_tri_native.p;
    d.native_ref = c->d_x_native_ref.p;
    d.tri_leaf = c->d_x_tri_leaf.p;
    d.subtree_end = c->d_x_subtree_end.p;
    d.tlas_parent = c->d_x_tlas_parent.p;
    d.inst_tlas_leaf = c->d_x_inst_leaf.p;
    d.ok = true;
    d.tlas_current = true;
    return JPT_OK;
}

// host RefScene (+ flatten) -> device
int upload_scene(jpt_ctx* c)
{
    std::string err;
    if (!flatten(c->ref, c->wide, err)) return fail(c, JPT_E_INVALID, "flatten: " + err);
    // the four-child collapse is for the native builder's trees; reference-exact trees keep the two-child
    // records so the kernels visit them node for node like main.glsl (event counters equal the oracle's)
    const bool use4 = !c->ref_is_exact && c->native_tree;
    if (use4) flatten4(c->wide);
    else {
        c->wide.blas_nodes4.clear();
        c->wide.tlas_nodes4.clear();
        c->wide.instances4.clear();
    }
    compute_stack_need(c->wide);
    {
        // the reference walks with two unchecked 64-entry stacks (main.glsl:272,307); here a deeper tree is an
        // error at upload time instead of undefined behaviour at render time
        const uint32_t need = use4 ? c->wide.stack_need4 : c->wide.stack_need2;
        if (need > trace_stack_capacity())
            return fail(c, JPT_E_LIMIT, "acceleration structure too deep: a traversal could need " + std::to_string(need) +
                                            " stack entries, the kernels hold " + std::to_string(trace_stack_capacity()));
    }
    c->host_scene_ready = true;
    c->tlas_dirty = false;
    c->refit_active = false;
    note_ties(c);
    if (c->device < 0) return JPT_OK;  // host-only context: arrays stay on the host, nothing can be rendered
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    HIP_TRY(c, c->d_tri_geom.upload(c->ref.tri_geom, s));
    {
        std::vector<ShadeTri> st(c->ref.tri_data.size());
        for (size_t i = 0; i < st.size(); i++) {
            const RefTriData& t = c->ref.tri_data[i];
            ShadeTri& o = st[i];
            for (int k = 0; k < 3; k++) o.n0[k] = t.n0[k];
            o.n1[0] = t.n1.x; o.n1[1] = t.n1.y; o.n1[2] = t.n1.z;
            o.n2[0] = t.n2.x; o.n2[1] = t.n2.y; o.n2[2] = t.n2.z;
            for (int k = 0; k < 3; k++) o.uvs[k][0] = t.uvs[k][0], o.uvs[k][1] = t.uvs[k][1];
            o.material_index = t.material_index;
        }
        HIP_TRY(c, c->d_shade_tris.upload(st, s));
        HIP_TRY(c, hipStreamSynchronize(s));   // (the staging vector goes out of scope)
    }
    HIP_TRY(c, c->d_materials.upload(c->ref.materials, s));
    HIP_TRY(c, c->d_bvh.upload(c->ref.bvh_nodes, s));
    HIP_TRY(c, c->d_instances.upload(c->ref.instances, s));
    HIP_TRY(c, c->d_tlas.upload(c->ref.tlas_nodes, s));
    HIP_TRY(c, c->d_tex.upload(c->ref.textures, s));
    HIP_TRY(c, c->d_wblas.upload(c->wide.blas_nodes, s));
    HIP_TRY(c, c->d_wtlas.upload(c->wide.tlas_nodes, s));
    HIP_TRY(c, c->d_wtris.upload(c->wide.tris, s));
    HIP_TRY(c, c->d_winst.upload(c->wide.instances, s));
    HIP_TRY(c, c->d_winst4.upload(c->wide.instances4, s));
    HIP_TRY(c, c->d_reach_tri.upload(c->ref.reach_tri, s));
    HIP_TRY(c, c->d_reach_inst.upload(c->ref.reach_inst, s));
    HIP_TRY(c, c->d_cut_boxes.upload(c->ref.inst_cut_boxes, s));
    HIP_TRY(c, c->d_cut_range.upload(c->ref.inst_cut_range, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    {
        const int rc4 = upload_nodes4(c, false);
        if (rc4 != JPT_OK) return rc4;
    }
    DeviceScene& d = c->ds;
    const bool reach = !c->ref.reach_tri.empty() && c->ref.reach_tri.size() == c->ref.tri_geom.size() &&
                       c->ref.reach_inst.size() == c->ref.instances.size();
    d.reach_tri = reach ? c->d_reach_tri.p : nullptr;
    d.reach_inst = reach ? c->d_reach_inst.p : nullptr;
    d.ref_tri_geom = c->d_tri_geom.p;
    d.shade_tris = c->d_shade_tris.p;
    d.ref_materials = c->d_materials.p;
    d.ref_bvh = c->d_bvh.p;
    d.ref_instances = c->d_instances.p;
    d.ref_tlas = c->d_tlas.p;
    d.tex = c->d_tex.p;
    d.n_tris = (uint32_t)c->ref.tri_geom.size();
    d.n_materials = (uint32_t)c->ref.materials.size();
    d.n_ref_bvh = (uint32_t)c->ref.bvh_nodes.size();
    d.n_instances = (uint32_t)c->ref.instances.size();
    d.n_ref_tlas = (uint32_t)c->ref.tlas_nodes.size();
    d.tex_res = c->ref.tex_res;
    d.n_layers = c->ref.n_layers;
    d.blas_nodes = c->d_wblas.p;
    d.wide_tris = c->d_wtris.p;
    d.tlas_nodes = c->d_wtlas.p;
    d.wide_instances = c->d_winst.p;
    d.tlas_root = c->wide.tlas_root;
    d.n_blas_nodes = (uint32_t)c->wide.blas_nodes.size();
    d.n_tlas_nodes = (uint32_t)c->wide.tlas_nodes.size();
    d.use4 = use4;
    d.stack_need4 = c->wide.stack_need4;
    d.wide_instances4 = c->d_winst4.p;
    {
        const int rcx = upload_shadow(c, false);
        if (rcx != JPT_OK) return rcx;
    }
    c->scene_ready = true;
    return JPT_OK;
}

int validate_ref_scene(jpt_ctx* c, const RefScene& r)
{
    if (r.tri_geom.size() != r.tri_data.size()) return fail(c, JPT_E_INVALID, "triangle geometry/data counts differ");
    if (r.materials.empty()) return fail(c, JPT_E_INVALID, "material table is empty (entry 0 is the default material)");
    for (const RefInstance& i : r.instances)
        if (i.blas_index >= r.bvh_nodes.size()) return fail(c, JPT_E_INVALID, "instance root index out of range");
    for (const RefBvhNode& n : r.bvh_nodes) {
        if (n.tri_count > 0) {
            if ((size_t)n.first_tri_index + n.tri_count > r.tri_geom.size())
                return fail(c, JPT_E_INVALID, "BVH leaf triangle range out of bounds");
        } else if (n.left_child >= r.bvh_nodes.size() || n.right_child >= r.bvh_nodes.size()) {
            return fail(c, JPT_E_INVALID, "BVH child index out of range");
        }
    }
    return JPT_OK;
}

// Screen rectangles of the boxes the TLAS root offers (SkyCull, jpt_kernels.h).  A primary ray is the half-line from
// camera.position through ivp * (ndc, 1, 1) (main.glsl:411-420).  The projection used here is the double-precision
// inverse of that same ivp, and every projected corner is checked to lie on the ray of its own projection; if the
// camera block does not behave like a pinhole seen from `position` (inconsistent vp / ivp / position, a corner at or
// behind the camera), culling is off for this render.  Rectangles are widened by two pixels: the jitter stays inside
// the pixel (main.glsl:183-187) and float rounding of the ray set-up is orders of magnitude below a pixel.
void compute_sky_cull(const jpt_ctx* c, SkyCull& out)
{
    out.n = -1;
    if (!tuning().sky_cull || c->width <= 0 || c->height <= 0) return;
    if (!c->cull_boxes_current) return;  // the root's boxes were last refitted on the device only: the host copy is stale
    // the boxes: children of the TLAS root record in the layout the kernels walk
    double lo[4][3], hi[4][3];
    int n = 0;
    const WideScene& w = c->wide;
    const bool use4 = !c->ref_is_exact && c->native_tree;
    if (w.instances.empty()) return;
    if (use4) {
        if (w.tlas_root4 < 0 || (size_t)w.tlas_root4 >= w.tlas_nodes4.size()) return;  // the root is a single instance
        const WideNode4& r = w.tlas_nodes4[(size_t)w.tlas_root4];
        for (int k = 0; k < 4; k++)
            if (r.child[k] != kEmptyChild) {
                lo[n][0] = r.lo_x[k]; lo[n][1] = r.lo_y[k]; lo[n][2] = r.lo_z[k];
                hi[n][0] = r.hi_x[k]; hi[n][1] = r.hi_y[k]; hi[n][2] = r.hi_z[k];
                n++;
            }
    } else {
        if (w.tlas_root < 0 || (size_t)w.tlas_root >= w.tlas_nodes.size()) return;
        const WideNode& r = w.tlas_nodes[(size_t)w.tlas_root];
        if (r.left == w.tlas_root && r.right == w.tlas_root) return;
        for (int k = 0; k < 3; k++) {
            lo[0][k] = r.lmin[k]; hi[0][k] = r.lmax[k];
            lo[1][k] = r.rmin[k]; hi[1][k] = r.rmax[k];
        }
        n = 2;
    }
    // vp' = inverse of ivp (column-major), in double
    double a[4][8];
    for (int r = 0; r < 4; r++)
        for (int col = 0; col < 4; col++) {
            a[r][col] = (double)c->camera.ivp[col * 4 + r];
            a[r][4 + col] = r == col ? 1.0 : 0.0;
        }
    for (int col = 0; col < 4; col++) {
        int piv = col;
        for (int r = col + 1; r < 4; r++)
            if (std::fabs(a[r][col]) > std::fabs(a[piv][col])) piv = r;
        if (!(std::fabs(a[piv][col]) > 0.0) || !std::isfinite(a[piv][col])) return;
        if (piv != col)
            for (int k = 0; k < 8; k++) std::swap(a[piv][k], a[col][k]);
        const double d = a[col][col];
        for (int k = 0; k < 8; k++) a[col][k] /= d;
        for (int r = 0; r < 4; r++)
            if (r != col) {
                const double f = a[r][col];
                for (int k = 0; k < 8; k++) a[r][k] -= f * a[col][k];
            }
    }
    auto vp = [&](int r, int col) { return a[r][4 + col]; };
    const double O[3] = {(double)c->camera.position.x, (double)c->camera.position.y, (double)c->camera.position.z};
    const double W = (double)c->width, H = (double)c->height;
    for (int b = 0; b < n; b++) {
        double x0 = 1e300, y0 = 1e300, x1 = -1e300, y1 = -1e300;
        for (int corner = 0; corner < 8; corner++) {
            const double C[3] = {(corner & 1) ? hi[b][0] : lo[b][0], (corner & 2) ? hi[b][1] : lo[b][1], (corner & 4) ? hi[b][2] : lo[b][2]};
            if (!std::isfinite(C[0]) || !std::isfinite(C[1]) || !std::isfinite(C[2])) return;
            double clip[4];
            for (int r = 0; r < 4; r++) clip[r] = vp(r, 0) * C[0] + vp(r, 1) * C[1] + vp(r, 2) * C[2] + vp(r, 3);
            if (!(std::fabs(clip[3]) > 1e-12)) return;
            const double nx = clip[0] / clip[3], ny = clip[1] / clip[3];
            if (!std::isfinite(nx) || !std::isfinite(ny) || std::fabs(nx) > 1e6 || std::fabs(ny) > 1e6) return;
            // the ray of that projection, as the kernel builds it: from O through F = ivp * (nx, ny, 1, 1) / w
            double F[4];
            for (int r = 0; r < 4; r++)
                F[r] = (double)c->camera.ivp[0 * 4 + r] * nx + (double)c->camera.ivp[1 * 4 + r] * ny + (double)c->camera.ivp[2 * 4 + r] +
                       (double)c->camera.ivp[3 * 4 + r];
            if (!(std::fabs(F[3]) > 1e-300)) return;
            const double D[3] = {F[0] / F[3] - O[0], F[1] / F[3] - O[1], F[2] / F[3] - O[2]};
            const double V[3] = {C[0] - O[0], C[1] - O[1], C[2] - O[2]};
            const double dd = D[0] * D[0] + D[1] * D[1] + D[2] * D[2], vv = V[0] * V[0] + V[1] * V[1] + V[2] * V[2];
            const double dv = D[0] * V[0] + D[1] * V[1] + D[2] * V[2];
            if (!(dd > 0.0) || !(vv > 0.0) || !(dv > 0.0)) return;  // corner at / behind the camera, or a degenerate ray
            // distance of the corner from that ray, relative to its distance from the camera
            const double cross2 = vv - dv * dv / dd;
            if (!(cross2 <= 1e-8 * vv)) return;  // the block is not a pinhole seen from `position`
            const double px = (nx + 1.0) * 0.5 * W, py = (1.0 - ny) * 0.5 * H;  // main.glsl:411-412 inverted
            x0 = std::min(x0, px); x1 = std::max(x1, px);
            y0 = std::min(y0, py); y1 = std::max(y1, py);
        }
        const double margin = 2.0;
        auto clampi = [](double v) { return (int32_t)std::max(-1.0e9, std::min(1.0e9, v)); };
        out.x0[b] = clampi(std::floor(x0 - margin));
        out.y0[b] = clampi(std::floor(y0 - margin));
        out.x1[b] = clampi(std::ceil(x1 + margin));
        out.y1[b] = clampi(std::ceil(y1 + margin));
    }
    for (int b = n; b < 4; b++) out.x0[b] = out.y0[b] = 1, out.x1[b] = out.y1[b] = 0;
    out.n = n;
}

int do_render_batch(jpt_ctx* c, int32_t n_frames, uint32_t first_frame_index, bool counted, bool blocking);

// frames of one wavefront render that fit the workspace budget (all frames of a batch are in flight at once)
int32_t frames_per_batch(const jpt_ctx* c, int32_t n_frames)
{
    const size_t budget = c->workspace_budget ? (size_t)c->workspace_budget : (size_t)tuning().workspace_budget_mb << 20;
    if (c->kernel_variant == JPT_KERNEL_REFERENCE_LAYOUT || c->debug_steps || n_frames <= 1) return n_frames;
    const size_t one = wf2_workspace_bytes(c->width, c->local_rows, 1, c->max_bounces);
    const size_t fit = std::max<size_t>(1, budget / std::max<size_t>(one, 1));
    return (int32_t)std::min<size_t>((size_t)n_frames, fit);
}

int do_render(jpt_ctx* c, int32_t n_frames, uint32_t first_frame_index, bool counted, bool blocking)
{
    if (!c) return JPT_E_INVALID;
    if (c->device < 0 || !c->scene_ready || !c->params_set || !c->camera_set || n_frames <= 0)
        return do_render_batch(c, n_frames, first_frame_index, counted, blocking);  // reports the error / no-op
    // a budget the caller set is a cap, not a hint: one frame is the smallest batch there is
    if (c->workspace_budget && c->kernel_variant != JPT_KERNEL_REFERENCE_LAYOUT && !c->debug_steps) {
        const size_t one = wf2_workspace_bytes(c->width, c->local_rows, 1, c->max_bounces);
        if (one > c->workspace_budget)
            return fail(c, JPT_E_LIMIT, "jpt_set_memory_policy: one frame of this resolution needs a workspace of " + std::to_string(one) +
                                            " bytes, the budget is " + std::to_string(c->workspace_budget) +
                                            " (raise it, or render with JPT_KERNEL_REFERENCE_LAYOUT, which needs none)");
    }
    const int32_t per = frames_per_batch(c, n_frames);
    if (per >= n_frames) return do_render_batch(c, n_frames, first_frame_index, counted, blocking);
    // more frames than the workspace budget holds at once: batches in frame order (the accumulation continues).  A
    // queued render queues its batches (each takes the next pipeline slot; the statistics are those of the last one) ...
    if (!blocking && !counted) {
        for (int32_t done = 0; done < n_frames; done += per) {
            const int rc = do_render_batch(c, std::min(per, n_frames - done), first_frame_index + (uint32_t)done, false, false);
            if (rc != JPT_OK) return rc;
        }
        return JPT_OK;
    }
    // ... a blocking one runs them one after another, statistics summed over the batches
    jpt_stats sum;
    std::memset(&sum, 0, sizeof sum);
    for (int32_t done = 0; done < n_frames; done += per) {
        const int32_t nb = std::min(per, n_frames - done);
        const int rc = do_render_batch(c, nb, first_frame_index + (uint32_t)done, counted, true);
        if (rc != JPT_OK) return rc;
        sum.rays += c->stats.rays;
        sum.blas_expand += c->stats.blas_expand;
        sum.tri_tests += c->stats.tri_tests;
        sum.tlas_expand += c->stats.tlas_expand;
        sum.inst_visits += c->stats.inst_visits;
        sum.shaded_hits += c->stats.shaded_hits;
        sum.last_render_ms += c->stats.last_render_ms;
        sum.last_trace_ms += c->stats.last_trace_ms;
        sum.set_aside += c->stats.set_aside;
        sum.set_aside_dropped += c->stats.set_aside_dropped;
    }
    c->stats.rays = sum.rays;
    c->stats.blas_expand = sum.blas_expand;
    c->stats.tri_tests = sum.tri_tests;
    c->stats.tlas_expand = sum.tlas_expand;
    c->stats.inst_visits = sum.inst_visits;
    c->stats.shaded_hits = sum.shaded_hits;
    c->stats.last_render_ms = sum.last_render_ms;
    c->stats.last_trace_ms = sum.last_trace_ms;
    c->stats.set_aside = sum.set_aside;
    c->stats.set_aside_dropped = sum.set_aside_dropped;
    return JPT_OK;
}

// Helper streams and events are made on first use: most contexts (tests, tools, one-off renders) never queue renders
// or render enough paths to split them.  NOTE on hardware queues: the HIP runtime multiplexes the streams of a process
// onto GPU_MAX_HW_QUEUES hardware queues (default 4) per stream priority level, in the order of their first use, and two
// streams that share a queue execute in submission order.  With all of this library's streams at the normal level the
// rate of queued renders depended on which streams happened to share a queue (round 1, C3: 1.32 ms per render with the
// most fortunate order, 1.60 and 2.39 with others); the pipeline slots now take their queues from the highest priority
// level's pool (ensure_pipe_slot below), and so do the helper streams of frame groups (round 5: below).
bool ensure_pipe_slot(jpt_ctx* c, int slot);
// frame groups: `groups - 1` helper streams (launch_wf2_render); false = not available, renders run serially
bool ensure_group_streams(jpt_ctx* c, int groups)
{
    if (!c->async.fork && hipEventCreateWithFlags(&c->async.fork, hipEventDisableTiming) != hipSuccess) {
        c->async.fork = nullptr;
        (void)hipGetLastError();
        return false;
    }
    for (int k = 0; k + 1 < groups && k < 3; k++) {
        if (!c->async.aux_stream[k]) {
            // The helper stream of group k + 1 IS pipeline slot k's stream (highest priority level by default, ensure_pipe_slot), not a
            // stream of its own at the context stream's level: the normal level's queues are dealt in the order of first use among ALL
            // the process's streams, and a helper that lands on the context stream's queue runs its group AFTER group 0 instead of
            // beside it -- 1.24 -> 1.48 ms per blocking C3 render after a counted render had run first, with six queues per level
            // (round 5, profiles/r05/r05an_...).  And not an extra stream at the slots' level either: a seventh stream there takes
            // one of the six queues the slots are measured to have to themselves (six_queues_probe).  A blocking render runs when
            // the renders queued before it have been ordered ahead of it on the context's stream; its groups queue behind whatever a
            // slot still holds.  (Normal / low slot priority chosen by the host: a stream of its own, as before.)
            const bool borrow = c->slot_priority != JPT_STREAM_PRIORITY_NORMAL && c->slot_priority != JPT_STREAM_PRIORITY_LOW;
            if (borrow) {
                if (!ensure_pipe_slot(c, k)) return false;
                c->async.aux_stream[k] = c->pipe_stream[k];
                c->aux_borrowed[k] = true;
            } else if (hipStreamCreateWithFlags(&c->async.aux_stream[k], hipStreamNonBlocking) != hipSuccess) {
                c->async.aux_stream[k] = nullptr;
                (void)hipGetLastError();
                return false;
            }
        }
        if (!c->async.join[k] && hipEventCreateWithFlags(&c->async.join[k], hipEventDisableTiming) != hipSuccess) {
            c->async.join[k] = nullptr;
            (void)hipGetLastError();
            return false;
        }
    }
    return true;
}
// render pipelining: stream + two events of slot `slot`
bool ensure_pipe_slot(jpt_ctx* c, int slot)
{
    if (c->pipe_stream[slot] && c->ev_paths_done[slot] && c->ev_acc_done[slot]) return true;
    bool ok = true;
    if (!c->pipe_stream[slot]) {
        // Streams of different PRIORITY levels take their hardware queues from different pools (three levels on this
        // device, GPU_MAX_HW_QUEUES queues each).  The slots' four streams are created at the HIGHEST level: they have that
        // level's pool to themselves -- the host's streams, torch's, RCCL's and this library's own helper streams are all
        // at the normal level -- so they never share a queue, even when the pool is the runtime's default of four, which
        // a library cannot change (the variable is read at the process's first HIP call: tools/hwq_probe.py) and should not.
        // C3 queued rate in bench.py (torch loaded, counted and blocking renders before the timed region), pool of 4 / 16:
        // all slots normal 1.339 / 1.036 ms, all high 1.052 / 1.051, dealt over the three levels 1.230 / 1.212 (the normal-level
        // slot shares a queue with host streams), all low 1.137 / 1.135 (tools/prio_probe.sh, profiles/r02/prio_probe.txt).
        int least = 0, greatest = 0;
        // (the embedding application decides per context with jpt_set_stream_priority; the default is the highest level)
        const bool normal = c->slot_priority == JPT_STREAM_PRIORITY_NORMAL, low = c->slot_priority == JPT_STREAM_PRIORITY_LOW;
        if (!normal && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least > greatest)
            ok = hipStreamCreateWithPriority(&c->pipe_stream[slot], hipStreamNonBlocking, low ? least : greatest) == hipSuccess;
        else
            ok = hipStreamCreateWithFlags(&c->pipe_stream[slot], hipStreamNonBlocking) == hipSuccess;
    }
    if (ok && !c->ev_paths_done[slot]) ok = hipEventCreateWithFlags(&c->ev_paths_done[slot], hipEventDisableTiming) == hipSuccess;
    if (ok && !c->ev_acc_done[slot]) ok = hipEventCreateWithFlags(&c->ev_acc_done[slot], hipEventDisableTiming) == hipSuccess;
    if (!ok) (void)hipGetLastError();
    return ok;
}

// Do SIX of the slots' streams run side by side?  A stream runs on one of the hardware queues of its priority level's pool --
// GPU_MAX_HW_QUEUES per level, four by default, fixed at the process's first HIP call -- and streams that share a queue run one after
// the other: with the default pool a fifth and sixth render in flight cost 15-20 % (r04ad), with a pool of six or more they are worth
// 4 % on C3 and 15 % on small renders (profiles/r05/r05aj_slots_hw_queues_chain_sweep.txt).  The environment is the host's to set
// (bench.py and the Python binding do; jpt_create tries, jpt.h); what the library does is LOOK: one wave per slot stream busy for
// 0.4 ms, all six launched together -- side by side they are done in about that time, on four queues in about twice that.  Once per
// context, ~1 ms, before its first queued render.
bool six_queues_probe(jpt_ctx* c)
{
    if (c->six_queues >= 0) return c->six_queues != 0;
    c->six_queues = 0;
    constexpr int kProbe = 6;
    for (int k = 0; k < kProbe; k++)
        if (!ensure_pipe_slot(c, k)) return false;
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device) != hipSuccess || khz <= 0) {
        (void)hipGetLastError();
        return false;
    }
    const long long ticks = (long long)khz * 400 / 1000;   // 0.4 ms
    for (int k = 0; k < kProbe; k++) launch_queue_spin(c->pipe_stream[k], 1);   // (the module is loaded, the queues exist)
    for (int k = 0; k < kProbe; k++)
        if (hipStreamSynchronize(c->pipe_stream[k]) != hipSuccess) return false;
    double best = 1e9;
    for (int rep = 0; rep < 2; rep++) {   // the better of two: a host hiccup can only make it look serial
        const auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < kProbe; k++) launch_queue_spin(c->pipe_stream[k], ticks);
        for (int k = 0; k < kProbe; k++)
            if (hipStreamSynchronize(c->pipe_stream[k]) != hipSuccess) return false;
        best = std::min(best, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    if (hipGetLastError() != hipSuccess) return false;
    c->six_queues = best < 0.4 * 1.6 ? 1 : 0;
    return c->six_queues != 0;
}

int do_render_batch(jpt_ctx* c, int32_t n_frames, uint32_t first_frame_index, bool counted, bool blocking)
{
    if (!c) return JPT_E_INVALID;
    if (c->device < 0) return fail(c, JPT_E_DEVICE, "host-only context (JPT_DEVICE_HOST_ONLY) cannot render: there is no CPU fallback");
    if (!c->scene_ready) return fail(c, JPT_E_STATE, "no scene: call jpt_scene_upload_reference_layout or jpt_scene_commit first");
    if (!c->params_set) return fail(c, JPT_E_STATE, "jpt_set_params not called");
    if (!c->camera_set) return fail(c, JPT_E_STATE, "jpt_set_camera not called");
    if (n_frames < 0) return fail(c, JPT_E_INVALID, "n_frames < 0");
    if (c->refit_active && (c->kernel_variant != JPT_KERNEL_WAVEFRONT || c->debug_steps))
        return fail(c, JPT_E_STATE, "jpt_scene_refit_tlas refits the default kernel's records only: call jpt_scene_update_tlas before rendering with another kernel");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    c->assembled = c->assembled_ldr = false;
    if (c->denoise == JPT_DENOISE_TEMPORAL) {
        // the pass runs once per displayed frame and reads other pixels' depth and history (temporal_reprojection.glsl:57-61)
        if (n_frames > 1) return fail(c, JPT_E_INVALID, "temporal reprojection renders one frame per call");
        if (c->world != 1) return fail(c, JPT_E_STATE, "temporal reprojection needs the whole image in one context (world == 1)");
        if (!c->temporal_set) return fail(c, JPT_E_STATE, "jpt_set_temporal_params not called");
        if (c->temporal.width != c->width || c->temporal.height != c->height)
            return fail(c, JPT_E_INVALID, "temporal RenderParameters width/height differ from jpt_set_params");
        const size_t npx = (size_t)c->width * c->height;
        if (!c->hist_valid || c->d_hist1.n != npx) {
            // Image::create(...) zero-filled history images (temporal_reprojection.cpp:42-43)
            HIP_TRY(c, hipStreamSynchronize(s));
            HIP_TRY(c, c->d_hist1.resize(npx));
            HIP_TRY(c, c->d_hist2.resize(npx));
            if (npx) {
                HIP_TRY(c, hipMemsetAsync(c->d_hist1.p, 0, npx * sizeof(float4), s));
                HIP_TRY(c, hipMemsetAsync(c->d_hist2.p, 0, npx * sizeof(float4), s));
            }
            c->hist_valid = true;
            c->hist_written = nullptr;
        }
    }
    DevCounters* cnt = nullptr;
    if (counted) {
        cnt = c->d_counters.p;
        HIP_TRY(c, hipMemsetAsync(cnt, 0, sizeof(DevCounters), s));
    }
    const bool wavefront = c->kernel_variant == JPT_KERNEL_WAVEFRONT && !c->debug_steps;   // (DEBUG_STEPS exists in the audit kernel only)
    const bool wf2 = wavefront;
    const int nq = c->max_bounces + 2;
    if (wavefront && c->local_rows > 0 && c->width > 0 && n_frames > 0) {
        const size_t need = wf2_workspace_bytes(c->width, c->local_rows, n_frames, c->max_bounces);
        if (c->d_workspace.n < need) {
            HIP_TRY(c, hipStreamSynchronize(s));
            HIP_TRY(c, c->d_workspace.resize(need));
        }
    }
    HIP_TRY(c, hipEventRecord(c->ev0, s));
    bool pipelined = false;
    if (c->local_rows > 0 && c->width > 0 && n_frames > 0) {
        FrameParams fp;
        fp.width = c->width;
        fp.height = c->height;
        fp.local_rows = c->local_rows;
        fp.rank = c->rank;
        fp.world = c->world;
        fp.max_bounces = c->max_bounces;
        fp.accum_mode = c->accum_mode;
        fp.display_mode = c->denoise == JPT_DENOISE_PROGRESSIVE ? 0 : 1;
        fp.debug_steps = c->debug_steps ? 1 : 0;
        // the depth image (main.glsl:432,435) has one reader, the temporal pass: off when the host said so (jpt_set_outputs)
        const bool want_depth = (c->outputs & JPT_OUTPUT_DEPTH) != 0u || c->denoise == JPT_DENOISE_TEMPORAL;
        float* const depth_img = want_depth ? c->d_depth.p : nullptr;
        c->depth_valid = want_depth;
        if (wavefront) {
            fp.frame_index = first_frame_index;
            fp.frame_count = c->frame_count + 1;
            fp.n_frames = n_frames;
            fp.depth_frame = want_depth ? n_frames - 1 : -1;
            const size_t need_ev = c->kernel_timing ? 2 * (size_t)(c->max_bounces + 1) : 0;
            while (c->trace_events.size() < need_ev) {
                hipEvent_t e;
                HIP_TRY(c, hipEventCreate(&e));
                c->trace_events.push_back(e);
            }
            c->trace_events_used = (int32_t)need_ev;
            if (wf2) compute_sky_cull(c, c->async.cull);
            // the sky cells of whole tiles, for wf2_accumulate (REF_LDR8 sums of several frames): on the context's stream, which every
            // accumulation is ordered behind; again only when the camera (its frame index aside), the image size or the partition changed
            c->async.sky_tiles = nullptr;
            if (wf2 && c->accum_mode == JPT_ACCUM_REF_LDR8 && n_frames > 1) {
                const int32_t key[5] = {c->width, c->height, c->local_rows, c->rank, c->world};
                RefCamera cam_key = c->camera;
                cam_key.frame_index = 0;
                const size_t n_tiles = wf2_sky_tile_count(c->width, c->local_rows);
                if (!c->sky_tiles_valid || c->d_sky_tiles.n < n_tiles || std::memcmp(key, c->sky_tiles_key, sizeof key) != 0 ||
                    std::memcmp(&cam_key, &c->sky_tiles_camera, sizeof cam_key) != 0) {
                    if (c->d_sky_tiles.n < n_tiles) {
                        HIP_TRY(c, hipStreamSynchronize(s));   // (renders in flight may read the old buffer)
                        for (int k = 0; k < jpt_ctx::kPipeSlots; k++)
                            if (c->pipe_stream[k]) HIP_TRY(c, hipStreamSynchronize(c->pipe_stream[k]));
                        HIP_TRY(c, c->d_sky_tiles.resize(n_tiles));
                    }
                    launch_sky_tiles(s, fp, c->camera, c->d_sky_tiles.p);
                    std::memcpy(c->sky_tiles_key, key, sizeof key);
                    c->sky_tiles_camera = cam_key;
                    c->sky_tiles_valid = true;
                }
                c->async.sky_tiles = c->d_sky_tiles.p;
            }
            const bool pipelining = tuning().pipelining;
            // renders in flight (JPT_PIPE_SLOTS overrides): a render is eleven dependent launches of >= 25-30 us each
            // however little work it holds, so several of them are needed to fill the chip
            const int forced_slots = [c] {
                if (c->max_slots > 0) return c->max_slots;   // the embedding application's cap (jpt_set_memory_policy)
                const int k = tuning().pipe_slots;
                return k <= 0 ? 0 : (k < 2 ? 2 : (k > jpt_ctx::kPipeSlots ? jpt_ctx::kPipeSlots : k));
            }();
            // Four renders in flight, each tracing with a quarter of the blocks (four consecutive segments per block: a
            // deeper queue keeps a block's lanes refilled for a larger share of its launch) beat two renders of full-width
            // launches on every size tried: C3 1.32 vs 1.45 ms, C2 0.36 vs 0.46, 1 frame 0.30 vs 0.37, 1920x136 0.185 vs
            // 0.193, 4K x 16 spp 11.1 vs 11.1.  Renders whose workspace exceeds 24 GiB keep two slots and full-width launches.
            const size_t one_workspace = wf2 ? wf2_workspace_bytes(c->width, c->local_rows, n_frames, c->max_bounces) : 0;
            const bool huge = one_workspace > ((size_t)24 << 30);  // 4 x 24 GiB of workspaces is where this stops
            // ... six where the slots' streams have six hardware queues to themselves (six_queues_probe), four otherwise
            const bool may_queue = wf2 && pipelining && !blocking && !counted;
            const int pipe_slots = forced_slots ? forced_slots : (huge ? 2 : ((may_queue && six_queues_probe(c)) ? 6 : 4));
            const int slot = (int)(c->async_seq % (uint64_t)pipe_slots);
            if (may_queue) c->last_pipe_slots = pipe_slots;
            // A host that queues ONE render at a time (jpt_render_async, its own work, jpt_sync or the split read-back, again)
            // never has a second render in flight: the pipelined form's quarter-width launches then run alone, and a render takes
            // half as long again as a blocking one (C3 1.93 against 1.26 ms, 3840 x 2160 x 4 spp 3.33 against 2.11:
            // tools/lone_async_probe.py).  The third queued render in a row that finds nothing in flight -- the event every render
            // leaves on the context's stream has completed -- is launched like a blocking one instead (frame groups, full-width
            // launches; nobody waits for it here), and so on until a render finds work in flight.  A queue of renders is not
            // affected: only its first render finds the pipeline empty.
            bool lone_async = false;
            if (wf2 && pipelining && !blocking && !counted && !need_ev) {
                const bool idle = hipEventQuery(c->ev1) == hipSuccess;
                (void)hipGetLastError();   // (hipErrorNotReady is an answer, not an error)
                c->idle_streak = idle ? c->idle_streak + 1 : 0;
                lone_async = c->idle_streak >= 3;
            }
            if (wf2 && pipelining && !blocking && !counted && !need_ev && !lone_async && ensure_pipe_slot(c, slot)) {
                // asynchronous render: path kernels on a helper stream + the other workspace; the accumulation on `s`
                hipStream_t ps = c->pipe_stream[slot];
                DevBuf<char>& ws = slot ? c->d_workspace_more[slot - 1] : c->d_workspace;
                const size_t need = wf2_workspace_bytes(c->width, c->local_rows, n_frames, c->max_bounces);
                bool slots_ready = true;
                for (int k = 0; k < pipe_slots; k++)
                    slots_ready = slots_ready && c->pipe_stream[k] && (k ? c->d_workspace_more[k - 1] : c->d_workspace).n >= need;
                if (!slots_ready) {
                    // The first queued render of this size prepares ALL the slots, so that none of it lands in the middle
                    // of a queue of renders: a stream that gets its own hardware queue costs ~6 ms to create, a device
                    // allocation waits for the device, and a fresh 2 GB allocation costs its first user ~10 ms.
                    HIP_TRY(c, hipStreamSynchronize(s));  // every earlier render ends with a kernel on `s`
                    for (int k = 0; k < pipe_slots; k++) {
                        DevBuf<char>& wk = k ? c->d_workspace_more[k - 1] : c->d_workspace;
                        if (wk.n < need) {
                            HIP_TRY(c, wk.resize(need));
                            HIP_TRY(c, hipMemsetAsync(wk.p, 0, need, s));  // first touch
                            c->acc_done_valid[k] = false;
                        }
                        if (!ensure_pipe_slot(c, k)) return fail(c, JPT_E_DEVICE, "cannot create the stream of a pipeline slot");
                    }
                }
                // this workspace was last read by the accumulation of the render `pipe_slots` renders ago; when that is
                // not on record (first use of the slot, or renders that went through `s` itself since), wait for
                // whatever `s` holds now
                if (!c->acc_done_valid[slot]) HIP_TRY(c, hipEventRecord(c->ev_acc_done[slot], s));
                HIP_TRY(c, hipStreamWaitEvent(ps, c->ev_acc_done[slot], 0));
                if (c->slot_refit_seen[slot] != c->refit_wait_seq) {
                    // the instance level this render reads was refitted on the refit stream (jpt_scene_refit_tlas)
                    HIP_TRY(c, hipStreamWaitEvent(ps, c->ev_refit_done, 0));
                    c->slot_refit_seen[slot] = c->refit_wait_seq;
                }
                // (one frame group: overlapping with the neighbouring render does what groups do, without extra launches:
                // 3840x2160x16 spp 11.56 ms against 11.96 with two groups, instanced scene 3.80 against 4.05)
                Wf2Async one_group = c->async;
                one_group.aux_stream[0] = nullptr;
                one_group.trace_chain = (huge || pipe_slots == 1) ? 1 : 4;   // (a single render in flight: full-width launches)
                // The accumulation runs on the slot's stream too, after whatever `s` holds now (the previous render's accumulation, an
                // upload, a read-back), and `s` then waits for it: the results are those of serial execution, and `s` itself carries no
                // kernels of a render (the hardware runs a handful of queues side by side; a busy `s` would be one more)
                HIP_TRY(c, hipEventRecord(c->ev_paths_done[slot], s));
                one_group.before_acc = c->ev_paths_done[slot];
                launch_wf2_render(ps, c->ds, fp, c->camera, ws.p, c->d_accum.p, c->d_ldr.p, depth_img, nullptr, nullptr, one_group);
                HIP_TRY(c, hipEventRecord(c->ev_acc_done[slot], ps));
                HIP_TRY(c, hipStreamWaitEvent(s, c->ev_acc_done[slot], 0));
                c->acc_done_valid[slot] = true;
                c->async_seq++;
                pipelined = true;
            } else if (wf2) {
                // (launch_wf2_render splits the frames into groups only if the helper streams exist)
                const int want_groups = wf2_wanted_groups(n_frames, (size_t)c->width * (size_t)c->local_rows * (size_t)n_frames);
                if (want_groups > 1 && !need_ev && !counted) (void)ensure_group_streams(c, want_groups);
                // A SMALL render that runs alone (the addon's use: one blocking 1-spp frame per Godot frame): two chained
                // segments per tracing block.  With a few rays per lane a segment's queue runs dry almost at once; half as
                // many waves with queues twice as deep keep their lanes fuller (C2 0.779 -> 0.738 ms, a 1-spp 1080p frame
                // 0.673 -> 0.648).  Not for windows of millions of paths (C3 in one group: 1.39 -> 1.51 ms) nor for scenes
                // past the caches, which are bound by the latency of their fetches and want every wave they can get (1 M
                // triangles, 2 spp: 9.49 -> 9.93 ms): one segment per block (profiles/r03/r03ao_lone_chain.txt).
                Wf2Async lone = c->async;
                {
                    FrameParams wfp;
                    wfp.width = c->width;
                    wfp.height = c->height;
                    wfp.local_rows = c->local_rows;
                    wfp.rank = c->rank;
                    wfp.world = c->world;
                    const uint64_t window_paths =
                        ((uint64_t)c->width * (uint64_t)c->local_rows - wf2_pixels_outside_window(c->async.cull, wfp)) * (uint64_t)n_frames;
                    const size_t walked_bytes = c->wide.blas_nodes4.size() * sizeof(WideNodeQ) + c->wide.tris.size() * sizeof(WideTri);
                    const bool small = window_paths <= 1500000u && walked_bytes <= ((size_t)32 << 20);
                    lone.trace_chain = (c->native_tree && !c->ref_is_exact && small) ? 2 : 1;
                }
                launch_wf2_render(s, c->ds, fp, c->camera, c->d_workspace.p, c->d_accum.p, c->d_ldr.p, depth_img, cnt,
                                  need_ev ? c->trace_events.data() : nullptr, lone);
            }
        } else {
            for (int32_t f = 0; f < n_frames; f++) {
                fp.frame_index = first_frame_index + (uint32_t)f;
                fp.frame_count = c->frame_count + (uint32_t)f + 1;
                fp.n_frames = 1;
                fp.depth_frame = 0;
                launch_ref_frame(s, c->ds, fp, c->camera, c->d_accum.p, c->d_ldr.p, depth_img, cnt);
            }
        }
        if (c->denoise == JPT_DENOISE_TEMPORAL) {
            // TemporalReprojection::render's dispatch (temporal_reprojection.cpp:71): screen + depth of this frame in,
            // blended history and the displayed screen out
            launch_temporal(s, c->temporal, c->d_ldr.p, c->d_depth.p, c->d_hist1.p, c->d_hist2.p);
            c->hist_written = (c->temporal.frame_count % 2u) == 0u ? c->d_hist2.p : c->d_hist1.p;
        }
        HIP_TRY(c, hipGetLastError());
    }
    HIP_TRY(c, hipEventRecord(c->ev1, s));
    if (wavefront && !pipelined) c->acc_done_valid[0] = false;  // d_workspace was used on `s` itself: see the pipelined branch
    c->frame_count += (uint32_t)n_frames;
    c->stats.frames = c->frame_count;
    if (blocking || counted) {
        HIP_TRY(c, hipEventSynchronize(c->ev1));
        float ms = 0.0f;
        HIP_TRY(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
        c->stats.last_render_ms = ms;
        c->stats.last_trace_ms = ms;
        if (wavefront && c->trace_events_used > 0 && n_frames > 0 && c->local_rows > 0) {
            double tms = 0.0;
            for (int32_t k = 0; k + 1 < c->trace_events_used; k += 2) {
                float t = 0.0f;
                if (hipEventElapsedTime(&t, c->trace_events[(size_t)k], c->trace_events[(size_t)k + 1]) == hipSuccess) {
                    tms += t;
                    if (k == 0) c->stats.last_primary_ms = t;
                }
            }
            c->stats.last_trace_ms = tms;
        }
        if (wavefront && c->d_workspace.p && n_frames > 0 && c->local_rows > 0) {
            // ray segments traced = sum of the per-bounce queue sizes (always available on this route)
            const size_t per_row = wf2 ? (size_t)wf2_segments() : 1u;
            c->h_qcount.assign((size_t)nq * per_row + 2u, 0u);   // queue sizes, then the set-aside counts
            HIP_TRY(c, hipMemcpy(c->h_qcount.data(), c->d_workspace.p, c->h_qcount.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
            c->stats.set_aside = c->h_qcount[(size_t)nq * per_row];
            c->stats.set_aside_dropped = c->h_qcount[(size_t)nq * per_row + 1u];
            // every in-image (pixel, frame) has one primary segment; rows 1.. hold the later bounces' queue sizes
            uint64_t rays = (uint64_t)c->width * (uint64_t)c->local_rows * (uint64_t)n_frames;
            for (int b = 1; b <= c->max_bounces; b++)
                for (size_t k = 0; k < per_row; k++) rays += c->h_qcount[(size_t)b * per_row + k];
            c->stats.rays = rays;
        }
        if (counted) {
            DevCounters h;
            HIP_TRY(c, hipMemcpy(&h, cnt, sizeof h, hipMemcpyDeviceToHost));
            // (pixel, frame) pairs outside the render's window were never enumerated on the device: each is one primary
            // segment that ends at the TLAS root, like the culled ones inside the window
            uint64_t outside = 0;
            if (wf2) {
                FrameParams wfp;
                wfp.width = c->width;
                wfp.height = c->height;
                wfp.local_rows = c->local_rows;
                wfp.rank = c->rank;
                wfp.world = c->world;
                outside = wf2_pixels_outside_window(c->async.cull, wfp) * (uint64_t)n_frames;
            }
            h.phase[7] += outside;
            c->stats.rays = h.rays + outside;
            c->stats.blas_expand = h.blas_expand;
            c->stats.tri_tests = h.tri_tests;
            c->stats.tlas_expand = h.tlas_expand + outside;
            c->stats.inst_visits = h.inst_visits;
            c->stats.shaded_hits = h.shaded_hits;
            for (int k = 0; k < 8; k++) c->stats.phase[k] = h.phase[k];
            c->stats.sky_culled = h.phase[7];
            c->stats.walk_steps_max = h.walk_max;
            for (int k = 0; k < 8; k++) c->stats.walk_steps_hist[k] = h.walk_hist[k];
            c->stats.zero_throughput = h.zero_thr;
        }
    }
    return JPT_OK;
}

// local strip-major rows -> full image rows, on the host
template <typename T>
void scatter_rows(const T* local, T* out, int32_t width, int32_t height, int32_t rank, int32_t world, int comps)
{
    const int32_t n_strips = (height + kStripRows - 1) / kStripRows;
    int32_t ly = 0;
    for (int32_t s = rank; s < n_strips; s += world) {
        for (int32_t r = 0; r < kStripRows; r++) {
            const int32_t y = s * kStripRows + r;
            if (y >= height) break;
            std::memcpy(out + (size_t)y * width * comps, local + (size_t)ly * width * comps, (size_t)width * comps * sizeof(T));
            ly++;
        }
    }
}
template <typename T>
void scatter_rows(const std::vector<T>& local, T* out, int32_t width, int32_t height, int32_t rank, int32_t world, int comps)
{
    scatter_rows(local.data(), out, width, height, rank, world, comps);
}

// Blocking read-backs (jpt_read_ldr_rgba8 / accum_f32 / depth_f32): device -> the context's pinned read buffer -> the caller's
// memory.  A device-to-host copy straight into pageable memory runs at ~3 GB/s (8 MB of display image: 2.7 ms; 33 MB of float4
// sums: 11 ms); through pinned memory the copy runs at the link's rate and the host copy at memory speed.  (A buffer of its
// own: the split read-back's staging buffer may hold an image in flight.)
int staged_read(jpt_ctx* c, const void* src, size_t bytes)
{
    if (c->h_read_bytes < bytes) {
        if (c->h_read_pinned) (void)hipHostFree(c->h_read_pinned);
        c->h_read_pinned = nullptr;
        c->h_read_bytes = 0;
        HIP_TRY(c, hipHostMalloc(&c->h_read_pinned, bytes, hipHostMallocDefault));
        c->h_read_bytes = bytes;
    }
    if (bytes) {
        HIP_TRY(c, hipMemcpyAsync(c->h_read_pinned, src, bytes, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return JPT_OK;
}

// the pinned staging buffer of the display image's split read-backs (one full image)
int ensure_ldr_pinned(jpt_ctx* c)
{
    if (c->h_ldr_pinned_px < (size_t)c->width * c->height) {
        if (c->h_ldr_pinned) (void)hipHostFree(c->h_ldr_pinned);
        c->h_ldr_pinned = nullptr;
        c->h_ldr_pinned_px = 0;
        HIP_TRY(c, hipHostMalloc((void**)&c->h_ldr_pinned, (size_t)c->width * c->height * sizeof(uint32_t), hipHostMallocDefault));
        c->h_ldr_pinned_px = (size_t)c->width * c->height;
    }
    return JPT_OK;
}

}  // namespace

extern "C" {

int jpt_abi_version(void) { return JPT_ABI_VERSION; }

int jpt_create(int device_id, jpt_ctx** out)
{
    if (!out) return JPT_E_INVALID;
    *out = nullptr;
    if (device_id == JPT_DEVICE_HOST_ONLY) {  // builder-only context: no device is touched
        jpt_ctx* c = new (std::nothrow) jpt_ctx();
        if (!c) return JPT_E_DEVICE;
        c->device = -1;
        std::memset(&c->stats, 0, sizeof c->stats);
        std::memset(&c->camera, 0, sizeof c->camera);
        *out = c;
        return JPT_OK;
    }
    // (Six renders in flight want six hardware queues for their streams; the runtime makes GPU_MAX_HW_QUEUES per priority level, four
    // by default, and reads the variable at the process's first HIP call.  That is the HOST's to export before it starts threads --
    // jpt.h "Process environment"; the library never writes the environment, it only measures what it got: six_queues_probe.)
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        g_create_error = std::string("no HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
        return JPT_E_DEVICE;
    }
    if (device_id < 0 || device_id >= n) {
        g_create_error = "device id out of range";
        return JPT_E_INVALID;
    }
    jpt_ctx* c = new (std::nothrow) jpt_ctx();
    if (!c) return JPT_E_DEVICE;
    c->device = device_id;
    std::memset(&c->stats, 0, sizeof c->stats);
    std::memset(&c->camera, 0, sizeof c->camera);
    if ((e = hipSetDevice(device_id)) != hipSuccess || (e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreate(&c->ev0)) != hipSuccess || (e = hipEventCreate(&c->ev1)) != hipSuccess) {
        g_create_error = std::string("HIP init: ") + hipGetErrorString(e);
        delete c;
        return JPT_E_DEVICE;
    }
    c->stream = c->own_stream;
    (void)tuning();  // environment switches are read here, once per process
    *out = c;
    return JPT_OK;
}

void jpt_destroy(jpt_ctx* c)
{
    if (!c) return;
    if (c->device < 0) {
        delete c;
        return;
    }
    (void)hipSetDevice(c->device);
    if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    for (hipEvent_t e : c->trace_events) (void)hipEventDestroy(e);
    if (c->ev_readback) (void)hipEventDestroy(c->ev_readback);
    if (c->h_ldr_pinned) (void)hipHostFree(c->h_ldr_pinned);
    if (c->h_read_pinned) (void)hipHostFree(c->h_read_pinned);
    if (c->refit_stream) { (void)hipStreamSynchronize(c->refit_stream); (void)hipStreamDestroy(c->refit_stream); }
    for (int k = 0; k < jpt_ctx::kInstanceSets; k++) if (c->ev_set_retired[k]) (void)hipEventDestroy(c->ev_set_retired[k]);
    if (c->ev_refit_done) (void)hipEventDestroy(c->ev_refit_done);
    for (int k = 0; k < jpt_ctx::kRefitStages; k++) {
        if (c->h_refit_t12[k]) (void)hipHostFree(c->h_refit_t12[k]);
        if (c->ev_refit_copied[k]) (void)hipEventDestroy(c->ev_refit_copied[k]);
    }
    for (int k = 0; k < 3; k++) {
        if (c->async.aux_stream[k]) {
            (void)hipStreamSynchronize(c->async.aux_stream[k]);
            if (!c->aux_borrowed[k]) (void)hipStreamDestroy(c->async.aux_stream[k]);   // (a borrowed one goes with the pipeline slots below)
        }
        if (c->async.join[k]) (void)hipEventDestroy(c->async.join[k]);
    }
    if (c->async.fork) (void)hipEventDestroy(c->async.fork);
    for (int k = 0; k < jpt_ctx::kPipeSlots; k++) {
        if (c->pipe_stream[k]) {
            (void)hipStreamSynchronize(c->pipe_stream[k]);
            (void)hipStreamDestroy(c->pipe_stream[k]);
        }
        if (c->ev_paths_done[k]) (void)hipEventDestroy(c->ev_paths_done[k]);
        if (c->ev_acc_done[k]) (void)hipEventDestroy(c->ev_acc_done[k]);
    }
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

const char* jpt_last_error(const jpt_ctx* c) { return c ? c->error.c_str() : g_create_error.c_str(); }

int jpt_set_stream(jpt_ctx* c, void* hip_stream)
{
    if (!c || c->device < 0) return JPT_E_INVALID;
    hipStream_t next = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    if (next != c->stream) {
        // everything this context has queued is ordered by the old stream (uploads, accumulation kernels, the
        // events its helper streams wait on): drain it before work starts appearing on another one
        HIP_TRY(c, hipSetDevice(c->device));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        for (int k = 0; k < jpt_ctx::kPipeSlots; k++) c->acc_done_valid[k] = false;
        c->stream = next;
    }
    return JPT_OK;
}

int jpt_renders_in_flight(const jpt_ctx* c) { return c ? c->last_pipe_slots : 0; }

int jpt_set_stream_priority(jpt_ctx* c, int32_t priority)
{
    if (!c) return JPT_E_INVALID;
    if (priority < JPT_STREAM_PRIORITY_DEFAULT || priority > JPT_STREAM_PRIORITY_LOW) return fail(c, JPT_E_INVALID, "unknown stream priority");
    if (priority == c->slot_priority) return JPT_OK;
    c->slot_priority = priority;
    if (c->device < 0) return JPT_OK;
    // the pipeline slots' streams are made on first use: drop the ones that exist, the next queued render makes new ones
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (int k = 0; k < 3; k++) {   // the frame groups' helper streams follow: borrowed slot streams go with the slots, own ones are re-made
        if (c->async.aux_stream[k]) {
            (void)hipStreamSynchronize(c->async.aux_stream[k]);
            if (!c->aux_borrowed[k]) (void)hipStreamDestroy(c->async.aux_stream[k]);
            c->async.aux_stream[k] = nullptr;
            c->aux_borrowed[k] = false;
        }
    }
    c->six_queues = -1;   // (measured again for the new streams)
    for (int k = 0; k < jpt_ctx::kPipeSlots; k++) {
        if (c->pipe_stream[k]) {
            (void)hipStreamSynchronize(c->pipe_stream[k]);
            (void)hipStreamDestroy(c->pipe_stream[k]);
            c->pipe_stream[k] = nullptr;
        }
        c->acc_done_valid[k] = false;
    }
    return JPT_OK;
}

int jpt_set_memory_policy(jpt_ctx* c, int32_t renders_in_flight, uint64_t workspace_budget_bytes)
{
    if (!c) return JPT_E_INVALID;
    if (renders_in_flight < 0 || renders_in_flight > jpt_ctx::kPipeSlots) return fail(c, JPT_E_INVALID, "renders_in_flight must be 0 (the library's rule) or 1..8");
    c->max_slots = renders_in_flight;
    c->workspace_budget = workspace_budget_bytes;
    if (c->device < 0) return JPT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    // drain, then give back what the new policy no longer allows: the workspaces of slots past the cap, and any workspace
    // larger than the new budget (the next render allocates what it needs)
    for (int k = 0; k < jpt_ctx::kPipeSlots; k++)
        if (c->pipe_stream[k]) HIP_TRY(c, hipStreamSynchronize(c->pipe_stream[k]));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const int keep = renders_in_flight > 0 ? renders_in_flight : jpt_ctx::kPipeSlots;
    for (int k = 0; k < jpt_ctx::kPipeSlots; k++) {
        DevBuf<char>& w = k ? c->d_workspace_more[k - 1] : c->d_workspace;
        if (k >= keep || (workspace_budget_bytes && w.n > workspace_budget_bytes)) {
            w.release();
            c->acc_done_valid[k] = false;
        }
    }
    c->async_seq = 0;   // the next queued render starts at slot 0 again
    release_read_staging(c);   // (the pinned staging of the blocking read-backs: re-made by the next one)
    return JPT_OK;
}

int jpt_get_workspace_bytes(jpt_ctx* c, uint64_t* bytes_out)
{
    if (!c || !bytes_out) return JPT_E_INVALID;
    uint64_t n = c->d_workspace.n;
    for (int k = 0; k + 1 < jpt_ctx::kPipeSlots; k++) n += c->d_workspace_more[k].n;
    *bytes_out = n;
    return JPT_OK;
}

int jpt_get_stream(jpt_ctx* c, void** hip_stream)
{
    if (!c || !hip_stream) return JPT_E_INVALID;
    if (c->device < 0) return fail(c, JPT_E_DEVICE, "host-only context has no stream");
    *hip_stream = (void*)c->stream;
    return JPT_OK;
}

int jpt_scene_upload_reference_layout(jpt_ctx* c, const void* tri_geometry, uint32_t n_triangles, const void* tri_data,
                                      const void* materials, uint32_t n_materials, const void* bvh_nodes, uint32_t n_bvh_nodes,
                                      const void* blas_instances, uint32_t n_instances, const void* tlas_nodes,
                                      uint32_t n_tlas_nodes, const uint8_t* tex_rgba8, int32_t tex_res, int32_t n_layers)
{
    if (!c) return JPT_E_INVALID;
    if ((n_triangles && (!tri_geometry || !tri_data)) || (n_materials && !materials) || (n_bvh_nodes && !bvh_nodes) ||
        (n_instances && !blas_instances) || (n_tlas_nodes && !tlas_nodes))
        return fail(c, JPT_E_INVALID, "null buffer with non-zero count");
    if (n_tlas_nodes > 65536) return fail(c, JPT_E_LIMIT, "TLAS has more nodes than 16-bit child indices address (bvh.h:59)");
    if (n_instances > 32767) return fail(c, JPT_E_LIMIT, "more instances than a TLAS with 16-bit child indices can hold (bvh.h:59; 32 767)");
    const auto t0 = std::chrono::steady_clock::now();
    c->scene_ready = c->host_scene_ready = false;
    c->from_commit = false;
    c->upload_note.clear();
    RefScene up;
    auto put = [](auto& vec, const void* src, uint32_t n) {
        using T = typename std::remove_reference<decltype(vec)>::type::value_type;
        vec.resize(n);
        if (n) std::memcpy(vec.data(), src, (size_t)n * sizeof(T));
    };
    put(up.tri_geom, tri_geometry, n_triangles);
    put(up.tri_data, tri_data, n_triangles);
    put(up.materials, materials, n_materials);
    put(up.bvh_nodes, bvh_nodes, n_bvh_nodes);
    put(up.instances, blas_instances, n_instances);
    put(up.tlas_nodes, tlas_nodes, n_tlas_nodes);
    if (tex_rgba8 && tex_res > 0 && n_layers > 0) {
        up.textures.assign(tex_rgba8, tex_rgba8 + (size_t)tex_res * tex_res * 4 * n_layers);
        up.tex_res = tex_res;
        up.n_layers = n_layers;
    }
    int rc = validate_ref_scene(c, up);
    if (rc != JPT_OK) return rc;
    // Default: the kernels walk the NATIVE tree (four-child quantised records over the uploaded triangles) and the boxes of
    // the uploaded leaves / TLAS leaves become the reach records that keep the image the reference's (jpt_builder.h,
    // native_from_uploaded) -- the addon keeps GeometryGroup3D::build() and gets the fast route's rate.  Arrays the reach
    // rule does not apply to, JPT_UPLOAD_WALK_AS_GIVEN and JPT_UPLOAD_WALK=given are walked node for node as uploaded.
    bool native = c->upload_mode == JPT_UPLOAD_NATIVE_TREE && !tuning().upload_as_given;
    if (native) {
        std::string why;
        native = native_from_uploaded(up, c->ref, why);
        if (!native) c->upload_note = "reference-layout upload is walked as given: " + why;
    } else {
        c->upload_note = "reference-layout upload is walked as given: requested";
    }
    if (!native) {
        c->ref.clear();
        c->ref.tri_geom = std::move(up.tri_geom);
        c->ref.tri_data = std::move(up.tri_data);
        c->ref.materials = std::move(up.materials);
        c->ref.bvh_nodes = std::move(up.bvh_nodes);
        c->ref.instances = std::move(up.instances);
        c->ref.tlas_nodes = std::move(up.tlas_nodes);
        c->ref.textures = std::move(up.textures);
        c->ref.tex_res = up.tex_res;
        c->ref.n_layers = up.n_layers;
    }
    c->ref_is_exact = false;
    c->native_tree = native;
    c->build_mode = native ? BuildMode::Sah : BuildMode::ReferenceExact;
    rc = upload_scene(c);
    c->stats.last_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

int jpt_set_upload_mode(jpt_ctx* c, int32_t mode)
{
    if (!c) return JPT_E_INVALID;
    if (mode != JPT_UPLOAD_NATIVE_TREE && mode != JPT_UPLOAD_WALK_AS_GIVEN) return fail(c, JPT_E_INVALID, "unknown upload mode");
    c->upload_mode = mode;
    return JPT_OK;
}

int jpt_scene_tree_kind(jpt_ctx* c)
{
    if (!c) return JPT_E_INVALID;
    if (!c->host_scene_ready) return JPT_TREE_NONE;
    if (c->ref_is_exact) return JPT_TREE_REFERENCE_EXACT;
    if (!c->native_tree) return JPT_TREE_AS_GIVEN;
    return c->build_mode == BuildMode::Sah ? JPT_TREE_NATIVE_REACH : JPT_TREE_NATIVE_WATERTIGHT;
}

const char* jpt_scene_upload_note(const jpt_ctx* c) { return c ? c->upload_note.c_str() : ""; }

int jpt_scene_ties_exact(jpt_ctx* c, const char** why_out)
{
    if (why_out) *why_out = "";
    if (!c) return JPT_E_INVALID;
    if (!c->host_scene_ready) return fail(c, JPT_E_STATE, "no scene");
    if (why_out) *why_out = c->ties_note.c_str();
    return c->ties_note.empty() ? 1 : 0;
}

int jpt_scene_begin(jpt_ctx* c)
{
    if (!c) return JPT_E_INVALID;
    c->builder.begin();
    c->pending_materials.clear();
    c->pending_tex.clear();
    c->pending_tex_res = c->pending_layers = 0;
    c->building = true;
    return JPT_OK;
}

int jpt_scene_add_mesh(jpt_ctx* c, const jpt_surface* surfaces, int32_t n_surfaces, uint32_t* mesh_id_out)
{
    if (!c || !c->building) return fail(c, JPT_E_STATE, "jpt_scene_begin not called");
    if (n_surfaces < 0 || (n_surfaces && !surfaces)) return fail(c, JPT_E_INVALID, "bad surface array");
    std::vector<SurfaceView> sv((size_t)n_surfaces);
    for (int32_t i = 0; i < n_surfaces; i++) {
        const jpt_surface& s = surfaces[i];
        if (s.n_indices < 0 || s.n_vertices < 0 || (s.n_indices && (!s.indices || !s.vertices || !s.normals || !s.uvs)))
            return fail(c, JPT_E_INVALID, "surface needs vertex, normal, uv and index arrays (bvh.cpp:195-198)");
        if (s.n_indices % 3) return fail(c, JPT_E_INVALID, "index count is not a multiple of 3");
        for (int32_t k = 0; k < s.n_indices; k++)
            if (s.indices[k] < 0 || s.indices[k] >= s.n_vertices) return fail(c, JPT_E_INVALID, "vertex index out of range");
        sv[i] = SurfaceView{s.vertices, s.normals, s.uvs, s.indices, s.n_vertices, s.n_indices};
    }
    const uint32_t id = c->builder.add_mesh(sv.data(), n_surfaces);
    if (mesh_id_out) *mesh_id_out = id;
    return JPT_OK;
}

int jpt_scene_add_instance(jpt_ctx* c, uint32_t mesh_id, const float* transform12, const int32_t* material_ids, int32_t n_ids)
{
    if (!c || !c->building) return fail(c, JPT_E_STATE, "jpt_scene_begin not called");
    if (!transform12 || n_ids < 0 || (n_ids && !material_ids)) return fail(c, JPT_E_INVALID, "bad instance arguments");
    if (!c->builder.add_instance(mesh_id, transform12, material_ids, n_ids)) return fail(c, JPT_E_INVALID, "unknown mesh id");
    return JPT_OK;
}

int jpt_scene_set_materials(jpt_ctx* c, const void* materials, uint32_t n_materials)
{
    if (!c || !c->building) return fail(c, JPT_E_STATE, "jpt_scene_begin not called");
    if (n_materials && !materials) return fail(c, JPT_E_INVALID, "null materials");
    c->pending_materials.resize(n_materials);
    if (n_materials) std::memcpy(c->pending_materials.data(), materials, (size_t)n_materials * sizeof(RefMaterial));
    return JPT_OK;
}

int jpt_scene_set_textures(jpt_ctx* c, const uint8_t* tex, int32_t tex_res, int32_t n_layers)
{
    if (!c || !c->building) return fail(c, JPT_E_STATE, "jpt_scene_begin not called");
    c->pending_tex.clear();
    c->pending_tex_res = c->pending_layers = 0;
    if (tex && tex_res > 0 && n_layers > 0) {
        c->pending_tex.assign(tex, tex + (size_t)tex_res * tex_res * 4 * n_layers);
        c->pending_tex_res = tex_res;
        c->pending_layers = n_layers;
    }
    return JPT_OK;
}

int jpt_scene_commit(jpt_ctx* c, int32_t builder)
{
    if (!c || !c->building) return fail(c, JPT_E_STATE, "jpt_scene_begin not called");
    if (builder != JPT_BUILD_REFERENCE_EXACT && builder != JPT_BUILD_SAH && builder != JPT_BUILD_SAH_WATERTIGHT)
        return fail(c, JPT_E_INVALID, "unknown builder");
    c->build_mode = builder == JPT_BUILD_SAH ? BuildMode::Sah : (builder == JPT_BUILD_SAH_WATERTIGHT ? BuildMode::SahWatertight : BuildMode::ReferenceExact);
    const auto t0 = std::chrono::steady_clock::now();
    c->scene_ready = c->host_scene_ready = false;
    std::string err;
    if (!c->builder.commit(c->build_mode, c->ref, err))
        return fail(c, JPT_E_LIMIT, err);
    c->ref.materials = c->pending_materials;
    c->ref.textures = c->pending_tex;
    c->ref.tex_res = c->pending_tex_res;
    c->ref.n_layers = c->pending_layers;
    c->ref_is_exact = (builder == JPT_BUILD_REFERENCE_EXACT);
    c->native_tree = is_native(c->build_mode);
    int rc = validate_ref_scene(c, c->ref);
    if (rc != JPT_OK) return rc;
    c->from_commit = true;
    c->upload_note.clear();
    rc = upload_scene(c);
    c->building = false;
    c->stats.last_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

int jpt_scene_share(jpt_ctx* dst, jpt_ctx* src)
{
    if (!dst || !src) return JPT_E_INVALID;
    if (dst == src) return JPT_OK;
    if (!src->host_scene_ready || src->building) return fail(dst, JPT_E_STATE, "the source context holds no committed scene");
    if (src->from_commit && (src->tlas_dirty || src->refit_active)) {  // bring the source's host arrays up to date with its last transforms
        const int rc = jpt_scene_update_tlas(src);
        if (rc != JPT_OK) return fail(dst, rc, std::string("source context: ") + src->error);
    }
    const auto t0 = std::chrono::steady_clock::now();
    dst->scene_ready = dst->host_scene_ready = false;
    dst->builder = src->builder;   // so that jpt_scene_set_instance_transform / update_tlas keep working on the copy
    dst->ref = src->ref;
    dst->ref_is_exact = src->ref_is_exact;
    dst->native_tree = src->native_tree;
    dst->build_mode = src->build_mode;
    dst->from_commit = src->from_commit;
    dst->upload_note = src->upload_note;
    dst->ties_note = src->ties_note;
    dst->building = false;
    const int rc = upload_scene(dst);   // flatten + upload to dst's device; no builder runs
    dst->stats.last_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

namespace {

// instances / TLAS of c->ref changed (BLASes did not): re-flatten that part and upload it
int upload_tlas_update(jpt_ctx* c)
{
    std::string err;
    const bool use4 = !c->ref_is_exact && c->native_tree;
    if (!reflatten_tlas(c->ref, c->wide, use4, err)) return fail(c, JPT_E_INVALID, "tlas update: " + err);
    const uint32_t need = use4 ? c->wide.stack_need4 : c->wide.stack_need2;
    if (need > trace_stack_capacity()) {
        c->scene_ready = c->host_scene_ready = false;
        return fail(c, JPT_E_LIMIT, "acceleration structure too deep after the TLAS update: a traversal could need " +
                                        std::to_string(need) + " stack entries, the kernels hold " + std::to_string(trace_stack_capacity()));
    }
    c->ds.stack_need4 = c->wide.stack_need4;
    if (c->device < 0) return JPT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    // renders already queued on the stream still read the old records: let them finish before the buffers move
    HIP_TRY(c, hipStreamSynchronize(s));
    HIP_TRY(c, c->d_instances.upload(c->ref.instances, s));
    HIP_TRY(c, c->d_tlas.upload(c->ref.tlas_nodes, s));
    HIP_TRY(c, c->d_wtlas.upload(c->wide.tlas_nodes, s));
    HIP_TRY(c, c->d_winst.upload(c->wide.instances, s));
    HIP_TRY(c, c->d_winst4.upload(c->wide.instances4, s));
    HIP_TRY(c, c->d_reach_inst.upload(c->ref.reach_inst, s));
    HIP_TRY(c, c->d_cut_boxes.upload(c->ref.inst_cut_boxes, s));
    HIP_TRY(c, c->d_cut_range.upload(c->ref.inst_cut_range, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    {
        const int rc4 = upload_nodes4(c, true);
        if (rc4 != JPT_OK) return rc4;
    }
    if (c->ds.reach_tri) c->ds.reach_inst = c->d_reach_inst.p;
    c->refit_active = false;
    DeviceScene& d = c->ds;
    d.ref_instances = c->d_instances.p;
    d.ref_tlas = c->d_tlas.p;
    d.n_ref_tlas = (uint32_t)c->ref.tlas_nodes.size();
    d.tlas_nodes = c->d_wtlas.p;
    d.wide_instances = c->d_winst.p;
    d.tlas_root = c->wide.tlas_root;
    d.n_tlas_nodes = (uint32_t)c->wide.tlas_nodes.size();
    d.wide_instances4 = c->d_winst4.p;
    return upload_shadow(c, true);
}

}  // namespace

int jpt_scene_set_instance_transform(jpt_ctx* c, uint32_t instance, const float* transform12)
{
    if (!c) return JPT_E_INVALID;
    if (!c->host_scene_ready || c->building || !c->from_commit)
        return fail(c, JPT_E_STATE, "jpt_scene_set_instance_transform needs a scene made by jpt_scene_commit");
    if (!transform12) return fail(c, JPT_E_INVALID, "null transform");
    if (!c->builder.set_instance_transform(instance, transform12)) return fail(c, JPT_E_INVALID, "no such instance");
    c->tlas_dirty = true;
    return JPT_OK;
}

int jpt_scene_update_tlas(jpt_ctx* c)
{
    if (!c) return JPT_E_INVALID;
    if (!c->host_scene_ready || c->building || !c->from_commit)
        return fail(c, JPT_E_STATE, "jpt_scene_update_tlas needs a scene made by jpt_scene_commit");
    if (!c->tlas_dirty) return JPT_OK;
    const auto t0 = std::chrono::steady_clock::now();
    std::string err;
    if (!c->builder.rebuild_instances(c->build_mode, c->ref, err)) {
        c->scene_ready = c->host_scene_ready = false;
        return fail(c, JPT_E_LIMIT, err);
    }
    const int rc = upload_tlas_update(c);
    if (rc == JPT_OK) c->tlas_dirty = false;
    c->stats.last_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

int jpt_scene_refit_tlas(jpt_ctx* c, const float* transforms12, uint32_t n_instances)
{
    if (!c) return JPT_E_INVALID;
    if (!c->host_scene_ready || c->building || !c->native_tree || !c->from_commit)
        return fail(c, JPT_E_STATE, "jpt_scene_refit_tlas needs a scene made by jpt_scene_commit with the native builder");
    if (!transforms12 && n_instances) return fail(c, JPT_E_INVALID, "null transforms");
    if (n_instances != c->ref.instances.size()) return fail(c, JPT_E_INVALID, "one transform per instance of the committed scene");
    if (c->device < 0) return fail(c, JPT_E_DEVICE, "host-only context: the refit runs on the device (jpt_scene_update_tlas is the host route)");
    // the host keeps the transforms (a later jpt_scene_update_tlas rebuilds from them); its arrays are stale from here on
    for (uint32_t i = 0; i < n_instances; i++) (void)c->builder.set_instance_transform(i, transforms12 + (size_t)i * 12);
    c->tlas_dirty = true;
    if (!c->ds.use4 || !c->scene_ready) return jpt_scene_update_tlas(c);  // no four-child records to refit
    if (n_instances == 0) return JPT_OK;
    const auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const size_t floats = (size_t)n_instances * 12;
    // a ring of pinned staging buffers, so that the host can queue refits ahead of the device
    const int st = (int)(c->refit_seq++ % (uint64_t)jpt_ctx::kRefitStages);
    if (!c->ev_refit_copied[st]) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_refit_copied[st], hipEventDisableTiming));
    else HIP_TRY(c, hipEventSynchronize(c->ev_refit_copied[st]));  // the refit that used this stage last has run
    if (c->h_refit_floats[st] < floats) {
        if (c->h_refit_t12[st]) (void)hipHostFree(c->h_refit_t12[st]);
        c->h_refit_t12[st] = nullptr;
        c->h_refit_floats[st] = 0;
        HIP_TRY(c, hipHostMalloc((void**)&c->h_refit_t12[st], floats * sizeof(float), hipHostMallocDefault));
        c->h_refit_floats[st] = floats;
    }
    std::memcpy(c->h_refit_t12[st], transforms12, floats * sizeof(float));
    // the kernel reads the transforms straight from the pinned buffer (48 B per instance over the host link)
    float* dev_view = nullptr;
    HIP_TRY(c, hipHostGetDevicePointer((void**)&dev_view, c->h_refit_t12[st], 0));
    if (!c->refit_stream) HIP_TRY(c, hipStreamCreateWithFlags(&c->refit_stream, hipStreamNonBlocking));
    for (int k = 0; k < jpt_ctx::kInstanceSets; k++)
        if (!c->ev_set_retired[k]) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_set_retired[k], hipEventDisableTiming));
    if (!c->ev_refit_done) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_refit_done, hipEventDisableTiming));
    if (!c->set_b_ready) {
        // first refit since the last host upload: the other copies of the instance arrays start as copies of copy 0
        // (the TLAS tails were all uploaded).  Once per upload, so the drain does not matter.
        HIP_TRY(c, hipStreamSynchronize(s));
        for (int k = 0; k + 1 < jpt_ctx::kInstanceSets; k++) {
            HIP_TRY(c, c->d_instances_more[k].resize(c->d_instances.n));
            HIP_TRY(c, c->d_winst4_more[k].resize(c->d_winst4.n));
            HIP_TRY(c, hipMemcpyAsync(c->d_instances_more[k].p, c->d_instances.p, c->d_instances.n * sizeof(RefInstance), hipMemcpyDeviceToDevice, s));
            HIP_TRY(c, hipMemcpyAsync(c->d_winst4_more[k].p, c->d_winst4.p, c->d_winst4.n * sizeof(WideInstance), hipMemcpyDeviceToDevice, s));
            HIP_TRY(c, c->d_reach_inst_more[k].resize(c->d_reach_inst.n));
            if (c->d_reach_inst.n)
                HIP_TRY(c, hipMemcpyAsync(c->d_reach_inst_more[k].p, c->d_reach_inst.p, c->d_reach_inst.n * sizeof(ReachInst), hipMemcpyDeviceToDevice, s));
        }
        HIP_TRY(c, hipStreamSynchronize(s));
        c->set_b_ready = true;
        for (int k = 0; k < jpt_ctx::kInstanceSets; k++) c->set_retired_valid[k] = false;
    }
    // The refit writes the copy that new renders do not read yet (`next`); its last readers are the renders queued
    // before the refit that retired it, i.e. before ev_set_retired[next] was recorded on the context's stream (which
    // waits for every render).  The renders queued since read the current copy and keep running meanwhile.
    const int next = (c->cur_set + 1) % jpt_ctx::kInstanceSets;
    HIP_TRY(c, hipEventRecord(c->ev_set_retired[c->cur_set], s));
    c->set_retired_valid[c->cur_set] = true;
    hipStream_t rs = c->refit_stream;
    if (c->set_retired_valid[next]) HIP_TRY(c, hipStreamWaitEvent(rs, c->ev_set_retired[next], 0));
    RefInstance* inst_next = next ? c->d_instances_more[next - 1].p : c->d_instances.p;
    WideInstance* winst4_next = next ? c->d_winst4_more[next - 1].p : c->d_winst4.p;
    const uint32_t tail_base = (uint32_t)(c->wide.blas_nodes4.size() + (size_t)next * c->tlas4_cap);
    ReachInst* reach_next = c->ds.reach_tri ? (next ? c->d_reach_inst_more[next - 1].p : c->d_reach_inst.p) : nullptr;
    launch_tlas4_refit(rs, dev_view, n_instances, c->d_bvh.p, inst_next, nullptr, winst4_next, c->d_nodes4.p, tail_base,
                       c->d_tlas4_order.p, c->d_tlas4_levels.p, c->n_tlas4_levels, reach_next, c->d_nodesq.p, (uint32_t)c->wide.tlas_nodes4.size(),
                       c->ref.inst_cut_range.size() == 2 * (size_t)n_instances && !c->ref.inst_cut_boxes.empty() ? c->d_cut_boxes.p : nullptr,
                       c->ref.inst_cut_range.size() == 2 * (size_t)n_instances && !c->ref.inst_cut_boxes.empty() ? c->d_cut_range.p : nullptr);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(c->ev_refit_copied[st], rs));
    HIP_TRY(c, hipEventRecord(c->ev_refit_done, rs));
    // the context's stream (blocking renders, read-backs, uploads, foreign work) is ordered after the refit; queued
    // renders wait for it on their own streams (do_render_batch), not through the context's stream
    HIP_TRY(c, hipStreamWaitEvent(s, c->ev_refit_done, 0));
    c->refit_wait_seq++;
    c->cur_set = next;
    c->ds.ref_instances = inst_next;
    c->ds.wide_instances4 = winst4_next;
    if (reach_next) c->ds.reach_inst = reach_next;
    if (c->wide.tlas_root4 >= 0) c->ds.tlas_root4 = c->wide.tlas_root4 + (int32_t)tail_base;
    // The sky cull (compute_sky_cull) projects the boxes the TLAS root offers, on the host.  For a modest number of
    // instances the host repeats the refit on its own copy of the four-child TLAS records (same arithmetic, same
    // schedule: ~0.1 us per instance); beyond that the cull is off until the next jpt_scene_update_tlas.
    if (n_instances <= 4096u && !c->tlas4_levels_h.empty()) {
        WideScene& w = c->wide;
        std::vector<RefInstance> boxes(n_instances);
        for (uint32_t i = 0; i < n_instances; i++) {
            const RefBvhNode& root = c->ref.bvh_nodes[c->ref.instances[i].blas_index];
            instance_record(transforms12 + (size_t)i * 12, root.aabbMin, root.aabbMax, true, boxes[i]);
        }
        for (size_t l = 0; l + 1 < c->tlas4_levels_h.size(); l++)
            for (uint32_t k = c->tlas4_levels_h[l]; k < c->tlas4_levels_h[l + 1]; k++) {
                WideNode4& node = w.tlas_nodes4[c->tlas4_order_h[k]];
                for (int slot = 0; slot < 4; slot++) {
                    const int32_t ch = node.child[slot];
                    if (ch == kEmptyChild) continue;
                    float lo[3], hi[3];
                    if (ch < 0) {
                        const RefInstance& in = boxes[(uint32_t)~ch];
                        lo[0] = in.aabbMin.x; lo[1] = in.aabbMin.y; lo[2] = in.aabbMin.z;
                        hi[0] = in.aabbMax.x; hi[1] = in.aabbMax.y; hi[2] = in.aabbMax.z;
                    } else {
                        const WideNode4& below = w.tlas_nodes4[(size_t)ch];
                        bool any = false;
                        for (int j = 0; j < 4; j++) {
                            if (below.child[j] == kEmptyChild) continue;
                            const float bl[3] = {below.lo_x[j], below.lo_y[j], below.lo_z[j]};
                            const float bh[3] = {below.hi_x[j], below.hi_y[j], below.hi_z[j]};
                            for (int a = 0; a < 3; a++) {
                                lo[a] = any ? imin_(lo[a], bl[a]) : bl[a];
                                hi[a] = any ? imax_(hi[a], bh[a]) : bh[a];
                            }
                            any = true;
                        }
                        if (!any) continue;
                    }
                    node.lo_x[slot] = lo[0]; node.lo_y[slot] = lo[1]; node.lo_z[slot] = lo[2];
                    node.hi_x[slot] = hi[0]; node.hi_y[slot] = hi[1]; node.hi_z[slot] = hi[2];
                }
            }
        c->cull_boxes_current = true;
    } else {
        c->cull_boxes_current = false;
    }
    c->refit_active = true;
    c->ds.x.tlas_current = false;   // the copy of the reference's instance level is the last HOST update's: until the next one exact ties are decided inside one instance only (jpt_tie_walk.h)
    c->stats.last_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return JPT_OK;
}

int jpt_scene_update_reference_tlas(jpt_ctx* c, const void* blas_instances, uint32_t n_instances, const void* tlas_nodes,
                                    uint32_t n_tlas_nodes)
{
    if (!c) return JPT_E_INVALID;
    if (!c->host_scene_ready || c->building) return fail(c, JPT_E_STATE, "no scene to update");
    if ((n_instances && !blas_instances) || (n_tlas_nodes && !tlas_nodes)) return fail(c, JPT_E_INVALID, "null buffer with non-zero count");
    if (n_tlas_nodes > 65536) return fail(c, JPT_E_LIMIT, "TLAS has more nodes than 16-bit child indices address (bvh.h:59)");
    if (c->from_commit) return fail(c, JPT_E_STATE, "the scene was made by jpt_scene_commit: use jpt_scene_set_instance_transform / jpt_scene_update_tlas");
    if (n_instances != c->ref.instances.size())
        return fail(c, JPT_E_INVALID, "instance count changed: upload the whole scene again");
    const bool native = c->native_tree;  // the upload built the native tree: instances are matched by the roots the caller named then
    const RefInstance* in = static_cast<const RefInstance*>(blas_instances);
    for (uint32_t i = 0; i < n_instances; i++) {
        RefInstance one;
        std::memcpy(&one, reinterpret_cast<const char*>(in) + (size_t)i * sizeof(RefInstance), sizeof one);
        if (one.blas_index != (native ? c->ref.up_blas_index[i] : c->ref.instances[i].blas_index))
            return fail(c, JPT_E_INVALID, "an instance now names another BLAS: upload the whole scene again");
    }
    const auto t0 = std::chrono::steady_clock::now();
    // keep the old arrays until the new ones are known to flatten
    std::vector<RefInstance> old_inst = c->ref.instances;
    std::vector<RefTlasNode> old_tlas = c->ref.tlas_nodes;
    std::vector<ReachInst> old_reach = c->ref.reach_inst;
    // ... and the shadow's instance level with them (native_instances_from_uploaded rewrites it): after a
    // rejected update exact ties must still be decided on the TLAS of the scene that is rendered
    std::vector<RefInstance> old_x_inst;
    std::vector<RefTlasNode> old_x_tlas;
    std::vector<uint32_t> old_up_index;
    if (native) {
        old_x_inst = c->ref.exact.instances;
        old_x_tlas = c->ref.exact.tlas_nodes;
        old_up_index = c->ref.up_blas_index;
    }
    if (native) {
        std::vector<RefInstance> ni(n_instances);
        std::vector<RefTlasNode> nt(n_tlas_nodes);
        if (n_instances) std::memcpy(ni.data(), blas_instances, (size_t)n_instances * sizeof(RefInstance));
        if (n_tlas_nodes) std::memcpy(nt.data(), tlas_nodes, (size_t)n_tlas_nodes * sizeof(RefTlasNode));
        std::string why;
        if (!native_instances_from_uploaded(ni, nt, c->ref, why))  // (c->ref is untouched when this fails)
            return fail(c, JPT_E_INVALID, "the new instance level does not fit the native tree of the last upload (" + why +
                                              "): upload the whole scene again, or with JPT_UPLOAD_WALK_AS_GIVEN");
    } else {
        if (n_instances) std::memcpy(c->ref.instances.data(), blas_instances, (size_t)n_instances * sizeof(RefInstance));
        c->ref.tlas_nodes.resize(n_tlas_nodes);
        if (n_tlas_nodes) std::memcpy(c->ref.tlas_nodes.data(), tlas_nodes, (size_t)n_tlas_nodes * sizeof(RefTlasNode));
    }
    int rc = upload_tlas_update(c);
    if (rc != JPT_OK && rc != JPT_E_DEVICE) {
        const std::string msg = c->error;
        c->ref.instances = old_inst;
        c->ref.tlas_nodes = old_tlas;
        c->ref.reach_inst = old_reach;
        if (native) {
            c->ref.exact.instances = old_x_inst;
            c->ref.exact.tlas_nodes = old_x_tlas;
            c->ref.up_blas_index = old_up_index;
            if (c->ref.exact.valid) c->ref.exact.finish(c->ref.triangles.size());
        }
        const int back = upload_tlas_update(c);
        c->host_scene_ready = (back == JPT_OK);
        c->scene_ready = c->host_scene_ready && c->device >= 0;
        c->error = msg;
    }
    c->stats.last_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

int jpt_scene_get_reference_buffer(jpt_ctx* c, int32_t which, void* out, size_t capacity, size_t* size_out)
{
    if (!c) return JPT_E_INVALID;
    if (c->refit_active && (which == JPT_BUF_INSTANCES || which == JPT_BUF_TLAS_NODES || which == JPT_BUF_REACH_INSTANCES)) {
        // the instance level was last refitted on the device: bring the host mirrors (and the device) to a fresh build
        const int rc = jpt_scene_update_tlas(c);
        if (rc != JPT_OK) return rc;
    }
    const void* src = nullptr;
    size_t bytes = 0;
    const RefScene& r = c->ref;
    switch (which) {
        case JPT_BUF_TRI_GEOMETRY: src = r.tri_geom.data(); bytes = r.tri_geom.size() * sizeof(RefTriGeometry); break;
        case JPT_BUF_TRI_DATA: src = r.tri_data.data(); bytes = r.tri_data.size() * sizeof(RefTriData); break;
        case JPT_BUF_MATERIALS: src = r.materials.data(); bytes = r.materials.size() * sizeof(RefMaterial); break;
        case JPT_BUF_BVH_NODES: src = r.bvh_nodes.data(); bytes = r.bvh_nodes.size() * sizeof(RefBvhNode); break;
        case JPT_BUF_INSTANCES: src = r.instances.data(); bytes = r.instances.size() * sizeof(RefInstance); break;
        case JPT_BUF_TLAS_NODES: src = r.tlas_nodes.data(); bytes = r.tlas_nodes.size() * sizeof(RefTlasNode); break;
        case JPT_BUF_TRIANGLES: src = r.triangles.data(); bytes = r.triangles.size() * sizeof(RefTriangle); break;
        case JPT_BUF_REACH_TRIANGLES: src = r.reach_tri.data(); bytes = r.reach_tri.size() * sizeof(ReachTri); break;
        case JPT_BUF_REACH_INSTANCES: src = r.reach_inst.data(); bytes = r.reach_inst.size() * sizeof(ReachInst); break;
        default: return fail(c, JPT_E_INVALID, "unknown buffer id");
    }
    if (size_out) *size_out = bytes;
    if (out) {
        if (capacity < bytes) return fail(c, JPT_E_INVALID, "buffer too small");
        if (bytes) std::memcpy(out, src, bytes);
    }
    return JPT_OK;
}

int jpt_set_params(jpt_ctx* c, int32_t width, int32_t height, int32_t max_bounces, int32_t accum_mode, int32_t sampler_mode)
{
    if (!c) return JPT_E_INVALID;
    if (width < 0 || height < 0 || width > 65536 || height > 65536) return fail(c, JPT_E_INVALID, "bad resolution");
    if (max_bounces < 0 || max_bounces > 64) return fail(c, JPT_E_INVALID, "max_bounces must be in [0,64]");
    if (accum_mode != JPT_ACCUM_REF_LDR8 && accum_mode != JPT_ACCUM_HDR_F32) return fail(c, JPT_E_INVALID, "unknown accum_mode");
    if (sampler_mode < JPT_SAMPLER_NEAREST_CLAMP || sampler_mode > JPT_SAMPLER_LINEAR_REPEAT) return fail(c, JPT_E_INVALID, "unknown sampler_mode");
    if (c->device < 0) return fail(c, JPT_E_DEVICE, "host-only context has no framebuffers");
    HIP_TRY(c, hipSetDevice(c->device));
    c->width = width;
    c->height = height;
    c->max_bounces = max_bounces;
    c->accum_mode = accum_mode;
    c->sampler_mode = sampler_mode;
    c->ds.sampler_mode = sampler_mode;
    c->params_set = true;
    c->depth_valid = false;
    return alloc_framebuffers(c);
}

int jpt_set_partition(jpt_ctx* c, int32_t rank, int32_t world)
{
    if (!c) return JPT_E_INVALID;
    if (world < 1 || rank < 0 || rank >= world) return fail(c, JPT_E_INVALID, "need 0 <= rank < world");
    c->rank = rank;
    c->world = world;
    if (c->params_set && c->device >= 0) {
        HIP_TRY(c, hipSetDevice(c->device));
        return alloc_framebuffers(c);
    }
    return JPT_OK;
}

int jpt_set_camera(jpt_ctx* c, const void* camera160)
{
    if (!c || !camera160) return fail(c, JPT_E_INVALID, "null camera");
    std::memcpy(&c->camera, camera160, sizeof(RefCamera));
    c->camera_set = true;
    return JPT_OK;
}

int jpt_render(jpt_ctx* c, int32_t n_frames, uint32_t first_frame_index) { return do_render(c, n_frames, first_frame_index, false, true); }
int jpt_render_counted(jpt_ctx* c, int32_t n_frames, uint32_t first_frame_index) { return do_render(c, n_frames, first_frame_index, true, true); }
int jpt_render_async(jpt_ctx* c, int32_t n_frames, uint32_t first_frame_index) { return do_render(c, n_frames, first_frame_index, false, false); }

int jpt_sync(jpt_ctx* c)
{
    if (!c || c->device < 0) return JPT_E_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, c->ev0, c->ev1) == hipSuccess) c->stats.last_render_ms = c->stats.last_trace_ms = ms;
    if (c->kernel_variant != JPT_KERNEL_REFERENCE_LAYOUT && c->trace_events_used > 0) {
        double tms = 0.0;
        for (int32_t k = 0; k + 1 < c->trace_events_used; k += 2) {
            float t = 0.0f;
            if (hipEventElapsedTime(&t, c->trace_events[(size_t)k], c->trace_events[(size_t)k + 1]) == hipSuccess) tms += t;
        }
        c->stats.last_trace_ms = tms;
    }
    return JPT_OK;
}

int jpt_set_kernel(jpt_ctx* c, int32_t variant)
{
    if (!c) return JPT_E_INVALID;
    if (variant != JPT_KERNEL_WAVEFRONT && variant != JPT_KERNEL_REFERENCE_LAYOUT) return fail(c, JPT_E_INVALID, "unknown kernel variant");
    c->kernel_variant = variant;
    return JPT_OK;
}

int jpt_set_debug_steps(jpt_ctx* c, int32_t enable)
{
    if (!c) return JPT_E_INVALID;
    c->debug_steps = enable != 0;
    return JPT_OK;
}

int jpt_set_kernel_timing(jpt_ctx* c, int32_t enable)
{
    if (!c) return JPT_E_INVALID;
    c->kernel_timing = enable != 0;
    if (!c->kernel_timing) c->trace_events_used = 0;
    return JPT_OK;
}

int jpt_accum_reset(jpt_ctx* c)
{
    if (!c) return JPT_E_INVALID;
    c->frame_count = 0;  // frame_count == 1 on the next frame overwrites the sum (progressive_rendering.glsl:34)
    c->stats.frames = 0;
    c->assembled = c->assembled_ldr = false;
    c->hist_valid = false;  // temporal mode: history images start from zero again (a new TemporalReprojection object)
    return JPT_OK;
}

int jpt_set_progressive_frame_count(jpt_ctx* c, uint32_t next_frame_count)
{
    if (!c) return JPT_E_INVALID;
    if (next_frame_count == 0) return fail(c, JPT_E_INVALID, "frame_count starts at 1 (progressive_rendering.cpp:20)");
    c->frame_count = next_frame_count - 1u;
    c->stats.frames = c->frame_count;
    c->assembled = c->assembled_ldr = false;
    return JPT_OK;
}

int jpt_set_denoising_mode(jpt_ctx* c, int32_t mode)
{
    if (!c) return JPT_E_INVALID;
    if (mode != JPT_DENOISE_PROGRESSIVE && mode != JPT_DENOISE_TEMPORAL && mode != JPT_DENOISE_NONE)
        return fail(c, JPT_E_INVALID, "unknown denoising mode");
    if (mode != c->denoise) {
        // the modes do not share state: the progressive sum and the temporal history both start over
        c->frame_count = 0;
        c->stats.frames = 0;
        c->hist_valid = false;
        c->assembled = c->assembled_ldr = false;
    }
    c->denoise = mode;
    return JPT_OK;
}

int jpt_set_outputs(jpt_ctx* c, uint32_t outputs)
{
    if (!c) return JPT_E_INVALID;
    if (outputs & ~(uint32_t)JPT_OUTPUT_DEPTH) return fail(c, JPT_E_INVALID, "jpt_set_outputs: unknown output bit");
    c->outputs = outputs;   // (takes effect with the next render; a depth image already written stays readable until then)
    return JPT_OK;
}

int jpt_set_temporal_params(jpt_ctx* c, const void* render_parameters)
{
    if (!c) return JPT_E_INVALID;
    if (!render_parameters) return fail(c, JPT_E_INVALID, "null RenderParameters");
    std::memcpy(&c->temporal, render_parameters, sizeof(RefTemporalParams));
    c->temporal_set = true;
    return JPT_OK;
}

static int read_common(jpt_ctx* c, void* out)
{
    if (!c || !out) return fail(c, JPT_E_INVALID, "null output");
    if (!c->params_set) return fail(c, JPT_E_STATE, "jpt_set_params not called");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return JPT_OK;
}

int jpt_read_ldr_rgba8(jpt_ctx* c, uint8_t* out)
{
    int rc = read_common(c, out);
    if (rc) return rc;
    // What the reference does every frame (get_image_uniform_buffer, path_tracing_camera.cpp:228-229): through staged_read.
    const size_t full = (size_t)c->width * c->height;
    const bool whole = c->assembled || c->assembled_ldr;
    const size_t px = whole ? full : (size_t)c->local_rows * c->width;
    rc = staged_read(c, whole ? c->d_full_ldr.p : c->d_ldr.p, px * 4);
    if (rc) return rc;
    if (whole || c->world == 1) std::memcpy(out, c->h_read_pinned, px * 4);
    else {
        std::memset(out, 0, full * 4);
        scatter_rows(static_cast<const uint32_t*>(c->h_read_pinned), reinterpret_cast<uint32_t*>(out), c->width, c->height, c->rank, c->world, 1);
    }
    return JPT_OK;
}

int jpt_readback_ldr_begin(jpt_ctx* c)
{
    if (!c) return JPT_E_INVALID;
    if (c->device < 0) return fail(c, JPT_E_DEVICE, "host-only context");
    if (!c->params_set) return fail(c, JPT_E_STATE, "jpt_set_params not called");
    if (c->readback_pending) return fail(c, JPT_E_STATE, "a read-back is already in flight (call jpt_readback_ldr_end)");
    HIP_TRY(c, hipSetDevice(c->device));
    c->readback_full = c->assembled || c->assembled_ldr;
    const size_t px = c->readback_full ? (size_t)c->width * c->height : (size_t)c->local_rows * c->width;
    {
        const int rc = ensure_ldr_pinned(c);
        if (rc) return rc;
    }
    if (!c->ev_readback) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_readback, hipEventDisableTiming));
    if (px)
        HIP_TRY(c, hipMemcpyAsync(c->h_ldr_pinned, c->readback_full ? c->d_full_ldr.p : c->d_ldr.p, px * sizeof(uint32_t),
                                  hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipEventRecord(c->ev_readback, c->stream));
    c->readback_pending = true;
    return JPT_OK;
}

int jpt_readback_ldr_end(jpt_ctx* c, uint8_t* out)
{
    if (!c || !out) return fail(c, JPT_E_INVALID, "null output");
    if (!c->readback_pending) return fail(c, JPT_E_STATE, "no read-back in flight (call jpt_readback_ldr_begin)");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipEventSynchronize(c->ev_readback));
    c->readback_pending = false;
    const size_t full = (size_t)c->width * c->height;
    if (c->readback_full || c->world == 1) {
        std::memcpy(out, c->h_ldr_pinned, full * 4);
    } else {
        std::memset(out, 0, full * 4);
        scatter_rows(c->h_ldr_pinned, reinterpret_cast<uint32_t*>(out), c->width, c->height, c->rank, c->world, 1);
    }
    return JPT_OK;
}

int jpt_read_accum_f32(jpt_ctx* c, float* out)
{
    int rc = read_common(c, out);
    if (rc) return rc;
    const size_t full = (size_t)c->width * c->height;
    if (c->denoise == JPT_DENOISE_TEMPORAL) {
        // the rgba32f image of this mode is the history image the last pass wrote
        if (!c->hist_written) return fail(c, JPT_E_STATE, "no temporal pass has run since the last reset");
        rc = staged_read(c, c->hist_written, full * sizeof(float4));
        if (rc) return rc;
        std::memcpy(out, c->h_read_pinned, full * sizeof(float4));
        return JPT_OK;
    }
    const size_t n = c->assembled ? full * 4 : (size_t)c->local_rows * c->width * 4;   // floats
    rc = staged_read(c, c->assembled ? (const void*)c->d_full_accum.p : (const void*)c->d_accum.p, n * sizeof(float));
    if (rc) return rc;
    if (c->assembled || c->world == 1) std::memcpy(out, c->h_read_pinned, n * sizeof(float));
    else {
        std::memset(out, 0, full * 16);
        scatter_rows(static_cast<const float*>(c->h_read_pinned), out, c->width, c->height, c->rank, c->world, 4);
    }
    return JPT_OK;
}

int jpt_read_depth_f32(jpt_ctx* c, float* out)
{
    int rc = read_common(c, out);
    if (rc) return rc;
    if (!c->depth_valid)
        return fail(c, JPT_E_STATE, (c->outputs & JPT_OUTPUT_DEPTH) ? "no render has written the depth image yet"
                                                                     : "the depth image is switched off (jpt_set_outputs): no render writes it");
    const size_t full = (size_t)c->width * c->height;
    const size_t n = (size_t)c->local_rows * c->width;
    rc = staged_read(c, c->d_depth.p, n * sizeof(float));
    if (rc) return rc;
    if (c->world == 1) std::memcpy(out, c->h_read_pinned, n * sizeof(float));
    else {
        std::memset(out, 0, full * 4);
        scatter_rows(static_cast<const float*>(c->h_read_pinned), out, c->width, c->height, c->rank, c->world, 1);
    }
    return JPT_OK;
}

void* jpt_device_accum(jpt_ctx* c, size_t* bytes_out)
{
    if (!c) return nullptr;
    if (bytes_out) *bytes_out = c->d_accum.n * sizeof(float4);
    return c->d_accum.p;
}

void* jpt_device_ldr(jpt_ctx* c, size_t* bytes_out)
{
    if (!c) return nullptr;
    if (bytes_out) *bytes_out = c->d_ldr.n * sizeof(uint32_t);
    return c->d_ldr.p;
}

int32_t jpt_local_rows(jpt_ctx* c) { return c ? c->local_rows : 0; }

int jpt_assemble_from_ranks(jpt_ctx* c, const void* device_gathered, int32_t world)
{
    if (!c || !device_gathered) return fail(c, JPT_E_INVALID, "null gathered buffer");
    if (c->device < 0) return fail(c, JPT_E_DEVICE, "host-only context");
    if (!c->params_set) return fail(c, JPT_E_STATE, "jpt_set_params not called");
    if (world != c->world) return fail(c, JPT_E_INVALID, "world differs from jpt_set_partition");
    if (c->denoise != JPT_DENOISE_PROGRESSIVE)
        return fail(c, JPT_E_STATE, "assembling re-derives the progressive display image; the other denoising modes run on one context");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t full = (size_t)c->width * c->height;
    HIP_TRY(c, c->d_full_accum.resize(full));
    HIP_TRY(c, c->d_full_ldr.resize(full));
    launch_assemble(c->stream, (const float4*)device_gathered, c->d_accum.p, c->rank, world, c->width, c->height,
                    max_rows_of_any_rank(c->height, world), c->d_full_accum.p, c->d_full_ldr.p, c->frame_count);
    HIP_TRY(c, hipGetLastError());
    c->assembled = true;  // asynchronous on the context's stream; jpt_read_* / jpt_sync wait for it
    return JPT_OK;
}

int jpt_assemble_ldr_from_ranks(jpt_ctx* c, const void* device_gathered, int32_t world)
{
    if (!c || !device_gathered) return fail(c, JPT_E_INVALID, "null gathered buffer");
    if (c->device < 0) return fail(c, JPT_E_DEVICE, "host-only context");
    if (!c->params_set) return fail(c, JPT_E_STATE, "jpt_set_params not called");
    if (world != c->world) return fail(c, JPT_E_INVALID, "world differs from jpt_set_partition");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, c->d_full_ldr.resize((size_t)c->width * c->height));
    launch_assemble_ldr(c->stream, (const uint32_t*)device_gathered, c->d_ldr.p, c->rank, world, c->width, c->height,
                        max_rows_of_any_rank(c->height, world), c->d_full_ldr.p);
    HIP_TRY(c, hipGetLastError());
    c->assembled_ldr = true;  // asynchronous on the context's stream; jpt_read_ldr_rgba8 / jpt_sync wait for it
    return JPT_OK;
}

int jpt_get_stats(jpt_ctx* c, jpt_stats* out)
{
    if (!c || !out) return JPT_E_INVALID;
    *out = c->stats;
    return JPT_OK;
}

}  // extern "C"
