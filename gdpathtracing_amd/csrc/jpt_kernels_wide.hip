// jpt_kernels_wide.hip -- first wavefront pipeline (kept for A/B: JPT_KERNEL_WAVEFRONT_V1); the default
// native route is jpt_kernels_wf2.hip.
//
// All `n_frames` frames of a render are in flight at once as independent paths (pixel x frame); the
// per-pixel RNG streams of the reference (main.glsl:409, :386) are kept, so results do not depend on
// the schedule.  Per bounce, two kernels exchange compact queues in HBM:
//
//   wf_trace   persistent waves pull 64 rays at a time from the bounce's queue and walk the two-level
//              BVH: one 64-byte record per expansion (both children's boxes), per-lane stacks staged in
//              LDS, Moller-Trumbore on precomputed-edge triangles.  Writes one 32-byte hit record per ray.
//   wf_shade   shading fetch, BRDF sample / pdf / eval (brdfs.glsl), emission accumulation, and the next
//              segment's ray, packed to the front of the next queue with a wave ballot + mbcnt prefix and
//              one atomic per wave (active-ray compaction).
//
// wf_generate builds bounce 0's queue; wf_accumulate replays the frames IN ORDER per pixel so the float sum
// is the same sequence of additions as the reference's frame-by-frame accumulation
// (progressive_rendering.glsl:33-37).  Replaces main.glsl:270-436 + progressive_rendering.glsl.
#include "jpt_trace_core.h"

namespace jpt {

namespace {

constexpr int kBlock = kTraceBlock;
constexpr uint32_t kInactive = 0xffffffffu;  // path id of a padding queue entry
constexpr int kShadeBlock = 1024;

template <bool COUNT>
__device__ __forceinline__ void trace_ray(const WideSceneDev& sc, const f3 wo, const f3 wd, const f3 wrD, TraceHit& hit,
                                          int32_t* __restrict__ lds_stack, DevCounters& cnt)
{
    (void)wrD;
    int32_t spill[kStackSpill];
    const typename Traversal<COUNT>::Stack st{lds_stack, spill, kTraceBlock, kStackLds};
    Traversal<COUNT> tr;
    tr.begin(sc, wo, wd);
    while (tr.step(sc, st, cnt)) {
    }
    hit = tr.hit;
}

// ---- path / queue storage -----------------------------------------------------------------------

struct WfBuffers {
    float4* ray_o[2];     // queue entry: origin.xyz, w unused
    float4* ray_d[2];     // direction.xyz, w = path id bits
    float4* hit_a;        // t, u, v, tri bits
    uint32_t* hit_b;      // inst | front << 31
    float4* thr;          // per path: throughput.xyz, w = seed.x bits
    float4* rad;          // per path: radiance.xyz,   w = seed.y bits   ([frame][slot]: also the per-frame output)
    float* first_depth;   // per slot of the LAST frame: distance of the first hit (or far)
    uint32_t* qcount;     // [max_bounces + 2] queue sizes
    uint32_t* cursor;     // [max_bounces + 2] fetch cursors of wf_trace
};

struct WfDims {
    int32_t tiles_x, tiles_y;
    uint32_t slots_per_frame;  // tiles_x * tiles_y * 64
    uint32_t n_paths;          // slots_per_frame * n_frames
};

__device__ __forceinline__ void slot_to_pixel(uint32_t slot, const WfDims& dm, int& px, int& ly)
{
    const uint32_t tile = slot >> 6, lane = slot & 63u;
    const uint32_t ty = tile / (uint32_t)dm.tiles_x, tx = tile - ty * (uint32_t)dm.tiles_x;
    px = (int)(tx * 8u + (lane & 7u));
    ly = (int)(ty * 8u + (lane >> 3));
}

// bounce 0 queue: one primary ray per (pixel, frame) (main.glsl:405-421)
__global__ __launch_bounds__(kBlock) void wf_generate(WfBuffers wb, WfDims dm, FrameParams fp, RefCamera cam)
{
    const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    bool active = false;
    float4 ro, rd, thr, rad;
    if (p < dm.n_paths) {
        const uint32_t f = p / dm.slots_per_frame, slot = p - f * dm.slots_per_frame;
        int px, ly;
        slot_to_pixel(slot, dm, px, ly);
        if (px < fp.width && ly < fp.local_rows) {
            const int py = local_to_global_row(ly, fp);
            uint32_t sx, sy;
            const Ray ray = primary_ray(cam, fp.width, fp.height, px, py, fp.frame_index + f, sx, sy);
            ro = make_float4(ray.o.x, ray.o.y, ray.o.z, 0.0f);
            rd = make_float4(ray.d.x, ray.d.y, ray.d.z, __uint_as_float(p));
            thr = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(sx));
            rad = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(sy));
            active = true;
        }
    }
    // bounce 0's queue is indexed by path id; out-of-image padding lanes are marked inactive (d.w = ~0)
    if (p < dm.n_paths) {
        if (active) {
            wb.ray_o[0][p] = ro;
            wb.ray_d[0][p] = rd;
            wb.thr[p] = thr;
            wb.rad[p] = rad;
        } else {
            wb.ray_d[0][p] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(kInactive));
        }
    }
    if (p == 0) wb.qcount[0] = dm.n_paths;
}

template <bool COUNT>
__global__ __launch_bounds__(kBlock) void wf_trace(WideSceneDev sc, WfBuffers wb, int bounce, DevCounters* __restrict__ counters)
{
    __shared__ int32_t stack[kStackLds * kBlock];
    const int lane = threadIdx.x & 63;
    const uint32_t n = wb.qcount[bounce];
    const float4* __restrict__ qo = wb.ray_o[bounce & 1];
    const float4* __restrict__ qd = wb.ray_d[bounce & 1];
    DevCounters cnt = {};
    // waves stride over 64-ray chunks of the queue (no shared cursor: a single-address atomic caps at
    // ~88 M/s on this chip, which is slower than the tracing itself)
    const uint32_t wave_id = (blockIdx.x * kBlock + threadIdx.x) >> 6, n_waves = (gridDim.x * kBlock) >> 6;
    for (uint32_t base = wave_id * 64u; base < n; base += n_waves * 64u) {
        const uint32_t i = base + (uint32_t)lane;
        if (i < n) {
            const float4 rd = qd[i];
            if (__float_as_uint(rd.w) == kInactive) continue;
            const float4 ro = qo[i];
            const f3 o = mk3(ro.x, ro.y, ro.z), d = mk3(rd.x, rd.y, rd.z);
            TraceHit hit;
            trace_ray<COUNT>(sc, o, d, rcp3(d), hit, &stack[threadIdx.x], cnt);
            wb.hit_a[i] = make_float4(hit.t, hit.u, hit.v, __uint_as_float(hit.tri));
            wb.hit_b[i] = hit.inst | (hit.front ? 0x80000000u : 0u);
        }
    }
    if (COUNT) flush_counters(cnt, counters);
}

// one path vertex per queue entry (main.glsl:378-397); survivors are packed into the next queue
template <bool COUNT>
__global__ __launch_bounds__(kShadeBlock) void wf_shade(SceneShading sh, WfBuffers wb, WfDims dm, FrameParams fp, RefCamera cam,
                                                   int bounce, DevCounters* __restrict__ counters)
{
    __shared__ uint32_t wave_alive[kShadeBlock / 64];
    __shared__ uint32_t block_base;
    const uint32_t n = wb.qcount[bounce];
    const uint32_t i = blockIdx.x * kShadeBlock + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (blockIdx.x * kShadeBlock >= n) return;  // whole block past the end
    bool alive = false;
    float4 no, nd;
    DevCounters cnt = {};
    if (i < n) {
        const int in = bounce & 1;
        const float4 rd = wb.ray_d[in][i];
        const uint32_t p = __float_as_uint(rd.w);
        if (p != kInactive) {
        const float4 ro = wb.ray_o[in][i];
        const float4 ha = wb.hit_a[i];
        float4 t4 = wb.thr[p], r4 = wb.rad[p];
        f3 throughput = mk3(t4.x, t4.y, t4.z), radiance = mk3(r4.x, r4.y, r4.z);
        uint32_t sx = __float_as_uint(t4.w), sy = __float_as_uint(r4.w);
        Ray ray;
        ray.o = mk3(ro.x, ro.y, ro.z);
        ray.d = mk3(rd.x, rd.y, rd.z);
        const bool is_hit = ha.x < 1e9f;  // main.glsl:349
        float first_dist = cam.far_;
        if (COUNT) cnt.rays++;
        if (!is_hit) {
            radiance = radiance + throughput * sample_sky(ray.d);
        } else {
            if (COUNT) cnt.shaded_hits++;
            const uint32_t hb = wb.hit_b[i];
            Hit h;
            h.t = ha.x;
            h.u = ha.y;
            h.v = ha.z;
            h.tri = __float_as_uint(ha.w);
            h.inst = hb & kInstMask;                               // hitInfo.blas
            const uint32_t found_in = (hb >> kInstBits) & kInstMask;   // the instance whose local ray found the triangle
            // the hit instance's local ray: same expression ray_trace_tlas evaluates (main.glsl:319-320)
            const RefInstance& b = sh.instances[found_in];
            h.lo = xform_point(b.inverse_transform, ray.o);
            h.ld = xform_dir(b.inverse_transform, ray.d);
            const Shading s = get_shading_data(sh, h, (hb >> 31) != 0u);
            radiance = radiance + throughput * s.emission;
            if (bounce == 0) first_dist = length3(s.position - ray.o);
            if (bounce < fp.max_bounces) alive = bounce_step(s, sx, sy, ray, throughput);
        }
        if (bounce == 0) {
            const uint32_t f = p / dm.slots_per_frame;
            if ((int)f == fp.depth_frame) wb.first_depth[p - f * dm.slots_per_frame] = first_dist;
        }
        wb.rad[p] = make_float4(radiance.x, radiance.y, radiance.z, __uint_as_float(sy));
        if (alive) {
            wb.thr[p] = make_float4(throughput.x, throughput.y, throughput.z, __uint_as_float(sx));
            no = make_float4(ray.o.x, ray.o.y, ray.o.z, 0.0f);
            nd = make_float4(ray.d.x, ray.d.y, ray.d.z, __uint_as_float(p));
        }
        }
    }
    // active-ray packing: wave ballot + prefix count, wave totals combined in LDS, ONE atomic per block
    const unsigned long long m = __ballot(alive);
    if (lane == 0) wave_alive[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (int w = 0; w < kShadeBlock / 64; w++) {
            const uint32_t c = wave_alive[w];
            wave_alive[w] = tot;
            tot += c;
        }
        block_base = tot ? atomicAdd(&wb.qcount[bounce + 1], tot) : 0u;
    }
    __syncthreads();
    if (alive) {
        const uint32_t j = block_base + wave_alive[wave] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        const int out = (bounce + 1) & 1;
        wb.ray_o[out][j] = no;
        wb.ray_d[out][j] = nd;
    }
    if (COUNT) flush_counters(cnt, counters);
}

// per pixel: frames in order -> accumulation buffer, display image, depth
__global__ __launch_bounds__(kBlock) void wf_accumulate(WfBuffers wb, WfDims dm, FrameParams fp, RefCamera cam,
                                                        float4* __restrict__ accum, uint32_t* __restrict__ ldr,
                                                        float* __restrict__ depth_out)
{
    const uint32_t slot = blockIdx.x * kBlock + threadIdx.x;
    if (slot >= dm.slots_per_frame) return;
    int px, ly;
    slot_to_pixel(slot, dm, px, ly);
    if (px >= fp.width || ly >= fp.local_rows) return;
    const size_t idx = (size_t)ly * fp.width + px;
    // fp.frame_count = ProgressiveRendering frame_count of the FIRST frame of this render
    f3 sum = mk3(0.0f, 0.0f, 0.0f);
    bool have_prev = fp.frame_count > 1;
    if (have_prev) {
        const float4 prev = accum[idx];
        sum = mk3(prev.x, prev.y, prev.z);
    }
    f3 last = mk3(0.0f, 0.0f, 0.0f);
    for (int f = 0; f < fp.n_frames; f++) {
        const float4 r = wb.rad[(size_t)f * dm.slots_per_frame + slot];
        f3 cur = mk3(r.x, r.y, r.z);
        last = cur;
        if (fp.accum_mode == 0)  // rgba8 store of main.glsl:434, load of progressive_rendering.glsl:33
            cur = mk3(from_unorm8(unorm8(cur.x)), from_unorm8(unorm8(cur.y)), from_unorm8(unorm8(cur.z)));
        sum = have_prev ? cur + sum : cur;  // progressive_rendering.glsl:34-36
        have_prev = true;
    }
    if (fp.n_frames > 0) {
        accum[idx] = make_float4(sum.x, sum.y, sum.z, 1.0f);
        const float fc = (float)(fp.frame_count + (uint32_t)fp.n_frames - 1u);
        const f3 col = fp.display_mode == 1 ? last : aces_film(mk3(sum.x / fc, sum.y / fc, sum.z / fc) * 1.0f);
        ldr[idx] = unorm8(col.x) | (unorm8(col.y) << 8) | (unorm8(col.z) << 16) | 0xFF000000u;
    }
    if (depth_out && fp.n_frames > 0) {
        const float dist = wb.first_depth[slot];
        depth_out[idx] = cam.far_ / (cam.far_ - cam.near_) * (1.0f - cam.near_ / dist);  // main.glsl:432
    }
}

}  // namespace

size_t wide_workspace_bytes(int width, int local_rows, int n_frames, int max_bounces)
{
    const size_t tiles_x = (width + 7) / 8, tiles_y = (local_rows + 7) / 8;
    const size_t slots = tiles_x * tiles_y * 64, paths = slots * (size_t)n_frames;
    size_t b = 0;
    b += paths * sizeof(float4) * 4;   // two ray queues
    b += paths * sizeof(float4);       // hit_a
    b += paths * sizeof(uint32_t);     // hit_b
    b += paths * sizeof(float4) * 2;   // thr, rad
    b += slots * sizeof(float);        // first_depth
    b += (size_t)(max_bounces + 2) * 2 * sizeof(uint32_t);
    return b + 4096;
}

void launch_wide_render(hipStream_t stream, const DeviceScene& ds, const FrameParams& fp, const RefCamera& cam, void* workspace,
                        float4* accum, uint32_t* ldr, float* depth, DevCounters* counters, hipEvent_t* trace_events)
{
    WfDims dm;
    dm.tiles_x = (fp.width + 7) / 8;
    dm.tiles_y = (fp.local_rows + 7) / 8;
    dm.slots_per_frame = (uint32_t)dm.tiles_x * (uint32_t)dm.tiles_y * 64u;
    dm.n_paths = dm.slots_per_frame * (uint32_t)fp.n_frames;
    if (dm.n_paths == 0) return;
    const size_t P = dm.n_paths;
    char* w = reinterpret_cast<char*>(workspace);
    auto carve = [&](size_t bytes) {
        void* p = w;
        w += (bytes + 255) & ~(size_t)255;
        return p;
    };
    WfBuffers wb;
    const int nq = fp.max_bounces + 2;
    wb.qcount = (uint32_t*)carve((size_t)nq * 2 * sizeof(uint32_t));
    wb.cursor = wb.qcount + nq;
    wb.ray_o[0] = (float4*)carve(P * sizeof(float4));
    wb.ray_o[1] = (float4*)carve(P * sizeof(float4));
    wb.ray_d[0] = (float4*)carve(P * sizeof(float4));
    wb.ray_d[1] = (float4*)carve(P * sizeof(float4));
    wb.hit_a = (float4*)carve(P * sizeof(float4));
    wb.hit_b = (uint32_t*)carve(P * sizeof(uint32_t));
    wb.thr = (float4*)carve(P * sizeof(float4));
    wb.rad = (float4*)carve(P * sizeof(float4));
    wb.first_depth = (float*)carve((size_t)dm.slots_per_frame * sizeof(float));

    WideSceneDev sc;
    sc.blas_nodes = ds.blas_nodes;
    sc.nodes4 = nullptr;
    sc.tris = ds.wide_tris;
    sc.tlas_nodes = ds.tlas_nodes;
    sc.instances = ds.wide_instances;
    sc.tlas_root = ds.tlas_root;
    sc.n_instances = ds.n_instances;
    const SceneShading sh = ds.shading();

    (void)hipMemsetAsync(wb.qcount, 0, (size_t)nq * 2 * sizeof(uint32_t), stream);
    const uint32_t blocks = (uint32_t)((P + kBlock - 1) / kBlock);
    const uint32_t sblocks = (uint32_t)((P + kShadeBlock - 1) / kShadeBlock);
    hipLaunchKernelGGL(wf_generate, dim3(blocks), dim3(kBlock), 0, stream, wb, dm, fp, cam);
    // persistent trace grid: enough waves to fill every CU at the kernel's occupancy
    const uint32_t trace_blocks = 256u * 6u;
    for (int b = 0; b <= fp.max_bounces; b++) {
        if (trace_events) (void)hipEventRecord(trace_events[2 * b], stream);
        if (counters) hipLaunchKernelGGL(wf_trace<true>, dim3(trace_blocks), dim3(kBlock), 0, stream, sc, wb, b, counters);
        else hipLaunchKernelGGL(wf_trace<false>, dim3(trace_blocks), dim3(kBlock), 0, stream, sc, wb, b, counters);
        if (trace_events) (void)hipEventRecord(trace_events[2 * b + 1], stream);
        if (counters) {
            hipLaunchKernelGGL(wf_shade<true>, dim3(sblocks), dim3(kShadeBlock), 0, stream, sh, wb, dm, fp, cam, b, counters);
        } else {
            hipLaunchKernelGGL(wf_shade<false>, dim3(sblocks), dim3(kShadeBlock), 0, stream, sh, wb, dm, fp, cam, b, counters);
        }
    }
    const uint32_t ablocks = (dm.slots_per_frame + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(wf_accumulate, dim3(ablocks), dim3(kBlock), 0, stream, wb, dm, fp, cam, accum, ldr, depth);
}

}  // namespace jpt
