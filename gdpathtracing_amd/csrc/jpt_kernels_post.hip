// jpt_kernels_post.hip -- post-processing that is not fused into the render: temporal reprojection
// (temporal_reprojection.glsl) and the multi-GPU epilogue: the rank-major pieces an RCCL gather delivers are
// scattered back to image rows, and the display image (ACES of the mean, progressive_rendering.glsl:39-45)
// is re-derived from the assembled sums.  No reference counterpart (the reference is single-device).
#include "jpt_kernels.h"
#include "jpt_instance_math.h"

namespace jpt {

// (`own`: the gathering rank's own rows, read where its render left them -- its piece crosses no link and is not copied)
__global__ __launch_bounds__(256) void assemble_kernel(const float4* __restrict__ gathered, const float4* __restrict__ own, int own_rank,
                                                       int world, int width, int height,
                                                       int max_local_rows, float4* __restrict__ accum_full,
                                                       uint32_t* __restrict__ ldr_full, uint32_t frame_count)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= width || y >= height) return;
    const int strip = y / kStripRows;
    const int rank = strip % world;
    const int ly = (strip / world) * kStripRows + (y - strip * kStripRows);
    const float4 v = rank == own_rank ? own[(size_t)ly * width + x] : gathered[((size_t)rank * max_local_rows + ly) * width + x];
    const size_t idx = (size_t)y * width + x;
    accum_full[idx] = v;
    const float fc = (float)(frame_count ? frame_count : 1u);
    const f3 col = aces_film(mk3(v.x / fc, v.y / fc, v.z / fc) * 1.0f);
    ldr_full[idx] = unorm8(col.x) | (unorm8(col.y) << 8) | (unorm8(col.z) << 16) | 0xFF000000u;
}

// the same scatter for the display image alone: 4 bytes per pixel cross the links instead of 16
__global__ __launch_bounds__(256) void assemble_ldr_kernel(const uint32_t* __restrict__ gathered, const uint32_t* __restrict__ own,
                                                           int own_rank, int world, int width, int height,
                                                           int max_local_rows, uint32_t* __restrict__ ldr_full)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= width || y >= height) return;
    const int strip = y / kStripRows;
    const int rank = strip % world;
    const int ly = (strip / world) * kStripRows + (y - strip * kStripRows);
    ldr_full[(size_t)y * width + x] = rank == own_rank ? own[(size_t)ly * width + x] : gathered[((size_t)rank * max_local_rows + ly) * width + x];
}

// temporal_reprojection.glsl:30-72, one thread per pixel.  Reads its own screen pixel, the depth image at its own
// and at the reprojected position, and the previous history image at the reprojected position; writes its own
// pixel of the other history image and of the screen -- no pixel is written by one thread and read by another.
__global__ __launch_bounds__(256) void temporal_kernel(RefTemporalParams tp, uint32_t* __restrict__ screen,
                                                       const float* __restrict__ depth, const float4* __restrict__ prev,
                                                       float4* __restrict__ next)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    const int W = tp.width, H = tp.height;
    if (x >= W || y >= H) return;
    const size_t i = (size_t)y * W + x;
    const uint32_t q = screen[i];
    const f3 cur = mk3(from_unorm8(q & 255u), from_unorm8((q >> 8) & 255u), from_unorm8((q >> 16) & 255u));  // :35
    f3 rep = cur;                                                                                              // :48
    const float d = depth[i];                                                                                  // :36
    const float fw = (float)(uint32_t)W, fh = (float)(uint32_t)H;
    const float nx = ((float)x + 0.5f) / fw * 2.0f - 1.0f;   // :38-43
    const float ny = ((float)y + 0.5f) / fh * -2.0f + 1.0f;
    if (tp.frame_count > 0u) {                                // :49
        const float* m = tp.deltaMatrix;                      // column-major mat4 * vec4(ndc, 1), summed left to right
        float cx = m[0] * nx + m[4] * ny + m[8] * d + m[12] * 1.0f;
        float cy = m[1] * nx + m[5] * ny + m[9] * d + m[13] * 1.0f;
        float cz = m[2] * nx + m[6] * ny + m[10] * d + m[14] * 1.0f;
        const float cw = m[3] * nx + m[7] * ny + m[11] * d + m[15] * 1.0f;
        cx = cx / cw; cy = cy / cw; cz = cz / cw;             // :51
        const float u = (cx + 1.0f) * 0.5f;                   // :53-56
        const float v = (1.0f - cy) * 0.5f;
        const int32_t px = f2i_sat(u * fw), py = f2i_sat(v * fh);  // :57
        if (px >= 0 && px < W && py >= 0 && py < H) {         // :59
            const size_t j = (size_t)py * W + px;
            if (fabsf(depth[j] - cz) < 0.1f) {
                const float4 h = prev[j];                     // :60
                rep = mk3(h.x, h.y, h.z);
            }
        }
    }
    // mix(cur, rep, 0.75) = cur * (1 - 0.75) + rep * 0.75 (:64, the literal -- blendFactor is not read)
    const f3 blended = mk3(cur.x * (1.0f - 0.75f) + rep.x * 0.75f, cur.y * (1.0f - 0.75f) + rep.y * 0.75f,
                           cur.z * (1.0f - 0.75f) + rep.z * 0.75f);
    next[i] = make_float4(blended.x, blended.y, blended.z, 1.0f);  // :66
    const f3 col = aces_film(blended);                             // :68-70
    screen[i] = unorm8(col.x) | (unorm8(col.y) << 8) | (unorm8(col.z) << 16) | 0xFF000000u;
}

// ---- how many of the pipeline slots' streams run side by side (jpt_capi.cpp, six_queues_probe) ---------------------------------

// one wave that keeps its hardware queue busy for `ticks` of the constant-rate wall clock
__global__ __launch_bounds__(64) void queue_spin_kernel(long long ticks)
{
    const long long t0 = (long long)wall_clock64();
    while ((long long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
void launch_queue_spin(hipStream_t stream, long long ticks) { hipLaunchKernelGGL(queue_spin_kernel, dim3(1), dim3(64), 0, stream, ticks); }

// ---- device-side refit of the instance level (jpt_scene_refit_tlas) -------------------------------------------------

// one thread per instance: the BLASInstance record (jpt_instance_math.h, the host builder's own code) and the hot
// traversal records' inverse matrices
__global__ __launch_bounds__(256) void instance_refit_kernel(const float* __restrict__ t12, uint32_t n, const RefBvhNode* __restrict__ bvh,
                                                             RefInstance* __restrict__ ref_inst, WideInstance* __restrict__ winst,
                                                             WideInstance* __restrict__ winst4, ReachInst* __restrict__ reach,
                                                             const float* __restrict__ cut_boxes, const uint32_t* __restrict__ cut_range)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    RefInstance inst = ref_inst[i];  // blas_index and materials stay
    float t[12];
    for (int k = 0; k < 12; k++) t[k] = t12[(size_t)i * 12 + k];
    const RefBvhNode root = bvh[inst.blas_index];
    if (reach) {
        // the world box the reference computes from ITS root box (reach records, jpt_types.h): same code as the host's
        ReachInst r = reach[i];
        RefInstance tmp = inst;
        instance_record(t, Vec4{r.root_lo[0], r.root_lo[1], r.root_lo[2], 1.0f}, Vec4{r.root_hi[0], r.root_hi[1], r.root_hi[2], 1.0f}, false, tmp);
        r.lo[0] = tmp.aabbMin.x; r.lo[1] = tmp.aabbMin.y; r.lo[2] = tmp.aabbMin.z;
        r.hi[0] = tmp.aabbMax.x; r.hi[1] = tmp.aabbMax.y; r.hi[2] = tmp.aabbMax.z;
        reach[i] = r;
    }
    instance_record(t, root.aabbMin, root.aabbMax, true, inst);
    // ... bound more tightly by the boxes of the mesh's tree the host chose at the last commit (jpt_builder.cpp, InstanceCuts::tighten:
    // the same arithmetic -- centre' -+ |M| half extent, the union padded, intersected with the root rule's box); affine transforms
    // only, which is what a Transform3D is
    const uint32_t n_cut = cut_range ? cut_range[2 * i + 1] : 0u;
    if (n_cut >= 2u) {
        const float* m = inst.transform;   // column-major 4 x 4
        float am[9];
        for (int r = 0; r < 3; r++)
            for (int k = 0; k < 3; k++) am[k * 3 + r] = __builtin_fabsf(m[k * 4 + r]);
        float lo[3] = {1e34f, 1e34f, 1e34f}, hi[3] = {-1e34f, -1e34f, -1e34f};
        const float* b = cut_boxes + (size_t)cut_range[2 * i] * 6;
        for (uint32_t k = 0; k < n_cut; k++, b += 6)
            for (int r = 0; r < 3; r++) {
                const float wc = m[r] * b[0] + m[4 + r] * b[1] + m[8 + r] * b[2] + m[12 + r];
                const float we = am[r] * b[3] + am[3 + r] * b[4] + am[6 + r] * b[5];
                lo[r] = fminf(lo[r], wc - we);
                hi[r] = fmaxf(hi[r], wc + we);
            }
        float big = 0.0f;
        for (int r = 0; r < 3; r++) big = fmaxf(big, fmaxf(__builtin_fabsf(lo[r]), __builtin_fabsf(hi[r])));
        const float pad = big * 4e-6f;
        if (lo[0] <= hi[0] && lo[1] <= hi[1] && lo[2] <= hi[2]) {
            inst.aabbMin = Vec4{fmaxf(inst.aabbMin.x, lo[0] - pad), fmaxf(inst.aabbMin.y, lo[1] - pad), fmaxf(inst.aabbMin.z, lo[2] - pad), inst.aabbMin.w};
            inst.aabbMax = Vec4{fminf(inst.aabbMax.x, hi[0] + pad), fminf(inst.aabbMax.y, hi[1] + pad), fminf(inst.aabbMax.z, hi[2] + pad), inst.aabbMax.w};
        }
    }
    ref_inst[i] = inst;
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 3; r++) {
            const float v = inst.inverse_transform[c * 4 + r];
            if (winst) winst[i].inv[c * 3 + r] = v;
            if (winst4) winst4[i].inv[c * 3 + r] = v;
        }
}

// One block walks the levels of the TLAS deepest first: the box of a child slot is the instance's world box, or the
// union of the boxes of the record below (min / max are exact, so these are the boxes a host build of the same
// topology stores).  A level only reads records of deeper levels; __syncthreads orders the levels.
__global__ __launch_bounds__(1024) void tlas4_refit_kernel(WideNode4* __restrict__ nodes4, uint32_t nb, const uint32_t* __restrict__ order,
                                                           const uint32_t* __restrict__ level_start, uint32_t n_levels,
                                                           const RefInstance* __restrict__ inst)
{
    for (uint32_t l = 0; l < n_levels; l++) {
        for (uint32_t i = level_start[l] + threadIdx.x; i < level_start[l + 1]; i += blockDim.x) {
            WideNode4* node = nodes4 + nb + order[i];
            for (int k = 0; k < 4; k++) {
                const int32_t c = node->child[k];
                if (c == kEmptyChild) continue;
                float lo[3], hi[3];
                if (c < 0) {
                    const RefInstance& in = inst[(uint32_t)~c];
                    lo[0] = in.aabbMin.x; lo[1] = in.aabbMin.y; lo[2] = in.aabbMin.z;
                    hi[0] = in.aabbMax.x; hi[1] = in.aabbMax.y; hi[2] = in.aabbMax.z;
                } else {
                    const WideNode4* below = nodes4 + c;  // (internal references already count from the array's start)
                    bool any = false;
                    for (int j = 0; j < 4; j++) {
                        if (below->child[j] == kEmptyChild) continue;
                        const float bl[3] = {below->lo_x[j], below->lo_y[j], below->lo_z[j]};
                        const float bh[3] = {below->hi_x[j], below->hi_y[j], below->hi_z[j]};
                        for (int a = 0; a < 3; a++) {
                            lo[a] = any ? imin_(lo[a], bl[a]) : bl[a];
                            hi[a] = any ? imax_(hi[a], bh[a]) : bh[a];
                        }
                        any = true;
                    }
                    if (!any) continue;
                }
                node->lo_x[k] = lo[0]; node->lo_y[k] = lo[1]; node->lo_z[k] = lo[2];
                node->hi_x[k] = hi[0]; node->hi_y[k] = hi[1]; node->hi_z[k] = hi[2];
            }
        }
        __syncthreads();
    }
}

// the refitted float records of one TLAS tail -> their quantised form (the same function the host uses at upload)
__global__ __launch_bounds__(256) void quantize_tail_kernel(const WideNode4* __restrict__ nodes4, WideNodeQ* __restrict__ nodesq, uint32_t first,
                                                            uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    WideNodeQ q;
    quantize_node4(nodes4[first + i], q);
    nodesq[first + i] = q;
}

void launch_tlas4_refit(hipStream_t stream, const float* transforms12, uint32_t n_instances, const RefBvhNode* bvh,
                        RefInstance* ref_instances, WideInstance* wide_instances, WideInstance* wide_instances4, WideNode4* nodes4,
                        uint32_t n_blas_records, const uint32_t* order, const uint32_t* level_start, uint32_t n_levels, ReachInst* reach,
                        WideNodeQ* nodesq, uint32_t n_tlas_records, const float* cut_boxes, const uint32_t* cut_range)
{
    if (n_instances == 0) return;
    hipLaunchKernelGGL(instance_refit_kernel, dim3((n_instances + 255) / 256), dim3(256), 0, stream, transforms12, n_instances, bvh,
                       ref_instances, wide_instances, wide_instances4, reach, cut_boxes, cut_range);
    if (n_levels) {
        hipLaunchKernelGGL(tlas4_refit_kernel, dim3(1), dim3(1024), 0, stream, nodes4, n_blas_records, order, level_start, n_levels,
                           ref_instances);
        if (nodesq && n_tlas_records)
            hipLaunchKernelGGL(quantize_tail_kernel, dim3((n_tlas_records + 255) / 256), dim3(256), 0, stream, nodes4, nodesq, n_blas_records,
                               n_tlas_records);
    }
}

void launch_temporal(hipStream_t stream, const RefTemporalParams& tp, uint32_t* screen, const float* depth, float4* hist1,
                     float4* hist2)
{
    if (tp.width <= 0 || tp.height <= 0) return;
    const bool use_first = (tp.frame_count % 2u) == 0u;  // :46
    dim3 grid((tp.width + 255) / 256, tp.height), block(256);
    hipLaunchKernelGGL(temporal_kernel, grid, block, 0, stream, tp, screen, depth, use_first ? hist1 : hist2,
                       use_first ? hist2 : hist1);
}

void launch_assemble_ldr(hipStream_t stream, const uint32_t* gathered, const uint32_t* own, int own_rank, int world, int width,
                         int height, int max_local_rows, uint32_t* ldr_full)
{
    dim3 grid((width + 255) / 256, height), block(256);
    hipLaunchKernelGGL(assemble_ldr_kernel, grid, block, 0, stream, gathered, own, own_rank, world, width, height, max_local_rows,
                       ldr_full);
}

void launch_assemble(hipStream_t stream, const float4* gathered, const float4* own, int own_rank, int world, int width, int height,
                     int max_local_rows, float4* accum_full, uint32_t* ldr_full, uint32_t frame_count)
{
    dim3 grid((width + 255) / 256, height), block(256);
    hipLaunchKernelGGL(assemble_kernel, grid, block, 0, stream, gathered, own, own_rank, world, width, height, max_local_rows,
                       accum_full, ldr_full, frame_count);
}

}  // namespace jpt
