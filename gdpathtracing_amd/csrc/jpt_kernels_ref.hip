// jpt_kernels_ref.hip -- reference-layout kernels: one thread per pixel over the six byte buffers
// exactly as GeometryGroup3D emits them.  This is the drop-in / audit route: it visits the
// reference's own BVH nodes in the reference's order (main.glsl:270-350), so its images are
// bit-comparable with the oracle including exact-t ties.  The fast route is jpt_kernels_wide.hip.
#include "jpt_kernels.h"
#include "jpt_shade.h"
#include "jpt_tie_walk.h"
#include "jpt_tuning.h"

namespace jpt {

struct RefSceneDev {
    const RefTriGeometry* __restrict__ tri_geom;
    const RefBvhNode* __restrict__ bvh;
    const RefInstance* __restrict__ instances;
    const RefTlasNode* __restrict__ tlas;
    uint32_t n_instances, n_tlas;
    // reach records (jpt_types.h), null unless the scene is a native tree committed with JPT_BUILD_SAH: then this kernel
    // walks the native tree in reference layout and applies the reference's two reachability tests itself
    const ReachTri* __restrict__ reach_tri;
    const ReachInst* __restrict__ reach_inst;
};

struct RefHit {
    float t, u, v;
    uint32_t tri, inst;
    uint32_t steps;   // hitInfo.steps: intersectTriangle calls (main.glsl:225)
    bool front;
    bool tied;        // the triangle kept has EXACTLY the distance of one accepted before it (see kHitTied, jpt_trace_core.h)
    f3 lo, ld;  // local ray of the best hit so far (candidate per instance, committed with `inst`)
};

// main.glsl:259-268
__device__ __forceinline__ float intersect_aabb(const Ray& ray, f3 bmin, f3 bmax)
{
    const float tx1 = (bmin.x - ray.o.x) * ray.rD.x, tx2 = (bmax.x - ray.o.x) * ray.rD.x;
    float tmin = fmin_(tx1, tx2), tmax = fmax_(tx1, tx2);
    const float ty1 = (bmin.y - ray.o.y) * ray.rD.y, ty2 = (bmax.y - ray.o.y) * ray.rD.y;
    tmin = fmax_(tmin, fmin_(ty1, ty2)), tmax = fmin_(tmax, fmax_(ty1, ty2));
    const float tz1 = (bmin.z - ray.o.z) * ray.rD.z, tz2 = (bmax.z - ray.o.z) * ray.rD.z;
    tmin = fmax_(tmin, fmin_(tz1, tz2)), tmax = fmin_(tmax, fmax_(tz1, tz2));
    return (tmax >= tmin && tmax > 0.0f) ? tmin : 1e30f;
}

// main.glsl:224-257
// `ties` (native trees whose scene keeps the reference's own trees beside it): with hitInfo.t preset to a tying distance
// the walk notes, per accepted triangle, the REFERENCE leaf that holds it (jpt_tie_walk.h, step 1)
template <bool COUNT>
__device__ __forceinline__ void intersect_triangle(const RefSceneDev& sc, const Ray& ray, uint32_t tri_index, RefHit& hit,
                                                   bool& improved, DevCounters& cnt, const TieShadowDev* ties = nullptr,
                                                   TieLeaves* tl = nullptr, uint32_t inst = 0u)
{
    if (COUNT) cnt.tri_tests++;
    hit.steps++;
    const RefTriGeometry& tri = sc.tri_geom[tri_index];
    const f3 v0 = mk3(tri.vertices[0].x, tri.vertices[0].y, tri.vertices[0].z);
    const f3 v1 = mk3(tri.vertices[1].x, tri.vertices[1].y, tri.vertices[1].z);
    const f3 v2 = mk3(tri.vertices[2].x, tri.vertices[2].y, tri.vertices[2].z);
    const f3 edge1 = v1 - v0;
    const f3 edge2 = v2 - v0;
    const f3 pvec = cross3(ray.d, edge2);
    const float det = dot3(edge1, pvec);
    if (__builtin_fabsf(det) < 1e-5f) return;
    const float invDet = 1.0f / det;
    const f3 tvec = ray.o - v0;
    const float u = dot3(tvec, pvec) * invDet;
    if (u < 0.0f || u > 1.0f) return;
    const f3 qvec = cross3(tvec, edge1);
    const float v = dot3(ray.d, qvec) * invDet;
    if (v < 0.0f || u + v > 1.0f) return;
    const float t = dot3(edge2, qvec) * invDet;
    if (t < 0.0f || t > hit.t) return;
    if (sc.reach_tri) {
        const ReachTri& r = sc.reach_tri[tri_index];
        if (!r.always && !(intersect_aabb(ray, mk3(r.lo[0], r.lo[1], r.lo[2]), mk3(r.hi[0], r.hi[1], r.hi[2])) < 1e30f)) return;
    }
    hit.tied = !(t < hit.t) && hit.t < 1e9f;
    hit.t = t;
    hit.tri = tri_index;
    hit.u = u;
    hit.v = v;
    hit.front = dot3(cross3(edge1, edge2), ray.d) > 0.0f;
    improved = true;
    if (tl) {
        const uint32_t rt = ties->native_ref[tri_index];
        if (rt != 0xffffffffu) tl->add(ties->tri_leaf[rt], inst);
    }
}

// main.glsl:270-303
template <bool COUNT>
__device__ __forceinline__ void ray_trace_blas(const RefSceneDev& sc, uint32_t root, const Ray& ray, RefHit& hit,
                                               bool& improved, DevCounters& cnt, const TieShadowDev* ties = nullptr,
                                               TieLeaves* tl = nullptr, uint32_t inst = 0u)
{
    uint32_t stack[64];
    uint32_t sp = 0;
    stack[sp++] = root;
    while (sp > 0) {
        const RefBvhNode& node = sc.bvh[stack[--sp]];
        const uint32_t tri_count = node.tri_count;
        if (tri_count > 0) {
            const uint32_t first = node.first_tri_index;
            for (uint32_t i = 0; i < tri_count; i++) intersect_triangle<COUNT>(sc, ray, first + i, hit, improved, cnt, ties, tl, inst);
            continue;
        }
        if (COUNT) cnt.blas_expand++;
        const uint32_t li = node.left_child, ri = node.right_child;
        const RefBvhNode& childL = sc.bvh[li];
        const RefBvhNode& childR = sc.bvh[ri];
        const float d1 = intersect_aabb(ray, mk3(childL.aabbMin.x, childL.aabbMin.y, childL.aabbMin.z),
                                        mk3(childL.aabbMax.x, childL.aabbMax.y, childL.aabbMax.z));
        const float d2 = intersect_aabb(ray, mk3(childR.aabbMin.x, childR.aabbMin.y, childR.aabbMin.z),
                                        mk3(childR.aabbMax.x, childR.aabbMax.y, childR.aabbMax.z));
        const bool leftValid = d1 < hit.t;
        const bool rightValid = d2 < hit.t;
        if (d1 < d2) {
            if (rightValid && sp < 64) stack[sp++] = ri;
            if (leftValid && sp < 64) stack[sp++] = li;
        } else {
            if (leftValid && sp < 64) stack[sp++] = li;
            if (rightValid && sp < 64) stack[sp++] = ri;
        }
    }
}

// main.glsl:305-350
template <bool COUNT>
__device__ __forceinline__ bool ray_trace_tlas(const RefSceneDev& sc, const Ray& ray, RefHit& hit, DevCounters& cnt,
                                               const TieShadowDev* ties = nullptr, TieLeaves* tl = nullptr, float preset_t = 1e9f)
{
    hit.t = preset_t;
    hit.tied = false;
    hit.steps = 0;
    if (sc.n_tlas == 0 || sc.n_instances == 0) return false;
    uint32_t stack[64];
    uint32_t sp = 0;
    stack[sp++] = 0;
    float minT = 1e9f;
    while (sp > 0) {
        const RefTlasNode& node = sc.tlas[stack[--sp]];
        const uint32_t lr = node.leftRight;
        if (lr == 0) {
            if (COUNT) cnt.inst_visits++;
            const uint32_t inst = node.blas;
            if (sc.reach_inst && sc.n_instances > 1u) {
                const ReachInst& r = sc.reach_inst[inst];
                if (!(intersect_aabb(ray, mk3(r.lo[0], r.lo[1], r.lo[2]), mk3(r.hi[0], r.hi[1], r.hi[2])) < 1e30f)) continue;
            }
            const RefInstance& b = sc.instances[inst];
            Ray b_ray;
            b_ray.o = xform_point(b.inverse_transform, ray.o);
            b_ray.d = xform_dir(b.inverse_transform, ray.d);
            b_ray.rD = rcp3(b_ray.d);
            bool improved = false;
            ray_trace_blas<COUNT>(sc, b.blas_index, b_ray, hit, improved, cnt, ties, tl, inst);
            // hitInfo.position / out_dir come from the local ray of the last accepted triangle (main.glsl:249,253)
            if (improved) {
                hit.lo = b_ray.o;
                hit.ld = b_ray.d;
            }
            // main.glsl:324-327: the instance id follows strict improvements of t only
            if (hit.t < minT) {
                hit.inst = inst;
                minT = hit.t;
            }
            continue;
        }
        if (COUNT) cnt.tlas_expand++;
        const uint32_t left = lr & 0xFFFFu, right = lr >> 16;
        const RefTlasNode& childL = sc.tlas[left];
        const RefTlasNode& childR = sc.tlas[right];
        const float d1 = intersect_aabb(ray, mk3(childL.aabbMin[0], childL.aabbMin[1], childL.aabbMin[2]),
                                        mk3(childL.aabbMax[0], childL.aabbMax[1], childL.aabbMax[2]));
        const float d2 = intersect_aabb(ray, mk3(childR.aabbMin[0], childR.aabbMin[1], childR.aabbMin[2]),
                                        mk3(childR.aabbMax[0], childR.aabbMax[1], childR.aabbMax[2]));
        const bool leftValid = d1 < hit.t;
        const bool rightValid = d2 < hit.t;
        if (d1 < d2) {
            if (rightValid && sp < 64) stack[sp++] = right;
            if (leftValid && sp < 64) stack[sp++] = left;
        } else {
            if (leftValid && sp < 64) stack[sp++] = left;
            if (rightValid && sp < 64) stack[sp++] = right;
        }
    }
    return hit.t < 1e9f;
}


// One dispatch of main.glsl (main.glsl:404-436) fused with one dispatch of progressive_rendering.glsl
// (:28-46) for the pixels of this context's partition.
template <bool COUNT, bool TIES>
__global__ __launch_bounds__(256, 5) void ref_frame_kernel(RefSceneDev sc, TieShadowDev shadow, SceneShading sh, FrameParams fp, RefCamera cam,
                                                        float4* __restrict__ accum, uint32_t* __restrict__ ldr,
                                                        float* __restrict__ depth_out, DevCounters* __restrict__ counters)
{
    // 8x32 pixel tiles: a wave covers 8x8 pixels
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = blockIdx.x * 32 + wave * 8 + (lane & 7);
    const int ly = blockIdx.y * 8 + (lane >> 3);  // local row
    DevCounters cnt = {};
    if (px < fp.width && ly < fp.local_rows) {
        const int py = local_to_global_row(ly, fp);
        uint32_t sx, sy;
        Ray ray = primary_ray(cam, fp.width, fp.height, px, py, fp.frame_index, sx, sy);
        float depth = cam.far_;
        f3 radiance = mk3(0.0f, 0.0f, 0.0f);
        f3 throughput = mk3(1.0f, 1.0f, 1.0f);
        if (fp.debug_steps) {   // #ifdef DEBUG_STEPS (main.glsl:358-361, 423-427): the primary ray's triangle tests / 256, depth = far
            RefHit hit;
            if (COUNT) cnt.rays++;
            (void)ray_trace_tlas<COUNT>(sc, ray, hit, cnt);
            const float g = clamp_((float)hit.steps / 256.0f, 0.0f, 1.0f);
            radiance = mk3(g, g, g);
        } else
        for (int i = 0; i < fp.max_bounces + 1; i++) {  // main.glsl:377
            RefHit hit;
            if (COUNT) cnt.rays++;
            const bool is_hit = ray_trace_tlas<COUNT>(sc, ray, hit, cnt);
            if (!is_hit) {
                radiance = radiance + throughput * sample_sky(ray.d);
                break;
            }
            if (TIES && hit.tied) {
                // an exact distance tie: decided where the reference decides it (jpt_tie_walk.h) -- the leaves that hold
                // the tying triangles from one more walk with hitInfo.t preset, then the reference's own walk through
                // their ancestors (event counters: not those of a reference tree anyway)
                TieLeaves tl;
                RefHit again;
                DevCounters none = {};
                (void)ray_trace_tlas<false>(sc, ray, again, none, &shadow, &tl, hit.t);
                TraceHit xh;
                if (tl.n > 0 && tie_walk(shadow, sh.instances, shadow.tlas_current, tl, ray.o, ray.d, xh) && xh.t == hit.t) {
                    const uint32_t found_in = (xh.inst >> kInstBits) & kInstMask;
                    hit.u = xh.u;
                    hit.v = xh.v;
                    hit.tri = shadow.tri_native[xh.tri];
                    hit.front = xh.front;
                    hit.inst = xh.inst & kInstMask;
                    const RefInstance& fb = sc.instances[found_in];
                    hit.lo = xform_point(fb.inverse_transform, ray.o);
                    hit.ld = xform_dir(fb.inverse_transform, ray.d);
                }
            }
            if (COUNT) cnt.shaded_hits++;
            Hit h;
            h.t = hit.t; h.u = hit.u; h.v = hit.v; h.tri = hit.tri; h.inst = hit.inst; h.lo = hit.lo; h.ld = hit.ld;
            const Shading s = get_shading_data(sh, h, hit.front, load_shade_tri(sh, h.tri));
            radiance = radiance + throughput * s.emission;
            if (i == 0) depth = length3(s.position - ray.o);
            if (!bounce_step(s, sx, sy, ray, throughput)) break;
        }
        depth = cam.far_ / (cam.far_ - cam.near_) * (1.0f - cam.near_ / depth);
        const size_t idx = (size_t)ly * fp.width + px;
        accumulate_pixel(fp, idx, radiance, accum, ldr);
        if (depth_out) depth_out[idx] = depth;
    }
    if (COUNT) flush_counters(cnt, counters);
}

void launch_ref_frame(hipStream_t stream, const DeviceScene& ds, const FrameParams& fp, const RefCamera& cam, float4* accum,
                      uint32_t* ldr, float* depth, DevCounters* counters)
{
    RefSceneDev sc;
    sc.tri_geom = ds.ref_tri_geom;
    sc.bvh = ds.ref_bvh;
    sc.instances = ds.ref_instances;
    sc.tlas = ds.ref_tlas;
    sc.n_instances = ds.n_instances;
    sc.n_tlas = ds.n_ref_tlas;
    sc.reach_tri = ds.reach_tri;
    sc.reach_inst = ds.reach_inst;
    const SceneShading sh = ds.shading();
    dim3 grid((fp.width + 31) / 32, (fp.local_rows + 7) / 8), block(256);
    const bool ties = ds.x.ok && ds.reach_tri != nullptr && !fp.debug_steps;
    if (ties) {
        if (counters) hipLaunchKernelGGL((ref_frame_kernel<true, true>), grid, block, 0, stream, sc, ds.x, sh, fp, cam, accum, ldr, depth, counters);
        else hipLaunchKernelGGL((ref_frame_kernel<false, true>), grid, block, 0, stream, sc, ds.x, sh, fp, cam, accum, ldr, depth, counters);
    } else {
        if (counters) hipLaunchKernelGGL((ref_frame_kernel<true, false>), grid, block, 0, stream, sc, ds.x, sh, fp, cam, accum, ldr, depth, counters);
        else hipLaunchKernelGGL((ref_frame_kernel<false, false>), grid, block, 0, stream, sc, ds.x, sh, fp, cam, accum, ldr, depth, counters);
    }
}

}  // namespace jpt
