// jpt_trace_core.h -- the traversal state machine shared by the wavefront kernels.
//
// One `Traversal` holds one ray's walk over the flattened two-level BVH and advances one step at a time
// (an internal record, a triangle leaf, or an instance entry), so a wave can keep lanes busy by handing a
// finished lane a new ray between steps.  Visit order, box tests and the triangle test are those of
// ray_trace_tlas / ray_trace_blas / intersectTriangle (main.glsl:224-350) on the same tree.
#pragma once

#include "jpt_kernels.h"
#include "jpt_shade.h"

namespace jpt {

constexpr int kTraceBlock = 256;
#ifndef JPT_STACK_LDS
#define JPT_STACK_LDS 20
#endif
#ifndef JPT_WAVES_PER_SIMD
#define JPT_WAVES_PER_SIMD 7
#endif
constexpr int kStackLds = JPT_STACK_LDS;              // per-lane entries kept in LDS ([entry][lane], conflict-free)
constexpr int kStackSpill = 76;            // deeper entries go to scratch (rare)
constexpr int32_t kSentinel = 0x7fffffff;  // "leave the instance" marker on the stack

struct WideSceneDev {
    const WideNode* __restrict__ blas_nodes;   // two-child records (W4 = false)
    const WideNode* __restrict__ tlas_nodes;
    const WideNodeQ* __restrict__ nodesq;      // four-child records (W4 = true), 64-byte quantised form (jpt_nodeq.h): BLAS
                                               // records, then the TLAS records, in ONE array (internal child references
                                               // of both levels index it), so a step needs no per-lane choice of base pointer
    const WideTri* __restrict__ tris;
    const WideInstance* __restrict__ instances;  // roots refer to the record kind in use
    int32_t tlas_root;
    uint32_t n_instances;
    const ReachTri* __restrict__ reach_tri;      // reach records (jpt_types.h); read by Traversal<.., REACH = true> only
    const ReachInst* __restrict__ reach_inst;
};

__device__ __forceinline__ float4 ld4(const void* p) { return *reinterpret_cast<const float4*>(p); }

// slab test of main.glsl:259-268 on one child box
__device__ __forceinline__ float slab(const f3& o, const f3& rD, float mnx, float mny, float mnz, float mxx, float mxy, float mxz)
{
    const float tx1 = (mnx - o.x) * rD.x, tx2 = (mxx - o.x) * rD.x;
    float tmin = fmin_(tx1, tx2), tmax = fmax_(tx1, tx2);
    const float ty1 = (mny - o.y) * rD.y, ty2 = (mxy - o.y) * rD.y;
    tmin = fmax_(tmin, fmin_(ty1, ty2)), tmax = fmin_(tmax, fmax_(ty1, ty2));
    const float tz1 = (mnz - o.z) * rD.z, tz2 = (mxz - o.z) * rD.z;
    tmin = fmax_(tmin, fmin_(tz1, tz2)), tmax = fmin_(tmax, fmax_(tz1, tz2));
    return (tmax >= tmin && tmax > 0.0f) ? tmin : 1e30f;
}

// TraceHit::inst holds two instance ids (a TLAS addresses at most 32 767 instances, bvh.h:59): bits 0..14 the
// instance hitInfo.blas names -- it moves only on a strictly smaller distance -- and bits 15..29 the instance whose
// local ray found the triangle kept, which is where hitInfo.position and hitInfo.out_dir come from (main.glsl:
// 246-252).  The two differ only after a tie between instances.
constexpr int kInstBits = 15;
constexpr uint32_t kInstMask = (1u << kInstBits) - 1u;
// Bit 30: the triangle kept has EXACTLY the distance of a triangle found before it (four-child records).  Which of the two
// the reference keeps depends on the order of ITS walk (`t > hitInfo.t` rejects, so the later test wins, main.glsl:247; and a
// box entered at exactly hitInfo.t is not entered, :290): the native walk cannot know, it only says so.  Cleared when a
// strictly closer triangle is found.
constexpr uint32_t kHitTied = 1u << 30;

struct TraceHit {
    float t, u, v;
    uint32_t tri, inst;
    bool front;
};

// REACH = true is the slow, exact form used to re-trace the few rays whose hit the reference's own traversal could not
// have reached (wf2_redo, jpt_kernels_wf2.hip): an instance is entered only if the world ray passes the world box the
// reference gives it, and a triangle is accepted only if the local ray passes the box of its reference leaf -- the two
// tests that decide reachability in main.glsl:270-350 (jpt_types.h, reach records).
template <bool COUNT, bool W4 = false, bool REACH = false>
struct Traversal {
    // The native route's four-child records are 64-byte quantised records (jpt_nodeq.h: four 16-byte loads instead of
    // seven and a third fewer VALU instructions -- the kernel sits at two thirds of its VALU issue capacity with the CU's
    // load-instruction rate next) walked with box tests that only have to be
    // CONSERVATIVE: the native boxes are padded and rounded outwards, and what the reference's own boxes let through is
    // decided by the reach records.  Entry / exit distances are one fused multiply-add per plane, t = q * (scale * rD) +
    // (origin * rD + ood) with ood = -o * rD, on the near and far plane words picked by the sign of the direction, with a
    // reciprocal from v_rcp_f32.  (The two-child records of reference trees use intersectAABB as written, with the exact
    // quotient 1 / d, and so do the REACH tests.)  On gfx950 an fma costs 2.3 SIMD cycles, a min / max / compare / select
    // 4.2 (tools/micro/valu_issue.hip).
    static constexpr bool kLean = W4;
    f3 wo, wd, wrD;   // world ray; wrD = its exact reciprocal direction (two-child records and REACH only)
    f3 o, d, rD;      // current-level ray (world at TLAS level, instance-local below); kLean: rD from v_rcp_f32
    f3 ood;           // kLean: -o * rD
    f3 xrD;           // REACH on four-child records: the exact 1 / d of the current level (the reference's intersectAABB)
    TraceHit hit;
    int32_t cur;
    int sp;
    uint32_t cur_inst;
    bool in_blas, have;

    // The stack: entries 0..kStackLds-1 in LDS (`lds` = this lane's column), deeper ones in `spill`, a
    // per-lane scratch array owned by the kernel (kept OUT of this struct so the struct stays in registers).
    // Capacity: lds_entries + spill_entries.  A push past it is dropped and the matching pop returns the sentinel, i.e. a
    // too-deep walk would silently lose subtrees -- which cannot happen: every upload and TLAS update computes the worst
    // case for the tree at hand (compute_stack_need) and refuses a scene that needs more than trace_stack_capacity(),
    // and jpt_scene_refit_tlas keeps the topology (only boxes move), so the bound of the last build still holds.
    struct Stack {
        int32_t* __restrict__ lds;    // entry k of this ray at lds[k * stride]
        int32_t* __restrict__ spill;  // entries past lds_entries (scratch)
        int stride = kTraceBlock;
        int lds_entries = kStackLds;
        int spill_entries = kStackSpill;
    };

    __device__ __forceinline__ void push(const Stack& st, int32_t v)
    {
        if (sp < st.lds_entries) st.lds[sp * st.stride] = v;
        else if (sp < st.lds_entries + st.spill_entries) st.spill[sp - st.lds_entries] = v;
        sp++;
    }
    __device__ __forceinline__ int32_t pop(const Stack& st)
    {
        sp--;
        if (st.lds_entries <= 0) return sp < st.spill_entries ? st.spill[sp] : kSentinel;  // (all-scratch stack: wf2_redo)
        // the LDS read is issued unconditionally (row clamped); only a walk deeper than the LDS part branches
        const int row = sp < st.lds_entries ? sp : st.lds_entries - 1;
        int32_t v = st.lds[row * st.stride];
        if (__builtin_expect(sp >= st.lds_entries, 0)) v = sp < st.lds_entries + st.spill_entries ? st.spill[sp - st.lds_entries] : kSentinel;
        return v;
    }

    __device__ __forceinline__ void begin(const WideSceneDev& sc, f3 ro, f3 rd)
    {
        wo = ro;
        wd = rd;
        o = ro;
        d = rd;
        set_level();
        wrD = (kLean && REACH) ? xrD : rD;
        hit.t = 1e9f;  // main.glsl:354
        hit.u = hit.v = 0.0f;
        hit.tri = hit.inst = 0;
        hit.front = false;
        sp = 0;
        in_blas = false;
        cur_inst = 0;
        cur = sc.tlas_root;
        have = sc.n_instances != 0;
    }

    // slab constants of the current-level ray (o, d)
    __device__ __forceinline__ void set_level()
    {
        if (kLean) {
            // Reciprocals are kept FINITE (|rD| <= 2^100: kRcpClamp, jpt_nodeq.h).  With rD = inf for a direction component of exactly 0
            // the plane distances q * (scale * rD) + (origin * rD - o * rD) are inf - inf = NaN, the min / max drop them, and the axis
            // is not tested at all: still conservative, but a ray that lies IN a coordinate plane (a camera on an axis of the scene
            // and the centre row or column of the image) then enters every box that lines up on the other two axes -- up to 1 900
            // records on a world-space tree of C3 (round 6, profiles/r06/r06f_*: eight of 16.6 M primary rays held a launch for 1.7 ms
            // instead of 0.2).  Clamped, the component counts as 2^-100 instead of 0: the ray leaves its plane by less than 1e-20 over
            // any walk, and planes off the ray's own coordinate get +-huge distances of the right signs.
            rD = mk3(__builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(d.x), -kRcpClamp, kRcpClamp), __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(d.y), -kRcpClamp, kRcpClamp),
                     __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(d.z), -kRcpClamp, kRcpClamp));
            ood = mk3(-(o.x * rD.x), -(o.y * rD.y), -(o.z * rD.z));
            if (REACH) xrD = rcp3(d);
        } else {
            rD = rcp3(d);
        }
    }

    // ---- the three kinds of records, as separate pieces so a kernel can run them in phases ----------------

    __device__ __forceinline__ bool wants_node() const { return have && cur >= 0; }
    __device__ __forceinline__ bool wants_leaf() const { return have && cur < 0 && in_blas; }
    __device__ __forceinline__ bool wants_instance() const { return have && cur < 0 && !in_blas; }
    __device__ __forceinline__ bool finished() const { return !have && sp == 0; }

    // four-child record: four box tests on the quantised planes, children visited nearest first.  The order among
    // children is a performance choice only (the closest hit does not depend on it).
    __device__ __forceinline__ void node_step4(const WideSceneDev& sc, const Stack& st, DevCounters& cnt)
    {
        const char* __restrict__ base = reinterpret_cast<const char*>(sc.nodesq);
        const uint32_t rec = (uint32_t)cur << 6;
        const float4 h0 = ld4(base + rec);           // origin.xyz, scale.x
        const float4 h1 = ld4(base + (rec + 16u));   // scale.y, scale.z, lo_x, lo_y
        const float4 h2 = ld4(base + (rec + 32u));   // lo_z, hi_x, hi_y, hi_z
        const float4 cf = ld4(base + (rec + 48u));   // child references
        node_step4_rec(h0, h1, h2, cf, st, cnt);
    }
    // ... the step itself, on a record that is already in registers (the pooled launches ask for the records of their NEXT turn
    // before they compute the current one: wf2_trace_pool)
    __device__ __forceinline__ void node_step4_rec(const float4 h0, const float4 h1, const float4 h2, const float4 cf, const Stack& st, DevCounters& cnt)
    {
        if (COUNT) {
            if (in_blas) cnt.blas_expand++;
            else cnt.tlas_expand++;
        }
        // t = (origin + q * scale - o) * rD = q * (scale * rD) + (origin * rD + ood)
        const float ax = h0.w * rD.x, ay = h1.x * rD.y, az = h1.y * rD.z;
        const float bx = __builtin_fmaf(h0.x, rD.x, ood.x), by = __builtin_fmaf(h0.y, rD.y, ood.y), bz = __builtin_fmaf(h0.z, rD.z, ood.z);
        // outwards, per axis: the planes' slack of 1/256 step and the evaluation's own error bound kWalkEps * (|b| + |o * rD|)
        // (jpt_nodeq.h): entry planes earlier, exit planes later.  (The slack alone can drop a far, small box whose corner the ray grazes:
        // tests/test_quantized_walk.py.)
        const float mx = __builtin_fmaf(kWalkSlackOverEps, __builtin_fabsf(ax), __builtin_fabsf(bx)) + __builtin_fabsf(ood.x);
        const float my = __builtin_fmaf(kWalkSlackOverEps, __builtin_fabsf(ay), __builtin_fabsf(by)) + __builtin_fabsf(ood.y);
        const float mz = __builtin_fmaf(kWalkSlackOverEps, __builtin_fabsf(az), __builtin_fabsf(bz)) + __builtin_fabsf(ood.z);
        const float nbx = __builtin_fmaf(-kWalkEps, mx, bx), fbx = __builtin_fmaf(kWalkEps, mx, bx);
        const float nby = __builtin_fmaf(-kWalkEps, my, by), fby = __builtin_fmaf(kWalkEps, my, by);
        const float nbz = __builtin_fmaf(-kWalkEps, mz, bz), fbz = __builtin_fmaf(kWalkEps, mz, bz);
        // the plane words a ray meets first / last on each axis: lo / hi, swapped where the direction is negative
        const bool negx = __float_as_int(d.x) < 0, negy = __float_as_int(d.y) < 0, negz = __float_as_int(d.z) < 0;
        const uint32_t lox = __float_as_uint(h1.z), loy = __float_as_uint(h1.w), loz = __float_as_uint(h2.x);
        const uint32_t hix = __float_as_uint(h2.y), hiy = __float_as_uint(h2.z), hiz = __float_as_uint(h2.w);
        const uint32_t nwx = negx ? hix : lox, fwx = negx ? lox : hix;
        const uint32_t nwy = negy ? hiy : loy, fwy = negy ? loy : hiy;
        const uint32_t nwz = negz ? hiz : loz, fwz = negz ? loz : hiz;
        int32_t r0 = __float_as_int(cf.x), r1 = __float_as_int(cf.y), r2 = __float_as_int(cf.z), r3 = __float_as_int(cf.w);
        constexpr uint32_t kInvalid = 0x7f800000u;
        const float tcur = hit.t;
        // entry = max over the axes of the near-plane distances (and 0), exit = min of the far-plane distances (and the
        // current hit); usable when entry <= exit and the slot is not empty
        auto key = [&](int k, int32_t ch) -> uint32_t {
            const float qnx = (float)((nwx >> (8 * k)) & 255u), qny = (float)((nwy >> (8 * k)) & 255u), qnz = (float)((nwz >> (8 * k)) & 255u);
            const float qfx = (float)((fwx >> (8 * k)) & 255u), qfy = (float)((fwy >> (8 * k)) & 255u), qfz = (float)((fwz >> (8 * k)) & 255u);
            const float t_in = fmax_(fmax_(fmax_(__builtin_fmaf(qnx, ax, nbx), __builtin_fmaf(qny, ay, nby)), __builtin_fmaf(qnz, az, nbz)), 0.0f);
            const float t_out = fmin_(fmin_(fmin_(__builtin_fmaf(qfx, ax, fbx), __builtin_fmaf(qfy, ay, fby)), __builtin_fmaf(qfz, az, fbz)), tcur);
            return ((t_in <= t_out) & (ch != kEmptyChild)) ? __float_as_uint(t_in) : kInvalid;
        };
        uint32_t k0 = key(0, r0), k1 = key(1, r1), k2 = key(2, r2), k3 = key(3, r3);
        // 5-comparator network on (key, child reference) pairs: one compare, min, max and two selects per
        // exchange.  (Looking the references up by slot after a key-only sort compiled to nested branches.)
        auto cswap = [](uint32_t& ka, uint32_t& kb, int32_t& ra, int32_t& rb) {
            const bool lt = ka < kb;
            const uint32_t klo = lt ? ka : kb, khi = lt ? kb : ka;
            const int32_t rlo = lt ? ra : rb, rhi = lt ? rb : ra;
            ka = klo; kb = khi; ra = rlo; rb = rhi;
        };
        cswap(k0, k1, r0, r1);
        cswap(k2, k3, r2, r3);
        cswap(k0, k2, r0, r2);
        cswap(k1, k3, r1, r3);
        cswap(k1, k2, r1, r2);
        // usable children are now a prefix: descend into the nearest, push the others farthest first
        have = k0 < kInvalid;
        cur = have ? r0 : cur;
        if (k1 < kInvalid) {
            if (k2 < kInvalid) {
                if (k3 < kInvalid) push(st, r3);
                push(st, r2);
            }
            push(st, r1);
        }
    }

    // internal record: both children's boxes in one 64-byte fetch; descends into the nearer valid child
    __device__ __forceinline__ void node_step(const WideSceneDev& sc, const Stack& st, DevCounters& cnt)
    {
        if (W4) {
            node_step4(sc, st, cnt);
            return;
        }
        const WideNode* n = (in_blas ? sc.blas_nodes : sc.tlas_nodes) + cur;
        const float4 a = ld4(&n->lmin[0]);  // lmin.xyz lmax.x
        const float4 b = ld4(&n->lmax[1]);  // lmax.yz rmin.xy
        const float4 c = ld4(&n->rmin[2]);  // rmin.z rmax.xyz
        const float4 e = ld4(&n->left);     // left right pad pad
        if (COUNT) {
            if (in_blas) cnt.blas_expand++;
            else cnt.tlas_expand++;
        }
        const float d1 = slab(o, rD, a.x, a.y, a.z, a.w, b.x, b.y);
        const float d2 = slab(o, rD, b.z, b.w, c.x, c.y, c.z, c.w);
        const int32_t left = __float_as_int(e.x), right = __float_as_int(e.y);
        const bool lv = d1 < hit.t, rv = d2 < hit.t;
        // `if (d1 < d2) {push right; push left} else {push left; push right}` then pop (main.glsl:293-299)
        const bool left_first = d1 < d2;
        const int32_t near_ref = left_first ? left : right, far_ref = left_first ? right : left;
        const bool near_v = left_first ? lv : rv, far_v = left_first ? rv : lv;
        have = false;
        if (near_v) {
            if (far_v) push(st, far_ref);
            cur = near_ref;
            have = true;
        } else if (far_v) {
            cur = far_ref;
            have = true;
        }
    }

    // triangle leaf (main.glsl:280-283, intersectTriangle :224-257).  Two forms of the same test.  On the native
    // tree (leaves of one or two triangles, W4) the early returns of the shader are folded into one predicate,
    // each comparison as written there so NaNs take the same side: lanes of a wave rarely agree on where to leave,
    // so the straight-line form costs the wave the same arithmetic and no branches (-3.5 % on the traversal
    // launches).  On reference-layout trees (leaves of up to 64 triangles, few lanes per leaf turn) the early
    // returns do skip work, and the branching form is 30 % faster.
    __device__ __forceinline__ void leaf_step(const WideSceneDev& sc, DevCounters& cnt)
    {
        const uint32_t bits = (uint32_t)~cur;
        const uint32_t first = bits & kLeafFirstMask;
        const uint32_t count = (bits >> kLeafCountShift) + 1u;
        have = false;
        for (uint32_t i = 0; i < count; i++) {
            const uint32_t ti = first + i;
            const WideTri* tp = sc.tris + ti;
            const float4 q0 = ld4(&tp->v0[0]);
            const float4 q1 = ld4(&tp->e1[0]);
            const float4 q2 = ld4(&tp->e2[0]);
            if (COUNT) cnt.tri_tests++;
            const f3 v0 = mk3(q0.x, q0.y, q0.z), edge1 = mk3(q1.x, q1.y, q1.z), edge2 = mk3(q2.x, q2.y, q2.z);
            if (W4) {
                const f3 pvec = cross3(d, edge2);
                const float det = dot3(edge1, pvec);
                const float invDet = 1.0f / det;
                const f3 tvec = o - v0;
                const float u = dot3(tvec, pvec) * invDet;
                const f3 qvec = cross3(tvec, edge1);
                const float v = dot3(d, qvec) * invDet;
                const float t = dot3(edge2, qvec) * invDet;
                bool out = (__builtin_fabsf(det) < 1e-5f) | (u < 0.0f) | (u > 1.0f) | (v < 0.0f) | (u + v > 1.0f) |
                           (t < 0.0f) | (t > hit.t);
                if (REACH) {
                    if (!out && !reaches_triangle(sc, ti)) out = true;
                }
                float facing = dot3(mk3(q0.w, q1.w, q2.w), d);  // the record carries cross(edge1, edge2)
                asm volatile("" : "+v"(facing));                // (keeps this, and with it the whole 16-byte loads, ahead of the branch below)
                const bool front = facing > 0.0f;
                // hitInfo.blas follows `if (hitInfo.t < minT)` after the instance's walk (main.glsl:324-327): a hit that
                // only TIES the distance found in an earlier instance replaces the triangle but not the instance
                // (under exec: moves instead of selects, skipped when no lane of the wave has a hit.  `front` is computed
                // before the branch so that the record stays three 16-byte loads: inside it the compiler splits them.)
                if (!out) {
                    const bool closer = t < hit.t;
                    hit.inst = ((closer ? cur_inst : hit.inst) & kInstMask) | (cur_inst << kInstBits) | (closer ? 0u : kHitTied);
                    hit.t = t;
                    hit.u = u;
                    hit.v = v;
                    hit.tri = ti;
                    hit.front = front;
                }
            } else {
                const f3 pvec = cross3(d, edge2);
                const float det = dot3(edge1, pvec);
                if (__builtin_fabsf(det) < 1e-5f) continue;
                const float invDet = 1.0f / det;
                const f3 tvec = o - v0;
                const float u = dot3(tvec, pvec) * invDet;
                if (u < 0.0f || u > 1.0f) continue;
                const f3 qvec = cross3(tvec, edge1);
                const float v = dot3(d, qvec) * invDet;
                if (v < 0.0f || u + v > 1.0f) continue;
                const float t = dot3(edge2, qvec) * invDet;
                if (t < 0.0f || t > hit.t) continue;
                if (REACH) {
                    if (!reaches_triangle(sc, ti)) continue;
                }
                hit.inst = (((t < hit.t) ? cur_inst : hit.inst) & kInstMask) | (cur_inst << kInstBits);  // main.glsl:324-327, see above
                hit.t = t;
                hit.u = u;
                hit.v = v;
                hit.tri = ti;
                hit.front = dot3(cross3(edge1, edge2), d) > 0.0f;
            }
        }
    }

    // the local ray against the box of the reference leaf that holds triangle ti (REACH)
    __device__ __forceinline__ bool reaches_triangle(const WideSceneDev& sc, uint32_t ti) const
    {
        const float4 a = ld4(&sc.reach_tri[ti].lo[0]), b = ld4(&sc.reach_tri[ti].hi[0]);
        if (__float_as_uint(a.w) != 0u) return true;  // the reference's BLAS root is this leaf: no box test on the way
        return slab(o, kLean ? xrD : rD, a.x, a.y, a.z, b.x, b.y, b.z) < 1e30f;   // intersectAABB with the exact 1 / d
    }

    // TLAS leaf: enter the instance (main.glsl:316-322)
    __device__ __forceinline__ void instance_step(const WideSceneDev& sc, const Stack& st, DevCounters& cnt)
    {
        cur_inst = (uint32_t)~cur;
        if (REACH) {
            // the reference enters the instance only through its TLAS leaf's box (no test when the TLAS is one leaf)
            if (sc.n_instances > 1u) {
                const float4 a = ld4(&sc.reach_inst[cur_inst].lo[0]), b = ld4(&sc.reach_inst[cur_inst].hi[0]);
                if (!(slab(wo, wrD, a.x, a.y, a.z, b.x, b.y, b.z) < 1e30f)) {
                    // not entered: the next record off the stack may be a TLAS record, whose box tests read the current-level
                    // ray -- which pop_next left as the PREVIOUS instance's local ray, counting on this entry to replace it
                    if (kLean) {
                        o = wo;
                        d = wd;
                        set_level();
                    }
                    have = false;
                    return;
                }
            }
        }
        const WideInstance* ip = sc.instances + cur_inst;
        const float4 m0 = ld4(&ip->inv[0]);
        const float4 m1 = ld4(&ip->inv[4]);
        const float4 m2 = ld4(&ip->inv[8]);
        const int32_t root = ip->root;
        if (COUNT) cnt.inst_visits++;
        // columns c0 = (m0.x m0.y m0.z) c1 = (m0.w m1.x m1.y) c2 = (m1.z m1.w m2.x) c3 = (m2.y m2.z m2.w)
        o = mk3(m0.x * wo.x + m0.w * wo.y + m1.z * wo.z + m2.y, m0.y * wo.x + m1.x * wo.y + m1.w * wo.z + m2.z,
                m0.z * wo.x + m1.y * wo.y + m2.x * wo.z + m2.w);
        d = mk3(m0.x * wd.x + m0.w * wd.y + m1.z * wd.z, m0.y * wd.x + m1.x * wd.y + m1.w * wd.z,
                m0.z * wd.x + m1.y * wd.y + m2.x * wd.z);
        set_level();
        in_blas = true;
        push(st, kSentinel);
        cur = root;
        have = true;
    }

    // take the next record off the stack (precondition: !have && sp > 0).  Leaving an instance restores
    // the world ray.  Afterwards either `have` or finished().
    __device__ __forceinline__ void pop_next(const Stack& st)
    {
        cur = pop(st);
        if (cur == kSentinel) {
            in_blas = false;
            if (sp == 0) return;   // the instance was the last record of the walk: nobody needs the world ray's constants any more
            cur = pop(st);
            // Only a TLAS RECORD's box tests read the current-level ray and its slab constants; an instance entry starts
            // from the world ray again (instance_step).  With a handful of instances under one TLAS record the next entry
            // is nearly always another instance: the world ray's constants are then not restored at all.
            if (!kLean || cur >= 0) {   // (C3 -1.5 %, close-up -1.7 %: profiles/r03/r03i_ab_pop_restore.txt)
                o = wo;
                d = wd;
                if (kLean) set_level();   // (recomputed rather than kept: three v_rcp_f32 per instance left, registers saved)
                else rD = wrD;
            }
        }
        have = true;
    }

    // Advances by one record.  Returns false when the walk is complete (hit holds the closest hit or t = 1e9).
    __device__ __forceinline__ bool step(const WideSceneDev& sc, const Stack& st, DevCounters& cnt)
    {
        if (!have) {
            if (sp == 0) return false;
            pop_next(st);
            if (!have) return false;
        }
        if (cur >= 0) node_step(sc, st, cnt);
        else if (in_blas) leaf_step(sc, cnt);
        else instance_step(sc, st, cnt);
        return true;
    }
};

}  // namespace jpt
