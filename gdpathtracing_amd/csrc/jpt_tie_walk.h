// jpt_tie_walk.h -- who wins an exact distance tie, decided where the reference decides it.
//
// Of two triangles at exactly the same distance main.glsl keeps the one its walk tests LATER (`t > hitInfo.t` rejects, :247),
// and a box whose entry distance equals hitInfo.t is not entered at all (`d < hitInfo.t`, :290): the winner is a property of
// the reference's visiting order, which the native tree does not have.  The native walk only flags such a hit (kHitTied);
// wf2_finish then
//   1. walks the native tree once more with hitInfo.t preset to the tying distance and notes the leaves whose triangles are
//      accepted at it (the tying triangles: nothing reachable is closer), and
//   2. repeats the REFERENCE's walk -- its nodes (the copy kept beside the native scene: ExactShadow, jpt_builder.h), its
//      order, its arithmetic as written -- through the ancestors of the reference leaves that hold those triangles only.
// What step 2 skips cannot change its outcome: every other subtree holds no reachable triangle at or below the tying
// distance, so in the reference's full walk it can only contribute FARTHER hits, and a farther hitInfo.t never keeps a box
// on the way to a tying triangle from being entered (its entry distance is at most the tying distance).  Among the boxes
// that are visited, which tying triangle is tested first and whether a later one's box is still entered once hitInfo.t
// equals the tying distance come out as in the reference, because it is the reference's walk.
#pragma once
#include "jpt_trace_core.h"

namespace jpt {

#if defined(__HIPCC__)

constexpr int kTieLeaves = 8;    // most tying leaves a vertex may have (more: the reference's whole walk, see tie_walk)
constexpr int kTieTlas = 96;     // most TLAS nodes on the way to them

struct TieLeaves {
    uint32_t leaf[kTieLeaves];   // reference BVH node
    uint32_t inst[kTieLeaves];
    int n = 0;
    bool overflow = false;
    __device__ __forceinline__ void add(uint32_t l, uint32_t i)
    {
        for (int k = 0; k < n; k++)
            if (leaf[k] == l && inst[k] == i) return;
        if (n == kTieLeaves) {
            overflow = true;
            return;
        }
        leaf[n] = l;
        inst[n] = i;
        n++;
    }
};

// step 1: the leaves of the native tree whose triangles the REACH walk accepts at distance t_tie
template <bool COUNT, bool W4>
__device__ __forceinline__ void tie_leaves(const WideSceneDev& sc, const TieShadowDev& x, const typename Traversal<COUNT, W4, true>::Stack& st, f3 ro,
                                           f3 rd, float t_tie, TieLeaves& out, DevCounters& cnt)
{
    Traversal<COUNT, W4, true> w;
    w.begin(sc, ro, rd);
    w.hit.t = t_tie;
    for (;;) {
        if (!w.have) {
            if (w.sp == 0) break;
            w.pop_next(st);
            if (!w.have) break;
        }
        if (w.cur >= 0) w.node_step(sc, st, cnt);
        else if (!w.in_blas) w.instance_step(sc, st, cnt);
        else {
            const uint32_t bits = (uint32_t)~w.cur;
            const uint32_t first = bits & kLeafFirstMask, count = (bits >> kLeafCountShift) + 1u;
            w.hit.tri = 0xffffffffu;
            w.leaf_step(sc, cnt);
            if (w.hit.tri != 0xffffffffu)   // (every triangle of the leaf: a superset of the tying ones is as good)
                for (uint32_t i = 0; i < count; i++) {
                    const uint32_t rt = x.native_ref[first + i];
                    if (rt != 0xffffffffu) out.add(x.tri_leaf[rt], w.cur_inst);
                }
        }
    }
}

// step 2: main.glsl:305-350 / :270-303 / :224-257 on the reference's arrays, through the ancestors of `tl` only.  false: the
// walk could not be set up (an out-of-date TLAS and ties in more than one instance): the caller keeps what it has.
// `inst_records`: the instances' CURRENT records (the shading pass's array: after a device refit the copy's own are out of
// date).  `tlas_current` false -- a device refit since the last host update, the reference's TLAS of the moved scene is not
// known -- limits the walk to ties inside ONE instance, which the instance level has no say in: the BLAS part alone.
// `tl.overflow` (more tying leaves than the list holds: many coincident copies of a triangle): no restriction at all -- the
// reference's whole walk, slow and exact.
__device__ __forceinline__ bool tie_walk(const TieShadowDev& x, const RefInstance* __restrict__ inst_records, bool tlas_current, const TieLeaves& tl,
                                         f3 wo, f3 wd, TraceHit& hit)
{
    bool everything = tl.overflow;   // (also: more TLAS nodes on the way than `way` holds)
    hit.t = 1e9f;
    hit.u = hit.v = 0.0f;
    hit.tri = 0u;
    hit.front = false;
    uint32_t blas_of_hit = 0u, found_in = 0u;
    float minT = 1e9f;
    uint32_t bs[64];
    // main.glsl:316-327: one instance's BLAS, through the ancestors of its tying leaves
    auto visit_instance = [&](uint32_t inst) {
        const RefInstance& b = inst_records[inst];
        const f3 o = xform_point(b.inverse_transform, wo), d = xform_dir(b.inverse_transform, wd);
        const f3 rD = rcp3(d);
        int bsp = 0;
        bs[bsp++] = x.instances[inst].blas_index;   // (the mesh an instance shows does not change when it moves)
        while (bsp > 0) {
            const RefBvhNode& bn = x.bvh[bs[--bsp]];
            if (bn.tri_count > 0u) {
                for (uint32_t i = 0; i < bn.tri_count; i++) {
                    const uint32_t ti = bn.first_tri_index + i;
                    const RefTriGeometry& tri = x.tri_geom[ti];
                    const f3 v0 = mk3(tri.vertices[0].x, tri.vertices[0].y, tri.vertices[0].z);
                    const f3 v1 = mk3(tri.vertices[1].x, tri.vertices[1].y, tri.vertices[1].z);
                    const f3 v2 = mk3(tri.vertices[2].x, tri.vertices[2].y, tri.vertices[2].z);
                    const f3 edge1 = v1 - v0, edge2 = v2 - v0;
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
                    hit.t = t;
                    hit.u = u;
                    hit.v = v;
                    hit.tri = ti;
                    hit.front = dot3(cross3(edge1, edge2), d) > 0.0f;
                    found_in = inst;
                }
                continue;
            }
            const uint32_t li = bn.left_child, ri = bn.right_child;
            // a node without triangles AND without children is the empty leaf that stands for an empty mesh
            // (SceneBuilder::commit, native_from_uploaded): nothing below it -- read as an interior node it would send this
            // instance's local ray through node 0, another mesh's tree (reachable in the whole-walk mode)
            if (li == 0u && ri == 0u) continue;
            const RefBvhNode& cl = x.bvh[li];
            const RefBvhNode& cr = x.bvh[ri];
            const float d1 = slab(o, rD, cl.aabbMin.x, cl.aabbMin.y, cl.aabbMin.z, cl.aabbMax.x, cl.aabbMax.y, cl.aabbMax.z);
            const float d2 = slab(o, rD, cr.aabbMin.x, cr.aabbMin.y, cr.aabbMin.z, cr.aabbMax.x, cr.aabbMax.y, cr.aabbMax.z);
            // (a subtree is on the way iff a tying leaf of this instance is numbered inside it)
            const uint32_t le = x.subtree_end[li], re = x.subtree_end[ri];
            bool lw = everything, rw = everything;
            for (int k = 0; k < tl.n; k++) {
                const bool here = tl.inst[k] == inst;
                lw = lw || (here && tl.leaf[k] >= li && tl.leaf[k] < le);
                rw = rw || (here && tl.leaf[k] >= ri && tl.leaf[k] < re);
            }
            const bool leftValid = d1 < hit.t && lw, rightValid = d2 < hit.t && rw;
            if (d1 < d2) {
                if (rightValid && bsp < 64) bs[bsp++] = ri;
                if (leftValid && bsp < 64) bs[bsp++] = li;
            } else {
                if (leftValid && bsp < 64) bs[bsp++] = li;
                if (rightValid && bsp < 64) bs[bsp++] = ri;
            }
        }
        if (hit.t < minT) {   // main.glsl:324-327
            blas_of_hit = inst;
            minT = hit.t;
        }
    };
    if (!tlas_current) {
        if (everything) return false;
        for (int k = 1; k < tl.n; k++)
            if (tl.inst[k] != tl.inst[0]) return false;
        visit_instance(tl.inst[0]);
        hit.inst = (blas_of_hit & kInstMask) | (found_in << kInstBits);
        return true;
    }
    // TLAS nodes on the way to the tying instances
    uint32_t way[kTieTlas];
    int n_way = 0;
    for (int k = 0; k < tl.n && !everything; k++) {
        uint32_t node = x.inst_tlas_leaf[tl.inst[k]];
        for (;;) {
            bool seen = false;
            for (int j = 0; j < n_way; j++) seen = seen || way[j] == node;
            if (seen) break;   // (and with it everything above)
            if (n_way == kTieTlas) {
                everything = true;
                break;
            }
            way[n_way++] = node;
            if (node == 0u) break;
            node = x.tlas_parent[node];
        }
    }
    auto on_way = [&](uint32_t node) {
        bool r = everything;
        for (int j = 0; j < n_way; j++) r = r || way[j] == node;
        return r;
    };
    const f3 wrD = rcp3(wd);
    uint32_t ts[64];
    int tsp = 0;
    ts[tsp++] = 0u;
    while (tsp > 0) {
        const RefTlasNode& node = x.tlas[ts[--tsp]];
        const uint32_t lr = node.leftRight;
        if (lr == 0u) {
            visit_instance(node.blas);
            continue;
        }
        const uint32_t left = lr & 0xffffu, right = lr >> 16;
        const RefTlasNode& cl = x.tlas[left];
        const RefTlasNode& cr = x.tlas[right];
        const float d1 = slab(wo, wrD, cl.aabbMin[0], cl.aabbMin[1], cl.aabbMin[2], cl.aabbMax[0], cl.aabbMax[1], cl.aabbMax[2]);
        const float d2 = slab(wo, wrD, cr.aabbMin[0], cr.aabbMin[1], cr.aabbMin[2], cr.aabbMax[0], cr.aabbMax[1], cr.aabbMax[2]);
        const bool leftValid = d1 < hit.t && on_way(left), rightValid = d2 < hit.t && on_way(right);
        if (d1 < d2) {
            if (rightValid && tsp < 64) ts[tsp++] = right;
            if (leftValid && tsp < 64) ts[tsp++] = left;
        } else {
            if (leftValid && tsp < 64) ts[tsp++] = left;
            if (rightValid && tsp < 64) ts[tsp++] = right;
        }
    }
    hit.inst = (blas_of_hit & kInstMask) | (found_in << kInstBits);
    return true;
}

#endif  // __HIPCC__

}  // namespace jpt
