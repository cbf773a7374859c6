// jpt_builder.cpp -- see jpt_builder.h.  Host C++ only (no device code).
#include "jpt_builder.h"
#include "jpt_instance_math.h"
#include "jpt_tuning.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <future>
#include <numeric>

namespace jpt {

void RefScene::clear()
{
    triangles.clear();
    tri_geom.clear();
    tri_data.clear();
    bvh_nodes.clear();
    instances.clear();
    tlas_nodes.clear();
    mesh_roots.clear();
    reach_tri.clear();
    reach_inst.clear();
    mesh_ref_root.clear();
    inst_cut_boxes.clear();
    inst_cut_range.clear();
    up_mesh_root.clear();
    up_blas_index.clear();
    exact.clear();
}

// ------------------------------------------------------------------------------------------------
// small vec4 helpers with the reference's semantics: std::min/std::max per component INCLUDING w
// (vec.h:78-86), arithmetic on w as well (vec.h:53-71).
namespace {

inline float lo_(float a, float b) { return (b < a) ? b : a; }  // std::min
inline float hi_(float a, float b) { return (a < b) ? b : a; }  // std::max
// bin of a scaled centroid coordinate, clamped BEFORE the conversion: (int) of a NaN or of a value past INT_MAX is
// undefined, and uploaded instances / triangles may hold either (found by the UBSan run of tests/test_upload_mutations.py);
// a finite in-range x converts exactly as the plain cast did, so valid scenes build the same trees
inline int bin_of(float x, int bins) { return !(x > 0.0f) ? 0 : (x >= (float)bins ? bins - 1 : (int)x); }
inline float comp(const Vec4& v, int axis) { return axis == 0 ? v.x : axis == 1 ? v.y : axis == 2 ? v.z : axis == 3 ? v.w : v.x; }

struct Box4 {
    Vec4 lo, hi;
    // bvh.cpp:6-10: `vec4(FLT_MAX)` / `vec4(FLT_MIN)` only set x (vec.h:49) -> y = z = 0, w = 1.
    static Box4 reference_default() { return Box4{{FLT_MAX, 0.0f, 0.0f, 1.0f}, {FLT_MIN, 0.0f, 0.0f, 1.0f}}; }
    void grow(const Vec4& p)  // bvh.cpp:12-16
    {
        lo = Vec4{lo_(lo.x, p.x), lo_(lo.y, p.y), lo_(lo.z, p.z), lo_(lo.w, p.w)};
        hi = Vec4{hi_(hi.x, p.x), hi_(hi.y, p.y), hi_(hi.z, p.z), hi_(hi.w, p.w)};
    }
    float half_area() const  // bvh.h:39-43
    {
        const float dx = hi.x - lo.x, dy = hi.y - lo.y, dz = hi.z - lo.z;
        return dx * dy + dy * dz + dz * dx;
    }
};

// ---- reference-exact BLAS build (bvh.cpp:39-185) ------------------------------------------------

struct ExactBlasBuilder {
    std::vector<RefBvhNode>& nodes;
    std::vector<RefTriangle>& tris;

    // EvaluateSAH (bvh.cpp:39-106): 8 bins over the node's VERTEX box, triangles binned by centroid
    float sweep_axis(const RefBvhNode& node, int axis, float& split_out) const
    {
        constexpr int kBins = 8;
        Box4 bin_box[kBins];
        int bin_n[kBins];
        for (int i = 0; i < kBins; i++) {
            bin_box[i] = Box4::reference_default();
            bin_n[i] = 0;
        }
        const float lo = comp(node.aabbMin, axis), hi = comp(node.aabbMax, axis);
        const float range = hi - lo;
        if (range < 1e-6f) return 1e+30f;
        const float inv_range = 1.0f / range;
        for (uint32_t i = 0; i < node.tri_count; i++) {
            const RefTriangle& t = tris[node.first_tri_index + i];
            // int(float) as bvh.cpp:64 writes it is undefined for a NaN or an out-of-range value (a mesh with such a vertex);
            // the reference's x86 build gets cvttss2si's "integer indefinite", INT_MIN, which the clamp turns into bin 0
            const float fb = float(kBins) * (comp(t.centroid, axis) - lo) * inv_range;
            const int b = std::clamp((fb >= -2147483648.0f && fb < 2147483648.0f) ? int(fb) : INT32_MIN, 0, kBins - 1);
            bin_n[b]++;
            bin_box[b].grow(t.vertices[0]);
            bin_box[b].grow(t.vertices[1]);
            bin_box[b].grow(t.vertices[2]);
        }
        Box4 prefix_box[kBins];
        int prefix_n[kBins];
        Box4 acc = Box4::reference_default();
        int n = 0;
        for (int i = 0; i < kBins - 1; i++) {
            acc.grow(bin_box[i].lo);
            acc.grow(bin_box[i].hi);
            n += bin_n[i];
            prefix_box[i] = acc;
            prefix_n[i] = n;
        }
        float best = 1e+30f;
        acc = Box4::reference_default();
        n = 0;
        for (int i = kBins - 1; i > 0; i--) {
            acc.grow(bin_box[i].lo);
            acc.grow(bin_box[i].hi);
            n += bin_n[i];
            const float cost = prefix_box[i - 1].half_area() * float(prefix_n[i - 1]) + acc.half_area() * float(n);
            if (cost < best) {
                best = cost;
                split_out = lo + (float(i) / float(kBins)) * range;
            }
        }
        return best;
    }

    // build_recursive (bvh.cpp:108-185): pre-order numbering, left child = parent + 1
    uint32_t build(int start, int end)
    {
        if (start >= end) return 0;
        const uint32_t self = (uint32_t)nodes.size();
        nodes.emplace_back();
        {
            Box4 box = Box4::reference_default();
            for (int i = start; i < end; i++)
                for (int j = 0; j < 3; j++) box.grow(tris[i].vertices[j]);
            RefBvhNode& n = nodes[self];
            n.aabbMin = box.lo;
            n.aabbMax = box.hi;
            n.left_child = n.right_child = 0;
            n.first_tri_index = (uint32_t)start;
            n.tri_count = (uint32_t)(end - start);
        }
        const RefBvhNode node = nodes[self];
        if (node.tri_count <= 4) return self;

        float best_split = 0.0f, best_cost = 1e30f;
        int best_axis = -1;
        for (int axis = 0; axis < 3; axis++) {
            float split = 0.0f;
            const float cost = sweep_axis(node, axis, split);
            if (cost < best_cost) {
                best_cost = cost;
                best_split = split;
                best_axis = axis;
            }
        }
        const float ex = node.aabbMax.x - node.aabbMin.x, ey = node.aabbMax.y - node.aabbMin.y,
                    ez = node.aabbMax.z - node.aabbMin.z;
        const float parent_cost = float(node.tri_count) * (ex * ey + ey * ez + ez * ex);
        if (best_cost * 0.8f >= parent_cost) return self;

        int i = start, j = end - 1;
        while (i <= j) {
            if (comp(tris[i].centroid, best_axis) < best_split) i++;
            else std::swap(tris[i], tris[j--]);
        }
        const int left_count = i - start;
        if (left_count == 0 || left_count == (int)node.tri_count) {
            const int mid = start + (end - start) / 2;
            // libstdc++'s element order (the reference on Linux); other STLs order ties differently
            std::nth_element(tris.begin() + start, tris.begin() + mid, tris.begin() + end,
                             [best_axis](const RefTriangle& a, const RefTriangle& b) {
                                 return comp(a.centroid, best_axis) < comp(b.centroid, best_axis);
                             });
            i = mid;
        }
        const uint32_t l = build(start, i);
        nodes[self].left_child = l;
        const uint32_t r = build(i, end);
        nodes[self].right_child = r;
        nodes[self].tri_count = 0;
        return self;
    }
};

// The reference's builder on a copy of one mesh: by_original[i] = box of the reference leaf that ends up holding
// triangle i of the mesh (`always` when that leaf is the root), root = the reference root's box.
// nodes_out: that tree (indices relative to its root = 0, first_tri_index relative to the mesh's first triangle);
// order_out[k] = which triangle of the mesh ends up at position k of the reference's order.
void reference_leaf_boxes(const std::vector<RefTriangle>& mesh_tris, std::vector<ReachTri>& by_original, ReachInst& root,
                          std::vector<RefBvhNode>& nodes, std::vector<uint32_t>& order_out)
{
    std::vector<RefTriangle> copy(mesh_tris);
    for (size_t i = 0; i < copy.size(); i++) copy[i]._pad = (uint32_t)i;  // travels with the triangle through the partition swaps
    nodes.clear();
    nodes.reserve(copy.size());
    ExactBlasBuilder eb{nodes, copy};
    (void)eb.build(0, (int)copy.size());
    by_original.assign(copy.size(), ReachTri{});
    order_out.resize(copy.size());
    for (size_t k = 0; k < copy.size(); k++) order_out[k] = copy[k]._pad;
    std::memset(&root, 0, sizeof root);
    if (nodes.empty()) return;
    root.root_lo[0] = nodes[0].aabbMin.x; root.root_lo[1] = nodes[0].aabbMin.y; root.root_lo[2] = nodes[0].aabbMin.z;
    root.root_hi[0] = nodes[0].aabbMax.x; root.root_hi[1] = nodes[0].aabbMax.y; root.root_hi[2] = nodes[0].aabbMax.z;
    for (size_t ni = 0; ni < nodes.size(); ni++) {
        const RefBvhNode& n = nodes[ni];
        if (n.tri_count == 0) continue;
        ReachTri r;
        r.lo[0] = n.aabbMin.x; r.lo[1] = n.aabbMin.y; r.lo[2] = n.aabbMin.z;
        r.hi[0] = n.aabbMax.x; r.hi[1] = n.aabbMax.y; r.hi[2] = n.aabbMax.z;
        r.always = ni == 0 ? 1u : 0u;
        r._pad = 0;
        for (uint32_t k = 0; k < n.tri_count; k++) by_original[copy[n.first_tri_index + k]._pad] = r;
    }
}

// ---- native binned-SAH BLAS build -----------------------------------------------------------------

struct Box3 {
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    void grow(const float* p)
    {
        for (int k = 0; k < 3; k++) {
            lo[k] = std::min(lo[k], p[k]);
            hi[k] = std::max(hi[k], p[k]);
        }
    }
    void grow(const Box3& b)
    {
        for (int k = 0; k < 3; k++) {
            lo[k] = std::min(lo[k], b.lo[k]);
            hi[k] = std::max(hi[k], b.hi[k]);
        }
    }
    float half_area() const
    {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
};

struct SahBlasBuilder {
    std::vector<RefBvhNode>& nodes;
    std::vector<RefTriangle>& tris;  // permuted in place, like the reference
    int base;                        // first triangle of this mesh in `tris`
    std::vector<Box3> tri_box;       // per triangle (mesh-local index)
    std::vector<float> centroid;     // 3 per triangle
    std::vector<uint32_t> order;     // permutation being built
    float pad = 0.0f;

#ifndef JPT_SAH_BINS
#define JPT_SAH_BINS 16
#endif
    static constexpr int kBins = JPT_SAH_BINS;
    // largest leaf the builder keeps without a split that pays
    static int max_leaf() { return Tuning::max_leaf; }

    void prepare(int start, int end)
    {
        base = start;
        const int n = end - start;
        tri_box.resize(n);
        centroid.resize((size_t)n * 3);
        order.resize(n);
        Box3 all;
        for (int i = 0; i < n; i++) {
            const RefTriangle& t = tris[start + i];
            Box3 b;
            for (int j = 0; j < 3; j++) b.grow(&t.vertices[j].x);
            tri_box[i] = b;
            all.grow(b);
            for (int k = 0; k < 3; k++) centroid[(size_t)i * 3 + k] = 0.5f * (b.lo[k] + b.hi[k]);
            order[i] = (uint32_t)i;
        }
        // boxes are padded so every Moller-Trumbore-accepted hit also passes the slab test in float
        float m = 0.0f;
        for (int k = 0; k < 3; k++) m = std::max(m, std::max(std::fabs(all.lo[k]), std::fabs(all.hi[k])));
        pad = m * 2e-6f + 1e-30f;
    }

    // Builds the subtree over order[lo, hi) into `out` (pre-order: left child = parent + 1); child indices are
    // relative to out's first element, first_tri_index is absolute.  Large subtrees are built by parallel
    // tasks on disjoint slices of `order` and spliced in a fixed order, so the result does not depend on timing.
    void build_into(std::vector<RefBvhNode>& out, int lo, int hi, int depth)
    {
        const size_t self = out.size();
        out.emplace_back();
        Box3 box, cbox;
        for (int i = lo; i < hi; i++) {
            box.grow(tri_box[order[i]]);
            cbox.grow(&centroid[(size_t)order[i] * 3]);
        }
        {
            RefBvhNode& n = out[self];
            n.aabbMin = Vec4{box.lo[0] - pad, box.lo[1] - pad, box.lo[2] - pad, 1.0f};
            n.aabbMax = Vec4{box.hi[0] + pad, box.hi[1] + pad, box.hi[2] + pad, 1.0f};
            n.left_child = n.right_child = 0;
            n.first_tri_index = (uint32_t)(base + lo);
            n.tri_count = (uint32_t)(hi - lo);
        }
        const int count = hi - lo;
        if (count <= 1) return;

        // one pass over the triangles bins all three axes
        Box3 bb[3][kBins];
        int bn[3][kBins] = {};
        float c0[3], scale[3];
        bool usable[3];
        for (int a = 0; a < 3; a++) {
            c0[a] = cbox.lo[a];
            usable[a] = cbox.hi[a] > cbox.lo[a];
            scale[a] = usable[a] ? float(kBins) / (cbox.hi[a] - cbox.lo[a]) : 0.0f;
        }
        for (int i = lo; i < hi; i++) {
            const uint32_t t = order[i];
            const Box3& tb = tri_box[t];
            for (int a = 0; a < 3; a++) {
                if (!usable[a]) continue;
                const int k = bin_of((centroid[(size_t)t * 3 + a] - c0[a]) * scale[a], kBins);
                bb[a][k].grow(tb);
                bn[a][k]++;
            }
        }
        int best_axis = -1, best_bin = -1;
        float best_cost = FLT_MAX;
        for (int axis = 0; axis < 3; axis++) {
            if (!usable[axis]) continue;
            float right_area[kBins];
            int right_n[kBins];
            Box3 acc;
            int n = 0;
            for (int k = kBins - 1; k > 0; k--) {
                acc.grow(bb[axis][k]);
                n += bn[axis][k];
                right_area[k] = n ? acc.half_area() : 0.0f;
                right_n[k] = n;
            }
            acc = Box3();
            n = 0;
            for (int k = 0; k < kBins - 1; k++) {
                acc.grow(bb[axis][k]);
                n += bn[axis][k];
                if (n == 0 || right_n[k + 1] == 0) continue;
                const float cost = acc.half_area() * float(n) + right_area[k + 1] * float(right_n[k + 1]);
                if (cost < best_cost) {
                    best_cost = cost;
                    best_axis = axis;
                    best_bin = k;
                }
            }
        }
        const float leaf_cost = float(count) * box.half_area();
        // traversal step ~ 1.2 triangle tests
        const bool split_pays = best_axis >= 0 && (best_cost + 1.2f * box.half_area()) < leaf_cost;
        if (count <= max_leaf() && !split_pays) return;

        int mid = lo;
        if (best_axis >= 0) {
            const float cc = c0[best_axis], sc = scale[best_axis];
            auto it = std::partition(order.begin() + lo, order.begin() + hi, [&](uint32_t t) {
                const int k = bin_of((centroid[(size_t)t * 3 + best_axis] - cc) * sc, kBins);
                return k <= best_bin;
            });
            mid = (int)(it - order.begin());
        }
        if (mid == lo || mid == hi) mid = lo + count / 2;  // all centroids coincide: split by index

        if (count >= kParallelMin && depth < kParallelDepth) {
            std::vector<RefBvhNode> left_nodes, right_nodes;
            auto fut = std::async(std::launch::async, [&] { build_into(left_nodes, lo, mid, depth + 1); });
            build_into(right_nodes, mid, hi, depth + 1);
            fut.get();
            const uint32_t l = (uint32_t)(self + 1), r = (uint32_t)(self + 1 + left_nodes.size());
            splice(out, left_nodes, l);
            splice(out, right_nodes, r);
            out[self].left_child = l;
            out[self].right_child = r;
        } else {
            const uint32_t l = (uint32_t)out.size();
            build_into(out, lo, mid, depth + 1);
            out[self].left_child = l;
            const uint32_t r = (uint32_t)out.size();
            build_into(out, mid, hi, depth + 1);
            out[self].right_child = r;
        }
        out[self].tri_count = 0;
    }

    // append `sub` (indices relative to its own start) at position `at` of `out`
    static void splice(std::vector<RefBvhNode>& out, const std::vector<RefBvhNode>& sub, uint32_t at)
    {
        for (const RefBvhNode& n : sub) {
            RefBvhNode m = n;
            if (m.tri_count == 0) {
                m.left_child += at;
                m.right_child += at;
            }
            out.push_back(m);
        }
    }

    static constexpr int kParallelMin = 32768;
    static constexpr int kParallelDepth = 4;  // up to 16 tasks

    // appends the tree to the shared node array; returns the root's index there
    uint32_t build(int lo, int hi)
    {
        std::vector<RefBvhNode> local;
        local.reserve((size_t)(hi - lo));
        build_into(local, lo, hi, 0);
        const uint32_t root = (uint32_t)nodes.size();
        splice(nodes, local, root);
        return root;
    }

    void apply_order(int start, int end)
    {
        std::vector<RefTriangle> tmp(tris.begin() + start, tris.begin() + end);
        for (int i = 0; i < end - start; i++) tris[start + i] = tmp[order[i]];
    }
};

// (Utils::transform_to_float, utils.h:15-37, is transform12_to_mat16 in jpt_instance_math.h)

// TLAS::FindBestMatch (bvh.cpp:319-340)
int nearest_partner(const std::vector<RefTlasNode>& nodes, const std::vector<int>& list, int n, int a)
{
    float smallest = 1e30f;
    int best = -1;
    const RefTlasNode& na = nodes[list[a]];
    for (int b = 0; b < n; b++) {
        if (b == a) continue;
        const RefTlasNode& nb = nodes[list[b]];
        const float ex = hi_(na.aabbMax[0], nb.aabbMax[0]) - lo_(na.aabbMin[0], nb.aabbMin[0]);
        const float ey = hi_(na.aabbMax[1], nb.aabbMax[1]) - lo_(na.aabbMin[1], nb.aabbMin[1]);
        const float ez = hi_(na.aabbMax[2], nb.aabbMax[2]) - lo_(na.aabbMin[2], nb.aabbMin[2]);
        const float area = ex * ey + ey * ez + ez * ex;
        if (area < smallest) {
            smallest = area;
            best = b;
        }
    }
    return best;
}

// TLAS::build (bvh.cpp:264-317): agglomerative clustering, 2N nodes, root copied to slot 0
bool build_tlas(const std::vector<RefInstance>& inst, std::vector<RefTlasNode>& out, std::string& err)
{
    int count = (int)inst.size();
    out.clear();
    out.emplace_back();  // slot 0 = root
    std::memset(&out[0], 0, sizeof(RefTlasNode));
    if (count == 0) return true;
    if (count * 2 > 65535) {  // 16-bit child indices (bvh.h:59, bvh.cpp:300)
        err = "TLAS: more than 32767 instances do not fit the reference's 16-bit child indices";
        return false;
    }
    std::vector<int> live;
    live.reserve(count);
    for (int i = 0; i < count; i++) {
        RefTlasNode n;
        std::memset(&n, 0, sizeof n);
        n.aabbMin[0] = inst[i].aabbMin.x; n.aabbMin[1] = inst[i].aabbMin.y; n.aabbMin[2] = inst[i].aabbMin.z;
        n.aabbMax[0] = inst[i].aabbMax.x; n.aabbMax[1] = inst[i].aabbMax.y; n.aabbMax[2] = inst[i].aabbMax.z;
        n.blas = (uint32_t)i;
        n.leftRight = 0;
        live.push_back((int)out.size());
        out.push_back(n);
    }
    // FindBestMatch keeps -1 when no union's half-area is below its 1e30 start value (NaN or huge instance boxes); the
    // reference then indexes its list with -1 (bvh.cpp:285-287).  Here that is an error of the scene, not of the process.
    const char* no_partner = "TLAS: an instance's world box is not finite (TLAS::FindBestMatch finds no partner, bvh.cpp:319-340)";
    int a = 0, b = nearest_partner(out, live, count, a);
    if (count > 1 && b < 0) {
        err = no_partner;
        return false;
    }
    while (count > 1) {
        const int c = nearest_partner(out, live, count, b);
        if (c < 0) {
            err = no_partner;
            return false;
        }
        if (a == c) {
            const int ia = live[a], ib = live[b];
            RefTlasNode n;
            std::memset(&n, 0, sizeof n);  // .blas of internal nodes is uninitialised in the reference
            n.leftRight = (uint32_t)ia + ((uint32_t)ib << 16);
            for (int k = 0; k < 3; k++) {
                n.aabbMin[k] = lo_(out[ia].aabbMin[k], out[ib].aabbMin[k]);
                n.aabbMax[k] = hi_(out[ia].aabbMax[k], out[ib].aabbMax[k]);
            }
            live[a] = (int)out.size();
            out.push_back(n);
            live[b] = live[--count];
            b = nearest_partner(out, live, count, a);
            if (count > 1 && b < 0) {
                err = no_partner;
                return false;
            }
        } else {
            a = b;
            b = c;
        }
    }
    out[0] = out[live[a]];
    return true;
}

// Native TLAS: top-down binned SAH over the instances' world boxes, emitted in the reference's TLASNode format
// (slot 0 = copy of the root, leaves 1..N, 16-bit child indices) so the rest of the pipeline is unchanged.
// The reference's agglomerative clustering (build_tlas) is O(N^2) and yields deep, overlapping trees for grids
// of instances; the closest hit does not depend on the TLAS shape.
struct TlasSahBuilder {
    const std::vector<RefInstance>& inst;
    std::vector<RefTlasNode>& out;
    std::vector<uint32_t> order;

    static void box_of(const RefInstance& i, float* lo, float* hi)
    {
        lo[0] = i.aabbMin.x; lo[1] = i.aabbMin.y; lo[2] = i.aabbMin.z;
        hi[0] = i.aabbMax.x; hi[1] = i.aabbMax.y; hi[2] = i.aabbMax.z;
    }
    uint32_t build(int lo, int hi)
    {
        if (hi - lo == 1) return 1u + order[(size_t)lo];
        Box3 box, cbox;
        for (int i = lo; i < hi; i++) {
            float a[3], b[3];
            box_of(inst[order[(size_t)i]], a, b);
            box.grow(a);
            box.grow(b);
            const float c[3] = {0.5f * (a[0] + b[0]), 0.5f * (a[1] + b[1]), 0.5f * (a[2] + b[2])};
            cbox.grow(c);
        }
        constexpr int kBins = 16;
        constexpr int kSweepMax = 4096;
        int best_axis = -1, best_bin = -1;
        float best_cost = FLT_MAX;
        for (int axis = 0; axis < 3; axis++) {
            const float c0 = cbox.lo[axis], c1 = cbox.hi[axis];
            if (!(c1 > c0)) continue;
            const float scale = float(kBins) / (c1 - c0);
            Box3 bb[kBins];
            int bn[kBins] = {0};
            for (int i = lo; i < hi; i++) {
                float a[3], b[3];
                box_of(inst[order[(size_t)i]], a, b);
                const int k = bin_of((0.5f * (a[axis] + b[axis]) - c0) * scale, kBins);
                bb[k].grow(a);
                bb[k].grow(b);
                bn[k]++;
            }
            float right_area[kBins];
            int right_n[kBins];
            Box3 acc;
            int n = 0;
            for (int k = kBins - 1; k > 0; k--) {
                acc.grow(bb[k]);
                n += bn[k];
                right_area[k] = n ? acc.half_area() : 0.0f;
                right_n[k] = n;
            }
            acc = Box3();
            n = 0;
            for (int k = 0; k < kBins - 1; k++) {
                acc.grow(bb[k]);
                n += bn[k];
                if (n == 0 || right_n[k + 1] == 0) continue;
                const float cost = acc.half_area() * float(n) + right_area[k + 1] * float(right_n[k + 1]);
                if (cost < best_cost) {
                    best_cost = cost;
                    best_axis = axis;
                    best_bin = k;
                }
            }
        }
        int mid = lo + (hi - lo) / 2;
        if (hi - lo <= kSweepMax) {
            // every split position of every axis (instances sorted by their boxes' centres), not 15 bin borders per axis: a scene
            // has thousands of instances, not millions of triangles, and the TLAS is walked by every ray (round 5: C4's TLAS steps
            // 144.5 -> 126.6 M per render, 2.65 -> 2.56 ms; profiles/r05/r05r_tlas_sweep_*.txt)
            const int n = hi - lo;
            std::vector<float> right_area((size_t)n);
            float sweep_cost = FLT_MAX;
            int sweep_axis = -1, sweep_at = -1;
            auto by_centre = [&](int axis) {
                std::sort(order.begin() + lo, order.begin() + hi, [&](uint32_t x, uint32_t y) {
                    float a[3], b[3], c[3], d[3];
                    box_of(inst[x], a, b);
                    box_of(inst[y], c, d);
                    const float cx = a[axis] + b[axis], cy = c[axis] + d[axis];
                    return cx < cy || (cx == cy && x < y);
                });
            };
            for (int axis = 0; axis < 3; axis++) {
                by_centre(axis);
                Box3 acc;
                for (int i = n - 1; i > 0; i--) {
                    float a[3], b[3];
                    box_of(inst[order[(size_t)(lo + i)]], a, b);
                    acc.grow(a);
                    acc.grow(b);
                    right_area[(size_t)i] = acc.half_area();
                }
                acc = Box3();
                for (int i = 0; i < n - 1; i++) {
                    float a[3], b[3];
                    box_of(inst[order[(size_t)(lo + i)]], a, b);
                    acc.grow(a);
                    acc.grow(b);
                    const float cost = acc.half_area() * float(i + 1) + right_area[(size_t)(i + 1)] * float(n - i - 1);
                    if (cost < sweep_cost) {
                        sweep_cost = cost;
                        sweep_axis = axis;
                        sweep_at = i + 1;
                    }
                }
            }
            if (sweep_axis >= 0) {
                if (sweep_axis != 2) by_centre(sweep_axis);
                mid = lo + sweep_at;
            }
        } else if (best_axis >= 0) {
            const float c0 = cbox.lo[best_axis], scale = float(kBins) / (cbox.hi[best_axis] - c0);
            auto it = std::partition(order.begin() + lo, order.begin() + hi, [&](uint32_t id) {
                float a[3], b[3];
                box_of(inst[id], a, b);
                const int k = bin_of((0.5f * (a[best_axis] + b[best_axis]) - c0) * scale, kBins);
                return k <= best_bin;
            });
            const int m = (int)(it - order.begin());
            if (m > lo && m < hi) mid = m;
        }
        const uint32_t l = build(lo, mid), r = build(mid, hi);
        RefTlasNode n;
        std::memset(&n, 0, sizeof n);
        n.leftRight = l + (r << 16);
        for (int k = 0; k < 3; k++) {
            n.aabbMin[k] = lo_(out[l].aabbMin[k], out[r].aabbMin[k]);
            n.aabbMax[k] = hi_(out[l].aabbMax[k], out[r].aabbMax[k]);
        }
        out.push_back(n);
        return (uint32_t)out.size() - 1u;
    }
};

bool build_tlas_sah(const std::vector<RefInstance>& inst, std::vector<RefTlasNode>& out, std::string& err)
{
    const int count = (int)inst.size();
    out.clear();
    out.emplace_back();
    std::memset(&out[0], 0, sizeof(RefTlasNode));
    if (count == 0) return true;
    if (count * 2 > 65535) {
        err = "TLAS: more than 32767 instances do not fit the reference's 16-bit child indices";
        return false;
    }
    for (int i = 0; i < count; i++) {
        RefTlasNode n;
        std::memset(&n, 0, sizeof n);
        n.aabbMin[0] = inst[(size_t)i].aabbMin.x; n.aabbMin[1] = inst[(size_t)i].aabbMin.y; n.aabbMin[2] = inst[(size_t)i].aabbMin.z;
        n.aabbMax[0] = inst[(size_t)i].aabbMax.x; n.aabbMax[1] = inst[(size_t)i].aabbMax.y; n.aabbMax[2] = inst[(size_t)i].aabbMax.z;
        n.blas = (uint32_t)i;
        out.push_back(n);
    }
    TlasSahBuilder b{inst, out, {}};
    b.order.resize((size_t)count);
    std::iota(b.order.begin(), b.order.end(), 0u);
    const uint32_t root = b.build(0, count);
    out[0] = out[root];
    return true;
}

}  // namespace

// Triangle -> GpuTriangleGeometry / GpuTriangleData (geometry_group3d.cpp:356-365)
static void split_triangles(RefScene& out)
{
    out.tri_geom.resize(out.triangles.size());
    out.tri_data.resize(out.triangles.size());
    for (size_t i = 0; i < out.triangles.size(); i++) {
        const RefTriangle& t = out.triangles[i];
        for (int k = 0; k < 3; k++) out.tri_geom[i].vertices[k] = t.vertices[k];
        RefTriData& d = out.tri_data[i];
        std::memset(&d, 0, sizeof d);
        d.n0[0] = t.normals[0].x; d.n0[1] = t.normals[0].y; d.n0[2] = t.normals[0].z;
        d.material_index = t.materialIndex;
        d.n1 = t.normals[1];
        d.n2 = t.normals[2];
        for (int k = 0; k < 3; k++) {
            d.uvs[k][0] = t.uvs[k][0];
            d.uvs[k][1] = t.uvs[k][1];
        }
    }
}

void SceneBuilder::begin()
{
    meshes_.clear();
    instances_.clear();
}

uint32_t SceneBuilder::add_mesh(const SurfaceView* surfaces, int32_t n_surfaces)
{
    // the de-indexing half of BuildBVH (bvh.cpp:192-213)
    PendingMesh pm;
    for (int32_t l = 0; l < n_surfaces; l++) {
        const SurfaceView& s = surfaces[l];
        for (int32_t i = 0; i + 2 < s.n_indices; i += 3) {
            RefTriangle t;
            std::memset(&t, 0, sizeof t);
            for (int j = 0; j < 3; j++) {
                const int32_t k = s.indices[i + j];
                t.vertices[j] = Vec4{s.vertices[k * 3], s.vertices[k * 3 + 1], s.vertices[k * 3 + 2], 1.0f};
                t.normals[j] = Vec4{s.normals[k * 3], s.normals[k * 3 + 1], s.normals[k * 3 + 2], 1.0f};
                t.uvs[j][0] = s.uvs[k * 2];
                t.uvs[j][1] = s.uvs[k * 2 + 1];
            }
            t.materialIndex = (uint32_t)l;
            // (v0 + v1 + v2) * 0.33333333f on all four components (bvh.cpp:210, vec.h:53-66)
            const Vec4 &a = t.vertices[0], &b = t.vertices[1], &c = t.vertices[2];
            t.centroid = Vec4{((a.x + b.x) + c.x) * 0.33333333f, ((a.y + b.y) + c.y) * 0.33333333f,
                              ((a.z + b.z) + c.z) * 0.33333333f, ((a.w + b.w) + c.w) * 0.33333333f};
            pm.tris.push_back(t);
        }
    }
    meshes_.push_back(std::move(pm));
    return (uint32_t)meshes_.size() - 1;
}

bool SceneBuilder::add_instance(uint32_t mesh_id, const float* t12, const int32_t* material_ids, int32_t n_ids)
{
    if (mesh_id >= meshes_.size()) return false;
    PendingInstance pi;
    pi.mesh = mesh_id;
    std::memcpy(pi.t12, t12, sizeof pi.t12);
    pi.mats[0] = pi.mats[1] = pi.mats[2] = 0;  // the reference leaves unset slots uninitialised (bvh.h:73-79)
    for (int i = 0; i < std::min(n_ids, 3); i++) pi.mats[i] = (uint32_t)material_ids[i];
    instances_.push_back(pi);
    return true;
}

bool SceneBuilder::set_instance_transform(uint32_t instance, const float* transform12)
{
    if (instance >= instances_.size() || !transform12) return false;
    std::memcpy(instances_[instance].t12, transform12, sizeof(float) * 12);
    return true;
}

// the world box BLASInstance::update_aabb (bvh.h:90-115) gives an instance whose BLAS root has the reference's box
ReachInst reach_instance(const float* t12, const ReachInst& mesh_root)
{
    ReachInst r = mesh_root;
    RefInstance tmp;
    std::memset(&tmp, 0, sizeof tmp);
    const Vec4 lo{mesh_root.root_lo[0], mesh_root.root_lo[1], mesh_root.root_lo[2], 1.0f};
    const Vec4 hi{mesh_root.root_hi[0], mesh_root.root_hi[1], mesh_root.root_hi[2], 1.0f};
    instance_record(t12, lo, hi, /*pad_box*/ false, tmp);
    r.lo[0] = tmp.aabbMin.x; r.lo[1] = tmp.aabbMin.y; r.lo[2] = tmp.aabbMin.z;
    r.hi[0] = tmp.aabbMax.x; r.hi[1] = tmp.aabbMax.y; r.hi[2] = tmp.aabbMax.z;
    return r;
}

// Native scenes: the world box of an instance from up to `n_boxes` boxes of its mesh's tree instead of the root's alone.  The
// reference's rule -- the box of the root box's eight transformed corners (bvh.h:90-115) -- grows by up to a factor of two in
// footprint when the instance is rotated; the native TLAS only has to contain the triangles, so it takes the union of the
// transformed boxes of a cut through the tree (the largest box opened first), each padded like the root's, and intersects that
// with the root's (InstanceCuts: the cut depends on the mesh alone, so it is made once per mesh).  What the reference's own box
// decides stays with the reach records (reach_instance).
struct InstanceCuts {
    struct Box {
        float c[3], e[3];   // centre, half extent
        float lo[3], hi[3];
    };
    const std::vector<RefBvhNode>& nodes;
    int n_boxes;
    std::vector<std::pair<uint32_t, std::vector<Box>>> per_root;   // (a scene has few meshes; the boxes side by side, not scattered over the node array)

    // 1 024 boxes per instance while the scene's instances x boxes stay below 2^18 box transforms (a TLAS
    // update is meant to take milliseconds); never fewer than 16
    static int boxes_for(size_t n_instances)
    {
        const int want = Tuning::instance_boxes;
        if (want <= 1) return want;
        const size_t share = ((size_t)1 << 18) / std::max<size_t>(n_instances, 1);
        return (int)std::max<size_t>(std::min<size_t>((size_t)want, share), std::min<size_t>((size_t)want, 16));
    }

    const std::vector<Box>& cut_of(uint32_t root)
    {
        for (const auto& e : per_root)
            if (e.first == root) return e.second;
        auto interior = [&](uint32_t n) {
            const RefBvhNode& b = nodes[n];
            return b.tri_count == 0 && !(b.left_child == 0 && b.right_child == 0) && b.left_child < nodes.size() && b.right_child < nodes.size();
        };
        auto area = [&](uint32_t n) {
            const RefBvhNode& b = nodes[n];
            const float dx = b.aabbMax.x - b.aabbMin.x, dy = b.aabbMax.y - b.aabbMin.y, dz = b.aabbMax.z - b.aabbMin.z;
            return dx * dy + dy * dz + dz * dx;
        };
        // the largest box opened first, until n_boxes (leaves and boxes that cannot be opened go straight to the cut)
        std::vector<uint32_t> cut;
        std::vector<std::pair<float, uint32_t>> open;   // a max-heap by area
        auto file = [&](uint32_t n) {
            if (interior(n)) {
                open.emplace_back(area(n), n);
                std::push_heap(open.begin(), open.end());
            } else {
                cut.push_back(n);
            }
        };
        file(root);
        while (!open.empty() && (int)(cut.size() + open.size()) < n_boxes) {
            std::pop_heap(open.begin(), open.end());
            const uint32_t n = open.back().second;
            open.pop_back();
            file(nodes[n].left_child);
            file(nodes[n].right_child);
        }
        for (const auto& o : open) cut.push_back(o.second);
        std::vector<Box> boxes;
        for (uint32_t n : cut) {
            const RefBvhNode& b = nodes[n];
            if (!(b.aabbMin.x <= b.aabbMax.x && b.aabbMin.y <= b.aabbMax.y && b.aabbMin.z <= b.aabbMax.z)) continue;   // an empty leaf's box
            Box x;
            const float lo[3] = {b.aabbMin.x, b.aabbMin.y, b.aabbMin.z}, hi[3] = {b.aabbMax.x, b.aabbMax.y, b.aabbMax.z};
            for (int k = 0; k < 3; k++) {
                x.lo[k] = lo[k];
                x.hi[k] = hi[k];
                x.c[k] = 0.5f * (lo[k] + hi[k]);
                x.e[k] = 0.5f * (hi[k] - lo[k]);
            }
            boxes.push_back(x);
        }
        per_root.emplace_back(root, std::move(boxes));
        return per_root.back().second;
    }

    // the cuts as flat arrays for the device's refit: boxes[6 * k] = centre.xyz, half extent.xyz; range[2 * i] = instance i's first
    // box and count (0: its box is the root's alone)
    void export_to(const std::vector<RefInstance>& instances, std::vector<float>& boxes, std::vector<uint32_t>& range)
    {
        boxes.clear();
        range.assign(instances.size() * 2, 0u);
        if (n_boxes <= 1) return;
        std::vector<std::pair<uint32_t, uint32_t>> placed(per_root.size(), {0u, 0u});
        for (size_t r = 0; r < per_root.size(); r++) {
            const std::vector<Box>& cut = per_root[r].second;
            placed[r] = {(uint32_t)(boxes.size() / 6), cut.size() < 2 ? 0u : (uint32_t)cut.size()};
            if (cut.size() < 2) continue;
            for (const Box& x : cut) {
                boxes.insert(boxes.end(), x.c, x.c + 3);
                boxes.insert(boxes.end(), x.e, x.e + 3);
            }
        }
        for (size_t i = 0; i < instances.size(); i++)
            for (size_t r = 0; r < per_root.size(); r++)
                if (per_root[r].first == instances[i].blas_index) {
                    range[2 * i] = placed[r].first;
                    range[2 * i + 1] = placed[r].second;
                    break;
                }
    }

    void tighten(RefInstance& inst)
    {
        if (n_boxes <= 1 || inst.blas_index >= nodes.size()) return;
        const std::vector<Box>& cut = cut_of(inst.blas_index);
        if (cut.size() < 2) return;
        float lo[3] = {1e34f, 1e34f, 1e34f}, hi[3] = {-1e34f, -1e34f, -1e34f};
        const float* m = inst.transform;   // column-major 4 x 4
        if (m[3] == 0.0f && m[7] == 0.0f && m[11] == 0.0f && m[15] == 1.0f) {
            // the image of a box under an affine map: centre' -+ |M| half-extent (9 + 9 products instead of eight corners x 16); a
            // few ulp of the coordinates away from the corner rule's box: twice its padding, on the union
            float am[9];
            for (int r = 0; r < 3; r++)
                for (int k = 0; k < 3; k++) am[k * 3 + r] = std::fabs(m[k * 4 + r]);
            for (const Box& x : cut)
                for (int r = 0; r < 3; r++) {
                    const float wc = m[r] * x.c[0] + m[4 + r] * x.c[1] + m[8 + r] * x.c[2] + m[12 + r];
                    const float we = am[r] * x.e[0] + am[3 + r] * x.e[1] + am[6 + r] * x.e[2];
                    lo[r] = std::min(lo[r], wc - we);
                    hi[r] = std::max(hi[r], wc + we);
                }
            float big = 0.0f;
            for (int r = 0; r < 3; r++) big = std::max(big, std::max(std::fabs(lo[r]), std::fabs(hi[r])));
            const float pad = big * 4e-6f;
            for (int r = 0; r < 3; r++) {
                lo[r] -= pad;
                hi[r] += pad;
            }
        } else {
            for (const Box& x : cut) {
                Vec4 l, h;
                instance_world_box(inst.transform, Vec4{x.lo[0], x.lo[1], x.lo[2], 1.0f}, Vec4{x.hi[0], x.hi[1], x.hi[2], 1.0f}, /*pad_box*/ true, l, h);
                lo[0] = std::min(lo[0], l.x); lo[1] = std::min(lo[1], l.y); lo[2] = std::min(lo[2], l.z);
                hi[0] = std::max(hi[0], h.x); hi[1] = std::max(hi[1], h.y); hi[2] = std::max(hi[2], h.z);
            }
        }
        if (!(lo[0] <= hi[0] && lo[1] <= hi[1] && lo[2] <= hi[2])) return;
        inst.aabbMin = Vec4{std::max(inst.aabbMin.x, lo[0]), std::max(inst.aabbMin.y, lo[1]), std::max(inst.aabbMin.z, lo[2]), inst.aabbMin.w};
        inst.aabbMax = Vec4{std::min(inst.aabbMax.x, hi[0]), std::min(inst.aabbMax.y, hi[1]), std::min(inst.aabbMax.z, hi[2]), inst.aabbMax.w};
    }
};

// instances (geometry_group3d.cpp:322-341) and TLAS::build (bvh.cpp:264-317) over the BLASes already in `out`
bool SceneBuilder::rebuild_instances(BuildMode mode, RefScene& out, std::string& err)
{
    if (out.mesh_roots.size() != meshes_.size()) {
        err = "instances can only be rebuilt over the scene of the last commit";
        return false;
    }
    out.instances.clear();
    out.tlas_nodes.clear();
    InstanceCuts cuts{out.bvh_nodes, InstanceCuts::boxes_for(instances_.size()), {}};
    for (const PendingInstance& pi : instances_) {
        RefInstance inst;
        std::memset(&inst, 0, sizeof inst);
        inst.blas_index = out.mesh_roots[pi.mesh];
        for (int k = 0; k < 3; k++) inst.material[k] = pi.mats[k];
        const RefBvhNode& root = out.bvh_nodes[inst.blas_index];
        instance_record(pi.t12, root.aabbMin, root.aabbMax, is_native(mode), inst);  // jpt_instance_math.h
        if (is_native(mode)) cuts.tighten(inst);
        out.instances.push_back(inst);
    }
    out.inst_cut_boxes.clear();
    out.inst_cut_range.clear();
    if (is_native(mode)) cuts.export_to(out.instances, out.inst_cut_boxes, out.inst_cut_range);
    out.reach_inst.clear();
    if (mode == BuildMode::Sah && out.mesh_ref_root.size() == meshes_.size())
        for (const PendingInstance& pi : instances_) out.reach_inst.push_back(reach_instance(pi.t12, out.mesh_ref_root[pi.mesh]));
    if (is_native(mode) ? !build_tlas_sah(out.instances, out.tlas_nodes, err) : !build_tlas(out.instances, out.tlas_nodes, err))
        return false;
    // the shadow's instance level: the records and the TLAS the reference itself would make (bvh.h:81-115, bvh.cpp:264-317)
    ExactShadow& x = out.exact;
    x.instances.clear();
    x.tlas_nodes.clear();
    x.valid = false;
    if (mode == BuildMode::Sah && x.mesh_roots.size() == meshes_.size()) {
        for (const PendingInstance& pi : instances_) {
            RefInstance inst;
            std::memset(&inst, 0, sizeof inst);
            inst.blas_index = x.mesh_roots[pi.mesh];
            for (int k = 0; k < 3; k++) inst.material[k] = pi.mats[k];
            const RefBvhNode& root = x.bvh_nodes[inst.blas_index];
            instance_record(pi.t12, root.aabbMin, root.aabbMax, /*pad_box*/ false, inst);
            x.instances.push_back(inst);
        }
        std::string xerr;
        x.valid = build_tlas(x.instances, x.tlas_nodes, xerr);
        x.tri_geom.resize(out.triangles.size());
        x.tri_native.resize(out.triangles.size(), 0xffffffffu);
        x.finish(out.triangles.size());
    }
    return true;
}

bool SceneBuilder::commit(BuildMode mode, RefScene& out, std::string& err)
{
    out.clear();
    // per unique mesh: append triangles, build its BLAS into the shared node array (geometry_group3d.cpp:308-313)
    for (const PendingMesh& pm : meshes_) {
        const int start = (int)out.triangles.size();
        out.triangles.insert(out.triangles.end(), pm.tris.begin(), pm.tris.end());
        const int end = (int)out.triangles.size();
        uint32_t root = 0;
        if (end > start) {
            if (mode == BuildMode::ReferenceExact) {
                ExactBlasBuilder b{out.bvh_nodes, out.triangles};
                root = b.build(start, end);
            } else {
                // reach records: the reference's own builder runs on a copy of the mesh (beside the native build), only
                // to learn which leaf box holds which triangle and what the root box is
                std::vector<ReachTri> by_original;
                ReachInst ref_root;
                std::vector<RefBvhNode> ref_nodes;
                std::vector<uint32_t> ref_order;
                std::future<void> reach;
                if (mode == BuildMode::Sah)
                    reach = std::async(std::launch::async, [&] { reference_leaf_boxes(pm.tris, by_original, ref_root, ref_nodes, ref_order); });
                SahBlasBuilder b{out.bvh_nodes, out.triangles};
                b.prepare(start, end);
                root = b.build(0, end - start);
                b.apply_order(start, end);
                if (mode == BuildMode::Sah) {
                    reach.get();
                    out.reach_tri.resize((size_t)end);
                    for (int i = 0; i < end - start; i++) out.reach_tri[(size_t)(start + i)] = by_original[b.order[(size_t)i]];
                    out.mesh_ref_root.resize(out.mesh_roots.size() + 1);
                    out.mesh_ref_root.back() = ref_root;
                    // the shadow: the reference's tree of this mesh, its triangle order, and where each of its triangles
                    // sits in the native order
                    ExactShadow& x = out.exact;
                    const uint32_t noff = (uint32_t)x.bvh_nodes.size();
                    for (RefBvhNode n : ref_nodes) {
                        if (n.tri_count == 0) {
                            n.left_child += noff;
                            n.right_child += noff;
                        } else {
                            n.first_tri_index += (uint32_t)start;
                        }
                        x.bvh_nodes.push_back(n);
                    }
                    std::vector<uint32_t> native_pos((size_t)(end - start));
                    for (int i = 0; i < end - start; i++) native_pos[b.order[(size_t)i]] = (uint32_t)(start + i);
                    x.tri_geom.resize((size_t)end);
                    x.tri_native.resize((size_t)end, 0xffffffffu);
                    for (int k = 0; k < end - start; k++) {
                        const RefTriangle& t = pm.tris[ref_order[(size_t)k]];
                        for (int j = 0; j < 3; j++) x.tri_geom[(size_t)(start + k)].vertices[j] = t.vertices[j];
                        x.tri_native[(size_t)(start + k)] = native_pos[ref_order[(size_t)k]];
                    }
                    x.mesh_roots.push_back(noff);
                }
            }
        } else {
            // an empty mesh: resolved after the loop
            if (mode == BuildMode::Sah) {
                out.mesh_ref_root.resize(out.mesh_roots.size() + 1);
                std::memset(&out.mesh_ref_root.back(), 0, sizeof(ReachInst));
                out.exact.mesh_roots.push_back(0u);
            }
        }
        out.mesh_roots.push_back(root);
    }
    // BuildBVH on a mesh without triangles pushes no node and returns 0 (bvh.cpp:111-112), so in the reference an instance of
    // such a mesh gets blas_index 0 (geometry_group3d.cpp:311,325) and SHOWS WHATEVER TREE STARTS AT NODE 0 -- the first mesh
    // that has triangles -- under its own transform and materials.  Reproduced: an empty mesh stands for that mesh in every
    // table (so the node arrays equal the reference's byte for byte and all routes render what the reference renders).  Only
    // when NO mesh has a triangle -- the reference's shader would then read a node past the end of an empty buffer -- a
    // single empty leaf is kept so that every root names a valid node.
    {
        size_t first = meshes_.size();
        for (size_t m = 0; m < meshes_.size(); m++)
            if (!meshes_[m].tris.empty()) {
                first = m;
                break;
            }
        const bool shadow = mode == BuildMode::Sah;
        if (first == meshes_.size() && !meshes_.empty()) {
            RefBvhNode n;
            std::memset(&n, 0, sizeof n);
            out.bvh_nodes.push_back(n);     // mesh_roots are all 0 already
            if (shadow) out.exact.bvh_nodes.push_back(n);
        } else {
            for (size_t m = 0; m < meshes_.size(); m++)
                if (meshes_[m].tris.empty()) {
                    out.mesh_roots[m] = out.mesh_roots[first];
                    if (shadow) {
                        out.mesh_ref_root[m] = out.mesh_ref_root[first];
                        out.exact.mesh_roots[m] = out.exact.mesh_roots[first];
                    }
                }
        }
    }
    if (!rebuild_instances(mode, out, err)) return false;
    split_triangles(out);
    return true;
}

// ------------------------------------------------------------------------------------------------
// route (i) on the native tree: native SAH trees + reach records from uploaded reference-layout arrays

namespace {

inline bool box_inside(const float* lo, const float* hi, const float* plo, const float* phi)
{
    // (written so that a NaN anywhere answers "no")
    return lo[0] >= plo[0] && lo[1] >= plo[1] && lo[2] >= plo[2] && hi[0] <= phi[0] && hi[1] <= phi[1] && hi[2] <= phi[2];
}

// transform * inverse_transform == identity, to float accuracy?  (column-major 4x4, as utils.h:15-37 writes them)
bool transforms_belong_together(const RefInstance& in)
{
    const float *a = in.transform, *b = in.inverse_transform;
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) {
            double p = 0.0, mag = 0.0;
            for (int k = 0; k < 4; k++) {
                const double t = (double)a[k * 4 + r] * (double)b[c * 4 + k];
                p += t;
                mag += std::fabs(t);
            }
            const double want = r == c ? 1.0 : 0.0;
            if (!(std::fabs(p - want) <= 1e-4 * (1.0 + mag))) return false;
        }
    return true;
}

}  // namespace

bool native_instances_from_uploaded(const std::vector<RefInstance>& up_inst, const std::vector<RefTlasNode>& up_tlas, RefScene& out,
                                    std::string& why)
{
    const size_t ni = up_inst.size();
    std::vector<RefInstance> instances;
    std::vector<ReachInst> reach;
    std::vector<uint32_t> up_index(ni);
    if (ni > 0) {
        if (up_tlas.empty()) {
            why = "instances without a TLAS";
            return false;
        }
        // the TLAS leaf of every instance (ray_trace_tlas starts at slot 0 and tests the boxes of the children it pushes,
        // main.glsl:307-350): each instance in exactly one leaf, no node reachable twice, child boxes inside their parent's
        // (the root's own box is never tested)
        constexpr uint32_t kNone = 0xffffffffu;
        std::vector<uint32_t> leaf_of(ni, kNone);
        std::vector<uint8_t> seen(up_tlas.size(), 0);
        std::vector<std::pair<uint32_t, uint32_t>> todo;  // (node, parent)
        todo.emplace_back(0u, kNone);
        while (!todo.empty()) {
            const uint32_t n = todo.back().first, parent = todo.back().second;
            todo.pop_back();
            if (seen[n]) {
                why = "a TLAS node is reachable twice";
                return false;
            }
            seen[n] = 1;
            const RefTlasNode& node = up_tlas[n];
            if (parent != kNone && parent != 0u && !box_inside(node.aabbMin, node.aabbMax, up_tlas[parent].aabbMin, up_tlas[parent].aabbMax)) {
                why = "TLAS boxes are not nested";
                return false;
            }
            if (node.leftRight == 0u) {
                if (node.blas >= ni) {
                    why = "a TLAS leaf names an instance that does not exist";
                    return false;
                }
                if (leaf_of[node.blas] != kNone) {
                    why = "an instance is in two TLAS leaves";
                    return false;
                }
                leaf_of[node.blas] = n;
                continue;
            }
            const uint32_t l = node.leftRight & 0xffffu, r = node.leftRight >> 16;
            if (l >= up_tlas.size() || r >= up_tlas.size()) {
                why = "TLAS child index out of range";
                return false;
            }
            todo.emplace_back(r, n);
            todo.emplace_back(l, n);
        }
        instances.resize(ni);
        reach.resize(ni);
        InstanceCuts cuts{out.bvh_nodes, InstanceCuts::boxes_for(ni), {}};
        for (size_t i = 0; i < ni; i++) {
            if (leaf_of[i] == kNone) {
                why = "an instance is in no TLAS leaf";
                return false;
            }
            const RefInstance& in = up_inst[i];
            size_t mesh = out.up_mesh_root.size();
            for (size_t m = 0; m < out.up_mesh_root.size(); m++)
                if (out.up_mesh_root[m] == in.blas_index) {
                    mesh = m;
                    break;
                }
            if (mesh == out.up_mesh_root.size()) {
                why = "an instance names a BLAS root the uploaded scene did not have";
                return false;
            }
            if (!transforms_belong_together(in)) {
                why = "an instance's inverse_transform is not the inverse of its transform";
                return false;
            }
            up_index[i] = in.blas_index;
            RefInstance n = in;  // matrices and material slots as uploaded (the device reads exactly these)
            n.blas_index = out.mesh_roots[mesh];
            const RefBvhNode& root = out.bvh_nodes[n.blas_index];
            instance_world_box(n.transform, root.aabbMin, root.aabbMax, /*pad_box*/ true, n.aabbMin, n.aabbMax);
            cuts.tighten(n);
            instances[i] = n;
            // the reach record: the box ray_trace_tlas tests before it enters the instance is its TLAS leaf's
            ReachInst r = out.mesh_ref_root[mesh];
            const RefTlasNode& leaf = up_tlas[leaf_of[i]];
            for (int k = 0; k < 3; k++) {
                r.lo[k] = leaf.aabbMin[k];
                r.hi[k] = leaf.aabbMax[k];
            }
            reach[i] = r;
        }
    }
    std::vector<RefTlasNode> tlas;
    std::string err;
    if (!build_tlas_sah(instances, tlas, err)) {
        why = err;
        return false;
    }
    out.instances = std::move(instances);
    out.reach_inst = std::move(reach);
    out.tlas_nodes = std::move(tlas);
    out.up_blas_index = std::move(up_index);
    // the shadow's instance level is the uploaded one
    out.exact.instances = up_inst;
    out.exact.tlas_nodes = up_tlas;
    if (out.exact.valid) out.exact.finish(out.triangles.size());   // (a TLAS update of an uploaded scene; the upload itself finishes below)
    return true;
}

bool native_from_uploaded(const RefScene& up, RefScene& out, std::string& why)
{
    out.clear();
    constexpr uint32_t kNone = 0xffffffffu;
    const size_t nn = up.bvh_nodes.size(), nt = up.tri_geom.size();
    if (up.tri_data.size() != nt) {
        why = "triangle geometry/data counts differ";
        return false;
    }
    std::vector<uint8_t> node_seen(nn, 0);
    std::vector<uint32_t> tri_leaf(nt, kNone);  // the uploaded leaf that holds each triangle
    std::vector<uint32_t> up_to_native(nt, kNone);
    for (const RefInstance& in : up.instances) {
        const uint32_t up_root = in.blas_index;
        if (std::find(out.up_mesh_root.begin(), out.up_mesh_root.end(), up_root) != out.up_mesh_root.end()) continue;
        if (up_root >= nn) {
            why = "instance root index out of range";
            return false;
        }
        // the triangles under this root, in the order ray_trace_blas would meet the leaves going left first
        std::vector<uint32_t> mesh_tris;
        std::vector<std::pair<uint32_t, uint32_t>> todo;  // (node, parent)
        todo.emplace_back(up_root, kNone);
        while (!todo.empty()) {
            const uint32_t n = todo.back().first, parent = todo.back().second;
            todo.pop_back();
            if (node_seen[n]) {
                why = "a BVH node is reachable twice (shared subtree or cycle)";
                return false;
            }
            node_seen[n] = 1;
            const RefBvhNode& node = up.bvh_nodes[n];
            // a leaf's box must lie inside the box of every ancestor that is tested on the way down (all but the root)
            if (parent != kNone && parent != up_root &&
                !box_inside(&node.aabbMin.x, &node.aabbMax.x, &up.bvh_nodes[parent].aabbMin.x, &up.bvh_nodes[parent].aabbMax.x)) {
                why = "BVH boxes are not nested";
                return false;
            }
            if (node.tri_count > 0) {
                if ((size_t)node.first_tri_index + node.tri_count > nt) {
                    why = "BVH leaf triangle range out of bounds";
                    return false;
                }
                for (uint32_t k = 0; k < node.tri_count; k++) {
                    const uint32_t t = node.first_tri_index + k;
                    if (tri_leaf[t] != kNone) {
                        why = "a triangle is in two leaves";
                        return false;
                    }
                    tri_leaf[t] = n;
                    mesh_tris.push_back(t);
                }
                continue;
            }
            if (node.left_child == 0 && node.right_child == 0) continue;  // a leaf without triangles
            if (node.left_child >= nn || node.right_child >= nn) {
                why = "BVH child index out of range";
                return false;
            }
            todo.emplace_back(node.right_child, n);
            todo.emplace_back(node.left_child, n);
        }
        const int start = (int)out.triangles.size();
        for (uint32_t t : mesh_tris) {
            const RefTriGeometry& g = up.tri_geom[t];
            const RefTriData& d = up.tri_data[t];
            RefTriangle tri;
            std::memset(&tri, 0, sizeof tri);
            for (int j = 0; j < 3; j++) tri.vertices[j] = g.vertices[j];
            tri.normals[0] = Vec4{d.n0[0], d.n0[1], d.n0[2], 1.0f};
            tri.normals[1] = d.n1;
            tri.normals[2] = d.n2;
            for (int j = 0; j < 3; j++) {
                tri.uvs[j][0] = d.uvs[j][0];
                tri.uvs[j][1] = d.uvs[j][1];
            }
            tri.materialIndex = d.material_index;
            const Vec4 &a = tri.vertices[0], &b = tri.vertices[1], &c = tri.vertices[2];
            tri.centroid = Vec4{((a.x + b.x) + c.x) * 0.33333333f, ((a.y + b.y) + c.y) * 0.33333333f,
                                ((a.z + b.z) + c.z) * 0.33333333f, ((a.w + b.w) + c.w) * 0.33333333f};
            out.triangles.push_back(tri);
        }
        const int end = (int)out.triangles.size();
        uint32_t root = 0;
        if (end > start) {
            SahBlasBuilder b{out.bvh_nodes, out.triangles};
            b.prepare(start, end);
            root = b.build(0, end - start);
            b.apply_order(start, end);
            out.reach_tri.resize((size_t)end);
            for (int i = 0; i < end - start; i++) {
                up_to_native[mesh_tris[b.order[(size_t)i]]] = (uint32_t)(start + i);
                const uint32_t leaf = tri_leaf[mesh_tris[b.order[(size_t)i]]];
                const RefBvhNode& ln = up.bvh_nodes[leaf];
                ReachTri r;
                r.lo[0] = ln.aabbMin.x; r.lo[1] = ln.aabbMin.y; r.lo[2] = ln.aabbMin.z;
                r.hi[0] = ln.aabbMax.x; r.hi[1] = ln.aabbMax.y; r.hi[2] = ln.aabbMax.z;
                r.always = leaf == up_root ? 1u : 0u;  // the root is popped without a box test (main.glsl:272-283)
                r._pad = 0;
                out.reach_tri[(size_t)(start + i)] = r;
            }
        } else {
            root = (uint32_t)out.bvh_nodes.size();  // nothing under this root: a valid empty leaf (as SceneBuilder::commit)
            RefBvhNode n;
            std::memset(&n, 0, sizeof n);
            out.bvh_nodes.push_back(n);
        }
        ReachInst rr;
        std::memset(&rr, 0, sizeof rr);
        const RefBvhNode& ur = up.bvh_nodes[up_root];
        rr.root_lo[0] = ur.aabbMin.x; rr.root_lo[1] = ur.aabbMin.y; rr.root_lo[2] = ur.aabbMin.z;
        rr.root_hi[0] = ur.aabbMax.x; rr.root_hi[1] = ur.aabbMax.y; rr.root_hi[2] = ur.aabbMax.z;
        out.mesh_ref_root.push_back(rr);
        out.mesh_roots.push_back(root);
        out.up_mesh_root.push_back(up_root);
    }
    out.reach_tri.resize(out.triangles.size());
    if (!native_instances_from_uploaded(up.instances, up.tlas_nodes, out, why)) return false;
    // the shadow: the uploaded trees themselves, and where each uploaded triangle sits in the native order
    out.exact.bvh_nodes = up.bvh_nodes;
    out.exact.tri_geom = up.tri_geom;
    out.exact.tri_native = std::move(up_to_native);
    out.exact.mesh_roots = out.up_mesh_root;
    out.exact.valid = true;
    out.exact.finish(out.triangles.size());
    split_triangles(out);
    out.materials = up.materials;
    out.textures = up.textures;
    out.tex_res = up.tex_res;
    out.n_layers = up.n_layers;
    return true;
}

// ------------------------------------------------------------------------------------------------
// flatten: reference layout -> 64-byte two-child records

namespace {

struct Flattener {
    const RefScene& ref;
    WideScene& out;
    std::string err;
    // Uploaded arrays are only range-checked when they get here, so the two recursions below must end on ANY input:
    // a node is expanded once (kDone: a second parent gets the record made for the first, so a shared subtree costs
    // its size, not 2^depth), a node met again while its own subtree is still open is a cycle (kOpen: an error --
    // main.glsl:270-350 would never leave it), and no chain is followed deeper than kMaxDepth frames of the host
    // stack.  That loses nothing: the kernels' traversal stack holds fewer than 100 entries and a tree that could need
    // more is refused after the flatten (compute_stack_need, jpt_capi.cpp upload_scene), so a tree this deep was never
    // going to be accepted; the bound only keeps collapse4 / need2 / need4, which recurse over what is made here, shallow.
    enum : uint8_t { kUnseen = 0, kOpen = 1, kDone = 2 };
    static constexpr uint32_t kMaxDepth = 512u;
    std::vector<uint8_t> blas_state, tlas_state;
    std::vector<uint8_t> blas_parented, tlas_parented;   // the node has been met as somebody's child
    std::vector<int32_t> blas_memo, tlas_memo;

    Flattener(const RefScene& r, WideScene& o) : ref(r), out(o) {}

    int32_t leaf_ref(uint32_t first, uint32_t count) { return ~(int32_t)((first & kLeafFirstMask) | ((count - 1) << kLeafCountShift)); }

    // returns the child reference that stands for reference node `ni`
    int32_t blas_child(uint32_t ni, uint32_t depth)
    {
        if (!err.empty()) return ~0;
        if (ni >= ref.bvh_nodes.size()) {
            err = "BVH child index out of range";
            return ~0;
        }
        if (blas_state.empty()) {
            blas_state.assign(ref.bvh_nodes.size(), kUnseen);
            blas_memo.assign(ref.bvh_nodes.size(), 0);
            blas_parented.assign(ref.bvh_nodes.size(), 0);
        }
        // The memo bounds THIS flatten; the walk on the device has no visited set and follows every path (main.glsl:270-303), so a
        // node WITH CHILDREN under two parents -- a DAG -- costs the walk its number of root-to-leaf paths: 2^60 box visits for a
        // chain of 60 nodes whose two children are the same next node, i.e. a render that never ends.  Refused.  What
        // stays legal is sharing that cannot multiply paths: several instances naming one BLAS root (depth 1: not a parent), and
        // leaves under several parents.
        if (blas_state[ni] == kOpen) {
            err = "BVH has a cycle";
            return ~0;
        }
        if (depth > 1) {
            const RefBvhNode& n = ref.bvh_nodes[ni];
            const bool interior = !(n.tri_count > 0 || (n.left_child == 0 && n.right_child == 0));
            if (interior && blas_parented[ni]) {
                err = "a BVH node with children is reachable from two parents (the traversal follows every path: a shared subtree is walked once per path)";
                return ~0;
            }
            blas_parented[ni] = 1;
        }
        if (blas_state[ni] == kDone) return blas_memo[ni];
        if (blas_state[ni] == kOpen) {
            err = "BVH has a cycle";
            return ~0;
        }
        if (depth > kMaxDepth) {
            err = "BVH is deeper than 512 levels";
            return ~0;
        }
        blas_state[ni] = kOpen;
        const int32_t r = blas_expand(ni, depth);
        blas_state[ni] = kDone;
        blas_memo[ni] = r;
        return r;
    }

    int32_t blas_expand(uint32_t ni, uint32_t depth)
    {
        out.max_blas_depth = std::max(out.max_blas_depth, depth);
        const RefBvhNode& n = ref.bvh_nodes[ni];
        if (n.tri_count > 0 || (n.left_child == 0 && n.right_child == 0)) {
            uint32_t first = n.first_tri_index, count = n.tri_count;
            if (count == 0) return make_empty_leaf();
            if ((uint64_t)first + count > (uint64_t)kLeafFirstMask || (uint64_t)first + count > ref.tri_geom.size()) {
                err = (uint64_t)first + count > ref.tri_geom.size() ? "BVH leaf triangle range out of bounds"
                                                                     : "too many triangles for the 25-bit leaf reference";
                return ~0;
            }
            if (count <= (uint32_t)kMaxLeafTris) return leaf_ref(first, count);
            if (count > (uint32_t)kMaxLeafTris * kMaxDepth) {
                err = "a BVH leaf holds more than 32768 triangles";
                return ~0;
            }
            // oversized leaf (the reference makes them when all three SAH axes are rejected): chain of
            // records whose two boxes both equal the leaf box.  With equal entry distances the walk takes the
            // else branch of main.glsl:293-299 and visits the RIGHT child first, so the first 64 triangles go
            // right and the rest left: the triangles are tested in index order, as the leaf loop of
            // main.glsl:280-283 does (ties, t == hitInfo.t, go to the later triangle).  A loop, not a recursion:
            // the count is the caller's word and a chain may be 2^25 / 64 records long.
            const int32_t head = (int32_t)out.blas_nodes.size();
            while (true) {
                const int32_t self = (int32_t)out.blas_nodes.size();
                WideNode w;
                std::memset(&w, 0, sizeof w);
                set_box(w.lmin, w.lmax, n);
                set_box(w.rmin, w.rmax, n);
                w.right = leaf_ref(first, kMaxLeafTris);
                first += kMaxLeafTris;
                count -= kMaxLeafTris;
                w.left = count <= (uint32_t)kMaxLeafTris ? leaf_ref(first, count) : self + 1;
                out.blas_nodes.push_back(w);
                if (count <= (uint32_t)kMaxLeafTris) break;
            }
            return head;
        }
        const int32_t self = (int32_t)out.blas_nodes.size();
        out.blas_nodes.emplace_back();
        WideNode w;
        std::memset(&w, 0, sizeof w);
        if (n.left_child >= ref.bvh_nodes.size() || n.right_child >= ref.bvh_nodes.size()) {
            err = "BVH child index out of range";
            return ~0;
        }
        set_box(w.lmin, w.lmax, ref.bvh_nodes[n.left_child]);
        set_box(w.rmin, w.rmax, ref.bvh_nodes[n.right_child]);
        w.left = blas_child(n.left_child, depth + 1);
        w.right = blas_child(n.right_child, depth + 1);
        out.blas_nodes[self] = w;
        return self;
    }

    int32_t make_empty_leaf()
    {
        // a leaf with no triangles: reference it as an internal record whose boxes never hit
        const int32_t self = (int32_t)out.blas_nodes.size();
        WideNode w;
        std::memset(&w, 0, sizeof w);
        for (int k = 0; k < 3; k++) {
            w.lmin[k] = w.rmin[k] = FLT_MAX;
            w.lmax[k] = w.rmax[k] = -FLT_MAX;
        }
        w.left = w.right = self;
        out.blas_nodes.push_back(w);
        return self;
    }

    static void set_box(float* lo, float* hi, const RefBvhNode& n)
    {
        lo[0] = n.aabbMin.x; lo[1] = n.aabbMin.y; lo[2] = n.aabbMin.z;
        hi[0] = n.aabbMax.x; hi[1] = n.aabbMax.y; hi[2] = n.aabbMax.z;
    }

    int32_t tlas_child(uint32_t ni, uint32_t depth)
    {
        if (!err.empty()) return ~0;
        if (ni >= ref.tlas_nodes.size()) {
            err = "TLAS child index out of range";
            return ~0;
        }
        if (tlas_state.empty()) {
            tlas_state.assign(ref.tlas_nodes.size(), kUnseen);
            tlas_memo.assign(ref.tlas_nodes.size(), 0);
            tlas_parented.assign(ref.tlas_nodes.size(), 0);
        }
        if (tlas_state[ni] == kOpen) {
            err = "TLAS has a cycle";
            return ~0;
        }
        if (depth > 1) {   // (as for the BVH: a TLAS node with children under two parents multiplies the walk's paths)
            if (ref.tlas_nodes[ni].leftRight != 0u && tlas_parented[ni]) {
                err = "a TLAS node with children is reachable from two parents (the traversal follows every path: a shared subtree is walked once per path)";
                return ~0;
            }
            tlas_parented[ni] = 1;
        }
        if (tlas_state[ni] == kDone) return tlas_memo[ni];
        if (tlas_state[ni] == kOpen) {
            err = "TLAS has a cycle";
            return ~0;
        }
        if (depth > kMaxDepth) {
            err = "TLAS is deeper than 512 levels";
            return ~0;
        }
        tlas_state[ni] = kOpen;
        const int32_t r = tlas_expand(ni, depth);
        tlas_state[ni] = kDone;
        tlas_memo[ni] = r;
        return r;
    }

    int32_t tlas_expand(uint32_t ni, uint32_t depth)
    {
        out.max_tlas_depth = std::max(out.max_tlas_depth, depth);
        const RefTlasNode& n = ref.tlas_nodes[ni];
        if (n.leftRight == 0) {
            if (n.blas >= ref.instances.size()) {
                err = "TLAS leaf names an instance that does not exist";
                return ~0;
            }
            return ~(int32_t)n.blas;
        }
        const uint32_t l = n.leftRight & 0xFFFFu, r = n.leftRight >> 16;
        if (l >= ref.tlas_nodes.size() || r >= ref.tlas_nodes.size()) {
            err = "TLAS child index out of range";
            return ~0;
        }
        const int32_t self = (int32_t)out.tlas_nodes.size();
        out.tlas_nodes.emplace_back();
        WideNode w;
        std::memset(&w, 0, sizeof w);
        for (int k = 0; k < 3; k++) {
            w.lmin[k] = ref.tlas_nodes[l].aabbMin[k];
            w.lmax[k] = ref.tlas_nodes[l].aabbMax[k];
            w.rmin[k] = ref.tlas_nodes[r].aabbMin[k];
            w.rmax[k] = ref.tlas_nodes[r].aabbMax[k];
        }
        w.left = tlas_child(l, depth + 1);
        w.right = tlas_child(r, depth + 1);
        out.tlas_nodes[self] = w;
        return self;
    }
};

}  // namespace

bool flatten(const RefScene& ref, WideScene& out, std::string& err)
{
    out = WideScene();
    Flattener f(ref, out);
    // triangles: v0 + the two Moller-Trumbore edges (same subtractions as main.glsl:231-232)
    out.tris.resize(ref.tri_geom.size());
    for (size_t i = 0; i < ref.tri_geom.size(); i++) {
        const RefTriGeometry& g = ref.tri_geom[i];
        WideTri& t = out.tris[i];
        std::memset(&t, 0, sizeof t);
        t.v0[0] = g.vertices[0].x; t.v0[1] = g.vertices[0].y; t.v0[2] = g.vertices[0].z;
        t.e1[0] = g.vertices[1].x - g.vertices[0].x; t.e1[1] = g.vertices[1].y - g.vertices[0].y; t.e1[2] = g.vertices[1].z - g.vertices[0].z;
        t.e2[0] = g.vertices[2].x - g.vertices[0].x; t.e2[1] = g.vertices[2].y - g.vertices[0].y; t.e2[2] = g.vertices[2].z - g.vertices[0].z;
        // cross(e1, e2), each product and difference rounded on its own like the shader's (no contraction: Makefile)
        t.nx = t.e1[1] * t.e2[2] - t.e1[2] * t.e2[1];
        t.ny = t.e1[2] * t.e2[0] - t.e1[0] * t.e2[2];
        t.nz = t.e1[0] * t.e2[1] - t.e1[1] * t.e2[0];
    }
    // one BLAS tree per distinct root referenced by an instance
    std::vector<std::pair<uint32_t, int32_t>> root_refs;
    out.instances.resize(ref.instances.size());
    for (size_t i = 0; i < ref.instances.size(); i++) {
        const RefInstance& ri = ref.instances[i];
        int32_t rr = 0;
        bool found = false;
        for (auto& p : root_refs)
            if (p.first == ri.blas_index) {
                rr = p.second;
                found = true;
                break;
            }
        if (!found) {
            rr = f.blas_child(ri.blas_index, 1);
            if (!f.err.empty()) {
                err = f.err;
                return false;
            }
            root_refs.emplace_back(ri.blas_index, rr);
        }
        WideInstance& wi = out.instances[i];
        std::memset(&wi, 0, sizeof wi);
        const float* m = ri.inverse_transform;
        for (int c = 0; c < 4; c++)
            for (int r = 0; r < 3; r++) wi.inv[c * 3 + r] = m[c * 4 + r];
        wi.root = rr;
    }
    if (ref.tlas_nodes.empty() || ref.instances.empty()) {
        out.tlas_root = 0;
        return true;
    }
    out.tlas_root = f.tlas_child(0, 1);
    if (!f.err.empty()) {
        err = f.err;
        return false;
    }
    return true;
}

// the index arrays of the restricted walk (jpt_builder.h); `resolvable` stays false when the trees are not in the shape
// the walk indexes -- BLAS subtrees numbered in pre-order (left child = parent + 1: what bvh.cpp:108-185 emits), every
// instance in exactly one TLAS leaf reachable from slot 0
void ExactShadow::finish(size_t n_native)
{
    resolvable = false;
    native_ref.assign(n_native, 0xffffffffu);
    tri_leaf.assign(tri_geom.size(), 0xffffffffu);
    subtree_end.assign(bvh_nodes.size(), 0u);
    tlas_parent.assign(tlas_nodes.size(), 0xffffffffu);
    inst_tlas_leaf.assign(instances.size(), 0xffffffffu);
    if (!valid) return;
    for (size_t t = 0; t < tri_native.size(); t++)
        if (tri_native[t] < n_native) native_ref[tri_native[t]] = (uint32_t)t;
    // BLAS: pre-order check and subtree ends, one explicit-stack pass per root
    std::vector<uint8_t> done(bvh_nodes.size(), 0);
    for (uint32_t root : mesh_roots) {
        if (root >= bvh_nodes.size()) return;
        if (done[root]) continue;
        std::vector<uint32_t> open;   // nodes whose subtree is being numbered
        uint32_t next = root;         // the id the next visited node must have
        std::vector<uint32_t> todo{root};
        while (!todo.empty()) {
            const uint32_t n = todo.back();
            todo.pop_back();
            if (n != next || n >= bvh_nodes.size() || done[n]) return;   // not pre-order
            done[n] = 1;
            next++;
            const RefBvhNode& node = bvh_nodes[n];
            if (node.tri_count > 0) {
                for (uint32_t k = 0; k < node.tri_count; k++) {
                    const size_t t = (size_t)node.first_tri_index + k;
                    if (t >= tri_leaf.size()) return;
                    tri_leaf[t] = n;
                }
            } else if (!(node.left_child == 0 && node.right_child == 0)) {
                if (node.left_child != n + 1) return;
                todo.push_back(node.right_child);
                todo.push_back(node.left_child);
            }
        }
        // ends: a node's subtree ends where the next node that is not its descendant begins; in pre-order that is the
        // right sibling of the nearest ancestor-or-self that is a left child, else the end of the root's range
        std::vector<std::pair<uint32_t, uint32_t>> st{{root, next}};
        while (!st.empty()) {
            const uint32_t n = st.back().first, e = st.back().second;
            st.pop_back();
            subtree_end[n] = e;
            const RefBvhNode& node = bvh_nodes[n];
            if (node.tri_count == 0 && !(node.left_child == 0 && node.right_child == 0)) {
                if (!(node.right_child > node.left_child && node.right_child < e)) return;
                st.emplace_back(node.left_child, node.right_child);
                st.emplace_back(node.right_child, e);
            }
        }
    }
    // TLAS: parents and the leaf of every instance
    if (!instances.empty()) {
        if (tlas_nodes.empty()) return;
        std::vector<uint32_t> todo{0u};
        tlas_parent[0] = 0u;
        std::vector<uint8_t> seen(tlas_nodes.size(), 0);
        while (!todo.empty()) {
            const uint32_t n = todo.back();
            todo.pop_back();
            if (seen[n]) return;
            seen[n] = 1;
            const RefTlasNode& node = tlas_nodes[n];
            if (node.leftRight == 0u) {
                if (node.blas >= instances.size() || inst_tlas_leaf[node.blas] != 0xffffffffu) return;
                inst_tlas_leaf[node.blas] = n;
                continue;
            }
            const uint32_t l = node.leftRight & 0xffffu, r = node.leftRight >> 16;
            if (l >= tlas_nodes.size() || r >= tlas_nodes.size()) return;
            tlas_parent[l] = tlas_parent[r] = n;
            todo.push_back(r);
            todo.push_back(l);
        }
        for (uint32_t leaf : inst_tlas_leaf)
            if (leaf == 0xffffffffu) return;
    }
    resolvable = true;
}

namespace {

struct Child4 {
    float lo[3], hi[3];
    int32_t ref;
    float area() const
    {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
};

// WHICH two-child records become four-child records (the TLAS and the meshes' trees alike), chosen for the fewest expected record steps instead of
// greedily.  A ray that enters a kept record pays one step whatever the number of its slots, so the cost of a collapse is the sum
// of the kept records' surface areas; the leaves are the same either way.  F(n, k) = least such sum below record n when n's
// subtree may fill at most k slots of the kept record above it:
//     F(leaf, k) = 0
//     F(n, 1)    = A(n) + min over a of F(left, a) + F(right, 4 - a)               (n kept: its own four slots dealt to its sides)
//     F(n, k)    = min(F(n, 1), min over a in 1..k-1 of F(left, a) + F(right, k - a))   (n dissolved into the record above)
// (the dynamic programme of Ylitie, Karras, Laine 2017, section 3, for width 4 and fixed leaves).  pick[n][k-1] = the a taken, 0 =
// n is kept.  Evaluated lazily from the roots that collapse_root is asked for; records shared by instances are visited once.
struct CollapsePlan {
    const std::vector<WideNode>& src;
    std::vector<float> f;        // [n][3]: F(n, 1..3)
    std::vector<uint8_t> pick;   // [n][3]
    std::vector<uint8_t> done;
    explicit CollapsePlan(const std::vector<WideNode>& s) : src(s), f(s.size() * 3, 0.0f), pick(s.size() * 3, 0), done(s.size(), 0) {}
    bool is_leaf(int32_t r) const
    {
        if (r < 0) return true;
        const WideNode& w = src[(size_t)r];
        return w.left == r && w.right == r;   // empty-leaf record
    }
    float F(int32_t r, int k) const { return is_leaf(r) ? 0.0f : f[(size_t)r * 3 + (size_t)(k - 1)]; }
    void solve(int32_t n)
    {
        if (is_leaf(n) || done[(size_t)n]) return;
        done[(size_t)n] = 1;
        const WideNode& w = src[(size_t)n];
        solve(w.left);
        solve(w.right);
        float lo[3], hi[3];
        for (int k = 0; k < 3; k++) {
            lo[k] = std::min(w.lmin[k], w.rmin[k]);
            hi[k] = std::max(w.lmax[k], w.rmax[k]);
        }
        const float dx = std::max(hi[0] - lo[0], 0.0f), dy = std::max(hi[1] - lo[1], 0.0f), dz = std::max(hi[2] - lo[2], 0.0f);
        const float area = dx * dy + dy * dz + dz * dx;
        float best = FLT_MAX;
        uint8_t best_a = 2;
        for (int a = 1; a <= 3; a++) {
            const float v = F(w.left, a) + F(w.right, 4 - a);
            if (v < best || (v == best && a == 2)) {   // (ties: the even split)
                best = v;
                best_a = (uint8_t)a;
            }
        }
        float* fn = &f[(size_t)n * 3];
        uint8_t* pn = &pick[(size_t)n * 3];
        fn[0] = area + best;
        pn[0] = best_a;
        for (int k = 2; k <= 3; k++) {
            float bk = fn[0];
            uint8_t ak = 0;
            for (int a = 1; a < k; a++) {
                const float v = F(w.left, a) + F(w.right, k - a);
                if (v < bk) {
                    bk = v;
                    ak = (uint8_t)a;
                }
            }
            fn[k - 1] = bk;
            pn[k - 1] = ak;
        }
    }
};
thread_local CollapsePlan* tl_plan = nullptr;

// collapse the subtree under two-child record `ni` of `src` into `dst` by the plan's cut; returns the new record's index.
// Records are laid out depth first: a record, then its first child's whole subtree, ...  (Siblings side by side, two per
// 128-byte line, measured no better: LAB_NOTEBOOK.md, round 5.)
int32_t collapse4(const std::vector<WideNode>& src, std::vector<WideNode4>& dst, int32_t ni, std::vector<int32_t>& memo)
{
    if (memo[(size_t)ni] >= 0) return memo[(size_t)ni];
    const int32_t self = (int32_t)dst.size();
    memo[(size_t)ni] = self;
    dst.emplace_back();
    auto kids_of = [&](int32_t n, Child4* out) {
        const WideNode& w = src[(size_t)n];
        for (int k = 0; k < 3; k++) {
            out[0].lo[k] = w.lmin[k]; out[0].hi[k] = w.lmax[k];
            out[1].lo[k] = w.rmin[k]; out[1].hi[k] = w.rmax[k];
        }
        out[0].ref = w.left;
        out[1].ref = w.right;
    };
    Child4 c[4];
    int n = 2;
    kids_of(ni, c);
    {   // the plan's cut: slot budgets dealt down the two sides
        tl_plan->solve(ni);
        int budget[4] = {(int)tl_plan->pick[(size_t)ni * 3], 0, 0, 0};
        budget[1] = 4 - budget[0];
        for (int i = 0; i < n;) {
            const int32_t r = c[i].ref;
            const int a = (budget[i] > 1 && !tl_plan->is_leaf(r)) ? (int)tl_plan->pick[(size_t)r * 3 + (size_t)(budget[i] - 1)] : 0;
            if (a == 0) {   // a leaf, or kept as a record of its own
                i++;
                continue;
            }
            Child4 two[2];
            kids_of(r, two);
            const int b = budget[i] - a;
            c[i] = two[0];
            budget[i] = a;
            c[n] = two[1];
            budget[n] = b;
            n++;   // (slot i is looked at again with its smaller budget)
        }
    }
    WideNode4 w;
    std::memset(&w, 0, sizeof w);
    for (int i = 0; i < 4; i++) {
        if (i < n) {
            w.lo_x[i] = c[i].lo[0]; w.lo_y[i] = c[i].lo[1]; w.lo_z[i] = c[i].lo[2];
            w.hi_x[i] = c[i].hi[0]; w.hi_y[i] = c[i].hi[1]; w.hi_z[i] = c[i].hi[2];
            int32_t r = c[i].ref;
            if (r >= 0) {
                const WideNode& s2 = src[(size_t)r];
                if (s2.left == r && s2.right == r) r = kEmptyChild;  // empty-leaf record: nothing below
                else r = collapse4(src, dst, r, memo);
            }
            w.child[i] = r;
        } else {
            w.lo_x[i] = w.lo_y[i] = w.lo_z[i] = FLT_MAX;
            w.hi_x[i] = w.hi_y[i] = w.hi_z[i] = -FLT_MAX;
            w.child[i] = kEmptyChild;
        }
    }
    dst[(size_t)self] = w;
    return self;
}

int32_t collapse_root(const std::vector<WideNode>& src, std::vector<WideNode4>& dst, int32_t root, std::vector<int32_t>& memo)
{
    if (root < 0) return root;  // the whole tree is one leaf
    if ((size_t)root >= src.size()) return root;  // no tree at all (empty scene)
    const WideNode& w = src[(size_t)root];
    if (w.left == root && w.right == root) {  // empty mesh: a record whose four slots are empty
        const int32_t self = (int32_t)dst.size();
        WideNode4 e;
        std::memset(&e, 0, sizeof e);
        for (int i = 0; i < 4; i++) {
            e.lo_x[i] = e.lo_y[i] = e.lo_z[i] = FLT_MAX;
            e.hi_x[i] = e.hi_y[i] = e.hi_z[i] = -FLT_MAX;
            e.child[i] = kEmptyChild;
        }
        dst.push_back(e);
        return self;
    }
    return collapse4(src, dst, root, memo);
}

}  // namespace

namespace {

// worst-case pending entries below a record: near-first traversal keeps (children - 1) siblings pending while it
// descends into one child
uint32_t need2(const std::vector<WideNode>& nodes, int32_t ref, std::vector<int32_t>& memo)
{
    if (ref < 0 || (size_t)ref >= nodes.size()) return 0;
    if (memo[(size_t)ref] >= 0) return (uint32_t)memo[(size_t)ref];
    const WideNode& n = nodes[(size_t)ref];
    uint32_t r = 0;
    if (!(n.left == ref && n.right == ref)) {
        memo[(size_t)ref] = 0;
        r = 1u + std::max(need2(nodes, n.left, memo), need2(nodes, n.right, memo));
    }
    memo[(size_t)ref] = (int32_t)r;
    return r;
}
uint32_t need4(const std::vector<WideNode4>& nodes, int32_t ref, std::vector<int32_t>& memo)
{
    if (ref < 0 || (size_t)ref >= nodes.size()) return 0;
    if (memo[(size_t)ref] >= 0) return (uint32_t)memo[(size_t)ref];
    const WideNode4& n = nodes[(size_t)ref];
    memo[(size_t)ref] = 0;
    uint32_t kids = 0, deepest = 0;
    for (int i = 0; i < 4; i++)
        if (n.child[i] != kEmptyChild) {
            kids++;
            deepest = std::max(deepest, need4(nodes, n.child[i], memo));
        }
    const uint32_t r = kids ? (kids - 1u) + deepest : 0u;
    memo[(size_t)ref] = (int32_t)r;
    return r;
}

}  // namespace

void tlas4_refit_schedule(const WideScene& w, std::vector<uint32_t>& order, std::vector<uint32_t>& level_start)
{
    order.clear();
    level_start.clear();
    level_start.push_back(0);
    const size_t n = w.tlas_nodes4.size();
    if (w.tlas_root4 < 0 || (size_t)w.tlas_root4 >= n) return;  // no record: the root is an instance (or nothing)
    // breadth first from the root: the levels, shallowest first; a record has one parent, so each appears once
    std::vector<std::vector<uint32_t>> levels;
    levels.push_back({(uint32_t)w.tlas_root4});
    for (size_t l = 0; l < levels.size() && l <= n; l++) {
        std::vector<uint32_t> next;
        for (uint32_t ni : levels[l])
            for (int k = 0; k < 4; k++) {
                const int32_t c = w.tlas_nodes4[ni].child[k];
                if (c >= 0 && (size_t)c < n) next.push_back((uint32_t)c);
            }
        if (!next.empty()) levels.push_back(std::move(next));
    }
    for (size_t l = levels.size(); l-- > 0;) {
        order.insert(order.end(), levels[l].begin(), levels[l].end());
        level_start.push_back((uint32_t)order.size());
    }
}

void compute_stack_need(WideScene& out)
{
    {
        std::vector<int32_t> mb(out.blas_nodes.size(), -1), mt(out.tlas_nodes.size(), -1);
        uint32_t blas = 0;
        for (const WideInstance& i : out.instances) blas = std::max(blas, need2(out.blas_nodes, i.root, mb));
        out.stack_need2 = need2(out.tlas_nodes, out.tlas_root, mt) + 1u + blas;
    }
    {
        std::vector<int32_t> mb(out.blas_nodes4.size(), -1), mt(out.tlas_nodes4.size(), -1);
        uint32_t blas = 0;
        for (const WideInstance& i : out.instances4) blas = std::max(blas, need4(out.blas_nodes4, i.root, mb));
        out.stack_need4 = out.instances4.empty() ? 0u : need4(out.tlas_nodes4, out.tlas_root4, mt) + 1u + blas;
    }
}

bool reflatten_tlas(const RefScene& ref, WideScene& out, bool with4, std::string& err)
{
    if (out.instances.size() != ref.instances.size()) {
        err = "instance count changed: upload the whole scene again";
        return false;
    }
    for (size_t i = 0; i < ref.instances.size(); i++) {
        const float* m = ref.instances[i].inverse_transform;
        for (int c = 0; c < 4; c++)
            for (int r = 0; r < 3; r++) out.instances[i].inv[c * 3 + r] = m[c * 4 + r];
        if (with4) std::memcpy(out.instances4[i].inv, out.instances[i].inv, sizeof out.instances[i].inv);
    }
    out.tlas_nodes.clear();
    out.tlas_nodes4.clear();
    out.max_tlas_depth = 0;
    out.tlas_root = out.tlas_root4 = 0;
    if (!ref.tlas_nodes.empty() && !ref.instances.empty()) {
        Flattener f(ref, out);
        out.tlas_root = f.tlas_child(0, 1);
        if (!f.err.empty()) {
            err = f.err;
            return false;
        }
    }
    if (with4) {
        std::vector<int32_t> memo_t(out.tlas_nodes.size(), -1);
        CollapsePlan plan_t(out.tlas_nodes);
        tl_plan = &plan_t;
        out.tlas_root4 = collapse_root(out.tlas_nodes, out.tlas_nodes4, out.tlas_root, memo_t);
        tl_plan = nullptr;
    }
    compute_stack_need(out);
    return true;
}

void flatten4(WideScene& out)
{
    out.blas_nodes4.clear();
    out.tlas_nodes4.clear();
    std::vector<int32_t> memo(out.blas_nodes.size(), -1);
    out.instances4 = out.instances;
    {
        CollapsePlan plan(out.blas_nodes);
        tl_plan = &plan;
        for (size_t i = 0; i < out.instances.size(); i++)
            out.instances4[i].root = collapse_root(out.blas_nodes, out.blas_nodes4, out.instances[i].root, memo);
        tl_plan = nullptr;
    }
    std::vector<int32_t> memo_t(out.tlas_nodes.size(), -1);
    CollapsePlan plan_t(out.tlas_nodes);
    tl_plan = &plan_t;
    out.tlas_root4 = collapse_root(out.tlas_nodes, out.tlas_nodes4, out.tlas_root, memo_t);
    tl_plan = nullptr;
}

}  // namespace jpt
