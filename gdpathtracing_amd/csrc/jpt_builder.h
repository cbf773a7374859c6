// jpt_builder.h -- host-side acceleration-structure builder of libjpt_hip.so.
//
// Replaces src/bvh/bvh.{h,cpp} of the reference (BVHBuilder::BuildBVH, BLASInstance::set_*,
// TLAS::build) and the packing tail of GeometryGroup3D::build (geometry_group3d.cpp:305-365).
// Two modes:
//   reference-exact  the reference's algorithm restated (same split decisions, same pre-order node
//                    numbering, same default-box quirk), so the arrays equal what the addon's own
//                    builder emits and the kernels walk the same tree in the same order;
//   SAH              a binned-SAH builder over centroid bounds with tight boxes (fast path).
// Either way `flatten()` re-lays the tree out as 64-byte two-child records for the kernels.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "jpt_types.h"

namespace jpt {

struct SurfaceView {  // one ArrayMesh surface (bvh.cpp:192-198)
    const float* vertices;
    const float* normals;
    const float* uvs;
    const int32_t* indices;
    int32_t n_vertices, n_indices;
};

// The reference's OWN trees, kept beside a native scene ("shadow").  Of two triangles at exactly the same distance the
// reference keeps the one its walk tests LATER (`t > hitInfo.t` rejects, main.glsl:247), and a box entered at exactly
// hitInfo.t is not entered at all (:290): which triangle survives an exact tie is a property of the reference's visiting
// order, which a native tree does not have.  The native walk flags such hits; wf2_finish then repeats the reference's walk
// -- its nodes, its order, its arithmetic -- restricted to the ANCESTORS of the leaves that hold the tying triangles
// (everything else can only contribute farther hits, which never keep a box that holds a tying triangle from being
// entered), and takes what that walk keeps.  On route (ii) the trees come from the run of the reference's builder that also
// yields the reach records; on route (i) they are the uploaded arrays themselves.
struct ExactShadow {
    std::vector<RefBvhNode> bvh_nodes;
    std::vector<RefTriGeometry> tri_geom;   // in the reference's triangle order
    std::vector<uint32_t> tri_native;       // reference triangle -> the same triangle's index in the native order (~0u: none)
    std::vector<RefInstance> instances;     // the reference's records (blas_index into bvh_nodes above)
    std::vector<RefTlasNode> tlas_nodes;
    std::vector<uint32_t> mesh_roots;       // per unique mesh: its root in bvh_nodes
    // derived by finish() -- what the restricted walk needs to tell ancestors from the rest:
    std::vector<uint32_t> native_ref;       // native triangle -> reference triangle
    std::vector<uint32_t> tri_leaf;         // reference triangle -> the BVH node (a leaf) that holds it
    std::vector<uint32_t> subtree_end;      // per BVH node: one past the last node of its subtree (the trees are in pre-order)
    std::vector<uint32_t> tlas_parent;      // per TLAS node reachable from slot 0: its parent (slot 0: itself)
    std::vector<uint32_t> inst_tlas_leaf;   // per instance: its TLAS leaf
    bool valid = false;                     // trees present ...
    bool resolvable = false;                // ... and in the shape finish() can index (pre-order BLASes, every instance in one TLAS leaf)
    void clear()
    {
        bvh_nodes.clear(); tri_geom.clear(); tri_native.clear(); instances.clear(); tlas_nodes.clear(); mesh_roots.clear();
        native_ref.clear(); tri_leaf.clear(); subtree_end.clear(); tlas_parent.clear(); inst_tlas_leaf.clear();
        valid = resolvable = false;
    }
    void finish(size_t n_native_triangles);
};

// Scene in the reference layout: exactly the vectors GeometryGroup3D keeps (geometry_group3d.h:40-60).
struct RefScene {
    std::vector<RefTriangle> triangles;       // builder-internal (empty after a reference-layout upload)
    std::vector<RefTriGeometry> tri_geom;
    std::vector<RefTriData> tri_data;
    std::vector<RefMaterial> materials;
    std::vector<RefBvhNode> bvh_nodes;
    std::vector<RefInstance> instances;
    std::vector<RefTlasNode> tlas_nodes;
    std::vector<uint8_t> textures;
    int32_t tex_res = 0, n_layers = 0;
    std::vector<uint32_t> mesh_roots;         // BuildBVH return values, one per unique mesh
    // reach records (jpt_types.h; BuildMode::Sah only, else empty): per triangle the box of its reference leaf, per
    // mesh the box of the reference root, per instance the world box the reference computes from it
    std::vector<ReachTri> reach_tri;
    std::vector<ReachInst> reach_inst;
    std::vector<ReachInst> mesh_ref_root;     // root_lo / root_hi filled; one per unique mesh
    // native commits: the boxes that bound an instance's world box (jpt_builder.cpp, InstanceCuts) -- per box centre.xyz and half
    // extent.xyz in the mesh's space, per instance (first box, count) -- so that a device refit (jpt_scene_refit_tlas) gives a moved
    // instance the box a host update would
    std::vector<float> inst_cut_boxes;
    std::vector<uint32_t> inst_cut_range;
    // route (i) on the native tree (native_from_uploaded): per unique BLAS the node index the CALLER's arrays gave its
    // root, and per instance the blas_index the caller uploaded -- what jpt_scene_update_reference_tlas matches new
    // BLASInstance records against
    std::vector<uint32_t> up_mesh_root;
    std::vector<uint32_t> up_blas_index;
    ExactShadow exact;                        // BuildMode::Sah and native uploads
    void clear();
};

// Scene in the flattened device layout.
struct WideScene {
    std::vector<WideNode> blas_nodes;
    std::vector<WideTri> tris;
    std::vector<WideNode> tlas_nodes;
    std::vector<WideInstance> instances;
    int32_t tlas_root = 0;
    uint32_t max_blas_depth = 0, max_tlas_depth = 0;
    // 4-wide collapse of the same trees (flatten4); instances4[i].root refers to blas_nodes4
    std::vector<WideNode4> blas_nodes4, tlas_nodes4;
    std::vector<WideInstance> instances4;
    int32_t tlas_root4 = 0;
    // worst-case number of live traversal-stack entries (TLAS pending + sentinel + BLAS pending) for the
    // two-child and the four-child records; the kernels' stack capacity must cover it
    uint32_t stack_need2 = 0, stack_need4 = 0;
};

enum class BuildMode {
    ReferenceExact = 0,
    Sah = 1,            // native tree + reach records (one extra run of the reference's builder per mesh)
    SahWatertight = 2   // native tree alone: every Moller-Trumbore hit is found, the reference's cracks are not reproduced
};
inline bool is_native(BuildMode m) { return m != BuildMode::ReferenceExact; }

class SceneBuilder {
  public:
    void begin();
    // BVHBuilder::BuildBVH (bvh.cpp:187-223); returns mesh id
    uint32_t add_mesh(const SurfaceView* surfaces, int32_t n_surfaces);
    // BLASInstance::set_materials + set_transform (bvh.h:73-115)
    bool add_instance(uint32_t mesh_id, const float* transform12, const int32_t* material_ids, int32_t n_ids);
    // runs the per-mesh builds, instance AABBs, TLAS::build (bvh.cpp:264-317) and the GpuTriangle split
    // (geometry_group3d.cpp:356-365); fills `out`
    bool commit(BuildMode mode, RefScene& out, std::string& err);
    // moving instances: new transform for instance `instance` of the committed scene, then rebuild_instances()
    // recomputes every BLASInstance record and the TLAS over the unchanged BLASes of `out`
    bool set_instance_transform(uint32_t instance, const float* transform12);
    bool rebuild_instances(BuildMode mode, RefScene& out, std::string& err);
    size_t instance_count() const { return instances_.size(); }

  private:
    struct PendingMesh { std::vector<RefTriangle> tris; };
    struct PendingInstance { uint32_t mesh; float t12[12]; uint32_t mats[3]; };
    std::vector<PendingMesh> meshes_;
    std::vector<PendingInstance> instances_;
};

// (godot Transform3D::affine_inverse restated in float: affine_inverse12 in jpt_instance_math.h)

// reach record of one instance: mesh_root (root_lo / root_hi) + the world box the reference computes for this transform
ReachInst reach_instance(const float* transform12, const ReachInst& mesh_root);

// ---- route (i) on the native tree ------------------------------------------------------------------------------------
// The arrays GeometryGroup3D emits (get_*_buffer, geometry_group3d.cpp:40-68) hold everything the fast route needs: the
// triangles (GpuTriangleGeometry / GpuTriangleData), the instance matrices, and -- directly readable -- the two boxes that
// decide what the reference's traversal can reach: the box of the reference leaf that holds each triangle (BVHNode with
// tri_count > 0) and the box of each instance's TLAS leaf.  native_from_uploaded builds the native SAH trees over the
// uploaded triangles of every BLAS an instance names and takes the reach records from those boxes; no builder of the
// reference runs.  `out` is a scene in reference layout whose trees are the native ones (like a JPT_BUILD_SAH commit).
// Returns false, with the reason in `why`, when the uploaded arrays are not a tree the reach rule applies to (a node or
// a triangle reachable twice, boxes that are not nested, an instance in no or several TLAS leaves, transform and
// inverse_transform that do not belong together): the caller then walks the arrays as given.
bool native_from_uploaded(const RefScene& up, RefScene& out, std::string& why);
// The same for the instance level alone (jpt_scene_update_reference_tlas): new BLASInstance / TLASNode arrays over the
// BLASes `out` already holds; each instance must name the BLAS it named at upload time.
bool native_instances_from_uploaded(const std::vector<RefInstance>& up_instances, const std::vector<RefTlasNode>& up_tlas,
                                    RefScene& out, std::string& why);

// Bottom-up schedule of the four-child TLAS records for a refit on the device (jpt_kernels_post.hip): `order` lists the
// records of w.tlas_nodes4 deepest level first, level l is order[level_start[l] .. level_start[l + 1]).
void tlas4_refit_schedule(const WideScene& w, std::vector<uint32_t>& order, std::vector<uint32_t>& level_start);

// Reference layout -> flattened layout.  Keeps topology, boxes and child order, so traversal visits the
// same nodes in the same order as main.glsl:270-350 does on the reference arrays.
bool flatten(const RefScene& ref, WideScene& out, std::string& err);

// Instances and TLAS of `ref` changed, BLASes did not: rewrites the instance records (keeping each one's BLAS
// root) and the TLAS records of `out`, including the four-child collapse when `with4`, and the stack need.
bool reflatten_tlas(const RefScene& ref, WideScene& out, bool with4, std::string& err);

// Collapses the two-child records of `out` (after flatten) into four-child records: a node's children are
// replaced, largest box first, by their own children until four slots are used.  Boxes and leaves are kept as
// they are, so the set of triangles a ray can reach is unchanged; only the visiting order differs.
void flatten4(WideScene& out);

// fills WideScene::stack_need2 / stack_need4 (call after flatten / flatten4)
void compute_stack_need(WideScene& out);

}  // namespace jpt
