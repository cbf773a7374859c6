// jpt_host.hpp -- C++ host layer over the C ABI (include/jpt.h): the reference's classes for this path with
// the same names, members and call order, minus the Godot scene tree.
//
//   GeometryGroup3D        src/path_tracing/geometry_group3d.{h,cpp}
//   ProgressiveRendering   src/path_tracing/post_processing/progressive_rendering.{h,cpp}   (frame_count logic)
//   PathTracingCamera      src/path_tracing/path_tracing_camera.{h,cpp}
//   Camera                 src/path_tracing/render_parameters.h:14-47
//
// godot-cpp (Transform3D, Projection, Ref<>, PackedByteArray, StandardMaterial3D ...) is an absent submodule,
// so the few value types the path needs are plain structs here; their arithmetic (affine_inverse,
// create_perspective, Projection * Transform3D, Projection::inverse) restates godot's published algorithms in
// float.  It runs BEFORE the boundary: it only fills the 160-byte Camera block and the instance transforms.
// Header-only, C++17, no dependencies besides jpt.h.
#pragma once

#include <jpt.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

namespace jpt_host {

using PackedByteArray = std::vector<uint8_t>;

struct Vector3 {
    float x = 0, y = 0, z = 0;
};

struct Color {
    float r = 0, g = 0, b = 0, a = 1;
};

struct Transform3D {  // godot: basis rows + origin
    float basis[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    Vector3 origin;

    Transform3D affine_inverse() const
    {
        // Basis::invert via cofactors, then origin = basis.xform(-origin)
        const auto& m = basis;
        auto cof = [&](int r1, int c1, int r2, int c2) { return m[r1][c1] * m[r2][c2] - m[r1][c2] * m[r2][c1]; };
        const float co0 = cof(1, 1, 2, 2), co1 = cof(1, 2, 2, 0), co2 = cof(1, 0, 2, 1);
        const float det = m[0][0] * co0 + m[0][1] * co1 + m[0][2] * co2;
        const float s = 1.0f / det;
        Transform3D r;
        r.basis[0][0] = co0 * s; r.basis[0][1] = cof(0, 2, 2, 1) * s; r.basis[0][2] = cof(0, 1, 1, 2) * s;
        r.basis[1][0] = co1 * s; r.basis[1][1] = cof(0, 0, 2, 2) * s; r.basis[1][2] = cof(0, 2, 1, 0) * s;
        r.basis[2][0] = co2 * s; r.basis[2][1] = cof(0, 1, 2, 0) * s; r.basis[2][2] = cof(0, 0, 1, 1) * s;
        const float nx = -origin.x, ny = -origin.y, nz = -origin.z;
        r.origin.x = r.basis[0][0] * nx + r.basis[0][1] * ny + r.basis[0][2] * nz;
        r.origin.y = r.basis[1][0] * nx + r.basis[1][1] * ny + r.basis[1][2] * nz;
        r.origin.z = r.basis[2][0] * nx + r.basis[2][1] * ny + r.basis[2][2] * nz;
        return r;
    }
    bool is_equal_approx(const Transform3D& o) const  // godot: CMP_EPSILON 1e-5 style comparison
    {
        auto eq = [](float a, float b) {
            if (a == b) return true;
            const float tol = std::max(1e-5f * std::fabs(a), 1e-5f);
            return std::fabs(a - b) < tol;
        };
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++)
                if (!eq(basis[i][j], o.basis[i][j])) return false;
        return eq(origin.x, o.origin.x) && eq(origin.y, o.origin.y) && eq(origin.z, o.origin.z);
    }
    void to_float12(float* t) const
    {
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) t[i * 3 + j] = basis[i][j];
        t[9] = origin.x; t[10] = origin.y; t[11] = origin.z;
    }
};

struct Projection {  // godot: columns[4][4]
    float columns[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};

    static Projection create_perspective(float fovy_degrees, float aspect, float z_near, float z_far, bool /*flip_fov*/ = false)
    {
        Projection p;
        const float radians = (fovy_degrees / 2.0f) * (3.14159265358979323846f / 180.0f);
        const float delta_z = z_far - z_near;
        const float sine = std::sin(radians);
        if (delta_z == 0 || sine == 0 || aspect == 0) return p;
        const float cotangent = std::cos(radians) / sine;
        p.columns[0][0] = cotangent / aspect;
        p.columns[1][1] = cotangent;
        p.columns[2][2] = -(z_far + z_near) / delta_z;
        p.columns[2][3] = -1;
        p.columns[3][2] = -2 * z_near * z_far / delta_z;
        p.columns[3][3] = 0;
        return p;
    }
    Projection() = default;
    // Projection(const Transform3D&): the transform promoted to a 4x4 with last row 0 0 0 1
    explicit Projection(const Transform3D& t)
    {
        for (int c = 0; c < 3; c++) {
            columns[c][0] = t.basis[0][c]; columns[c][1] = t.basis[1][c]; columns[c][2] = t.basis[2][c]; columns[c][3] = 0;
        }
        columns[3][0] = t.origin.x; columns[3][1] = t.origin.y; columns[3][2] = t.origin.z; columns[3][3] = 1;
    }
    // Projection -> Transform3D keeps the upper 3x4 and drops the bottom row (godot: operator Transform3D())
    Transform3D to_transform3d() const
    {
        Transform3D t;
        for (int c = 0; c < 3; c++) {
            t.basis[0][c] = columns[c][0]; t.basis[1][c] = columns[c][1]; t.basis[2][c] = columns[c][2];
        }
        t.origin.x = columns[3][0]; t.origin.y = columns[3][1]; t.origin.z = columns[3][2];
        return t;
    }
    Projection operator*(const Projection& m) const
    {
        Projection r;
        for (int j = 0; j < 4; j++)
            for (int i = 0; i < 4; i++) {
                float ab = 0;
                for (int k = 0; k < 4; k++) ab += columns[k][i] * m.columns[j][k];
                r.columns[j][i] = ab;
            }
        return r;
    }
    Projection operator*(const Transform3D& t) const { return *this * Projection(t); }
    Projection inverse() const  // Gauss-Jordan with partial pivoting, double accumulators
    {
        double a[4][8];
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 4; c++) {
                a[r][c] = columns[c][r];
                a[r][4 + c] = (r == c) ? 1.0 : 0.0;
            }
        for (int col = 0; col < 4; col++) {
            int piv = col;
            for (int r = col + 1; r < 4; r++)
                if (std::fabs(a[r][col]) > std::fabs(a[piv][col])) piv = r;
            if (piv != col)
                for (int c = 0; c < 8; c++) std::swap(a[piv][c], a[col][c]);
            const double d = a[col][col];
            if (d == 0.0) return Projection();
            for (int c = 0; c < 8; c++) a[col][c] /= d;
            for (int r = 0; r < 4; r++)
                if (r != col) {
                    const double f = a[r][col];
                    for (int c = 0; c < 8; c++) a[r][c] -= f * a[col][c];
                }
        }
        Projection out;
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 4; c++) out.columns[c][r] = (float)a[r][4 + c];
        return out;
    }
};

// render_parameters.h:14-47
struct Camera {
    float vp[16];
    float ivp[16];
    float position[4];
    unsigned int frame_index = 0;  // the reference never initialises it (render_parameters.h:19)
    float near_ = 0.01f;
    float far_ = 1000.0f;
    unsigned int _pad = 0;

    void set_camera_transform(const Transform3D& model, const Projection& projection)
    {
        position[0] = model.origin.x; position[1] = model.origin.y; position[2] = model.origin.z; position[3] = 1.0f;
        const Projection t = projection * model.affine_inverse();
        const Projection it = t.inverse();
        for (int i = 0; i < 4; i++)  // Utils::projection_to_float (utils.h:39-49)
            for (int j = 0; j < 4; j++) {
                vp[i * 4 + j] = t.columns[i][j];
                ivp[i * 4 + j] = it.columns[i][j];
            }
    }
};
static_assert(sizeof(Camera) == 160, "Camera block");

struct StandardMaterial3D {  // the properties GeometryGroup3D reads (geometry_group3d.cpp:279-290); godot defaults
    Color albedo{1, 1, 1, 1};
    float metallic = 0.0f;
    float roughness = 1.0f;
    Color emission{0, 0, 0, 1};
    float emission_energy_multiplier = 1.0f;
    int albedo_texture = -1;  // index into GeometryGroup3D::textures, -1 = none
};

struct Surface {
    std::vector<float> vertices, normals, uvs;  // 3, 3, 2 floats per vertex
    std::vector<int32_t> indices;
};

struct ArrayMesh {
    std::vector<Surface> surfaces;
    int get_surface_count() const { return (int)surfaces.size(); }
};

struct MeshInstance3D {
    const ArrayMesh* mesh = nullptr;
    Transform3D global_transform;
    std::vector<const StandardMaterial3D*> surface_override_materials;  // null = default material
};

struct GpuMaterial {  // render_parameters.h:49-57
    float albedo[4];
    float emission[4];
    float metallic, roughness;
    int albedo_texture_index;
    float padding[5];
};
static_assert(sizeof(GpuMaterial) == 64, "GpuMaterial");

// ---- scene ingest without Godot ---------------------------------------------------------------
// In the addon, Godot's importers turn .obj / .mtl / images into ArrayMesh, StandardMaterial3D and Image objects
// before GeometryGroup3D::build ever runs (geometry_group3d.cpp:119-304 only reads them).  These three functions
// stand in for that step so a host can feed the path from files.  They are NOT a claim about Godot's importers:
// the conventions are written down here.

// Wavefront OBJ -> ArrayMesh: one surface per `usemtl` group in order of first use (`surface_materials` gets the
// names), corners de-indexed to unique (position, uv, normal) triples in order of appearance, polygons fan-
// triangulated and emitted with Godot's clockwise front faces (main.glsl:254-255 derives `front` from the winding),
// a missing normal replaced by the face normal of the polygon's first three corners, a missing uv by (0, 0).
inline void load_obj(const std::string& text, ArrayMesh& mesh, std::vector<std::string>& surface_materials)
{
    struct Group {
        std::map<std::tuple<long, long, long, long>, int32_t> lut;
        Surface s;
    };
    std::vector<std::array<double, 3>> pos, nrm;   // kept in double, rounded to float once when a corner is emitted
    std::vector<std::array<double, 2>> uvs;
    std::vector<std::string> order;
    std::map<std::string, Group> groups;
    Group* cur = nullptr;
    auto use = [&](const std::string& name) {
        if (!groups.count(name)) order.push_back(name);
        cur = &groups[name];
    };
    use("");
    std::istringstream in(text);
    std::string line;
    long face_counter = 0;
    while (std::getline(in, line)) {
        std::istringstream ls(line);
        std::string tag;
        if (!(ls >> tag) || tag[0] == '#') continue;
        if (tag == "v" || tag == "vn") {
            std::array<double, 3> v{0, 0, 0};
            ls >> v[0] >> v[1] >> v[2];
            (tag == "v" ? pos : nrm).push_back(v);
        } else if (tag == "vt") {
            std::array<double, 2> t{0, 0};
            ls >> t[0] >> t[1];
            uvs.push_back(t);
        } else if (tag == "usemtl") {
            std::string name;
            ls >> name;
            use(name);
        } else if (tag == "f") {
            struct Corner { long v, t, n; };
            std::vector<Corner> corners;
            std::string tok;
            while (ls >> tok) {
                long idx[3] = {0, 0, 0};
                bool has[3] = {false, false, false};
                size_t start = 0;
                for (int k = 0; k < 3 && start <= tok.size(); k++) {
                    const size_t slash = tok.find('/', start);
                    const std::string part = tok.substr(start, slash == std::string::npos ? std::string::npos : slash - start);
                    if (!part.empty()) {
                        idx[k] = std::strtol(part.c_str(), nullptr, 10);
                        has[k] = true;
                    }
                    if (slash == std::string::npos) break;
                    start = slash + 1;
                }
                if (!has[0]) throw std::runtime_error("OBJ: face corner without a vertex index");
                auto resolve = [](long i, size_t n) { return i > 0 ? i - 1 : (long)n + i; };
                Corner c{resolve(idx[0], pos.size()), has[1] ? resolve(idx[1], uvs.size()) : -1, has[2] ? resolve(idx[2], nrm.size()) : -1};
                if (c.v < 0 || c.v >= (long)pos.size() || c.t >= (long)uvs.size() || c.n >= (long)nrm.size()) throw std::runtime_error("OBJ: index out of range");
                corners.push_back(c);
            }
            if (corners.size() < 3) continue;
            std::array<double, 3> fn{0, 0, 0};
            bool need_fn = false;
            for (const Corner& c : corners) need_fn = need_fn || c.n < 0;
            if (need_fn) {
                const auto &a = pos[(size_t)corners[0].v], &b = pos[(size_t)corners[1].v], &c = pos[(size_t)corners[2].v];
                const double e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
                const double e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
                double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
                const double len = std::max(std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]), 1e-30);
                fn = {n[0] / len, n[1] / len, n[2] / len};
            }
            std::vector<int32_t> ids;
            for (size_t k = 0; k < corners.size(); k++) {
                const Corner& c = corners[k];
                // a corner without a normal is unique to its face (it carries that face's normal)
                const auto key = c.n >= 0 ? std::make_tuple(c.v, c.t, c.n, -1L) : std::make_tuple(c.v, c.t, -2L - face_counter, (long)k);
                auto it = cur->lut.find(key);
                if (it == cur->lut.end()) {
                    const int32_t id = (int32_t)(cur->s.vertices.size() / 3);
                    it = cur->lut.emplace(key, id).first;
                    const auto& p = pos[(size_t)c.v];
                    const auto n = c.n >= 0 ? nrm[(size_t)c.n] : fn;
                    cur->s.vertices.insert(cur->s.vertices.end(), {(float)p[0], (float)p[1], (float)p[2]});
                    cur->s.normals.insert(cur->s.normals.end(), {(float)n[0], (float)n[1], (float)n[2]});
                    if (c.t >= 0) cur->s.uvs.insert(cur->s.uvs.end(), {(float)uvs[(size_t)c.t][0], (float)uvs[(size_t)c.t][1]});
                    else cur->s.uvs.insert(cur->s.uvs.end(), {0.0f, 0.0f});
                }
                ids.push_back(it->second);
            }
            for (size_t k = 1; k + 1 < ids.size(); k++) cur->s.indices.insert(cur->s.indices.end(), {ids[0], ids[k + 1], ids[k]});
            face_counter++;
        }
    }
    mesh.surfaces.clear();
    surface_materials.clear();
    for (const std::string& name : order)
        if (!groups[name].s.indices.empty()) {
            mesh.surfaces.push_back(std::move(groups[name].s));
            surface_materials.push_back(name);
        }
}

// Wavefront MTL -> StandardMaterial3D (the fields GeometryGroup3D converts to GpuMaterial, geometry_group3d.cpp:271-292):
// Kd -> albedo; Ke -> emission colour (energy multiplier 1, or the largest component when it exceeds 1, the colour
// scaled back into [0,1]); Pr -> roughness, else from the Phong exponent Ns as sqrt(2 / (Ns + 2)); Pm -> metallic (else 0);
// map_Kd -> albedo_texture = index of that file name in `albedo_maps` (appended on first use).
inline std::map<std::string, StandardMaterial3D> load_mtl(const std::string& text, std::vector<std::string>& albedo_maps)
{
    std::map<std::string, StandardMaterial3D> out;
    StandardMaterial3D* cur = nullptr;
    bool has_pr = false;
    std::istringstream in(text);
    std::string line;
    while (std::getline(in, line)) {
        std::istringstream ls(line);
        std::string tag;
        if (!(ls >> tag) || tag[0] == '#') continue;
        if (tag == "newmtl") {
            std::string name;
            ls >> name;
            cur = &out[name];
            *cur = StandardMaterial3D();
            has_pr = false;
        } else if (!cur) {
            continue;
        } else if (tag == "Kd") {
            ls >> cur->albedo.r >> cur->albedo.g >> cur->albedo.b;
        } else if (tag == "Ke") {
            float r = 0, g = 0, b = 0;
            ls >> r >> g >> b;
            const float m = std::max(r, std::max(g, b));
            if (m > 1.0f) {
                cur->emission = Color{r / m, g / m, b / m, 1};
                cur->emission_energy_multiplier = m;
            } else {
                cur->emission = Color{r, g, b, 1};
                cur->emission_energy_multiplier = 1.0f;
            }
        } else if (tag == "Pr") {
            ls >> cur->roughness;
            has_pr = true;
        } else if (tag == "Ns" && !has_pr) {
            float ns = 0;
            ls >> ns;
            cur->roughness = std::sqrt(2.0f / (std::max(ns, 0.0f) + 2.0f));
        } else if (tag == "Pm") {
            ls >> cur->metallic;
        } else if (tag == "map_Kd") {
            std::string file;
            ls >> file;
            size_t k = std::find(albedo_maps.begin(), albedo_maps.end(), file) - albedo_maps.begin();
            if (k == albedo_maps.size()) albedo_maps.push_back(file);
            cur->albedo_texture = (int)k;
        }
    }
    return out;
}

// One layer of the texture array (geometry_group3d.cpp:294-300: decompress + Image::resize to texture_array_resolution):
// an RGBA8 image of any size -> res x res RGBA8, bilinear over pixel centres (an image that already has the array's size
// passes through unchanged).  Godot's Image::resize is engine code: no texel-for-texel claim for other sizes.
inline PackedByteArray resize_rgba8(const uint8_t* rgba, int w, int h, int res)
{
    PackedByteArray out((size_t)res * res * 4);
    if (w == res && h == res) {
        std::memcpy(out.data(), rgba, out.size());
        return out;
    }
    for (int y = 0; y < res; y++) {
        const double ys = (y + 0.5) * h / res - 0.5;
        const int y0 = std::min(std::max((int)std::floor(ys), 0), h - 1), y1 = std::min(y0 + 1, h - 1);
        const double fy = std::min(std::max(ys - std::floor(ys), 0.0), 1.0);
        for (int x = 0; x < res; x++) {
            const double xs = (x + 0.5) * w / res - 0.5;
            const int x0 = std::min(std::max((int)std::floor(xs), 0), w - 1), x1 = std::min(x0 + 1, w - 1);
            const double fx = std::min(std::max(xs - std::floor(xs), 0.0), 1.0);
            for (int c = 0; c < 4; c++) {
                auto px = [&](int xx, int yy) { return (double)rgba[((size_t)yy * w + xx) * 4 + c]; };
                const double top = px(x0, y0) * (1.0 - fx) + px(x1, y0) * fx, bot = px(x0, y1) * (1.0 - fx) + px(x1, y1) * fx;
                out[((size_t)y * res + x) * 4 + c] = (uint8_t)std::min(std::max(std::floor(top * (1.0 - fy) + bot * fy + 0.5), 0.0), 255.0);
            }
        }
    }
    return out;
}

inline void check(jpt_ctx* ctx, int rc, const char* what)
{
    if (rc != JPT_OK) throw std::runtime_error(std::string(what) + ": " + jpt_last_error(ctx));
}

class GeometryGroup3D {
  public:
    StandardMaterial3D default_material{Color{0.5f, 0.5f, 0.5f, 1}, 0.0f, 0.5f};  // geometry_group3d.cpp:239-245
    bool default_material_set = false;
    int texture_array_resolution = 1024;                 // geometry_group3d.h:64
    std::vector<PackedByteArray> textures;               // RGBA8 layers, res x res each
    int builder = JPT_BUILD_SAH;

    void set_default_material(const StandardMaterial3D& m) { default_material = m; default_material_set = true; }
    void add_child(const MeshInstance3D& node) { children.push_back(node); }
    // The scene lives on several devices (PathTracingCameraMulti::init attaches its jpt_multi): build() then commits on
    // rank 0's context and shares, and update_transforms() reaches EVERY rank's replica (jpt_multi_set_instance_transform,
    // jpt_multi_update_tlas / jpt_multi_refit_tlas) -- not just the context the scene was built on.
    void attach_multi(jpt_multi* m) { multi_ = m; }

    // geometry_group3d.cpp:228-366: collect instances, dedup meshes / materials by pointer, convert materials,
    // then builder + instances + TLAS + upload (the jpt_scene_* calls)
    void build(jpt_ctx* ctx)
    {
        ctx_ = ctx;
        std::vector<const ArrayMesh*> meshes;
        std::vector<const StandardMaterial3D*> mats;  // entry 0 = default material
        mats.push_back(&default_material);
        struct NodeRef { size_t mesh_id; std::vector<int32_t> material_ids; Transform3D t; };
        std::vector<NodeRef> nodes;
        for (const MeshInstance3D& n : children) {  // collect_mesh_instances (:150-214)
            if (!n.mesh) continue;
            size_t mid = std::find(meshes.begin(), meshes.end(), n.mesh) - meshes.begin();
            if (mid == meshes.size()) meshes.push_back(n.mesh);
            NodeRef r{mid, {}, n.global_transform};
            for (int s = 0; s < n.mesh->get_surface_count(); s++) {
                const StandardMaterial3D* m = s < (int)n.surface_override_materials.size() ? n.surface_override_materials[s] : nullptr;
                int32_t id = 0;  // only override materials are honoured; anything else maps to the default (:186-202)
                if (m) {
                    size_t k = std::find(mats.begin(), mats.end(), m) - mats.begin();
                    if (k == mats.size()) mats.push_back(m);
                    id = (int32_t)k;
                }
                r.material_ids.push_back(id);
            }
            nodes.push_back(std::move(r));
        }
        materials_.clear();
        for (const StandardMaterial3D* m : mats) {  // :271-292
            GpuMaterial g;
            std::memset(&g, 0, sizeof g);
            g.albedo[0] = m->albedo.r; g.albedo[1] = m->albedo.g; g.albedo[2] = m->albedo.b; g.albedo[3] = 1.0f;
            g.metallic = m->metallic;
            g.roughness = m->roughness;
            g.emission[0] = m->emission.r; g.emission[1] = m->emission.g; g.emission[2] = m->emission.b;
            g.emission[3] = m->emission_energy_multiplier;
            g.albedo_texture_index = m->albedo_texture;
            materials_.push_back(g);
        }
        check(ctx, jpt_scene_begin(ctx), "jpt_scene_begin");
        std::vector<uint32_t> ids;
        for (const ArrayMesh* mesh : meshes) {  // :308-313
            std::vector<jpt_surface> sv;
            for (const Surface& s : mesh->surfaces)
                sv.push_back(jpt_surface{s.vertices.data(), s.normals.data(), s.uvs.data(), s.indices.data(),
                                         (int32_t)(s.vertices.size() / 3), (int32_t)s.indices.size()});
            uint32_t id = 0;
            check(ctx, jpt_scene_add_mesh(ctx, sv.data(), (int32_t)sv.size(), &id), "jpt_scene_add_mesh");
            ids.push_back(id);
        }
        built_transforms_.clear();
        for (const NodeRef& r : nodes) {  // :322-341
            float t12[12];
            r.t.to_float12(t12);
            built_transforms_.emplace_back();
            std::memcpy(built_transforms_.back().data(), t12, sizeof t12);
            check(ctx, jpt_scene_add_instance(ctx, ids[r.mesh_id], t12, r.material_ids.data(), (int32_t)r.material_ids.size()),
                  "jpt_scene_add_instance");
        }
        check(ctx, jpt_scene_set_materials(ctx, materials_.data(), (uint32_t)materials_.size()), "jpt_scene_set_materials");
        if (!textures.empty()) {
            PackedByteArray all;
            for (const PackedByteArray& t : textures) all.insert(all.end(), t.begin(), t.end());
            check(ctx, jpt_scene_set_textures(ctx, all.data(), texture_array_resolution, (int32_t)textures.size()), "jpt_scene_set_textures");
        }
        check(ctx, jpt_scene_commit(ctx, builder), "jpt_scene_commit");
    }

    // Moving nodes.  The reference has no such call -- a moved MeshInstance3D needs build() again -- but lists a
    // runtime TLAS update as wanted (README.md:39-40): re-reads every child's global transform, hands the changed
    // ones to the library and lets it redo the BLASInstance records + TLAS (BLASes stay on the device).
    // Returns the number of instances that moved.
    MeshInstance3D& get_child(size_t i) { return children.at(i); }
    size_t get_child_count() const { return children.size(); }
    // refit_on_device: instance records + TLAS boxes recomputed by two kernels over the topology of the last build
    // (jpt_scene_refit_tlas: no host rebuild, no stall); default: host rebuild (jpt_scene_update_tlas)
    int update_transforms(bool refit_on_device = false)
    {
        if (!ctx_) throw std::runtime_error("GeometryGroup3D::update_transforms before build");
        int moved = 0;
        uint32_t instance = 0;
        for (const MeshInstance3D& n : children) {
            if (!n.mesh) continue;  // build() skipped it too
            float now[12];
            n.global_transform.to_float12(now);
            if (instance >= built_transforms_.size()) throw std::runtime_error("children changed since build(): build() again");
            if (std::memcmp(now, built_transforms_[instance].data(), sizeof now) != 0) {
                if (multi_) mcheck(jpt_multi_set_instance_transform(multi_, instance, now), "jpt_multi_set_instance_transform");
                else check(ctx_, jpt_scene_set_instance_transform(ctx_, instance, now), "jpt_scene_set_instance_transform");
                std::memcpy(built_transforms_[instance].data(), now, sizeof now);
                moved++;
            }
            instance++;
        }
        if (moved) {
            if (refit_on_device) {
                std::vector<float> all;
                all.reserve(built_transforms_.size() * 12);
                for (const auto& t : built_transforms_) all.insert(all.end(), t.begin(), t.end());
                if (multi_) mcheck(jpt_multi_refit_tlas(multi_, all.data(), (uint32_t)built_transforms_.size()), "jpt_multi_refit_tlas");
                else check(ctx_, jpt_scene_refit_tlas(ctx_, all.data(), (uint32_t)built_transforms_.size()), "jpt_scene_refit_tlas");
            } else {
                if (multi_) mcheck(jpt_multi_update_tlas(multi_), "jpt_multi_update_tlas");
                else check(ctx_, jpt_scene_update_tlas(ctx_), "jpt_scene_update_tlas");
            }
        }
        return moved;
    }

    // geometry_group3d.cpp:40-68 (bytes as the reference emits them after a REFERENCE_EXACT build)
    PackedByteArray get_triangles_geometry_buffer() const { return get_buffer(JPT_BUF_TRI_GEOMETRY); }
    PackedByteArray get_triangles_data_buffer() const { return get_buffer(JPT_BUF_TRI_DATA); }
    PackedByteArray get_materials_buffer() const { return get_buffer(JPT_BUF_MATERIALS); }
    PackedByteArray get_bvh_buffer() const { return get_buffer(JPT_BUF_BVH_NODES); }
    PackedByteArray get_blas_buffer() const { return get_buffer(JPT_BUF_INSTANCES); }
    PackedByteArray get_tlas_buffer() const { return get_buffer(JPT_BUF_TLAS_NODES); }
    int get_triangle_count() const { return (int)(get_triangles_geometry_buffer().size() / 48); }   // :17
    int get_blas_count() const { return (int)(get_blas_buffer().size() / 176); }                    // :7
    int get_material_count() const { return (int)materials_.size(); }                               // :12
    int get_bvh_node_count() const { return (int)(get_bvh_buffer().size() / 48); }                  // :22
    int get_tlas_node_count() const { return (int)(get_tlas_buffer().size() / 32); }                // :27

  private:
    PackedByteArray get_buffer(int which) const
    {
        size_t n = 0;
        check(ctx_, jpt_scene_get_reference_buffer(ctx_, which, nullptr, 0, &n), "jpt_scene_get_reference_buffer");
        PackedByteArray out(n);
        check(ctx_, jpt_scene_get_reference_buffer(ctx_, which, out.data(), n, &n), "jpt_scene_get_reference_buffer");
        return out;
    }
    void mcheck(int rc, const char* what) const
    {
        if (rc != JPT_OK) throw std::runtime_error(std::string(what) + ": " + jpt_multi_last_error(multi_));
    }
    jpt_multi* multi_ = nullptr;
    std::vector<MeshInstance3D> children;
    std::vector<GpuMaterial> materials_;
    std::vector<std::array<float, 12>> built_transforms_;  // per instance, as handed to the library
    jpt_ctx* ctx_ = nullptr;
};

// progressive_rendering.cpp:47-66: the host half (camera-moved test and frame_count); the device half is fused
// into jpt_render.
class ProgressiveRendering {
  public:
    unsigned int frame_count = 1;
    // returns true when the accumulation restarts
    bool render(const Transform3D& camera_transform)
    {
        const bool camera_moved = !previous_transform.is_equal_approx(camera_transform);
        previous_transform = camera_transform;
        if (camera_moved) frame_count = 1;
        else frame_count++;
        return camera_moved;
    }

  private:
    Transform3D previous_transform;  // identity initially (progressive_rendering.h:44)
};

// temporal_reprojection.h:11-60 / temporal_reprojection.cpp:16-73, host half: previous_vp, frame_count and the
// 88-byte RenderParameters; the dispatch itself is part of jpt_render in JPT_DENOISE_TEMPORAL mode.
class TemporalReprojection {
  public:
    struct RenderParameters {  // temporal_reprojection.h:16-23
        float deltaMatrix[16];
        int width = 0, height = 0;
        unsigned int frame_count = 1;
        float blendFactor = 0.75f;
        float nearPlane = 0.01f;
        float farPlane = 1000.0f;
    };
    RenderParameters render_parameters;

    void init(int w, int h)  // temporal_reprojection.cpp:16-28
    {
        render_parameters.width = w;
        render_parameters.height = h;
        render_parameters.frame_count = 1;
    }
    // temporal_reprojection.cpp:56-72; view_matrix = get_global_transform().affine_inverse() (path_tracing_camera.cpp:220)
    void render(jpt_ctx* ctx, const Transform3D& view_matrix, const Projection& projection_matrix)
    {
        advance(view_matrix, projection_matrix);
        check(ctx, jpt_set_temporal_params(ctx, &render_parameters), "jpt_set_temporal_params");  // :67
    }
    // the host arithmetic of that call alone: the next frame's RenderParameters (no library call)
    void advance(const Transform3D& view_matrix, const Projection& projection_matrix)
    {
        const Projection vp = projection_matrix * Projection(view_matrix);
        // `Transform3D deltaMatrix = previous_vp * vp.inverse()`: the conversion drops the projective row, and
        // projection_to_float() of that Transform3D writes 0 0 0 1 back (:63,66)
        const Projection delta((previous_vp * vp.inverse()).to_transform3d());
        previous_vp = vp;
        render_parameters.frame_count++;
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) render_parameters.deltaMatrix[i * 4 + j] = delta.columns[i][j];
    }

  private:
    Projection previous_vp;  // identity initially (temporal_reprojection.h:49)
};
static_assert(sizeof(TemporalReprojection::RenderParameters) == 88, "TemporalReprojection::RenderParameters");

class PathTracingCamera {
  public:
    enum Denoising { PROGRESSIVE_RENDERING, TEMPORAL_REPROJECTION, NONE };  // path_tracing_camera.h:30-34

    explicit PathTracingCamera(int device = 0) { check(nullptr, create(device), "jpt_create"); }
    ~PathTracingCamera() { jpt_destroy(ctx); }
    PathTracingCamera(const PathTracingCamera&) = delete;
    PathTracingCamera& operator=(const PathTracingCamera&) = delete;

    float get_fov() const { return fov; }
    void set_fov(float v) { fov = v; }
    void set_geometry_group(GeometryGroup3D* g) { geometry_group = g; }
    Denoising get_denoising_mode() const { return denoising_mode; }
    void set_denoising_mode(Denoising m) { denoising_mode = m; }
    void set_global_transform(const Transform3D& t) { global_transform = t; }
    jpt_ctx* context() const { return ctx; }

    int max_bounces = 4;                    // the literal 5 of main.glsl:377 is max_bounces + 1
    int accum_mode = JPT_ACCUM_REF_LDR8;    // what the reference does (rgba8 screen image before the sum)

    // path_tracing_camera.cpp:111-187 (resolution is passed in instead of DisplayServer::window_get_size)
    void init(int w, int h)
    {
        if (!geometry_group) throw std::runtime_error("No geometry group set.");  // :117-121
        width = w;
        height = h;
        geometry_group->build(ctx);                                                // :126
        projection_matrix = Projection::create_perspective(fov, float(width) / float(height), 0.01f, 1000.0f, false);  // :134
        check(ctx, jpt_set_params(ctx, width, height, max_bounces, accum_mode, JPT_SAMPLER_NEAREST_CLAMP), "jpt_set_params");
        ready = true;
    }

    // path_tracing_camera.cpp:193-232: one frame; returns the RGBA8 screen image (get_image_uniform_buffer)
    PackedByteArray render()
    {
        if (!ready) return {};                                                      // :195
        advance_frame();
        check(ctx, jpt_render(ctx, 1, camera.frame_index), "jpt_render");           // :204 + the post-processing pass
        PackedByteArray out((size_t)width * height * 4);
        check(ctx, jpt_read_ldr_rgba8(ctx, out.data()), "jpt_read_ldr_rgba8");      // :228-229
        return out;
    }

    // The same frame loop with one frame of latency and no stall: queues this frame (jpt_render_async) and its
    // read-back into pinned memory, and returns the image of the PREVIOUS call (empty on the first call).  The
    // reference stalls on get_image_uniform_buffer every frame (:228-229); here the GPU renders frame k while the host
    // displays frame k-1, and the library overlaps the launches of consecutive queued renders.  flush() returns the
    // last queued frame.
    PackedByteArray render_overlapped()
    {
        if (!ready) return {};
        PackedByteArray previous = flush();
        advance_frame();
        check(ctx, jpt_render_async(ctx, 1, camera.frame_index), "jpt_render_async");
        check(ctx, jpt_readback_ldr_begin(ctx), "jpt_readback_ldr_begin");
        readback_queued = true;
        return previous;
    }
    PackedByteArray flush()
    {
        if (!readback_queued) return {};
        PackedByteArray out((size_t)width * height * 4);
        check(ctx, jpt_readback_ldr_end(ctx, out.data()), "jpt_readback_ldr_end");
        readback_queued = false;
        return out;
    }

    Camera camera;
    ProgressiveRendering progressive_renderer;
    TemporalReprojection temporal_reprojection;

  private:
    // camera block, frame index and the post-processing mode's host half (path_tracing_camera.cpp:198-225)
    void advance_frame()
    {
        camera.set_camera_transform(global_transform, projection_matrix);           // :198
        camera.frame_index++;                                                       // :199
        check(ctx, jpt_set_camera(ctx, &camera), "jpt_set_camera");                 // :200
        // (the r32f depth image of main.glsl:435 is read by the temporal pass only: not produced in the other two modes)
        check(ctx, jpt_set_outputs(ctx, denoising_mode == TEMPORAL_REPROJECTION ? JPT_OUTPUT_DEPTH : 0u), "jpt_set_outputs");
        switch (denoising_mode) {                                                   // :207-225
            case PROGRESSIVE_RENDERING:
                check(ctx, jpt_set_denoising_mode(ctx, JPT_DENOISE_PROGRESSIVE), "jpt_set_denoising_mode");
                if (progressive_renderer.render(global_transform)) check(ctx, jpt_accum_reset(ctx), "jpt_accum_reset");
                break;
            case TEMPORAL_REPROJECTION:
                check(ctx, jpt_set_denoising_mode(ctx, JPT_DENOISE_TEMPORAL), "jpt_set_denoising_mode");
                if (!temporal_ready) {                                              // :216-219
                    temporal_reprojection.init(width, height);
                    temporal_ready = true;
                }
                temporal_reprojection.render(ctx, global_transform.affine_inverse(), projection_matrix);  // :220
                break;
            case NONE:
                check(ctx, jpt_set_denoising_mode(ctx, JPT_DENOISE_NONE), "jpt_set_denoising_mode");
                break;
        }
    }
    int create(int device) { return jpt_create(device, &ctx); }
    jpt_ctx* ctx = nullptr;
    GeometryGroup3D* geometry_group = nullptr;
    Transform3D global_transform;
    Projection projection_matrix;
    Denoising denoising_mode = PROGRESSIVE_RENDERING;
    float fov = 90.0f;  // path_tracing_camera.h:80
    int width = 0, height = 0;
    bool ready = false, temporal_ready = false, readback_queued = false;
};

// PathTracingCamera over several GPUs of one node, from this one process (jpt.h, jpt_multi_*): the scene is built once
// and shared with every device, each device renders its strips of the screen partition, rank 0 pulls the rows over
// xGMI and assembles them.  Progressive rendering (the default mode) and NONE; the temporal pass needs the whole image
// on one device (path_tracing_camera.cpp:215-221 reads neighbouring pixels' history).  The image is bit-identical to
// one GPU's.
class PathTracingCameraMulti {
  public:
    explicit PathTracingCameraMulti(const std::vector<int>& devices)
    {
        if (jpt_multi_create(devices.data(), (int)devices.size(), &multi) != JPT_OK)
            throw std::runtime_error(std::string("jpt_multi_create: ") + jpt_multi_last_error(nullptr));
    }
    ~PathTracingCameraMulti() { jpt_multi_destroy(multi); }
    PathTracingCameraMulti(const PathTracingCameraMulti&) = delete;
    PathTracingCameraMulti& operator=(const PathTracingCameraMulti&) = delete;

    void set_fov(float v) { fov = v; }
    void set_geometry_group(GeometryGroup3D* g) { geometry_group = g; }
    void set_global_transform(const Transform3D& t) { global_transform = t; }
    jpt_multi* handle() const { return multi; }
    int max_bounces = 4;
    int accum_mode = JPT_ACCUM_REF_LDR8;
    bool gather_display_rows_only = false;   // 4 B per pixel over the links instead of 16 (the sums stay on their ranks)

    void init(int w, int h)   // path_tracing_camera.cpp:111-187
    {
        if (!geometry_group) throw std::runtime_error("No geometry group set.");
        width = w;
        height = h;
        geometry_group->build(jpt_multi_ctx(multi, 0));     // one build ...
        mcheck(jpt_multi_share_scene(multi), "jpt_multi_share_scene");   // ... every device gets the arrays
        geometry_group->attach_multi(multi);                // ... and later update_transforms() calls reach every device
        projection_matrix = Projection::create_perspective(fov, float(width) / float(height), 0.01f, 1000.0f, false);
        mcheck(jpt_multi_set_params(multi, width, height, max_bounces, accum_mode, JPT_SAMPLER_NEAREST_CLAMP), "jpt_multi_set_params");
        mcheck(jpt_multi_set_gather(multi, gather_display_rows_only ? 1 : 0), "jpt_multi_set_gather");
        ready = true;
    }
    PackedByteArray render()  // path_tracing_camera.cpp:193-232, progressive mode
    {
        if (!ready) return {};
        camera.set_camera_transform(global_transform, projection_matrix);
        camera.frame_index++;
        mcheck(jpt_multi_set_camera(multi, &camera), "jpt_multi_set_camera");
        if (progressive_renderer.render(global_transform)) mcheck(jpt_multi_accum_reset(multi), "jpt_multi_accum_reset");
        mcheck(jpt_multi_render(multi, 1, camera.frame_index), "jpt_multi_render");
        PackedByteArray out((size_t)width * height * 4);
        mcheck(jpt_multi_read_ldr_rgba8(multi, out.data()), "jpt_multi_read_ldr_rgba8");
        return out;
    }
    Camera camera;
    ProgressiveRendering progressive_renderer;

  private:
    void mcheck(int rc, const char* what) const
    {
        if (rc != JPT_OK) throw std::runtime_error(std::string(what) + ": " + jpt_multi_last_error(multi));
    }
    jpt_multi* multi = nullptr;
    GeometryGroup3D* geometry_group = nullptr;
    Transform3D global_transform;
    Projection projection_matrix;
    float fov = 90.0f;
    int width = 0, height = 0;
    bool ready = false;
};

}  // namespace jpt_host
