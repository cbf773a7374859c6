// gdcs_adapter_test.cpp -- drives include/jpt_gdcs_adapter.hpp exactly the way the reference drives gdcs::ComputeShader:
// the bodies of Replay::PathTracingCamera::init()/render(), Replay::ProgressiveRendering::init()/render() and
// Replay::TemporalReprojection::init()/render() below make the ComputeShader calls of path_tracing_camera.cpp:139-186,
// 193-232, progressive_rendering.cpp:22-45,53-65 and temporal_reprojection.cpp:27-50,59-68 in the same order with the
// same argument shapes (Ref<Image>, Ref<RDTextureFormat>, Ref<RDTextureView>, RenderingDevice*, RID, braced group counts,
// `{"#define TESTe"}`), over small stand-ins of the godot-cpp types (godot-cpp itself is an absent submodule).  No call
// in those bodies is adapter-specific.  The scene's six byte buffers come from jpt_host.hpp's GeometryGroup3D on a
// host-only context (REFERENCE_EXACT = what the addon's own builder emits).
//   gdcs_adapter_test <scene.bin> <prefix> <w> <h> <frames> [denoising_mode: 0 progressive, 1 temporal, 2 none]
#include <jpt_gdcs_adapter.hpp>
#include <jpt_host.hpp>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <memory>

// ---- stand-ins for the godot-cpp types the call sites name -------------------------------------------------------
namespace godot_stub {

using PackedByteArray = jpt_host::PackedByteArray;
using String = std::string;
using jpt_host::Projection;
using jpt_host::Transform3D;

template <class T>
class Ref {
  public:
    Ref() = default;
    Ref(T* raw) : p_(raw) {}  // `Ref<RDTextureView> v = memnew(RDTextureView);`
    T* operator->() const { return p_.get(); }
    bool is_valid() const { return (bool)p_; }

  private:
    std::shared_ptr<T> p_;
};
#define memnew(T) (new T)

struct RID {
    uint64_t id = 0;
    uint64_t get_id() const { return id; }
};
struct Vector2i {
    int x = 0, y = 0;
};
struct RDTextureView {};
struct RDTextureFormat {
    int width = 0, height = 0, format = 0;
};
struct RenderingDevice {
    enum DataFormat { DATA_FORMAT_R8G8B8A8_UNORM = 37, DATA_FORMAT_R32_SFLOAT = 100, DATA_FORMAT_R32G32B32A32_SFLOAT = 109 };
    enum UniformType { UNIFORM_TYPE_IMAGE = 3 };
};
class Image {
  public:
    enum Format { FORMAT_RGBA8 = 5, FORMAT_RF = 8, FORMAT_RGBAF = 11 };
    static Ref<Image> create(int w, int h, bool /*mipmaps*/, Format f)
    {
        Image* im = new Image;
        im->w_ = w;
        im->h_ = h;
        im->f_ = f;
        im->data_.assign((size_t)w * h * (f == FORMAT_RGBA8 ? 4 : f == FORMAT_RF ? 4 : 16), 0);
        return Ref<Image>(im);
    }
    int get_width() const { return w_; }
    int get_height() const { return h_; }
    PackedByteArray get_data() const { return data_; }
    void set_data(int w, int h, bool, Format f, const PackedByteArray& d) { w_ = w; h_ = h; f_ = f; data_ = d; }

  private:
    int w_ = 0, h_ = 0;
    Format f_ = FORMAT_RGBA8;
    PackedByteArray data_;
};
class ImageTexture {
  public:
    static Ref<ImageTexture> create_from_image(const Ref<Image>& im)
    {
        ImageTexture* t = new ImageTexture;
        t->image = im;
        return Ref<ImageTexture>(t);
    }
    void update(const Ref<Image>& im) { image = im; }
    Ref<Image> image;
};

// what INTEGRATION.md section 4 calls GodotTraits, over the stand-ins
struct Traits {
    using Bytes = PackedByteArray;
    using RID = godot_stub::RID;
    using String = godot_stub::String;
    using Device = RenderingDevice;
    using ImageRef = Ref<Image>;
    using TextureFormatRef = Ref<RDTextureFormat>;
    using TextureViewRef = Ref<RDTextureView>;
    using DataFormat = RenderingDevice::DataFormat;
    using UniformType = RenderingDevice::UniformType;
    static const uint8_t* ptr(const Bytes& b) { return b.data(); }
    static uint8_t* ptrw(Bytes& b) { return b.data(); }
    static size_t size(const Bytes& b) { return b.size(); }
    static void resize(Bytes& b, size_t n) { b.resize(n); }
    static bool contains(const String& s, const char* needle) { return s.find(needle) != std::string::npos; }
    static std::string to_std(const String& s) { return s; }
    static RID make_rid(uint64_t id) { return RID{id}; }
    static uint64_t rid_id(const RID& r) { return r.get_id(); }
    static TextureFormatRef make_texture_format(int w, int h, DataFormat f)
    {
        RDTextureFormat* t = new RDTextureFormat;
        t->width = w;
        t->height = h;
        t->format = (int)f;
        return TextureFormatRef(t);
    }
    static int format_width(const TextureFormatRef& f) { return f->width; }
    static int format_height(const TextureFormatRef& f) { return f->height; }
    static int image_width(const ImageRef& i) { return i->get_width(); }
    static int image_height(const ImageRef& i) { return i->get_height(); }
    static Bytes image_data(const ImageRef& i) { return i->get_data(); }
};

}  // namespace godot_stub

using namespace godot_stub;
using ComputeShader = jpt_gdcs::ComputeShader<godot_stub::Traits>;  // what replaces #include "gdcs/include/gdcs.h"

template <typename T> static PackedByteArray bytes_of(const T& v)
{
    PackedByteArray b(sizeof(T));
    std::memcpy(b.data(), &v, sizeof(T));
    return b;
}

// ---- the reference's three classes, reduced to their ComputeShader traffic ----------------------------------------
namespace Replay {

// geometry_group3d.h:82-88 over jpt_host's GeometryGroup3D (whose textures are raw RGBA8 layers)
struct GeometryGroup {
    jpt_host::GeometryGroup3D* g = nullptr;
    PackedByteArray get_triangles_geometry_buffer() { return g->get_triangles_geometry_buffer(); }
    PackedByteArray get_triangles_data_buffer() { return g->get_triangles_data_buffer(); }
    PackedByteArray get_materials_buffer() { return g->get_materials_buffer(); }
    PackedByteArray get_bvh_buffer() { return g->get_bvh_buffer(); }
    PackedByteArray get_blas_buffer() { return g->get_blas_buffer(); }
    PackedByteArray get_tlas_buffer() { return g->get_tlas_buffer(); }
    int get_triangle_count() { return g->get_triangle_count(); }
    int get_blas_count() { return g->get_blas_count(); }
    int get_texture_array_resolution() const { return g->texture_array_resolution; }
    std::vector<Ref<Image>> get_textures_buffer()
    {
        std::vector<Ref<Image>> out;
        const int res = g->texture_array_resolution;
        for (const PackedByteArray& layer : g->textures) {
            Ref<Image> im = Image::create(res, res, false, Image::FORMAT_RGBA8);
            im->set_data(res, res, false, Image::FORMAT_RGBA8, layer);
            out.push_back(im);
        }
        if (out.empty()) out.push_back(Image::create(res, res, false, Image::FORMAT_RGBA8));  // the blank layer of geometry_group3d.cpp:301-303
        return out;
    }
};

class ProgressiveRendering {  // progressive_rendering.{h,cpp}
    struct RenderParameters {
        int width;
        int height;
        unsigned int frame_count;
        PackedByteArray to_packed_byte_array() { return bytes_of(*this); }
    };

  public:
    ~ProgressiveRendering() { delete cs; }
    void init(RenderingDevice* rd, const RID original_screen_texture_rid, const Vector2i size)
    {
        screen_texture_rid = original_screen_texture_rid;
        render_parameters.width = size.x;
        render_parameters.height = size.y;
        render_parameters.frame_count = 1;
        cs = new ComputeShader("res://addons/jar_path_tracing/src/shaders/progressive_rendering.glsl", rd);
        render_parameters_rid = cs->create_storage_buffer_uniform(render_parameters.to_packed_byte_array(), 0, 0);
        cs->add_existing_buffer(screen_texture_rid, RenderingDevice::UNIFORM_TYPE_IMAGE, 1, 0);
        auto frame_buffer_format = cs->create_texture_format(size.x, size.y, RenderingDevice::DATA_FORMAT_R32G32B32A32_SFLOAT);
        Ref<RDTextureView> frame_buffer_texture_view = memnew(RDTextureView);
        frame_buffer_image = Image::create(size.x, size.y, false, Image::FORMAT_RGBAF);
        frame_buffer_texture = ImageTexture::create_from_image(frame_buffer_image);
        frame_buffer_rid = cs->create_image_uniform(frame_buffer_image, frame_buffer_format, frame_buffer_texture_view, 2, 0);
        cs->finish_create_uniforms();
    }
    void render(Transform3D camera_transform)
    {
        if (cs == nullptr || !cs->check_ready()) return;
        bool camera_moved = !previous_transform.is_equal_approx(camera_transform);
        previous_transform = camera_transform;
        if (camera_moved) render_parameters.frame_count = 1;
        else render_parameters.frame_count++;
        cs->update_storage_buffer_uniform(render_parameters_rid, render_parameters.to_packed_byte_array());
        Vector2i Size = {render_parameters.width, render_parameters.height};
        cs->compute({static_cast<int32_t>(std::ceil(Size.x / 32.0f)), static_cast<int32_t>(std::ceil(Size.y / 32.0f)), 1});
    }
    unsigned int frame_count() const { return render_parameters.frame_count; }

  private:
    ComputeShader* cs = nullptr;
    Ref<Image> frame_buffer_image;
    Ref<ImageTexture> frame_buffer_texture;
    RenderParameters render_parameters;
    Transform3D previous_transform;
    RID render_parameters_rid, screen_texture_rid, frame_buffer_rid;
};

class TemporalReprojection {  // temporal_reprojection.{h,cpp}
  public:
    struct RenderParameters {
        float deltaMatrix[16];
        int width;
        int height;
        unsigned int frame_count;
        float blendFactor = 0.75f;
        float nearPlane = 0.01f;
        float farPlane = 1000.0f;
        PackedByteArray to_packed_byte_array() { return bytes_of(*this); }
    };
    ~TemporalReprojection() { delete cs; }
    void init(RenderingDevice* rd, const RID original_screen_texture_rid, const RID original_depth_texture_rid, const Vector2i size)
    {
        screen_texture_rid = original_screen_texture_rid;
        render_parameters.width = size.x;
        render_parameters.height = size.y;
        render_parameters.frame_count = 1;
        cs = new ComputeShader("res://addons/jar_path_tracing/src/shaders/temporal_reprojection.glsl", rd);
        render_parameters_rid = cs->create_storage_buffer_uniform(render_parameters.to_packed_byte_array(), 0, 0);
        cs->add_existing_buffer(screen_texture_rid, RenderingDevice::UNIFORM_TYPE_IMAGE, 1, 0);
        cs->add_existing_buffer(original_depth_texture_rid, RenderingDevice::UNIFORM_TYPE_IMAGE, 2, 0);
        auto frame_buffer_format = cs->create_texture_format(size.x, size.y, RenderingDevice::DATA_FORMAT_R32G32B32A32_SFLOAT);
        Ref<RDTextureView> frame_buffer_texture_view = memnew(RDTextureView);
        frame_buffer_image_1 = Image::create(size.x, size.y, false, Image::FORMAT_RGBAF);
        frame_buffer_image_2 = Image::create(size.x, size.y, false, Image::FORMAT_RGBAF);
        frame_buffer_texture_1 = ImageTexture::create_from_image(frame_buffer_image_1);
        frame_buffer_texture_2 = ImageTexture::create_from_image(frame_buffer_image_2);
        frame_buffer_rid_1 = cs->create_image_uniform(frame_buffer_image_1, frame_buffer_format, frame_buffer_texture_view, 3, 0);
        frame_buffer_rid_2 = cs->create_image_uniform(frame_buffer_image_2, frame_buffer_format, frame_buffer_texture_view, 4, 0);
        cs->finish_create_uniforms();
    }
    void render(Transform3D view_matrix, Projection projection_matrix)
    {
        if (cs == nullptr || !cs->check_ready()) return;
        Projection vp = projection_matrix * Projection(view_matrix);
        // `Transform3D deltaMatrix = previous_vp * vp.inverse();` then projection_to_float(deltaMatrix): the conversion
        // drops the projective row and writes 0 0 0 1 back (temporal_reprojection.cpp:61-64)
        const Projection delta((previous_vp * vp.inverse()).to_transform3d());
        previous_vp = vp;
        render_parameters.frame_count++;
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) render_parameters.deltaMatrix[i * 4 + j] = delta.columns[i][j];
        cs->update_storage_buffer_uniform(render_parameters_rid, render_parameters.to_packed_byte_array());
        Vector2i Size = {render_parameters.width, render_parameters.height};
        cs->compute({static_cast<int32_t>(std::ceil(Size.x / 32.0f)), static_cast<int32_t>(std::ceil(Size.y / 32.0f)), 1});
        error = cs->last_error();
    }
    RenderParameters render_parameters;
    std::string error;

  private:
    ComputeShader* cs = nullptr;
    Ref<Image> frame_buffer_image_1, frame_buffer_image_2;
    Ref<ImageTexture> frame_buffer_texture_1, frame_buffer_texture_2;
    Projection previous_vp;
    RID render_parameters_rid, screen_texture_rid, frame_buffer_rid_1, frame_buffer_rid_2;
};

class PathTracingCamera {  // path_tracing_camera.{h,cpp}
    struct RenderParameters {  // path_tracing_camera.h:36-52
        float backgroundColor[4];
        int width;
        int height;
        float fov;
        unsigned int triangleCount;
        unsigned int blasCount;
        PackedByteArray to_packed_byte_array() { return bytes_of(*this); }
    };

  public:
    enum Denoising { PROGRESSIVE_RENDERING, TEMPORAL_REPROJECTION, NONE };
    ~PathTracingCamera()
    {
        delete progressive_renderer;
        delete temporal_reprojection;
        delete cs;  // (the reference leaks this one, path_tracing_camera.cpp:189-191)
    }
    void init(Vector2i resolution)
    {
        _rd = &rendering_device;  // RenderingServer::get_singleton()->create_local_rendering_device()
        std::memset(&render_parameters, 0, sizeof render_parameters);
        render_parameters.width = resolution.x;
        render_parameters.height = resolution.y;
        render_parameters.fov = fov;
        render_parameters.triangleCount = geometry_group->get_triangle_count();
        render_parameters.blasCount = geometry_group->get_blas_count();
        projection_matrix = Projection::create_perspective(fov, static_cast<float>(render_parameters.width) / render_parameters.height, 0.01f, 1000.0f, false);
        camera.frame_index = 0;  // uninitialised in the reference (render_parameters.h:19)
        camera.set_camera_transform(global_transform.affine_inverse(), projection_matrix);

        cs = new ComputeShader("res://addons/jar_path_tracing/src/shaders/main.glsl", _rd, {"#define TESTe"});
        render_parameters_rid = cs->create_storage_buffer_uniform(render_parameters.to_packed_byte_array(), 2, 0);
        camera_rid = cs->create_storage_buffer_uniform(bytes_of(camera), 3, 0);

        Ref<RDTextureView> output_texture_view = memnew(RDTextureView);
        {
            auto output_format = cs->create_texture_format(render_parameters.width, render_parameters.height, RenderingDevice::DATA_FORMAT_R8G8B8A8_UNORM);
            output_image = Image::create(render_parameters.width, render_parameters.height, false, Image::FORMAT_RGBA8);
            output_texture = ImageTexture::create_from_image(output_image);
            output_texture_rid = cs->create_image_uniform(output_image, output_format, output_texture_view, 0, 0);
        }
        Ref<RDTextureView> depth_texture_view = memnew(RDTextureView);
        {
            auto depth_format = cs->create_texture_format(render_parameters.width, render_parameters.height, RenderingDevice::DATA_FORMAT_R32_SFLOAT);
            depth_image = Image::create(render_parameters.width, render_parameters.height, false, Image::FORMAT_RF);
            depth_texture_rid = cs->create_image_uniform(depth_image, depth_format, depth_texture_view, 1, 0);
        }
        {
            triangles_geometry_rid = cs->create_storage_buffer_uniform(geometry_group->get_triangles_geometry_buffer(), 0, 1);
            triangles_data_rid = cs->create_storage_buffer_uniform(geometry_group->get_triangles_data_buffer(), 1, 1);
            materials_rid = cs->create_storage_buffer_uniform(geometry_group->get_materials_buffer(), 2, 1);
            bvh_tree_rid = cs->create_storage_buffer_uniform(geometry_group->get_bvh_buffer(), 3, 1);
            blas_rid = cs->create_storage_buffer_uniform(geometry_group->get_blas_buffer(), 4, 1);
            tlas_rid = cs->create_storage_buffer_uniform(geometry_group->get_tlas_buffer(), 5, 1);
        }
        {
            Ref<RDTextureView> texture_view = memnew(RDTextureView);
            auto textures = geometry_group->get_textures_buffer();
            auto resolution = geometry_group->get_texture_array_resolution();
            auto textures_format = cs->create_texture_format(resolution, resolution, RenderingDevice::DATA_FORMAT_R8G8B8A8_UNORM);
            texture_array_rid = cs->create_layered_image_uniform(textures, textures_format, texture_view, 0, 2);
        }
        cs->finish_create_uniforms();
    }
    void render()
    {
        if (cs == nullptr || !cs->check_ready()) return;
        camera.set_camera_transform(global_transform, projection_matrix);
        camera.frame_index++;
        cs->update_storage_buffer_uniform(camera_rid, bytes_of(camera));

        Vector2i Size = {render_parameters.width, render_parameters.height};
        cs->compute({static_cast<int32_t>(std::ceil(Size.x / 32.0f)), static_cast<int32_t>(std::ceil(Size.y / 32.0f)), 1});
        switch (denoising_mode) {
            case PROGRESSIVE_RENDERING:
                if (progressive_renderer == nullptr) {
                    progressive_renderer = new ProgressiveRendering();
                    progressive_renderer->init(_rd, output_texture_rid, Size);
                }
                progressive_renderer->render(global_transform);
                break;
            case TEMPORAL_REPROJECTION:
                if (temporal_reprojection == nullptr) {
                    temporal_reprojection = new TemporalReprojection();
                    temporal_reprojection->init(_rd, output_texture_rid, depth_texture_rid, Size);
                }
                temporal_reprojection->render(global_transform.affine_inverse(), projection_matrix);
                break;
            case NONE: break;
        }
        output_image->set_data(Size.x, Size.y, false, Image::FORMAT_RGBA8, cs->get_image_uniform_buffer(output_texture_rid));
        output_texture->update(output_image);
    }

    GeometryGroup* geometry_group = nullptr;
    Transform3D global_transform;
    float fov = 90.0f;
    Denoising denoising_mode = PROGRESSIVE_RENDERING;
    jpt_host::Camera camera;
    ComputeShader* cs = nullptr;
    ProgressiveRendering* progressive_renderer = nullptr;
    TemporalReprojection* temporal_reprojection = nullptr;
    Ref<Image> output_image, depth_image;

  private:
    RenderingDevice rendering_device;
    RenderingDevice* _rd = nullptr;
    RenderParameters render_parameters;
    Projection projection_matrix;
    Ref<ImageTexture> output_texture;
    RID render_parameters_rid, camera_rid, output_texture_rid, depth_texture_rid, triangles_geometry_rid, triangles_data_rid, materials_rid,
        bvh_tree_rid, blas_rid, tlas_rid, texture_array_rid;
};

}  // namespace Replay

// the scene-file reader of host_demo.cpp, reduced
struct Reader {
    std::ifstream f;
    explicit Reader(const char* p) : f(p, std::ios::binary) {}
    template <typename T> T get() { T v; f.read(reinterpret_cast<char*>(&v), sizeof v); return v; }
    template <typename T> std::vector<T> vec(size_t n) { std::vector<T> v(n); f.read(reinterpret_cast<char*>(v.data()), (std::streamsize)(n * sizeof(T))); return v; }
};
static Transform3D read_transform(Reader& r)
{
    Transform3D t;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) t.basis[i][j] = r.get<float>();
    t.origin.x = r.get<float>(); t.origin.y = r.get<float>(); t.origin.z = r.get<float>();
    return t;
}

int main(int argc, char** argv)
{
    using namespace jpt_host;
    if (argc < 6) return 2;
    Reader r(argv[1]);
    const std::string prefix = argv[2];
    const int w = std::atoi(argv[3]), h = std::atoi(argv[4]), frames = std::atoi(argv[5]);
    const int denoise = argc >= 7 ? std::atoi(argv[6]) : 0;
    if (r.get<uint32_t>() != 0x5354504au) return 3;
    std::vector<std::unique_ptr<ArrayMesh>> meshes;
    for (uint32_t m = r.get<uint32_t>(); m > 0; m--) {
        auto mesh = std::make_unique<ArrayMesh>();
        for (uint32_t s = r.get<uint32_t>(); s > 0; s--) {
            const uint32_t nv = r.get<uint32_t>(), ni = r.get<uint32_t>();
            Surface su;
            su.vertices = r.vec<float>(3 * nv); su.normals = r.vec<float>(3 * nv); su.uvs = r.vec<float>(2 * nv); su.indices = r.vec<int32_t>(ni);
            mesh->surfaces.push_back(std::move(su));
        }
        meshes.push_back(std::move(mesh));
    }
    std::vector<StandardMaterial3D> mats(r.get<uint32_t>());
    for (auto& m : mats) {
        m.albedo.r = r.get<float>(); m.albedo.g = r.get<float>(); m.albedo.b = r.get<float>();
        m.metallic = r.get<float>(); m.roughness = r.get<float>();
        m.emission.r = r.get<float>(); m.emission.g = r.get<float>(); m.emission.b = r.get<float>();
        m.emission_energy_multiplier = r.get<float>(); m.albedo_texture = r.get<int32_t>();
    }
    GeometryGroup3D group;
    group.builder = JPT_BUILD_REFERENCE_EXACT;
    if (!mats.empty()) group.set_default_material(mats[0]);
    for (uint32_t i = r.get<uint32_t>(); i > 0; i--) {
        MeshInstance3D node;
        node.mesh = meshes.at(r.get<uint32_t>()).get();
        node.global_transform = read_transform(r);
        for (uint32_t k = r.get<uint32_t>(); k > 0; k--) {
            const int32_t id = r.get<int32_t>();
            node.surface_override_materials.push_back(id > 0 ? &mats.at((size_t)id) : nullptr);
        }
        group.add_child(node);
    }
    const Transform3D cam_t = read_transform(r);
    const float fov = r.get<float>();
    jpt_ctx* host_ctx = nullptr;
    if (jpt_create(JPT_DEVICE_HOST_ONLY, &host_ctx) != JPT_OK) return 4;
    group.build(host_ctx);  // geometry_group->build()  (path_tracing_camera.cpp:126)

    Replay::GeometryGroup gg{&group};
    Replay::PathTracingCamera cam;
    cam.geometry_group = &gg;
    cam.fov = fov;
    cam.global_transform = cam_t;
    cam.denoising_mode = (Replay::PathTracingCamera::Denoising)denoise;
    cam.init({w, h});
    if (!cam.cs || !cam.cs->check_ready()) { std::fprintf(stderr, "main not ready: %s\n", cam.cs ? cam.cs->last_error().c_str() : "no shader"); return 5; }

    for (int f = 0; f < frames; f++) {  // one NOTIFICATION_INTERNAL_PROCESS per frame (path_tracing_camera.cpp:54-57)
        if (denoise != 0 && f > 0) {    // same camera path as host_demo.cpp
            cam.global_transform.origin.x += 0.05f;
            cam.global_transform.origin.y += 0.01f * (float)f;
        }
        cam.render();
        if (!cam.cs->last_error().empty()) { std::fprintf(stderr, "main: %s\n", cam.cs->last_error().c_str()); return 6; }
        if (cam.temporal_reprojection && !cam.temporal_reprojection->error.empty()) { std::fprintf(stderr, "temporal: %s\n", cam.temporal_reprojection->error.c_str()); return 9; }
        if (denoise != 0) {
            std::ofstream(prefix + "_camera_f" + std::to_string(f) + ".bin", std::ios::binary).write(reinterpret_cast<const char*>(&cam.camera), sizeof cam.camera);
            TemporalReprojection::RenderParameters none;  // (host layer's struct: same 88 bytes) modes without the pass write defaults
            const void* tp = cam.temporal_reprojection ? (const void*)&cam.temporal_reprojection->render_parameters : (const void*)&none;
            std::ofstream(prefix + "_tp_f" + std::to_string(f) + ".bin", std::ios::binary).write(reinterpret_cast<const char*>(tp), 88);
        }
    }
    const PackedByteArray screen = cam.output_image->get_data();
    std::vector<float> accum((size_t)w * h * 4);
    if (denoise != 2 && jpt_read_accum_f32(cam.cs->context(), accum.data()) != JPT_OK) return 7;
    std::ofstream(prefix + "_accum.bin", std::ios::binary).write(reinterpret_cast<const char*>(accum.data()), (std::streamsize)(accum.size() * 4));
    std::ofstream(prefix + "_ldr.bin", std::ios::binary).write(reinterpret_cast<const char*>(screen.data()), (std::streamsize)screen.size());
    std::ofstream(prefix + "_camera.bin", std::ios::binary).write(reinterpret_cast<const char*>(&cam.camera), sizeof cam.camera);
    std::printf("adapter rendered %d frames, frame_count %u\n", frames, cam.progressive_renderer ? cam.progressive_renderer->frame_count() : 0u);
    jpt_destroy(host_ctx);
    return 0;
}
