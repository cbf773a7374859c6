// gdcs_adapter_test.cpp -- replays, call for call, what PathTracingCamera::init()/render() and
// ProgressiveRendering::init()/render() do with their ComputeShader objects (path_tracing_camera.cpp:139-232,
// progressive_rendering.cpp:25-65), against include/jpt_gdcs_adapter.hpp instantiated with std:: types.
// The six scene buffers come from the C++ host layer's GeometryGroup3D on a host-only context
// (REFERENCE_EXACT = what the addon's own builder emits).
//   gdcs_adapter_test <scene.bin> <prefix> <w> <h> <frames> [denoising_mode: 0 progressive, 1 temporal, 2 none]
#include <jpt_gdcs_adapter.hpp>
#include <jpt_host.hpp>

#include <cstdio>
#include <cstdlib>
#include <fstream>

using namespace jpt_host;

struct StdTraits {
    using Bytes = std::vector<uint8_t>;
    using RID = uint64_t;
    using String = std::string;
    static const uint8_t* ptr(const Bytes& b) { return b.data(); }
    static uint8_t* ptrw(Bytes& b) { return b.data(); }
    static size_t size(const Bytes& b) { return b.size(); }
    static void resize(Bytes& b, size_t n) { b.resize(n); }
    static bool contains(const String& s, const char* needle) { return s.find(needle) != std::string::npos; }
};
using CS = jpt_gdcs::ComputeShader<StdTraits>;

template <typename T> static std::vector<uint8_t> bytes_of(const T& v)
{
    std::vector<uint8_t> b(sizeof(T));
    std::memcpy(b.data(), &v, sizeof(T));
    return b;
}

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

    // ---- PathTracingCamera::init (path_tracing_camera.cpp:128-186)
    struct RenderParameters { float background[4]; int width, height; float fov; unsigned triangleCount, blasCount; } rp{};
    rp.width = w; rp.height = h; rp.fov = fov;
    rp.triangleCount = (unsigned)group.get_triangle_count(); rp.blasCount = (unsigned)group.get_blas_count();
    Camera camera;
    camera.frame_index = 0;
    const Projection projection = Projection::create_perspective(fov, float(w) / float(h), 0.01f, 1000.0f, false);
    auto dev = std::make_shared<jpt_gdcs::SharedDevice>();
    CS* cs = new CS("res://addons/jar_path_tracing/src/shaders/main.glsl", dev, {"#define TESTe"});
    cs->create_storage_buffer_uniform(bytes_of(rp), 2, 0);
    const uint64_t camera_rid = cs->create_storage_buffer_uniform(bytes_of(camera), 3, 0);
    const uint64_t output_texture_rid = cs->create_image_uniform(w, h, 0, 0);
    const uint64_t depth_texture_rid = cs->create_image_uniform(w, h, 1, 0);
    cs->create_storage_buffer_uniform(group.get_triangles_geometry_buffer(), 0, 1);
    cs->create_storage_buffer_uniform(group.get_triangles_data_buffer(), 1, 1);
    cs->create_storage_buffer_uniform(group.get_materials_buffer(), 2, 1);
    cs->create_storage_buffer_uniform(group.get_bvh_buffer(), 3, 1);
    cs->create_storage_buffer_uniform(group.get_blas_buffer(), 4, 1);
    cs->create_storage_buffer_uniform(group.get_tlas_buffer(), 5, 1);
    cs->create_layered_image_uniform({}, group.texture_array_resolution, 0, 2);
    cs->finish_create_uniforms();
    if (!cs->check_ready()) { std::fprintf(stderr, "main not ready: %s\n", cs->last_error().c_str()); return 5; }

    // ---- ProgressiveRendering::init (progressive_rendering.cpp:14-45)
    struct ProgParams { int width, height; unsigned frame_count; } pp{w, h, 1};
    CS* pcs = new CS("res://addons/jar_path_tracing/src/shaders/progressive_rendering.glsl", dev);
    const uint64_t pp_rid = pcs->create_storage_buffer_uniform(bytes_of(pp), 0, 0);
    pcs->add_existing_buffer(output_texture_rid, 0, 1, 0);
    pcs->create_image_uniform(w, h, 2, 0);
    pcs->finish_create_uniforms();
    ProgressiveRendering prog;  // host-side frame_count logic of jpt_host.hpp == progressive_rendering.cpp:53-60

    // ---- TemporalReprojection::init (temporal_reprojection.cpp:16-54), created on first use like :216-219
    CS* tcs = nullptr;
    uint64_t tp_rid = 0;
    TemporalReprojection temporal;  // host half of jpt_host.hpp: previous_vp, frame_count, deltaMatrix

    std::vector<uint8_t> screen;
    Transform3D t = cam_t;
    for (int f = 0; f < frames; f++) {  // PathTracingCamera::render (path_tracing_camera.cpp:193-232)
        if (!cs->check_ready()) return 6;
        if (denoise != 0 && f > 0) {    // same camera path as host_demo.cpp
            t.origin.x += 0.05f;
            t.origin.y += 0.01f * (float)f;
        }
        camera.set_camera_transform(t, projection);
        camera.frame_index++;
        cs->update_storage_buffer_uniform(camera_rid, bytes_of(camera));
        cs->compute({(w + 31) / 32, (h + 31) / 32, 1});
        if (denoise == 0) {
            prog.render(t);  // progressive_renderer->render(get_global_transform())
            pp.frame_count = prog.frame_count;
            pcs->update_storage_buffer_uniform(pp_rid, bytes_of(pp));
            pcs->compute({(w + 31) / 32, (h + 31) / 32, 1});
        } else if (denoise == 1) {
            if (!tcs) {
                temporal.init(w, h);
                tcs = new CS("res://addons/jar_path_tracing/src/shaders/temporal_reprojection.glsl", dev);
                tp_rid = tcs->create_storage_buffer_uniform(bytes_of(temporal.render_parameters), 0, 0);
                tcs->add_existing_buffer(output_texture_rid, 0, 1, 0);
                tcs->add_existing_buffer(depth_texture_rid, 0, 2, 0);
                tcs->create_image_uniform(w, h, 3, 0);
                tcs->create_image_uniform(w, h, 4, 0);
                tcs->finish_create_uniforms();
            }
            if (!tcs->check_ready()) return 8;
            temporal.render(host_ctx, t.affine_inverse(), projection);   // fills render_parameters (:62-66)
            tcs->update_storage_buffer_uniform(tp_rid, bytes_of(temporal.render_parameters));
            tcs->compute({(w + 31) / 32, (h + 31) / 32, 1});
            if (!tcs->last_error().empty()) { std::fprintf(stderr, "temporal: %s\n", tcs->last_error().c_str()); return 9; }
        }
        screen = cs->get_image_uniform_buffer(output_texture_rid);
        if (denoise != 0) {
            std::ofstream(prefix + "_camera_f" + std::to_string(f) + ".bin", std::ios::binary).write(reinterpret_cast<const char*>(&camera), sizeof camera);
            std::ofstream(prefix + "_tp_f" + std::to_string(f) + ".bin", std::ios::binary)
                .write(reinterpret_cast<const char*>(&temporal.render_parameters), sizeof temporal.render_parameters);
        }
    }
    std::vector<float> accum((size_t)w * h * 4);
    if (denoise != 2 && jpt_read_accum_f32(dev->ctx, accum.data()) != JPT_OK) return 7;
    std::ofstream(prefix + "_accum.bin", std::ios::binary).write(reinterpret_cast<const char*>(accum.data()), (std::streamsize)(accum.size() * 4));
    std::ofstream(prefix + "_ldr.bin", std::ios::binary).write(reinterpret_cast<const char*>(screen.data()), (std::streamsize)screen.size());
    std::ofstream(prefix + "_camera.bin", std::ios::binary).write(reinterpret_cast<const char*>(&camera), sizeof camera);
    std::printf("adapter rendered %d frames, frame_count %u\n", frames, prog.frame_count);
    delete tcs;
    delete pcs;
    delete cs;
    jpt_destroy(host_ctx);
    return 0;
}
