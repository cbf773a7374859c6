// jpt_gdcs_adapter.hpp -- a `ComputeShader`-shaped adapter over the C ABI, so that PathTracingCamera,
// ProgressiveRendering and TemporalReprojection of the reference compile against it with their call sites AS WRITTEN
// (SURVEY.md 8(f)-2).
//
// The reference drives its GPU work through `gdcs::ComputeShader` (submodule src/gdcs, absent).  Its contract is
// recoverable from the call sites; every method below has the parameter list of the call it serves:
//
//   ComputeShader(res_path, RenderingDevice*, std::vector<String> defines = {})   path_tracing_camera.cpp:139, progressive_rendering.cpp:25
//   RID  create_storage_buffer_uniform(PackedByteArray, binding, set)            path_tracing_camera.cpp:142-143,170-175
//   Ref<RDTextureFormat> create_texture_format(w, h, RenderingDevice::DataFormat)  :148,163,182; progressive_rendering.cpp:35
//   RID  create_image_uniform(Ref<Image>, Ref<RDTextureFormat>, Ref<RDTextureView>, binding, set)          :158,165
//   RID  create_layered_image_uniform(std::vector<Ref<Image>>, format, view, binding, set)                 :183
//   void add_existing_buffer(RID, RenderingDevice::UniformType, binding, set)    progressive_rendering.cpp:30
//   void finish_create_uniforms()                                                :186
//   bool check_ready()                                                           :195
//   void update_storage_buffer_uniform(RID, PackedByteArray)                     :200
//   void compute({gx, gy, gz})                                                   :204
//   PackedByteArray get_image_uniform_buffer(RID)                                :229
//
// The adapter is a template over a traits struct so it needs no Godot header: with godot-cpp the traits map to
// godot::PackedByteArray / RID / String / Ref<Image> / Ref<RDTextureFormat> / Ref<RDTextureView> / RenderingDevice
// (INTEGRATION.md section 4); the repository's test (tests/cpp/gdcs_adapter_test.cpp) instantiates it with small
// stand-ins of those types and replays the reference's init()/render() bodies call for call.  Descriptor roles are
// taken from the shaders' layouts (main.glsl:98-155, progressive_rendering.glsl:5-16):
//   main.glsl          set 0: b0 rgba8 out, b1 r32f depth, b2 Params, b3 Camera;  set 1: b0..b5 scene buffers;
//                      set 2: b0 texture array
//   progressive.glsl   set 0: b0 Params{w,h,frame_count}, b1 screen image (shared), b2 rgba32f frame buffer
//   temporal_reprojection.glsl  set 0: b0 RenderParameters (88 B), b1 screen image (shared), b2 depth image (shared),
//                      b3 / b4 the two rgba32f history images (temporal_reprojection.cpp:32-49)
//
// All ComputeShaders created on one RenderingDevice share one jpt context (the reference creates the three of them on
// the camera's local RenderingDevice, path_tracing_camera.cpp:114,139,211,218): the context hangs off the device
// pointer in a registry kept by this header and is destroyed with the last ComputeShader that uses it.
//
// Three `defines` strings are understood (anything else is ignored, like the reference's "#define TESTe"):
//   "#define DEBUG_STEPS"         the shader's own debug build (main.glsl:4,358-361): triangle tests of the primary ray / 256
//   "#define JPT_MAX_BOUNCES n"   path length (default 4 = the literal 5 of main.glsl:377)
//   "#define JPT_SAMPLER n"       JPT_SAMPLER_* of jpt.h for texture(textureArray, ...) (default NEAREST_CLAMP)
#pragma once

#include <jpt.h>

#include <array>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace jpt_gdcs {

// One jpt context shared by the main pass and the post-processing passes that add_existing_buffer() its images.
struct SharedDevice {
    jpt_ctx* ctx = nullptr;
    const void* key = nullptr;   // the RenderingDevice this context stands for
    int width = 0, height = 0;
    bool scene_ready = false, params_ready = false;
    uint32_t camera_frame_index = 0;
    uint32_t progressive_frame_count = 1;  // Params.frame_count as ProgressiveRendering::render last uploaded it
    bool frame_pending = false;  // main.glsl was dispatched, the frame has not been rendered yet
    ~SharedDevice();
};

// device pointer -> context; weak, so the context goes with the last ComputeShader on that device.  Guarded by a mutex
// (Godot creates local rendering devices from any thread); an entry is erased when its context is destroyed, so a
// RenderingDevice allocated later at the same address starts with a context of its own.
struct DeviceRegistry {
    std::mutex lock;
    std::map<const void*, std::weak_ptr<SharedDevice>> entries;
    static DeviceRegistry& get()
    {
        static DeviceRegistry r;
        return r;
    }
};
inline SharedDevice::~SharedDevice()
{
    {
        DeviceRegistry& r = DeviceRegistry::get();
        std::lock_guard<std::mutex> g(r.lock);
        auto it = r.entries.find(key);
        if (it != r.entries.end() && it->second.expired()) r.entries.erase(it);
    }
    jpt_destroy(ctx);
}
inline std::shared_ptr<SharedDevice> shared_device_of(const void* rendering_device)
{
    DeviceRegistry& r = DeviceRegistry::get();
    std::lock_guard<std::mutex> g(r.lock);
    std::shared_ptr<SharedDevice> d = r.entries[rendering_device].lock();
    if (!d) {
        d = std::make_shared<SharedDevice>();
        d->key = rendering_device;
        r.entries[rendering_device] = d;
    }
    return d;
}

template <class Traits>
class ComputeShader {
  public:
    using Bytes = typename Traits::Bytes;                  // PackedByteArray
    using RID = typename Traits::RID;
    using String = typename Traits::String;
    using Device = typename Traits::Device;                // RenderingDevice
    using ImageRef = typename Traits::ImageRef;            // Ref<Image>
    using TextureFormatRef = typename Traits::TextureFormatRef;  // Ref<RDTextureFormat>
    using TextureViewRef = typename Traits::TextureViewRef;      // Ref<RDTextureView>
    using DataFormat = typename Traits::DataFormat;        // RenderingDevice::DataFormat
    using UniformType = typename Traits::UniformType;      // RenderingDevice::UniformType

    ComputeShader(const String& res_path, Device* rd, const std::vector<String>& defines = {})
        : dev_(shared_device_of(rd)),
          progressive_(Traits::contains(res_path, "progressive_rendering")),
          temporal_(Traits::contains(res_path, "temporal_reprojection"))
    {
        for (const String& d : defines) {
            const std::string s = Traits::to_std(d);
            read_define(s, "JPT_MAX_BOUNCES", max_bounces_);
            read_define(s, "JPT_SAMPLER", sampler_);
            if (s.find("#define DEBUG_STEPS") != std::string::npos) debug_steps_ = true;   // the shader's own switch (main.glsl:4)
        }
        if (!dev_->ctx && jpt_create(0, &dev_->ctx) != JPT_OK) error_ = jpt_last_error(nullptr);
    }

    RID create_storage_buffer_uniform(const Bytes& data, int binding, int set)
    {
        const uint64_t id = next_id_++;
        Slot s{set, binding, std::vector<uint8_t>(Traits::ptr(data), Traits::ptr(data) + Traits::size(data))};
        slots_.emplace_back(id, std::move(s));
        return Traits::make_rid(id);
    }
    TextureFormatRef create_texture_format(int width, int height, DataFormat format)
    {
        return Traits::make_texture_format(width, height, format);
    }
    // images: the library owns the device images; what matters is the size of main.glsl's output image
    RID create_image_uniform(const ImageRef& image, const TextureFormatRef& format, const TextureViewRef& /*view*/, int binding, int set)
    {
        if (!progressive_ && !temporal_ && set == 0 && binding == 0) {
            dev_->width = Traits::format_width(format);
            dev_->height = Traits::format_height(format);
            if (Traits::image_width(image) != dev_->width || Traits::image_height(image) != dev_->height)
                error_ = "output image and its texture format differ in size";
        }
        return Traits::make_rid(next_id_++);
    }
    RID create_layered_image_uniform(const std::vector<ImageRef>& layers, const TextureFormatRef& format, const TextureViewRef& /*view*/,
                                     int /*binding*/, int /*set*/)
    {
        tex_res_ = Traits::format_width(format);
        tex_layers_ = (int)layers.size();
        tex_.clear();
        for (const ImageRef& l : layers) {
            const Bytes b = Traits::image_data(l);
            if ((int)Traits::size(b) != tex_res_ * tex_res_ * 4) error_ = "texture layer is not res x res RGBA8";
            tex_.insert(tex_.end(), Traits::ptr(b), Traits::ptr(b) + Traits::size(b));
        }
        return Traits::make_rid(next_id_++);
    }
    void add_existing_buffer(const RID&, UniformType, int /*binding*/, int /*set*/) {}

    void finish_create_uniforms()
    {
        if (!dev_->ctx) return;
        if (progressive_ || temporal_) {
            ready_ = true;  // both post-processing passes run inside jpt_render
            return;
        }
        const Slot* b[6] = {};
        for (auto& kv : slots_)
            if (kv.second.set == 1 && kv.second.binding >= 0 && kv.second.binding < 6) b[kv.second.binding] = &kv.second;
        for (const Slot* p : b)
            if (!p) {
                error_ = "main.glsl needs the six scene buffers of set 1";
                return;
            }
        // set 1: b0 GpuTriangleGeometry (48 B), b1 GpuTriangleData (80), b2 GpuMaterial (64), b3 BVHNode (48),
        //        b4 BLASInstance (176), b5 TLASNode (32)   (main.glsl:121-155)
        // (the DEBUG_STEPS build counts the triangle tests of the REFERENCE's tree: walk the uploaded arrays node for node)
        int rc = jpt_set_upload_mode(dev_->ctx, debug_steps_ ? JPT_UPLOAD_WALK_AS_GIVEN : JPT_UPLOAD_NATIVE_TREE);
        if (rc == JPT_OK)
            rc = jpt_scene_upload_reference_layout(dev_->ctx, b[0]->bytes.data(), (uint32_t)(b[0]->bytes.size() / 48),
                                                   b[1]->bytes.data(), b[2]->bytes.data(), (uint32_t)(b[2]->bytes.size() / 64),
                                                   b[3]->bytes.data(), (uint32_t)(b[3]->bytes.size() / 48), b[4]->bytes.data(),
                                                   (uint32_t)(b[4]->bytes.size() / 176), b[5]->bytes.data(),
                                                   (uint32_t)(b[5]->bytes.size() / 32), tex_.empty() ? nullptr : tex_.data(),
                                                   tex_res_, tex_layers_);
        if (rc == JPT_OK) rc = jpt_set_params(dev_->ctx, dev_->width, dev_->height, max_bounces_, JPT_ACCUM_REF_LDR8, sampler_);
        if (rc == JPT_OK) rc = jpt_set_debug_steps(dev_->ctx, debug_steps_ ? 1 : 0);
        if (rc != JPT_OK) {
            error_ = jpt_last_error(dev_->ctx);
            return;
        }
        dev_->scene_ready = dev_->params_ready = true;
        for (auto& kv : slots_)
            if (kv.second.set == 0 && kv.second.binding == 3) upload_camera(kv.second.bytes);
        ready_ = true;
    }

    bool check_ready() const { return ready_; }
    const std::string& last_error() const { return error_; }
    jpt_ctx* context() const { return dev_->ctx; }   // for callers that want the float sums (jpt_read_accum_f32) as well

    void update_storage_buffer_uniform(const RID& rid, const Bytes& data)
    {
        const uint64_t id = Traits::rid_id(rid);
        for (auto& kv : slots_)
            if (kv.first == id) {
                kv.second.bytes.assign(Traits::ptr(data), Traits::ptr(data) + Traits::size(data));
                if (!progressive_ && !temporal_ && kv.second.set == 0 && kv.second.binding == 3) upload_camera(kv.second.bytes);
                if (progressive_ && kv.second.set == 0 && kv.second.binding == 0 && kv.second.bytes.size() >= 12) {
                    uint32_t frame_count;  // Params{width, height, frame_count} (progressive_rendering.h:14-27)
                    std::memcpy(&frame_count, kv.second.bytes.data() + 8, 4);
                    dev_->progressive_frame_count = frame_count;  // 1 after a camera move (progressive_rendering.cpp:56-60)
                }
            }
    }

    // main shader: remembers that a frame is due.  progressive shader: runs the fused frame (trace + accumulate),
    // because only now is frame_count known (the reference dispatches main first, then the progressive pass,
    // path_tracing_camera.cpp:204,213).
    // temporal shader: the same, with the 88-byte RenderParameters of set 0 binding 0
    // (temporal_reprojection.cpp:32,67) handed over first.
    void compute(const std::array<int32_t, 3>& /*groups*/)
    {
        if (!ready_ || !dev_->scene_ready) return;
        if (!progressive_ && !temporal_) {
            dev_->frame_pending = true;
            return;
        }
        if (!dev_->frame_pending) return;
        int rc = jpt_set_denoising_mode(dev_->ctx, temporal_ ? JPT_DENOISE_TEMPORAL : JPT_DENOISE_PROGRESSIVE);
        // main.glsl's r32f depth image has one reader, temporal_reprojection.glsl (this adapter never hands it to the host:
        // get_image_uniform_buffer is only asked for the screen, path_tracing_camera.cpp:228-229): not produced in the other modes
        if (rc == JPT_OK) rc = jpt_set_outputs(dev_->ctx, temporal_ ? JPT_OUTPUT_DEPTH : 0u);
        if (rc == JPT_OK && temporal_) {
            const Slot* p = nullptr;
            for (auto& kv : slots_)
                if (kv.second.set == 0 && kv.second.binding == 0 && kv.second.bytes.size() >= 88) p = &kv.second;
            if (!p) {
                error_ = "temporal_reprojection.glsl needs its RenderParameters at set 0 binding 0";
                return;
            }
            rc = jpt_set_temporal_params(dev_->ctx, p->bytes.data());
        }
        // the frame is accumulated under the frame_count the caller's ProgressiveRendering::render computed, whatever it is
        // (1 restarts; the reference's first frame has 2 when the camera transform is the identity)
        if (rc == JPT_OK && progressive_) rc = jpt_set_progressive_frame_count(dev_->ctx, dev_->progressive_frame_count);
        if (rc == JPT_OK) rc = jpt_render(dev_->ctx, 1, dev_->camera_frame_index);
        if (rc != JPT_OK) error_ = jpt_last_error(dev_->ctx);
        dev_->frame_pending = false;
    }
    // denoising_mode == NONE: no progressive pass follows; the caller reads the image right after compute()
    Bytes get_image_uniform_buffer(const RID&)
    {
        Bytes out;
        Traits::resize(out, (size_t)dev_->width * dev_->height * 4);
        if (dev_->frame_pending) {  // no post-processing pass ran (denoising_mode == NONE): main.glsl's own rgba8 image
            int rc = jpt_set_denoising_mode(dev_->ctx, JPT_DENOISE_NONE);
            if (rc == JPT_OK) rc = jpt_set_outputs(dev_->ctx, 0u);   // (nobody reads the depth image in this mode)
            if (rc == JPT_OK) rc = jpt_render(dev_->ctx, 1, dev_->camera_frame_index);
            if (rc != JPT_OK) error_ = jpt_last_error(dev_->ctx);
            dev_->frame_pending = false;
        }
        if (jpt_read_ldr_rgba8(dev_->ctx, Traits::ptrw(out)) != JPT_OK) error_ = jpt_last_error(dev_->ctx);
        return out;
    }

  private:
    struct Slot {
        int set, binding;
        std::vector<uint8_t> bytes;
    };
    static void read_define(const std::string& s, const char* name, int& value)
    {
        const size_t at = s.find(name);
        if (at == std::string::npos) return;
        const char* p = s.c_str() + at + std::strlen(name);
        char* end = nullptr;
        const long v = std::strtol(p, &end, 10);
        if (end != p) value = (int)v;
    }
    void upload_camera(const std::vector<uint8_t>& bytes)
    {
        if (bytes.size() < 160 || !dev_->params_ready) return;
        jpt_set_camera(dev_->ctx, bytes.data());
        std::memcpy(&dev_->camera_frame_index, bytes.data() + 144, 4);  // Camera::frame_index (render_parameters.h:19)
    }
    std::shared_ptr<SharedDevice> dev_;
    bool progressive_, temporal_;
    bool ready_ = false;
    int max_bounces_ = 4, sampler_ = 0;
    bool debug_steps_ = false;
    std::vector<std::pair<uint64_t, Slot>> slots_;
    std::vector<uint8_t> tex_;
    int tex_res_ = 0, tex_layers_ = 0;
    uint64_t next_id_ = 1;
    std::string error_;
};

}  // namespace jpt_gdcs
