// gdcs/include/gdcs.h -- stands where the reference's `#include "gdcs/include/gdcs.h"` looks
// (path_tracing_camera.h:5, progressive_rendering.h:4, temporal_reprojection.h:4; include root `src/` of SConstruct:16):
// put this repository's include/ directory on the addon's include path AHEAD of src/ (or copy include/gdcs over the
// empty src/gdcs submodule directory) and PathTracingCamera / ProgressiveRendering / TemporalReprojection compile with
// their source untouched -- `ComputeShader` is the adapter of jpt_gdcs_adapter.hpp over godot-cpp's types, every call
// lands in libjpt_hip.so.
//
// Not compiled here against godot-cpp (the submodule is empty in this checkout); the same adapter is compiled and run
// against stand-ins of these godot-cpp types by tests/cpp/gdcs_adapter_test.cpp.
#pragma once

#include <godot_cpp/classes/image.hpp>
#include <godot_cpp/classes/rd_texture_format.hpp>
#include <godot_cpp/classes/rd_texture_view.hpp>
#include <godot_cpp/classes/rendering_device.hpp>
#include <godot_cpp/variant/packed_byte_array.hpp>
#include <godot_cpp/variant/rid.hpp>
#include <godot_cpp/variant/string.hpp>
#include <godot_cpp/variant/utility_functions.hpp>

#include <string>

#include "../../jpt_gdcs_adapter.hpp"

namespace jpt_gdcs {

struct GodotTraits {
    using Bytes = godot::PackedByteArray;
    using RID = godot::RID;
    using String = godot::String;
    using Device = godot::RenderingDevice;
    using ImageRef = godot::Ref<godot::Image>;
    using TextureFormatRef = godot::Ref<godot::RDTextureFormat>;
    using TextureViewRef = godot::Ref<godot::RDTextureView>;
    using DataFormat = godot::RenderingDevice::DataFormat;
    using UniformType = godot::RenderingDevice::UniformType;
    static const uint8_t* ptr(const Bytes& b) { return b.ptr(); }
    static uint8_t* ptrw(Bytes& b) { return b.ptrw(); }
    static size_t size(const Bytes& b) { return (size_t)b.size(); }
    static void resize(Bytes& b, size_t n) { b.resize((int64_t)n); }
    static bool contains(const String& s, const char* needle) { return s.contains(needle); }
    static std::string to_std(const String& s) { return std::string(s.utf8().get_data()); }
    static RID make_rid(uint64_t id) { return godot::UtilityFunctions::rid_from_int64((int64_t)id); }
    static uint64_t rid_id(const RID& r) { return (uint64_t)r.get_id(); }
    static TextureFormatRef make_texture_format(int w, int h, DataFormat f)
    {
        TextureFormatRef t;
        t.instantiate();
        t->set_width((uint32_t)w);
        t->set_height((uint32_t)h);
        t->set_format(f);
        return t;
    }
    static int format_width(const TextureFormatRef& f) { return (int)f->get_width(); }
    static int format_height(const TextureFormatRef& f) { return (int)f->get_height(); }
    static int image_width(const ImageRef& i) { return i->get_width(); }
    static int image_height(const ImageRef& i) { return i->get_height(); }
    static Bytes image_data(const ImageRef& i) { return i->get_data(); }
};

}  // namespace jpt_gdcs

// the name the reference's headers use, unqualified (path_tracing_camera.h:83, progressive_rendering.h:38)
using ComputeShader = jpt_gdcs::ComputeShader<jpt_gdcs::GodotTraits>;
