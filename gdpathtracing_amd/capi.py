"""ctypes loader for libjpt_hip.so (the C ABI of include/jpt.h).

There is no CPU fallback: if the HIP library is missing or no GPU is present, calls fail loudly.
`build()` compiles the library in-tree with hipcc for gfx950.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

# the library keeps six renders in flight where six of its streams get a hardware queue each: GPU_MAX_HW_QUEUES (per stream priority
# level, four by default) is read by the HIP runtime at the process's first HIP call -- set here, before this module's users can have
# made one through the library, never over a value the caller exported (jpt.h, jpt_create)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "6")

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("JPT_LIB", os.path.join(_HERE, "libjpt_hip.so"))  # JPT_LIB: A/B builds (tools/ab.sh)
CSRC = os.path.join(_HERE, "csrc")

OK = 0
ACCUM_REF_LDR8, ACCUM_HDR_F32 = 0, 1
BUILD_REFERENCE_EXACT, BUILD_SAH, BUILD_SAH_WATERTIGHT = 0, 1, 2
KERNEL_WAVEFRONT, KERNEL_REFERENCE_LAYOUT = 0, 1
SAMPLER_NEAREST_CLAMP, SAMPLER_NEAREST_REPEAT, SAMPLER_LINEAR_CLAMP, SAMPLER_LINEAR_REPEAT = 0, 1, 2, 3
DENOISE_PROGRESSIVE, DENOISE_TEMPORAL, DENOISE_NONE = 0, 1, 2
OUTPUT_DEPTH = 1
UPLOAD_NATIVE_TREE, UPLOAD_WALK_AS_GIVEN = 0, 1
STREAM_PRIORITY_DEFAULT, STREAM_PRIORITY_NORMAL, STREAM_PRIORITY_HIGH, STREAM_PRIORITY_LOW = 0, 1, 2, 3
TREE_NONE, TREE_AS_GIVEN, TREE_REFERENCE_EXACT, TREE_NATIVE_REACH, TREE_NATIVE_WATERTIGHT = range(5)
BUF_TRI_GEOMETRY, BUF_TRI_DATA, BUF_MATERIALS, BUF_BVH_NODES, BUF_INSTANCES, BUF_TLAS_NODES, BUF_TRIANGLES, BUF_REACH_TRIANGLES, BUF_REACH_INSTANCES = range(9)

# every symbol include/jpt.h declares
SYMBOLS = [
    "jpt_abi_version", "jpt_create", "jpt_destroy", "jpt_last_error", "jpt_set_stream", "jpt_get_stream", "jpt_set_stream_priority", "jpt_renders_in_flight", "jpt_set_memory_policy", "jpt_get_workspace_bytes",
    "jpt_scene_upload_reference_layout", "jpt_set_upload_mode", "jpt_scene_tree_kind", "jpt_scene_upload_note", "jpt_scene_ties_exact", "jpt_scene_begin", "jpt_scene_add_mesh", "jpt_scene_add_instance",
    "jpt_scene_set_materials", "jpt_scene_set_textures", "jpt_scene_commit", "jpt_scene_get_reference_buffer",
    "jpt_scene_set_instance_transform", "jpt_scene_update_tlas", "jpt_scene_refit_tlas", "jpt_scene_update_reference_tlas",
    "jpt_set_params", "jpt_set_kernel", "jpt_set_debug_steps", "jpt_set_kernel_timing", "jpt_set_partition", "jpt_set_camera", "jpt_render", "jpt_render_counted", "jpt_render_async",
    "jpt_sync", "jpt_accum_reset", "jpt_set_progressive_frame_count", "jpt_set_denoising_mode", "jpt_set_temporal_params", "jpt_set_outputs", "jpt_read_ldr_rgba8", "jpt_readback_ldr_begin", "jpt_readback_ldr_end", "jpt_read_accum_f32", "jpt_read_depth_f32",
    "jpt_device_accum", "jpt_assemble_from_ranks", "jpt_device_ldr", "jpt_assemble_ldr_from_ranks", "jpt_local_rows", "jpt_get_stats",
    "jpt_scene_share", "jpt_multi_create", "jpt_multi_destroy", "jpt_multi_last_error", "jpt_multi_world", "jpt_multi_ctx",
    "jpt_multi_share_scene", "jpt_multi_set_instance_transform", "jpt_multi_update_tlas", "jpt_multi_refit_tlas",
    "jpt_multi_update_reference_tlas", "jpt_multi_set_params", "jpt_multi_set_camera", "jpt_multi_accum_reset", "jpt_multi_set_gather",
    "jpt_multi_render", "jpt_multi_sync", "jpt_multi_gather_plan", "jpt_multi_read_ldr_rgba8", "jpt_multi_read_accum_f32",
    "jpt_debug_quantize_nodes4", "jpt_debug_node_step4", "jpt_debug_last_error",
]


class JptError(RuntimeError):
    pass


class Surface(C.Structure):
    _fields_ = [("vertices", C.c_void_p), ("normals", C.c_void_p), ("uvs", C.c_void_p), ("indices", C.c_void_p),
                ("n_vertices", C.c_int32), ("n_indices", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("frames", C.c_uint64), ("blas_expand", C.c_uint64), ("tri_tests", C.c_uint64),
                ("tlas_expand", C.c_uint64), ("inst_visits", C.c_uint64), ("shaded_hits", C.c_uint64),
                ("last_render_ms", C.c_double), ("last_trace_ms", C.c_double), ("last_build_ms", C.c_double),
                ("phase", C.c_uint64 * 8), ("sky_culled", C.c_uint64), ("last_primary_ms", C.c_double),
                ("set_aside", C.c_uint64), ("set_aside_dropped", C.c_uint64),
                ("walk_steps_max", C.c_uint64), ("walk_steps_hist", C.c_uint64 * 8), ("zero_throughput", C.c_uint64)]

    def as_dict(self):
        d = {n: getattr(self, n) for n, _ in self._fields_}
        d["phase"] = list(self.phase)
        d["walk_steps_hist"] = list(self.walk_steps_hist)
        return d


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile libjpt_hip.so in-tree (hipcc --offload-arch=gfx950)."""
    cmd = ["make", "-C", CSRC, "-j4"] + (["-B"] if force else [])
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode:
        print(res.stdout)
    if res.returncode:
        raise JptError("building libjpt_hip.so failed")
    return LIB_PATH


_lib = None


def lib():
    """Load the HIP library; raises JptError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64 (same SONAME as /opt/rocm's).  If this library pulled the system
    # runtime in first, a later `import torch` in the same process finds no GPU; loading torch's first
    # works for both.  Only relevant where torch is used next to the library (tests, bench: plumbing).
    try:
        import torch
        torch.cuda.is_available()
    except Exception:
        pass
    if not os.path.exists(LIB_PATH):
        raise JptError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(there is no CPU fallback for the HIP path)" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, i32, u32 = C.c_void_p, C.c_int32, C.c_uint32
    L.jpt_abi_version.restype = C.c_int
    L.jpt_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.jpt_destroy.argtypes = [vp]
    L.jpt_destroy.restype = None
    L.jpt_last_error.argtypes = [vp]
    L.jpt_last_error.restype = C.c_char_p
    L.jpt_set_stream.argtypes = [vp, vp]
    L.jpt_get_stream.argtypes = [vp, C.POINTER(vp)]
    L.jpt_set_stream_priority.argtypes = [vp, i32]
    L.jpt_set_memory_policy.argtypes = [vp, i32, C.c_uint64]
    L.jpt_get_workspace_bytes.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.jpt_scene_upload_reference_layout.argtypes = [vp, vp, u32, vp, vp, u32, vp, u32, vp, u32, vp, u32, vp, i32, i32]
    L.jpt_set_upload_mode.argtypes = [vp, i32]
    L.jpt_scene_tree_kind.argtypes = [vp]
    L.jpt_scene_upload_note.argtypes = [vp]
    L.jpt_scene_upload_note.restype = C.c_char_p
    L.jpt_scene_ties_exact.argtypes = [vp, C.POINTER(C.c_char_p)]
    L.jpt_scene_ties_exact.restype = C.c_int
    L.jpt_scene_begin.argtypes = [vp]
    L.jpt_scene_add_mesh.argtypes = [vp, C.POINTER(Surface), i32, C.POINTER(u32)]
    L.jpt_scene_add_instance.argtypes = [vp, u32, vp, vp, i32]
    L.jpt_scene_set_materials.argtypes = [vp, vp, u32]
    L.jpt_scene_set_textures.argtypes = [vp, vp, i32, i32]
    L.jpt_scene_commit.argtypes = [vp, i32]
    L.jpt_scene_get_reference_buffer.argtypes = [vp, i32, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.jpt_scene_set_instance_transform.argtypes = [vp, u32, vp]
    L.jpt_scene_update_tlas.argtypes = [vp]
    L.jpt_scene_refit_tlas.argtypes = [vp, vp, u32]
    L.jpt_scene_update_reference_tlas.argtypes = [vp, vp, u32, vp, u32]
    L.jpt_set_params.argtypes = [vp, i32, i32, i32, i32, i32]
    L.jpt_set_partition.argtypes = [vp, i32, i32]
    L.jpt_set_kernel.argtypes = [vp, i32]
    L.jpt_set_kernel_timing.argtypes = [vp, i32]
    L.jpt_set_debug_steps.argtypes = [vp, i32]
    L.jpt_set_camera.argtypes = [vp, vp]
    for n in ("jpt_render", "jpt_render_counted", "jpt_render_async"):
        getattr(L, n).argtypes = [vp, i32, u32]
    L.jpt_sync.argtypes = [vp]
    L.jpt_accum_reset.argtypes = [vp]
    L.jpt_set_progressive_frame_count.argtypes = [vp, u32]
    L.jpt_set_denoising_mode.argtypes = [vp, i32]
    L.jpt_set_temporal_params.argtypes = [vp, vp]
    if hasattr(L, "jpt_set_outputs") or "JPT_LIB" not in os.environ:   # (JPT_LIB: an A/B build of an earlier ABI may lack it)
        L.jpt_set_outputs.argtypes = [vp, C.c_uint32]
    if hasattr(L, "jpt_renders_in_flight") or "JPT_LIB" not in os.environ:
        L.jpt_renders_in_flight.argtypes = [vp]
        L.jpt_renders_in_flight.restype = C.c_int
    L.jpt_read_ldr_rgba8.argtypes = [vp, vp]
    L.jpt_read_accum_f32.argtypes = [vp, vp]
    L.jpt_readback_ldr_begin.argtypes = [vp]
    L.jpt_readback_ldr_end.argtypes = [vp, vp]
    L.jpt_read_depth_f32.argtypes = [vp, vp]
    L.jpt_device_accum.argtypes = [vp, C.POINTER(C.c_size_t)]
    L.jpt_device_accum.restype = vp
    L.jpt_assemble_from_ranks.argtypes = [vp, vp, i32]
    L.jpt_device_ldr.argtypes = [vp, C.POINTER(C.c_size_t)]
    L.jpt_device_ldr.restype = vp
    L.jpt_assemble_ldr_from_ranks.argtypes = [vp, vp, i32]
    L.jpt_local_rows.argtypes = [vp]
    L.jpt_local_rows.restype = i32
    L.jpt_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.jpt_scene_share.argtypes = [vp, vp]
    L.jpt_multi_create.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]
    L.jpt_multi_destroy.argtypes = [vp]
    L.jpt_multi_destroy.restype = None
    L.jpt_multi_last_error.argtypes = [vp]
    L.jpt_multi_last_error.restype = C.c_char_p
    L.jpt_multi_world.argtypes = [vp]
    L.jpt_multi_ctx.argtypes = [vp, C.c_int]
    L.jpt_multi_ctx.restype = vp
    L.jpt_multi_share_scene.argtypes = [vp]
    L.jpt_multi_set_instance_transform.argtypes = [vp, u32, vp]
    L.jpt_multi_update_tlas.argtypes = [vp]
    L.jpt_multi_refit_tlas.argtypes = [vp, vp, u32]
    L.jpt_multi_update_reference_tlas.argtypes = [vp, vp, u32, vp, u32]
    L.jpt_multi_set_params.argtypes = [vp, i32, i32, i32, i32, i32]
    L.jpt_multi_set_camera.argtypes = [vp, vp]
    L.jpt_multi_accum_reset.argtypes = [vp]
    L.jpt_multi_set_gather.argtypes = [vp, i32]
    L.jpt_multi_render.argtypes = [vp, i32, u32]
    L.jpt_multi_sync.argtypes = [vp]
    L.jpt_multi_gather_plan.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    L.jpt_multi_read_ldr_rgba8.argtypes = [vp, vp]
    L.jpt_multi_read_accum_f32.argtypes = [vp, vp]
    L.jpt_debug_quantize_nodes4.argtypes = [vp, u32, vp]
    L.jpt_debug_node_step4.argtypes = [C.c_int, vp, u32, vp, u32, i32, vp]
    L.jpt_debug_last_error.restype = C.c_char_p
    _lib = L
    return L


def check(ctx, rc: int, what: str):
    if rc != OK:
        msg = lib().jpt_last_error(ctx)
        raise JptError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))
