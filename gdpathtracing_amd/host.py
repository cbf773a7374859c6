"""Host-side mirror of the reference's interface for this path, over the C ABI (include/jpt.h).

Same names and meaning as the reference's C++ classes, minus the Godot scene tree:

  GeometryGroup3D      src/path_tracing/geometry_group3d.{h,cpp}   build(), get_*_buffer(), counts
  PathTracingCamera    src/path_tracing/path_tracing_camera.{h,cpp} init(), render(), denoising_mode
  ProgressiveRendering src/path_tracing/post_processing/progressive_rendering.{h,cpp}  frame_count logic

This is plumbing (ctypes + numpy); the product is libjpt_hip.so.  The C++ form of the same adapter, for
linking into the addon, is described in INTEGRATION.md.
"""
from __future__ import annotations

import atexit
import ctypes as C
import weakref
from typing import Optional

import numpy as np

from . import capi, scenes, wire


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


_live_contexts = weakref.WeakSet()


@atexit.register
def _close_live_contexts():
    # contexts the caller forgot to close are destroyed while the HIP runtime is still up (this handler is registered
    # after torch's, so it runs before it), not from __del__ during interpreter teardown
    for ctx in list(_live_contexts):
        try:
            ctx.close()
        except Exception:
            pass


class Context:
    """One jpt_ctx (one GPU).  Thin, explicit wrapper: every method is one C-ABI call."""

    def __init__(self, device: int = 0, _borrowed=None):
        self._lib = capi.lib()
        self._owned = _borrowed is None
        if _borrowed is not None:       # a rank's context inside a MultiContext: the jpt_multi owns it
            self.h = _borrowed
        else:
            h = C.c_void_p()
            rc = self._lib.jpt_create(device, C.byref(h))
            if rc != capi.OK:
                msg = self._lib.jpt_last_error(None)
                raise capi.JptError("jpt_create failed (%d): %s" % (rc, msg.decode() if msg else "?"))
            self.h = h
        self.width = self.height = 0
        self._keep = []
        if self._owned:
            _live_contexts.add(self)

    def close(self):
        if getattr(self, "h", None):
            if self._owned:
                self._lib.jpt_destroy(self.h)
            self.h = None

    def last_error(self) -> str:
        msg = self._lib.jpt_last_error(self.h)
        return msg.decode() if msg else ""

    def share_scene_from(self, other: "Context"):
        """jpt_scene_share: the committed scene of `other` becomes this context's scene too (no builder runs)."""
        self._ck(self._lib.jpt_scene_share(self.h, other.h), "jpt_scene_share")

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc, what):
        capi.check(self.h, rc, what)

    # ---- scene
    def upload_reference_layout(self, tri_geom, tri_data, materials, bvh_nodes, instances, tlas_nodes, textures=None,
                                as_given: bool = False):
        """Route (i).  Default: the kernels walk the native tree built over the uploaded triangles, with reach records
        from the uploaded leaf / TLAS-leaf boxes.  as_given=True: the uploaded trees are walked node for node (audits)."""
        self._ck(self._lib.jpt_set_upload_mode(self.h, capi.UPLOAD_WALK_AS_GIVEN if as_given else capi.UPLOAD_NATIVE_TREE),
                 "jpt_set_upload_mode")
        arrs = [np.ascontiguousarray(a) for a in (tri_geom, tri_data, materials, bvh_nodes, instances, tlas_nodes)]
        tex = None if textures is None else np.ascontiguousarray(textures, dtype=np.uint8)
        self._ck(self._lib.jpt_scene_upload_reference_layout(
            self.h, _ptr(arrs[0]), len(arrs[0]), _ptr(arrs[1]), _ptr(arrs[2]), len(arrs[2]), _ptr(arrs[3]), len(arrs[3]),
            _ptr(arrs[4]), len(arrs[4]), _ptr(arrs[5]), len(arrs[5]), _ptr(tex),
            0 if tex is None else tex.shape[1], 0 if tex is None else tex.shape[0]), "jpt_scene_upload_reference_layout")

    def set_memory_policy(self, renders_in_flight: int = 0, workspace_budget_bytes: int = 0):
        """Cap on the device memory spent on renders in flight: 1..8 workspaces (0: the library's rule -- 4, or 6 where six slot streams run side by side) and the most bytes
        one workspace may take (0: 24 GiB) -- jpt_set_memory_policy."""
        self._ck(self._lib.jpt_set_memory_policy(self.h, renders_in_flight, workspace_budget_bytes), "jpt_set_memory_policy")

    def workspace_bytes(self) -> int:
        n = C.c_uint64(0)
        self._ck(self._lib.jpt_get_workspace_bytes(self.h, C.byref(n)), "jpt_get_workspace_bytes")
        return int(n.value)

    def set_stream_priority(self, priority: int):
        """capi.STREAM_PRIORITY_*: the priority level of the streams queued renders run on (jpt_set_stream_priority)."""
        self._ck(self._lib.jpt_set_stream_priority(self.h, priority), "jpt_set_stream_priority")

    def renders_in_flight(self) -> int:
        """the pipeline slots the last queued render was dealt among: 4, or 6 where six slot streams run side by side (0 before the first)"""
        return int(self._lib.jpt_renders_in_flight(self.h)) if hasattr(self._lib, "jpt_renders_in_flight") else 0

    def set_debug_steps(self, enable: bool):
        """main.glsl's DEBUG_STEPS build: the image is the primary ray's triangle-test count / 256 (jpt_set_debug_steps)."""
        self._ck(self._lib.jpt_set_debug_steps(self.h, 1 if enable else 0), "jpt_set_debug_steps")

    def tree_kind(self) -> int:
        """capi.TREE_*: which tree the kernels walk for the current scene."""
        k = self._lib.jpt_scene_tree_kind(self.h)
        if k < 0:
            self._ck(k, "jpt_scene_tree_kind")
        return k

    def ties_exact(self):
        """(bool, why): are exact distance ties decided as the reference decides them (jpt_scene_ties_exact)?"""
        why = C.c_char_p()
        rc = self._lib.jpt_scene_ties_exact(self.h, C.byref(why))
        if rc < 0:
            self._ck(rc, "jpt_scene_ties_exact")
        return bool(rc), (why.value.decode() if why.value else "")

    def upload_note(self) -> str:
        msg = self._lib.jpt_scene_upload_note(self.h)
        return msg.decode() if msg else ""

    def build_scene(self, scene: scenes.Scene, builder: int = capi.BUILD_SAH):
        L = self._lib
        self._ck(L.jpt_scene_begin(self.h), "jpt_scene_begin")
        ids = []
        for mesh in scene.meshes:
            arr = (capi.Surface * len(mesh.surfaces))()
            for i, s in enumerate(mesh.surfaces):
                arr[i].vertices, arr[i].normals, arr[i].uvs, arr[i].indices = map(_ptr, (s.vertices, s.normals, s.uvs, s.indices))
                arr[i].n_vertices, arr[i].n_indices = len(s.vertices), len(s.indices)
            mid = C.c_uint32()
            self._ck(L.jpt_scene_add_mesh(self.h, arr, len(mesh.surfaces), C.byref(mid)), "jpt_scene_add_mesh")
            ids.append(mid.value)
        for inst in scene.instances:
            t = np.ascontiguousarray(inst.transform, dtype=np.float32)
            m = np.ascontiguousarray(inst.material_ids, dtype=np.int32)
            self._ck(L.jpt_scene_add_instance(self.h, ids[inst.mesh], _ptr(t), _ptr(m), len(m)), "jpt_scene_add_instance")
        mats = np.ascontiguousarray(scene.materials, dtype=wire.MATERIAL)
        self._ck(L.jpt_scene_set_materials(self.h, _ptr(mats), len(mats)), "jpt_scene_set_materials")
        if scene.textures is not None:
            tex = np.ascontiguousarray(scene.textures, dtype=np.uint8)
            self._ck(L.jpt_scene_set_textures(self.h, _ptr(tex), tex.shape[1], tex.shape[0]), "jpt_scene_set_textures")
        self._ck(L.jpt_scene_commit(self.h, builder), "jpt_scene_commit")

    # ---- moving instances (no full rebuild: BLASes stay on the device)
    def set_instance_transform(self, instance: int, transform12):
        t = np.ascontiguousarray(transform12, dtype=np.float32).reshape(12)
        self._ck(self._lib.jpt_scene_set_instance_transform(self.h, instance, _ptr(t)), "jpt_scene_set_instance_transform")

    def update_tlas(self):
        self._ck(self._lib.jpt_scene_update_tlas(self.h), "jpt_scene_update_tlas")

    def refit_tlas(self, transforms12):
        """All instance transforms at once, instance records + TLAS boxes recomputed on the device (no host rebuild,
        no synchronisation); transforms12: [n_instances, 12] float32."""
        t = np.ascontiguousarray(transforms12, dtype=np.float32).reshape(-1, 12)
        self._ck(self._lib.jpt_scene_refit_tlas(self.h, _ptr(t), t.shape[0]), "jpt_scene_refit_tlas")

    def update_reference_tlas(self, instances, tlas_nodes):
        a, b = np.ascontiguousarray(instances), np.ascontiguousarray(tlas_nodes)
        self._ck(self._lib.jpt_scene_update_reference_tlas(self.h, _ptr(a), len(a), _ptr(b), len(b)), "jpt_scene_update_reference_tlas")

    def reference_buffer(self, which: int, dtype) -> np.ndarray:
        n = C.c_size_t()
        self._ck(self._lib.jpt_scene_get_reference_buffer(self.h, which, None, 0, C.byref(n)), "jpt_scene_get_reference_buffer")
        out = np.zeros(n.value // np.dtype(dtype).itemsize, dtype=dtype)
        self._ck(self._lib.jpt_scene_get_reference_buffer(self.h, which, _ptr(out), out.nbytes, C.byref(n)),
                 "jpt_scene_get_reference_buffer")
        return out

    # ---- per-render state
    def set_params(self, width, height, max_bounces=4, accum_mode=capi.ACCUM_REF_LDR8, sampler_mode=0):
        self._ck(self._lib.jpt_set_params(self.h, width, height, max_bounces, accum_mode, sampler_mode), "jpt_set_params")
        self.width, self.height = width, height

    def set_kernel(self, variant):
        self._ck(self._lib.jpt_set_kernel(self.h, variant), "jpt_set_kernel")

    def set_kernel_timing(self, enable: bool):
        self._ck(self._lib.jpt_set_kernel_timing(self.h, 1 if enable else 0), "jpt_set_kernel_timing")

    def set_partition(self, rank, world):
        self._ck(self._lib.jpt_set_partition(self.h, rank, world), "jpt_set_partition")

    def set_camera(self, camera_block: np.ndarray):
        cam = np.ascontiguousarray(camera_block, dtype=wire.CAMERA).reshape(1)
        self._ck(self._lib.jpt_set_camera(self.h, _ptr(cam)), "jpt_set_camera")

    def set_stream(self, hip_stream: Optional[int]):
        self._ck(self._lib.jpt_set_stream(self.h, hip_stream), "jpt_set_stream")

    def get_stream(self) -> int:
        out = C.c_void_p()
        self._ck(self._lib.jpt_get_stream(self.h, C.byref(out)), "jpt_get_stream")
        return int(out.value or 0)

    def render(self, n_frames=1, first_frame_index=1, counted=False, asynchronous=False):
        fn = self._lib.jpt_render_counted if counted else (self._lib.jpt_render_async if asynchronous else self._lib.jpt_render)
        self._ck(fn(self.h, n_frames, first_frame_index), "jpt_render")

    def sync(self):
        self._ck(self._lib.jpt_sync(self.h), "jpt_sync")

    def accum_reset(self):
        self._ck(self._lib.jpt_accum_reset(self.h), "jpt_accum_reset")

    def set_denoising_mode(self, mode: int):
        self._ck(self._lib.jpt_set_denoising_mode(self.h, mode), "jpt_set_denoising_mode")

    def set_temporal_params(self, params: np.ndarray):
        p = np.ascontiguousarray(params, dtype=wire.TEMPORAL_PARAMS).reshape(1)
        self._ck(self._lib.jpt_set_temporal_params(self.h, _ptr(p)), "jpt_set_temporal_params")

    def set_outputs(self, depth: bool = True):
        """jpt_set_outputs: the r32f depth image (main.glsl:435) has one reader, TemporalReprojection; off saves its passes"""
        self._ck(self._lib.jpt_set_outputs(self.h, capi.OUTPUT_DEPTH if depth else 0), "jpt_set_outputs")

    # ---- outputs
    def read_ldr(self) -> np.ndarray:
        out = np.zeros((self.height, self.width, 4), dtype=np.uint8)
        self._ck(self._lib.jpt_read_ldr_rgba8(self.h, _ptr(out)), "jpt_read_ldr_rgba8")
        return out

    def readback_ldr_begin(self):
        self._ck(self._lib.jpt_readback_ldr_begin(self.h), "jpt_readback_ldr_begin")

    def readback_ldr_end(self) -> np.ndarray:
        out = np.zeros((self.height, self.width, 4), dtype=np.uint8)
        self._ck(self._lib.jpt_readback_ldr_end(self.h, _ptr(out)), "jpt_readback_ldr_end")
        return out

    def read_accum(self) -> np.ndarray:
        out = np.zeros((self.height, self.width, 4), dtype=np.float32)
        self._ck(self._lib.jpt_read_accum_f32(self.h, _ptr(out)), "jpt_read_accum_f32")
        return out

    def read_depth(self) -> np.ndarray:
        out = np.zeros((self.height, self.width), dtype=np.float32)
        self._ck(self._lib.jpt_read_depth_f32(self.h, _ptr(out)), "jpt_read_depth_f32")
        return out

    def device_accum(self):
        n = C.c_size_t()
        p = self._lib.jpt_device_accum(self.h, C.byref(n))
        return p, n.value

    def device_ldr(self):
        n = C.c_size_t()
        p = self._lib.jpt_device_ldr(self.h, C.byref(n))
        return p, n.value

    def assemble_ldr_from_ranks(self, device_ptr: int, world: int):
        self._ck(self._lib.jpt_assemble_ldr_from_ranks(self.h, device_ptr, world), "jpt_assemble_ldr_from_ranks")

    def local_rows(self) -> int:
        return self._lib.jpt_local_rows(self.h)

    def assemble_from_ranks(self, device_ptr: int, world: int):
        self._ck(self._lib.jpt_assemble_from_ranks(self.h, device_ptr, world), "jpt_assemble_from_ranks")

    def stats(self) -> dict:
        s = capi.Stats()
        self._ck(self._lib.jpt_get_stats(self.h, C.byref(s)), "jpt_get_stats")
        return s.as_dict()


class GeometryGroup3D:
    """geometry_group3d.h:17-95.  `build()` runs the builder + upload; the get_*_buffer() getters return
    the reference-layout byte buffers (valid after a REFERENCE_EXACT build)."""

    def __init__(self, scene: scenes.Scene, builder: int = capi.BUILD_SAH):
        self.scene = scene
        self.builder = builder
        self.ctx: Optional[Context] = None

    def build(self, ctx: Context):                      # geometry_group3d.cpp:228
        self.ctx = ctx
        ctx.build_scene(self.scene, self.builder)
        self._built = [np.array(i.transform, dtype=np.float32) for i in self.scene.instances]

    def update_transforms(self, refit: bool = False) -> int:
        """Moving nodes without build() again (the reference has no such call; README.md:39-40 wants one): hands
        the changed instance transforms to the library, which redoes BLASInstance records + TLAS only -- on the host
        (rebuild, the default) or, with refit=True, on the device over the topology of the last build."""
        moved = 0
        for i, inst in enumerate(self.scene.instances):
            now = np.asarray(inst.transform, dtype=np.float32)
            if now.tobytes() != self._built[i].tobytes():
                self.ctx.set_instance_transform(i, now)
                self._built[i] = now.copy()
                moved += 1
        if moved:
            if refit:
                self.ctx.refit_tlas(np.stack(self._built))
            else:
                self.ctx.update_tlas()
        return moved

    def get_triangles_geometry_buffer(self):            # geometry_group3d.cpp:40
        return self.ctx.reference_buffer(capi.BUF_TRI_GEOMETRY, wire.TRI_GEOMETRY)

    def get_triangles_data_buffer(self):                # :45
        return self.ctx.reference_buffer(capi.BUF_TRI_DATA, wire.TRI_DATA)

    def get_materials_buffer(self):                     # :50
        return self.ctx.reference_buffer(capi.BUF_MATERIALS, wire.MATERIAL)

    def get_bvh_buffer(self):                           # :55
        return self.ctx.reference_buffer(capi.BUF_BVH_NODES, wire.BVH_NODE)

    def get_blas_buffer(self):                          # :60
        return self.ctx.reference_buffer(capi.BUF_INSTANCES, wire.BLAS_INSTANCE)

    def get_tlas_buffer(self):                          # :65
        return self.ctx.reference_buffer(capi.BUF_TLAS_NODES, wire.TLAS_NODE)

    def get_triangle_count(self):                       # :17
        return len(self.get_triangles_geometry_buffer())

    def get_blas_count(self):                           # :7
        return len(self.get_blas_buffer())

    def get_bvh_node_count(self):                       # :22
        return len(self.get_bvh_buffer())

    def get_tlas_node_count(self):                      # :27
        return len(self.get_tlas_buffer())


class TemporalReprojection:
    """temporal_reprojection.h:11-60 / temporal_reprojection.cpp:16-73, host half: keeps previous_vp and
    frame_count and produces the 88-byte RenderParameters each frame; the dispatch itself is part of jpt_render."""

    def __init__(self, width: int, height: int):
        self.params = np.zeros((), dtype=wire.TEMPORAL_PARAMS)
        self.params["width"], self.params["height"] = width, height
        self.params["frame_count"] = 1                                   # temporal_reprojection.cpp:25
        self.params["blendFactor"], self.params["nearPlane"], self.params["farPlane"] = 0.75, 0.01, 1000.0
        self.previous_vp = np.eye(4)                                     # Projection() is the identity

    def render(self, vp: np.ndarray) -> np.ndarray:                      # temporal_reprojection.cpp:56-72
        self.params["deltaMatrix"] = scenes.temporal_delta(self.previous_vp, vp)
        self.previous_vp = vp.copy()
        self.params["frame_count"] += 1
        return self.params.copy()


class PathTracingCamera:
    """path_tracing_camera.h:25-112: init() creates the device state, render() advances one frame
    (frame_index pre-incremented, path_tracing_camera.cpp:199) and accumulates progressively; the
    accumulation restarts when the camera transform changes (progressive_rendering.cpp:53-60)."""

    PROGRESSIVE_RENDERING, TEMPORAL_REPROJECTION, NONE = 0, 1, 2   # path_tracing_camera.h:30-34

    def __init__(self, geometry_group: GeometryGroup3D, device: int = 0, max_bounces: int = 4,
                 accum_mode: int = capi.ACCUM_REF_LDR8):
        self.geometry_group = geometry_group
        self.camera_desc = geometry_group.scene.camera
        self.max_bounces = max_bounces
        self.accum_mode = accum_mode
        self.denoising_mode = self.PROGRESSIVE_RENDERING
        self.frame_index = 0          # the reference never initialises it (render_parameters.h:19); 0 here
        self.ctx = Context(device)
        self._prev_transform = None
        self.temporal_reprojection: Optional[TemporalReprojection] = None
        self.width = self.height = 0

    def init(self, width: int, height: int):                        # path_tracing_camera.cpp:111-187
        self.width, self.height = width, height
        self.geometry_group.build(self.ctx)
        self.ctx.set_params(width, height, self.max_bounces, self.accum_mode)

    def render(self, n_frames: int = 1):                            # path_tracing_camera.cpp:193-232
        self.ctx.set_denoising_mode(self.denoising_mode)            # the switch at :207-225
        self.ctx.set_outputs(depth=self.denoising_mode == self.TEMPORAL_REPROJECTION)   # (main.glsl:435's image has one reader)
        self.ctx.set_camera(scenes.camera_block(self.camera_desc, self.width, self.height))
        first = self.frame_index + 1                                # camera.frame_index++ before the dispatch
        if self.denoising_mode == self.PROGRESSIVE_RENDERING:
            t = np.asarray(self.camera_desc.transform, dtype=np.float32)
            moved = self._prev_transform is None or not np.allclose(self._prev_transform, t, rtol=0, atol=1e-5)
            self._prev_transform = t.copy()
            if moved:
                self.ctx.accum_reset()                              # frame_count = 1 (progressive_rendering.cpp:56-57)
        elif self.denoising_mode == self.TEMPORAL_REPROJECTION:
            if self.temporal_reprojection is None:                  # :216-219
                self.temporal_reprojection = TemporalReprojection(self.width, self.height)
            if n_frames != 1:
                raise ValueError("temporal reprojection advances one frame per render()")
            vp = scenes.view_projection(self.camera_desc, self.width, self.height)
            self.ctx.set_temporal_params(self.temporal_reprojection.render(vp))   # :220
        self.ctx.render(n_frames, first)
        self.frame_index += n_frames
        return self.ctx.read_ldr()                                  # get_image_uniform_buffer (:228-229)


class MultiContext:
    """One jpt_multi: one image tiled across several GPUs from this process (jpt.h, jpt_multi_*).  `devices` may name a
    device more than once (rehearsal on a box with fewer GPUs)."""

    def __init__(self, devices):
        self._lib = capi.lib()
        ids = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        rc = self._lib.jpt_multi_create(ids, len(devices), C.byref(h))
        if rc != capi.OK:
            msg = self._lib.jpt_multi_last_error(None)
            raise capi.JptError("jpt_multi_create failed (%d): %s" % (rc, msg.decode() if msg else "?"))
        self.h = h
        self.world = len(devices)
        self.width = self.height = 0
        _live_contexts.add(self)

    def close(self):
        if getattr(self, "h", None):
            self._lib.jpt_multi_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc, what):
        if rc != capi.OK:
            msg = self._lib.jpt_multi_last_error(self.h)
            raise capi.JptError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))

    def ctx(self, rank: int) -> Context:
        return Context(_borrowed=C.c_void_p(self._lib.jpt_multi_ctx(self.h, rank)))

    def build_scene(self, scene, builder=capi.BUILD_SAH):
        self.ctx(0).build_scene(scene, builder)
        self._ck(self._lib.jpt_multi_share_scene(self.h), "jpt_multi_share_scene")

    def upload_reference_layout(self, *arrays, **kw):
        self.ctx(0).upload_reference_layout(*arrays, **kw)
        self._ck(self._lib.jpt_multi_share_scene(self.h), "jpt_multi_share_scene")

    # moving instances: forwarded to every rank's replica
    def set_instance_transform(self, instance: int, transform12):
        t = np.ascontiguousarray(transform12, dtype=np.float32).reshape(12)
        self._ck(self._lib.jpt_multi_set_instance_transform(self.h, instance, _ptr(t)), "jpt_multi_set_instance_transform")

    def update_tlas(self):
        self._ck(self._lib.jpt_multi_update_tlas(self.h), "jpt_multi_update_tlas")

    def refit_tlas(self, transforms12):
        t = np.ascontiguousarray(transforms12, dtype=np.float32).reshape(-1, 12)
        self._ck(self._lib.jpt_multi_refit_tlas(self.h, _ptr(t), t.shape[0]), "jpt_multi_refit_tlas")

    def update_reference_tlas(self, instances, tlas_nodes):
        a, b = np.ascontiguousarray(instances), np.ascontiguousarray(tlas_nodes)
        self._ck(self._lib.jpt_multi_update_reference_tlas(self.h, _ptr(a), len(a), _ptr(b), len(b)), "jpt_multi_update_reference_tlas")

    def set_params(self, width, height, max_bounces=4, accum_mode=capi.ACCUM_REF_LDR8, sampler_mode=0):
        self._ck(self._lib.jpt_multi_set_params(self.h, width, height, max_bounces, accum_mode, sampler_mode), "jpt_multi_set_params")
        self.width, self.height = width, height

    def set_camera(self, camera_block):
        cam = np.ascontiguousarray(camera_block, dtype=wire.CAMERA).reshape(1)
        self._ck(self._lib.jpt_multi_set_camera(self.h, _ptr(cam)), "jpt_multi_set_camera")

    def set_gather(self, ldr_only: bool):
        self._ck(self._lib.jpt_multi_set_gather(self.h, 1 if ldr_only else 0), "jpt_multi_set_gather")

    def accum_reset(self):
        self._ck(self._lib.jpt_multi_accum_reset(self.h), "jpt_multi_accum_reset")

    def render(self, n_frames, first_frame_index):
        self._ck(self._lib.jpt_multi_render(self.h, n_frames, first_frame_index), "jpt_multi_render")

    def sync(self):
        self._ck(self._lib.jpt_multi_sync(self.h), "jpt_multi_sync")

    def gather_plan(self) -> dict:
        """what the last render issued for its gather: peer copies, distinct streams they went on, copies of rank 0's own piece"""
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        self._ck(self._lib.jpt_multi_gather_plan(self.h, C.byref(a), C.byref(b), C.byref(c)), "jpt_multi_gather_plan")
        return {"peer_copies": a.value, "distinct_streams": b.value, "own_piece_copies": c.value}

    def read_accum(self):
        out = np.zeros((self.height, self.width, 4), dtype=np.float32)
        self._ck(self._lib.jpt_multi_read_accum_f32(self.h, _ptr(out)), "jpt_multi_read_accum_f32")
        return out

    def read_ldr(self):
        out = np.zeros((self.height, self.width, 4), dtype=np.uint8)
        self._ck(self._lib.jpt_multi_read_ldr_rgba8(self.h, _ptr(out)), "jpt_multi_read_ldr_rgba8")
        return out
