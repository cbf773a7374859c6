"""ctypes binding of the CPU oracle (oracle/libjpt_oracle.so).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg, never by gdpathtracing_amd.  PARITY UNPINNED: see oracle/jpt_oracle.h.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

from gdpathtracing_amd import wire

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libjpt_oracle.so")


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("oracle_trace.c", "oracle_bvh.c", "jpt_oracle.h", "oracle_pins.h", "Makefile")]
    stale = (not os.path.exists(_LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "libjpt_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


class SceneView(C.Structure):
    _fields_ = [
        ("tri_geom", C.c_void_p), ("n_tri", C.c_uint32),
        ("tri_data", C.c_void_p),
        ("materials", C.c_void_p), ("n_mat", C.c_uint32),
        ("bvh_nodes", C.c_void_p), ("n_nodes", C.c_uint32),
        ("instances", C.c_void_p), ("n_inst", C.c_uint32),
        ("tlas_nodes", C.c_void_p), ("n_tlas", C.c_uint32),
        ("tex_rgba8", C.c_void_p),
        ("tex_res", C.c_int32), ("n_layers", C.c_int32), ("sampler_mode", C.c_int32),
    ]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("rays", "blas_expand", "tri_tests", "tlas_expand", "inst_visits", "shaded_hits", "stack_overflow")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class Surface(C.Structure):
    _fields_ = [("vertices", C.c_void_p), ("normals", C.c_void_p), ("uvs", C.c_void_p), ("indices", C.c_void_p),
                ("n_vertices", C.c_int32), ("n_indices", C.c_int32)]


class Shading(C.Structure):
    _fields_ = [("normal", C.c_float * 3), ("out_dir", C.c_float * 3), ("lambert_out", C.c_float),
                ("diffuse_albedo", C.c_float * 3), ("fresnel_0", C.c_float * 3), ("roughness", C.c_float)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.jpto_builder_create.restype = C.c_void_p
        L.jpto_builder_destroy.argtypes = [C.c_void_p]
        L.jpto_builder_add_mesh.argtypes = [C.c_void_p, C.POINTER(Surface), C.c_int32]
        L.jpto_builder_add_mesh.restype = C.c_uint32
        L.jpto_builder_add_instance.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int32]
        L.jpto_builder_add_instance.restype = C.c_uint32
        L.jpto_builder_finish.argtypes = [C.c_void_p]
        L.jpto_builder_counts.argtypes = [C.c_void_p] + [C.POINTER(C.c_uint32)] * 4
        for fn in ("triangles", "tri_geom", "tri_data", "nodes", "instances", "tlas"):
            f = getattr(L, "jpto_builder_" + fn)
            f.argtypes = [C.c_void_p]
            f.restype = C.c_void_p
        L.jpto_nth_element_centroid.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
        L.jpto_affine_inverse.argtypes = [C.c_void_p, C.c_void_p]
        L.jpto_trace_frame.argtypes = [C.POINTER(SceneView), C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_uint32,
                                       C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(Counters)]
        L.jpto_render.argtypes = [C.POINTER(SceneView), C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                  C.c_uint32, C.c_int32, C.c_uint32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.POINTER(Counters)]
        L.jpto_render.restype = C.c_int32
        L.jpto_prng_seed.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.jpto_pcg2d.argtypes = [C.c_void_p, C.c_void_p]
        L.jpto_sincos.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.jpto_intersect_aabb.argtypes = [C.c_void_p] * 4
        L.jpto_intersect_aabb.restype = C.c_float
        L.jpto_intersect_triangle.argtypes = [C.c_void_p] * 5 + [C.c_float, C.c_void_p, C.POINTER(C.c_int)]
        L.jpto_intersect_triangle.restype = C.c_int
        L.jpto_primary_ray.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                       C.c_void_p]
        L.jpto_brdf.argtypes = [C.POINTER(Shading), C.c_void_p, C.c_void_p]
        L.jpto_sample_brdf.argtypes = [C.POINTER(Shading), C.c_void_p, C.c_void_p]
        L.jpto_brdf_density.argtypes = [C.POINTER(Shading), C.c_void_p]
        L.jpto_brdf_density.restype = C.c_float
        L.jpto_screen_rgba8.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.jpto_temporal_reproject.argtypes = [C.c_void_p] * 5
        L.jpto_unorm8.argtypes = [C.c_float]
        L.jpto_unorm8.restype = C.c_uint8
        L.jpto_aces.argtypes = [C.c_void_p, C.c_void_p]
        L.jpto_sample_texture.argtypes = [C.POINTER(SceneView), C.c_float, C.c_float, C.c_int32, C.c_void_p]
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


@dataclass
class RefLayoutScene:
    """The six reference-layout byte buffers GeometryGroup3D::get_*_buffer() emit
    (geometry_group3d.cpp:40-68) + builder-internal Triangle array + textures."""
    triangles: np.ndarray     # wire.TRIANGLE
    tri_geom: np.ndarray      # wire.TRI_GEOMETRY
    tri_data: np.ndarray      # wire.TRI_DATA
    materials: np.ndarray     # wire.MATERIAL
    bvh_nodes: np.ndarray     # wire.BVH_NODE
    instances: np.ndarray     # wire.BLAS_INSTANCE
    tlas_nodes: np.ndarray    # wire.TLAS_NODE
    textures: np.ndarray | None = None
    roots: list | None = None

    def view(self, sampler_mode=0) -> SceneView:
        v = SceneView()
        v.tri_geom, v.n_tri = _ptr(self.tri_geom), len(self.tri_geom)
        v.tri_data = _ptr(self.tri_data)
        v.materials, v.n_mat = _ptr(self.materials), len(self.materials)
        v.bvh_nodes, v.n_nodes = _ptr(self.bvh_nodes), len(self.bvh_nodes)
        v.instances, v.n_inst = _ptr(self.instances), len(self.instances)
        v.tlas_nodes, v.n_tlas = _ptr(self.tlas_nodes), len(self.tlas_nodes)
        if self.textures is not None:
            v.tex_rgba8 = _ptr(self.textures)
            v.n_layers, v.tex_res = self.textures.shape[0], self.textures.shape[1]
        v.sampler_mode = sampler_mode
        return v


def _copy(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (n * dtype.itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=n).copy()


def build_scene(scene) -> RefLayoutScene:
    """Run the oracle's restatement of GeometryGroup3D::build's tail (geometry_group3d.cpp:305-365) on a
    gdpathtracing_amd.scenes.Scene."""
    L = lib()
    b = L.jpto_builder_create()
    try:
        roots = []
        for mesh in scene.meshes:
            arr = (Surface * len(mesh.surfaces))()
            for i, s in enumerate(mesh.surfaces):
                arr[i].vertices, arr[i].normals, arr[i].uvs, arr[i].indices = map(_ptr, (s.vertices, s.normals, s.uvs, s.indices))
                arr[i].n_vertices, arr[i].n_indices = len(s.vertices), len(s.indices)
            roots.append(L.jpto_builder_add_mesh(b, arr, len(mesh.surfaces)))
        for inst in scene.instances:
            t = np.ascontiguousarray(inst.transform, dtype=np.float32)
            m = np.ascontiguousarray(inst.material_ids, dtype=np.int32)
            L.jpto_builder_add_instance(b, roots[inst.mesh], _ptr(t), _ptr(m), len(m))
        L.jpto_builder_finish(b)
        nt, nn, ni, nl = (C.c_uint32() for _ in range(4))
        L.jpto_builder_counts(b, nt, nn, ni, nl)
        return RefLayoutScene(
            triangles=_copy(L.jpto_builder_triangles(b), nt.value, wire.TRIANGLE),
            tri_geom=_copy(L.jpto_builder_tri_geom(b), nt.value, wire.TRI_GEOMETRY),
            tri_data=_copy(L.jpto_builder_tri_data(b), nt.value, wire.TRI_DATA),
            materials=np.ascontiguousarray(scene.materials, dtype=wire.MATERIAL),
            bvh_nodes=_copy(L.jpto_builder_nodes(b), nn.value, wire.BVH_NODE),
            instances=_copy(L.jpto_builder_instances(b), ni.value, wire.BLAS_INSTANCE),
            tlas_nodes=_copy(L.jpto_builder_tlas(b), nl.value, wire.TLAS_NODE),
            textures=None if scene.textures is None else np.ascontiguousarray(scene.textures, dtype=np.uint8),
            roots=roots)
    finally:
        L.jpto_builder_destroy(b)


def render(ref: RefLayoutScene, camera: np.ndarray, width: int, height: int, max_bounces: int = 4, n_frames: int = 1,
           first_frame_index: int = 1, accum_mode: int = wire.ACCUM_REF_LDR8, flags: int = 0, n_threads: int = 0,
           sampler_mode: int = 0):
    """N frames of PathTracingCamera::render + ProgressiveRendering::render.
    Returns (accum f32 [H,W,4], ldr u8 [H,W,4], depth f32 [H,W], counters dict, threads used)."""
    L = lib()
    cam = np.ascontiguousarray(camera, dtype=wire.CAMERA).reshape(1)
    accum = np.zeros((height, width, 4), dtype=np.float32)
    ldr = np.zeros((height, width, 4), dtype=np.uint8)
    depth = np.zeros((height, width), dtype=np.float32)
    cnt = Counters()
    v = ref.view(sampler_mode)
    used = L.jpto_render(C.byref(v), _ptr(cam), width, height, max_bounces, n_frames, first_frame_index, accum_mode,
                         flags, n_threads, _ptr(accum), _ptr(ldr), _ptr(depth), C.byref(cnt))
    return accum, ldr, depth, cnt.as_dict(), used


def trace_frame(ref: RefLayoutScene, camera: np.ndarray, width: int, height: int, max_bounces: int = 4, flags: int = 0,
                rows=None):
    """One dispatch of main.glsl: float radiance (before the rgba8 store) + depth + counters.  rows = (y0, y1)
    restricts it to those image rows (the arrays keep the full size; other rows stay zero)."""
    L = lib()
    cam = np.ascontiguousarray(camera, dtype=wire.CAMERA).reshape(1)
    rad = np.zeros((height, width, 4), dtype=np.float32)
    depth = np.zeros((height, width), dtype=np.float32)
    cnt = Counters()
    v = ref.view()
    y0, y1 = (0, height) if rows is None else rows
    L.jpto_trace_frame(C.byref(v), _ptr(cam), width, height, max_bounces, flags, y0, y1, _ptr(rad), _ptr(depth),
                       C.byref(cnt))
    return rad, depth, cnt.as_dict()


def screen_rgba8(radiance: np.ndarray) -> np.ndarray:
    """imageStore(outputImage rgba8) of main.glsl:434 for a whole frame of float radiance."""
    rad = np.ascontiguousarray(radiance, dtype=np.float32)
    out = np.zeros(rad.shape[:-1] + (4,), dtype=np.uint8)
    lib().jpto_screen_rgba8(_ptr(rad), rad.size // 4, _ptr(out))
    return out


def temporal_reproject(params: np.ndarray, screen: np.ndarray, depth: np.ndarray, fb1: np.ndarray, fb2: np.ndarray):
    """One dispatch of temporal_reprojection.glsl; screen (u8 [H,W,4]) and fb1/fb2 (f32 [H,W,4]) are updated in place."""
    p = np.ascontiguousarray(params, dtype=wire.TEMPORAL_PARAMS).reshape(1)
    for a in (screen, fb1, fb2):
        assert a.flags["C_CONTIGUOUS"]
    d = np.ascontiguousarray(depth, dtype=np.float32)
    lib().jpto_temporal_reproject(_ptr(p), _ptr(screen), _ptr(d), _ptr(fb1), _ptr(fb2))
