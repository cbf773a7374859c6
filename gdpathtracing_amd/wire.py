"""Wire formats of the path (little-endian f32/u32), as numpy structured dtypes.

Byte-compatible with the reference structs (sizes in SURVEY.md 8(a) T-1..T-11):
  BVH::Triangle            src/bvh/bvh.h:22-29                      144 B
  BVH::BVHNode             src/bvh/bvh.h:46-54   main.glsl:45-52     48 B
  BVH::TLASNode            src/bvh/bvh.h:56-62   main.glsl:54-60     32 B
  BVH::BLASInstance        src/bvh/bvh.h:64-72   main.glsl:84-93    176 B
  GpuTriangleGeometry      render_parameters.h:59-62                 48 B
  GpuTriangleData          render_parameters.h:64-71                 80 B
  GpuMaterial              render_parameters.h:49-57                 64 B
  Camera                   render_parameters.h:14-21                160 B
"""
import numpy as np

VEC4 = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("w", "<f4")])

TRIANGLE = np.dtype([
    ("vertices", "<f4", (3, 4)), ("centroid", "<f4", (4,)), ("normals", "<f4", (3, 4)),
    ("uvs", "<f4", (3, 2)), ("materialIndex", "<u4"), ("_pad", "<u4")])
BVH_NODE = np.dtype([
    ("aabbMin", "<f4", (4,)), ("aabbMax", "<f4", (4,)), ("left_child", "<u4"), ("right_child", "<u4"),
    ("first_tri_index", "<u4"), ("tri_count", "<u4")])
TLAS_NODE = np.dtype([
    ("aabbMin", "<f4", (3,)), ("leftRight", "<u4"), ("aabbMax", "<f4", (3,)), ("blas", "<u4")])
BLAS_INSTANCE = np.dtype([
    ("transform", "<f4", (16,)), ("inverse_transform", "<f4", (16,)), ("aabbMin", "<f4", (4,)),
    ("aabbMax", "<f4", (4,)), ("blas_index", "<u4"), ("material", "<u4", (3,))])
TRI_GEOMETRY = np.dtype([("vertices", "<f4", (3, 4))])
TRI_DATA = np.dtype([
    ("n0", "<f4", (3,)), ("material_index", "<u4"), ("n1", "<f4", (4,)), ("n2", "<f4", (4,)),
    ("uvs", "<f4", (3, 2)), ("_pad", "<u4", (2,))])
MATERIAL = np.dtype([
    ("albedo", "<f4", (4,)), ("emission", "<f4", (4,)), ("metallic", "<f4"), ("roughness", "<f4"),
    ("albedo_texture_index", "<i4"), ("padding", "<f4", (5,))])
TEMPORAL_PARAMS = np.dtype([   # TemporalReprojection::RenderParameters, temporal_reprojection.h:16-23
    ("deltaMatrix", "<f4", (16,)), ("width", "<i4"), ("height", "<i4"), ("frame_count", "<u4"),
    ("blendFactor", "<f4"), ("nearPlane", "<f4"), ("farPlane", "<f4")])
CAMERA = np.dtype([
    ("vp", "<f4", (16,)), ("ivp", "<f4", (16,)), ("position", "<f4", (4,)), ("frame_index", "<u4"),
    ("near", "<f4"), ("far", "<f4"), ("_pad", "<u4")])

assert TRIANGLE.itemsize == 144 and BVH_NODE.itemsize == 48 and TLAS_NODE.itemsize == 32
assert BLAS_INSTANCE.itemsize == 176 and TRI_GEOMETRY.itemsize == 48 and TRI_DATA.itemsize == 80
assert MATERIAL.itemsize == 64 and CAMERA.itemsize == 160

ACCUM_REF_LDR8 = 0   # per-frame clamp + 8-bit quantise before the sum (what the reference does)
ACCUM_HDR_F32 = 1    # pure float sum
