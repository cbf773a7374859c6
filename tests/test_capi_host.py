"""The C-ABI library: loads, exports every symbol include/jpt.h declares, and its host-side logic
(argument checks, host-only contexts, error reporting) behaves -- no compute calls, no GPU needed."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from gdpathtracing_amd import capi, host, scenes, wire

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_declared_symbol_is_exported(hiplib):
    header = open(os.path.join(ROOT, "include", "jpt.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = sorted(set(re.findall(r"\b(jpt_[a-z0-9_]+)\s*\(", header)))
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(hiplib, name), "libjpt_hip.so does not export %s" % name
    assert sorted(capi.SYMBOLS) == declared
    assert hiplib.jpt_abi_version() == 6


def test_no_gpu_means_loud_failure_not_fallback(hiplib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.JptError, match="no HIP device|no ROCm"):
        host.Context(0)


def test_host_only_context_builds_but_never_renders(hiplib):
    ctx = host.Context(-1)
    sc = scenes.cornell_scene()
    ctx.build_scene(sc, capi.BUILD_REFERENCE_EXACT)
    assert len(ctx.reference_buffer(capi.BUF_TRI_GEOMETRY, wire.TRI_GEOMETRY)) == 36
    assert ctx.stats()["last_build_ms"] >= 0
    with pytest.raises(capi.JptError, match="no CPU fallback|host-only"):
        ctx.render(1, 1)
    with pytest.raises(capi.JptError, match="host-only"):
        ctx.set_params(64, 64)
    ctx.close()


def test_argument_validation(hiplib):
    L = hiplib
    ctx = host.Context(-1)
    h = ctx.h
    assert L.jpt_scene_add_mesh(h, None, 0, None) == capi.OK - 4      # JPT_E_STATE: begin not called
    assert b"jpt_scene_begin" in L.jpt_last_error(h)
    assert L.jpt_scene_begin(h) == capi.OK
    v = np.zeros((3, 3), np.float32); n = np.zeros((3, 3), np.float32); uv = np.zeros((3, 2), np.float32)
    bad_idx = np.array([0, 1, 7], np.int32)
    s = capi.Surface()
    s.vertices, s.normals, s.uvs, s.indices = (a.ctypes.data_as(C.c_void_p) for a in (v, n, uv, bad_idx))
    s.n_vertices, s.n_indices = 3, 3
    assert L.jpt_scene_add_mesh(h, C.byref(s), 1, None) == -1         # vertex index out of range
    s.n_indices = 2
    assert L.jpt_scene_add_mesh(h, C.byref(s), 1, None) == -1         # not a multiple of 3
    assert L.jpt_scene_add_instance(h, 5, v.ctypes.data_as(C.c_void_p), None, 0) == -1   # unknown mesh
    assert L.jpt_scene_commit(h, 99) == -1
    assert L.jpt_set_kernel(h, 7) == -1
    assert L.jpt_set_partition(h, 2, 2) == -1
    assert L.jpt_set_camera(h, None) == -1
    # empty material table is rejected at commit (entry 0 is the default material)
    assert L.jpt_scene_commit(h, capi.BUILD_SAH) == -1 and b"material" in L.jpt_last_error(h)
    ctx.close()


def test_reference_layout_upload_is_validated(hiplib, oracle):
    ref = oracle.build_scene(scenes.cornell_scene())
    ctx = host.Context(-1)
    ctx.upload_reference_layout(ref.tri_geom, ref.tri_data, ref.materials, ref.bvh_nodes, ref.instances, ref.tlas_nodes)
    bad = ref.bvh_nodes.copy()
    bad["first_tri_index"][1] = 10_000
    with pytest.raises(capi.JptError, match="out of bounds"):
        ctx.upload_reference_layout(ref.tri_geom, ref.tri_data, ref.materials, bad, ref.instances, ref.tlas_nodes)
    bad_inst = ref.instances.copy()
    bad_inst["blas_index"][0] = 999
    with pytest.raises(capi.JptError, match="root index"):
        ctx.upload_reference_layout(ref.tri_geom, ref.tri_data, ref.materials, ref.bvh_nodes, bad_inst, ref.tlas_nodes)
    bad_tlas = ref.tlas_nodes.copy()
    bad_tlas["blas"][1] = 77
    with pytest.raises(capi.JptError, match="instance that does not exist"):
        ctx.upload_reference_layout(ref.tri_geom, ref.tri_data, ref.materials, ref.bvh_nodes, ref.instances, bad_tlas)
    ctx.close()


def test_tlas_16bit_limit_is_reported(hiplib):
    """More than 32767 instances do not fit the reference's 2x16-bit child word (bvh.h:59, bvh.cpp:300)."""
    base = scenes.cornell_scene()
    mesh = scenes.plane_mesh()
    inst = [scenes.Instance(0, scenes.transform12(None, (i % 200, 0, i // 200)), [0]) for i in range(32768)]
    sc = scenes.Scene("many", [mesh], inst, base.materials, base.camera)
    ctx = host.Context(-1)
    with pytest.raises(capi.JptError, match="16-bit"):
        ctx.build_scene(sc, capi.BUILD_SAH)
    ctx.close()


def test_host_mirror_classes(hiplib):
    """GeometryGroup3D getters mirror geometry_group3d.cpp:7-68 (on a host-only context)."""
    sc = scenes.demo_scene(512)
    g = host.GeometryGroup3D(sc, capi.BUILD_REFERENCE_EXACT)
    g.build(host.Context(-1))
    assert g.get_triangle_count() == sc.n_unique_tris
    assert g.get_blas_count() == 4 and g.get_tlas_node_count() == 8
    assert g.get_bvh_node_count() == len(g.get_bvh_buffer())
    assert g.get_materials_buffer().dtype == wire.MATERIAL and len(g.get_materials_buffer()) == 7
    assert g.get_triangles_data_buffer().nbytes == 80 * sc.n_unique_tris


def test_too_deep_tree_is_rejected_at_upload(hiplib, oracle):
    """The reference walks with unchecked 64-entry stacks (main.glsl:272,307).  A tree whose worst-case stack
    exceeds what the kernels hold is refused at upload (JPT_E_LIMIT) instead of rendering garbage."""
    ref = oracle.build_scene(scenes.cornell_scene())
    depth = 200
    tri = np.repeat(ref.tri_geom[:1], depth + 1)
    dat = np.repeat(ref.tri_data[:1], depth + 1)
    nodes = np.zeros(2 * depth + 1, dtype=wire.BVH_NODE)
    nodes["aabbMin"][:, :3] = -1
    nodes["aabbMax"][:, :3] = 1
    for i in range(depth):          # node 2i: internal, left = leaf 2i+1, right = 2i+2
        nodes[2 * i]["left_child"], nodes[2 * i]["right_child"] = 2 * i + 1, 2 * i + 2
        nodes[2 * i + 1]["first_tri_index"], nodes[2 * i + 1]["tri_count"] = i, 1
    nodes[2 * depth]["first_tri_index"], nodes[2 * depth]["tri_count"] = depth, 1
    inst = ref.instances[:1].copy()
    inst["blas_index"] = 0
    tlas = np.zeros(2, dtype=wire.TLAS_NODE)
    tlas["aabbMin"], tlas["aabbMax"] = -1, 1
    ctx = host.Context(-1)
    with pytest.raises(capi.JptError, match="too deep"):
        ctx.upload_reference_layout(tri, dat, ref.materials, nodes, inst, tlas, as_given=True)
    shallow = nodes.copy()
    shallow[2 * 40]["left_child"] = shallow[2 * 40]["right_child"] = 0   # cut the chain at depth 40
    shallow[2 * 40]["first_tri_index"], shallow[2 * 40]["tri_count"] = 40, 1
    ctx.upload_reference_layout(tri, dat, ref.materials, shallow, inst, tlas, as_given=True)
    # the default upload walks its own tree over the same triangles: the depth of the uploaded one does not matter
    ctx.upload_reference_layout(tri, dat, ref.materials, nodes, inst, tlas)
    assert ctx.tree_kind() == capi.TREE_NATIVE_REACH
    ctx.close()


def test_device_refit_is_refused_without_a_device(hiplib):
    """jpt_scene_refit_tlas runs on the device: a host-only context reports that instead of doing anything else."""
    from gdpathtracing_amd import scenes
    ctx = host.Context(-1)
    sc = scenes.instanced_scene(n_side=3, n_unique=2, tris_per_mesh=32)
    ctx.build_scene(sc, capi.BUILD_SAH)
    t = np.stack([np.asarray(i.transform, dtype=np.float32) for i in sc.instances])
    with pytest.raises(capi.JptError, match="host-only"):
        ctx.refit_tlas(t)
    with pytest.raises(capi.JptError, match="one transform per instance"):
        ctx.refit_tlas(t[:-1])
    ctx.close()
