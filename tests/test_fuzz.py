"""Fuzzed scenes (scenes.random_scene: triangle soups with slivers, degenerate and coincident triangles, odd normals,
multi-surface meshes, sheared / mirrored instances, random materials and textures).
CPU part: the library's reference-exact builder against the oracle's builder, byte for byte.
GPU part: every device route against the oracle's image."""
import os

import numpy as np
import pytest

from gdpathtracing_amd import capi, host, scenes, wire

SEEDS = list(range(8))


def rel_l2(a, b):
    a = a[..., :3].astype(np.float64)
    b = b[..., :3].astype(np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.mark.parametrize("seed", SEEDS + [100, 101, 102, 103])
def test_builder_arrays_equal_the_oracles(hiplib, oracle, seed):
    sc = scenes.random_scene(seed, n_meshes=3 + seed % 3, n_instances=5 + seed % 7, tris_per_surface=17 + 13 * (seed % 5))
    ref = oracle.build_scene(sc)
    ctx = host.Context(-1)
    ctx.build_scene(sc, capi.BUILD_REFERENCE_EXACT)
    assert ctx.reference_buffer(capi.BUF_TRI_GEOMETRY, wire.TRI_GEOMETRY).tobytes() == ref.tri_geom.tobytes()
    assert ctx.reference_buffer(capi.BUF_TRI_DATA, wire.TRI_DATA).tobytes() == ref.tri_data.tobytes()
    assert ctx.reference_buffer(capi.BUF_BVH_NODES, wire.BVH_NODE).tobytes() == ref.bvh_nodes.tobytes()
    inst = ctx.reference_buffer(capi.BUF_INSTANCES, wire.BLAS_INSTANCE)
    for f in ("transform", "inverse_transform", "aabbMin", "aabbMax", "blas_index"):
        assert np.array_equal(inst[f], ref.instances[f]), f
    for k in range(3):   # material[k] beyond the mesh's surfaces is uninitialised in the reference (SURVEY A-6)
        has = np.array([k < len(sc.meshes[i.mesh].surfaces) for i in sc.instances])
        assert np.array_equal(inst["material"][has, k], ref.instances["material"][has, k])
    tlas = ctx.reference_buffer(capi.BUF_TLAS_NODES, wire.TLAS_NODE)
    for f in ("aabbMin", "aabbMax", "leftRight"):
        assert np.array_equal(tlas[f], ref.tlas_nodes[f]), f
    # the native builder accepts the same soups (degenerate triangles included) and stays within the stack limit
    ctx.build_scene(sc, capi.BUILD_SAH)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", [capi.KERNEL_WAVEFRONT, capi.KERNEL_REFERENCE_LAYOUT])
@pytest.mark.parametrize("seed", SEEDS)
def test_reference_tree_routes_are_bit_exact(oracle, hiplib, seed, kernel):
    sc = scenes.random_scene(seed)
    w, h, bounces, frames = 112, 80, 4, 3
    cam = scenes.camera_block(sc.camera, w, h)
    ref = oracle.build_scene(sc)
    mode = wire.ACCUM_REF_LDR8 if seed % 2 == 0 else wire.ACCUM_HDR_F32
    want, want_ldr, want_depth, cnt, _ = oracle.render(ref, cam, w, h, bounces, frames, 1 + seed, mode)
    for route in ("upload", "exact"):
        ctx = host.Context(0)
        try:
            ctx.set_kernel(kernel)
            if route == "upload":
                ctx.upload_reference_layout(ref.tri_geom, ref.tri_data, ref.materials, ref.bvh_nodes, ref.instances,
                                            ref.tlas_nodes, ref.textures, as_given=True)   # node for node: the counters too
            else:
                ctx.build_scene(sc, capi.BUILD_REFERENCE_EXACT)
            ctx.set_params(w, h, bounces, mode)
            ctx.set_camera(cam)
            ctx.render(frames, 1 + seed, counted=True)
            got, got_ldr, got_depth, st = ctx.read_accum(), ctx.read_ldr(), ctx.read_depth(), ctx.stats()
        finally:
            ctx.close()
        assert np.array_equal(got, want, equal_nan=True), route
        assert np.array_equal(got_ldr, want_ldr) and np.array_equal(got_depth, want_depth, equal_nan=True), route
        for k in ("rays", "blas_expand", "tri_tests", "tlas_expand", "inst_visits", "shaded_hits"):
            assert st[k] == cnt[k], (route, k)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", SEEDS)
def test_native_tree_matches_the_tree_independent_answer(oracle, hiplib, seed):
    """The native builder's tree against the oracle with every box test passing (JPTO_FLAG_NO_CULL: all triangles of
    all instances are tested, so no tree is involved).  Scenes without coincident triangles: with exact t ties the
    later-tested triangle wins (main.glsl:247), which depends on the visiting order."""
    sc = scenes.random_scene(seed, coincident=False)
    w, h, bounces, frames = 96, 64, 3, 2
    cam = scenes.camera_block(sc.camera, w, h)
    ref = oracle.build_scene(sc)
    want, _, _, _, _ = oracle.render(ref, cam, w, h, bounces, frames, 1, wire.ACCUM_HDR_F32, flags=1)
    ctx = host.Context(0)
    try:
        ctx.build_scene(sc, capi.BUILD_SAH_WATERTIGHT)
        ctx.set_params(w, h, bounces, wire.ACCUM_HDR_F32)
        ctx.set_camera(cam)
        ctx.render(frames, 1)
        got = ctx.read_accum()
    finally:
        ctx.close()
    # soups with degenerate triangles and arbitrary normals produce a few NaN pixels (0/0 in the shading frame); they
    # must be the same pixels on both sides, and the finite ones are compared by relative L2
    nan_got, nan_want = np.isnan(got).any(axis=-1), np.isnan(want).any(axis=-1)
    ok = ~(nan_got | nan_want)
    ndiff = int((got[ok] != want[ok]).any(axis=-1).sum())
    err = rel_l2(got[ok], want[ok])
    print("fuzz", seed, "differing pixels", ndiff, "rel_l2", err, "nan pixels", int(nan_got.sum()), int(nan_want.sum()))
    assert np.array_equal(nan_got, nan_want)
    assert err <= 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("route", ["commit", "upload"])
@pytest.mark.parametrize("kernel", [capi.KERNEL_WAVEFRONT, capi.KERNEL_REFERENCE_LAYOUT])
@pytest.mark.parametrize("seed", SEEDS)
def test_native_tree_with_reach_records_answers_like_the_reference(oracle, hiplib, seed, kernel, route):
    """JPT_BUILD_SAH = native tree + reach records: a triangle counts only if the reference's traversal can reach it (the
    world ray passes the instance's reference world box, the local ray passes the box of the triangle's reference leaf).
    Checked against the oracle's statement of exactly that rule on the REFERENCE arrays (JPTO_FLAG_REACH_ONLY: no internal
    box, no distance cull, so no tree order is involved) bit for bit, and against the oracle's normal walk of the
    reference tree within the north-star tolerance.  Soups without coincident triangles (exact ties depend on the order).
    route "upload": the same through jpt_scene_upload_reference_layout, whose reach records are the uploaded leaf /
    TLAS-leaf boxes themselves (no builder of the reference runs in the library)."""
    sc = scenes.random_scene(seed, coincident=False)
    w, h, bounces, frames = 96, 64, 3, 2
    cam = scenes.camera_block(sc.camera, w, h)
    ref = oracle.build_scene(sc)
    want, _, _, _, _ = oracle.render(ref, cam, w, h, bounces, frames, 1, wire.ACCUM_HDR_F32, flags=2)
    walk, _, _, _, _ = oracle.render(ref, cam, w, h, bounces, frames, 1, wire.ACCUM_HDR_F32)
    ctx = host.Context(0)
    try:
        ctx.set_kernel(kernel)
        if route == "upload":
            ctx.upload_reference_layout(ref.tri_geom, ref.tri_data, ref.materials, ref.bvh_nodes, ref.instances,
                                        ref.tlas_nodes, ref.textures)
            assert ctx.tree_kind() == capi.TREE_NATIVE_REACH, ctx.upload_note()
        else:
            ctx.build_scene(sc, capi.BUILD_SAH)
        ctx.set_params(w, h, bounces, wire.ACCUM_HDR_F32)
        ctx.set_camera(cam)
        ctx.render(frames, 1)
        got = ctx.read_accum()
    finally:
        ctx.close()
    nan_got, nan_want = np.isnan(got).any(axis=-1), np.isnan(want).any(axis=-1)
    ok = ~(nan_got | nan_want)
    ndiff = int((got[ok] != want[ok]).any(axis=-1).sum())
    ok2 = ok & ~np.isnan(walk).any(axis=-1)
    print("fuzz reach", seed, "vs REACH_ONLY: differing pixels", ndiff, "rel_l2", rel_l2(got[ok], want[ok]),
          "| vs the reference walk: differing pixels", int((got[ok2] != walk[ok2]).any(axis=-1).sum()), "rel_l2", rel_l2(got[ok2], walk[ok2]))
    assert np.array_equal(nan_got, nan_want)
    assert ndiff == 0
    assert rel_l2(got[ok2], walk[ok2]) <= 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", [capi.KERNEL_WAVEFRONT, capi.KERNEL_REFERENCE_LAYOUT])
@pytest.mark.parametrize("route", ["commit", "upload"])
@pytest.mark.parametrize("seed", SEEDS)
def test_native_tree_against_the_reference_walk_with_coincident_triangles(oracle, hiplib, seed, route, kernel):
    """Soups WITH coincident triangles: exact distance ties everywhere, where `t > hitInfo.t` rejects and the later-tested
    triangle wins (main.glsl:247) -- the order of the reference's own walk decides.  The native walk flags every hit that ties
    with an earlier one, and wf2_finish decides those on the reference's own trees (kept beside the native ones), walking
    only the ancestors of the tying leaves (jpt_tie_walk.h): the native routes land on the oracle's NORMAL walk of the
    reference tree, bit for bit -- not only on its order-free modes.  (Seed 2, pixel (41, 50), is also the regression test
    of a walk that kept the previous instance's local ray after a rejected instance entry.)"""
    sc = scenes.random_scene(seed, coincident=True)
    w, h, bounces, frames = 96, 64, 3, 2
    cam = scenes.camera_block(sc.camera, w, h)
    ref = oracle.build_scene(sc)
    want, _, want_depth, _, _ = oracle.render(ref, cam, w, h, bounces, frames, 1, wire.ACCUM_HDR_F32)
    ctx = host.Context(0)
    try:
        ctx.set_kernel(kernel)   # (the audit kernel decides its ties the same way)
        if route == "upload":
            ctx.upload_reference_layout(ref.tri_geom, ref.tri_data, ref.materials, ref.bvh_nodes, ref.instances, ref.tlas_nodes, ref.textures)
        else:
            ctx.build_scene(sc, capi.BUILD_SAH)
        ctx.set_params(w, h, bounces, wire.ACCUM_HDR_F32)
        ctx.set_camera(cam)
        ctx.render(frames, 1)
        got, got_depth, st = ctx.read_accum(), ctx.read_depth(), ctx.stats()
    finally:
        ctx.close()
    nan_got, nan_want = np.isnan(got).any(axis=-1), np.isnan(want).any(axis=-1)
    ok = ~(nan_got | nan_want)
    ndiff = int((got[ok] != want[ok]).any(axis=-1).sum())
    print("coincident soup", seed, route, "kernel", kernel, "set aside", st["set_aside"], "dropped", st["set_aside_dropped"], "differing pixels", ndiff,
          "rel_l2", rel_l2(got[ok], want[ok]))
    assert st["set_aside_dropped"] == 0
    assert np.array_equal(nan_got, nan_want)
    assert ndiff == 0


@pytest.mark.gpu
@pytest.mark.parametrize("route", ["commit", "upload"])
def test_many_coincident_copies_fall_back_to_the_whole_reference_walk(oracle, hiplib, route):
    """Forty coincident copies of one large triangle (different normals and uvs, so the winner shows) in every mesh of a
    soup.  The reference's builder puts at most four triangles in a leaf (bvh.cpp:125), so a ray through them ties in at
    least ten leaves: more than the tie walk's list holds (8).  Such a vertex is decided by the reference's WHOLE walk of
    its own trees (jpt_tie_walk.h, `overflow`) -- slow, exact: the native routes still equal the oracle's normal walk bit
    for bit."""
    import copy
    sc = copy.deepcopy(scenes.random_scene(3, n_meshes=2, n_instances=4, tris_per_surface=60, coincident=True))
    for mesh in sc.meshes:
        surf = mesh.surfaces[0]
        v = np.asarray(surf.vertices, dtype=np.float32).reshape(-1, 3, 3).copy()
        v[4:44] = np.array([[-1.5, -1.2, 0.1], [1.6, -1.1, -0.2], [0.1, 1.7, 0.3]], dtype=np.float32)
        surf.vertices = v.reshape(-1, 3)
    w, h, bounces, frames = 96, 64, 2, 2
    cam = scenes.camera_block(sc.camera, w, h)
    ref = oracle.build_scene(sc)
    want, _, want_depth, _, _ = oracle.render(ref, cam, w, h, bounces, frames, 1, wire.ACCUM_HDR_F32)
    ctx = host.Context(0)
    try:
        if route == "upload":
            ctx.upload_reference_layout(ref.tri_geom, ref.tri_data, ref.materials, ref.bvh_nodes, ref.instances, ref.tlas_nodes, ref.textures)
        else:
            ctx.build_scene(sc, capi.BUILD_SAH)
        ctx.set_params(w, h, bounces, wire.ACCUM_HDR_F32)
        ctx.set_camera(cam)
        ctx.render(frames, 1)
        got, got_depth, st = ctx.read_accum(), ctx.read_depth(), ctx.stats()
    finally:
        ctx.close()
    nan_got, nan_want = np.isnan(got).any(axis=-1), np.isnan(want).any(axis=-1)
    ok = ~(nan_got | nan_want)
    print("forty coincident copies", route, "set aside", st["set_aside"], "differing pixels", int((got[ok] != want[ok]).any(axis=-1).sum()))
    assert st["set_aside"] > 300 and st["set_aside_dropped"] == 0
    assert np.array_equal(nan_got, nan_want) and np.array_equal(got[ok], want[ok])
    assert np.array_equal(got_depth, want_depth, equal_nan=True)


@pytest.mark.gpu
@pytest.mark.parametrize("route", ["commit", "upload", "exact", "audit"])
def test_instance_of_an_empty_mesh_shows_what_the_reference_shows(oracle, hiplib, route):
    """BuildBVH of a mesh without triangles returns root 0 (bvh.cpp:111-112), so the reference draws the first mesh with
    triangles in its place.  With forty coincident copies in that mesh the tie walk runs in its whole-walk mode, where every
    instance is visited (an empty leaf read as an interior node sent the ray through node 0 by accident; now the
    instance IS an instance of that mesh, in the oracle's arrays and in the product's)."""
    import copy
    from tests.test_oracle_builder import with_empty_mesh
    sc = copy.deepcopy(scenes.random_scene(3, n_meshes=2, n_instances=4, tris_per_surface=60, coincident=True))
    for mesh in sc.meshes:
        surf = mesh.surfaces[0]
        v = np.asarray(surf.vertices, dtype=np.float32).reshape(-1, 3, 3).copy()
        v[4:44] = np.array([[-1.5, -1.2, 0.1], [1.6, -1.1, -0.2], [0.1, 1.7, 0.3]], dtype=np.float32)
        surf.vertices = v.reshape(-1, 3)
    sc = with_empty_mesh(sc, "last")
    w, h, bounces, frames = 96, 64, 2, 2
    cam = scenes.camera_block(sc.camera, w, h)
    ref = oracle.build_scene(sc)
    assert int(ref.instances["blas_index"][-1]) == 0
    want, _, want_depth, _, _ = oracle.render(ref, cam, w, h, bounces, frames, 1, wire.ACCUM_HDR_F32)
    ctx = host.Context(0)
    try:
        if route == "upload":
            ctx.upload_reference_layout(ref.tri_geom, ref.tri_data, ref.materials, ref.bvh_nodes, ref.instances, ref.tlas_nodes, ref.textures)
        else:
            ctx.build_scene(sc, capi.BUILD_REFERENCE_EXACT if route == "exact" else capi.BUILD_SAH)
        if route == "audit":
            ctx.set_kernel(capi.KERNEL_REFERENCE_LAYOUT)
        ctx.set_params(w, h, bounces, wire.ACCUM_HDR_F32)
        ctx.set_camera(cam)
        ctx.render(frames, 1)
        got, got_depth = ctx.read_accum(), ctx.read_depth()
    finally:
        ctx.close()
    nan_got, nan_want = np.isnan(got).any(axis=-1), np.isnan(want).any(axis=-1)
    ok = ~(nan_got | nan_want)
    assert np.array_equal(nan_got, nan_want) and np.array_equal(got[ok], want[ok])
    assert np.array_equal(got_depth, want_depth, equal_nan=True)


@pytest.mark.gpu
def test_extended_fuzz_slice(hiplib):
    """tests/tools/fuzz_more.py on seeds 8..39 (the whole tool runs 1 280 seeds by hand: profiles/r03/r03ab_fuzz_more.txt): varying
    sizes, bounce and frame counts, accumulation modes, queued and blocking renders; every reference-tree route and both
    native routes bit for bit against the oracle's walk of the reference tree, the watertight builder against the
    tree-independent mode."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FUZZ_FROM="8", FUZZ_TO="40")
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "tools", "fuzz_more.py")], capture_output=True, text=True, env=env, cwd=root, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    last = r.stdout.strip().splitlines()[-1]
    print(last)
    assert last.startswith("extended fuzz done") and last.endswith("mismatches: 0"), r.stdout[-2000:]
