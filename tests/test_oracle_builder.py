"""The BLAS/TLAS builder: oracle restatement (oracle/oracle_bvh.c) vs the product's host builder
(gdpathtracing_amd/csrc/jpt_builder.cpp, through the C ABI on a host-only context), structural
invariants, the reference's quirks, and libstdc++'s std::nth_element as a known answer.  CPU only."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np
import pytest

from gdpathtracing_amd import capi, host, scenes, wire

def with_empty_mesh(sc, where):
    """`sc` plus a mesh without triangles and one instance of it.  BuildBVH pushes no node for it and returns 0
    (bvh.cpp:111-112): in the reference that instance shows the tree that starts at node 0 -- the first mesh with triangles."""
    import copy
    sc = copy.deepcopy(sc)
    none = scenes.Surface(np.zeros((0, 3)), np.zeros((0, 3)), np.zeros((0, 2)), np.zeros((0,), dtype=np.int32))
    if where == "first":
        sc.meshes.insert(0, scenes.Mesh([none]))
        for i in sc.instances:
            i.mesh += 1
        mesh = 0
    else:
        sc.meshes.append(scenes.Mesh([none]))
        mesh = len(sc.meshes) - 1
    sc.instances.append(scenes.Instance(mesh, scenes.transform12(None, (0.3, 0.2, 0.1)), [0]))
    return sc


SCENES = {
    "cornell": lambda: scenes.cornell_scene(),
    "demo2k": lambda: scenes.demo_scene(2048),
    "demo20k": lambda: scenes.demo_scene(20000),
    "inst": lambda: scenes.instanced_scene(6, 3, 200),
    "empty_mesh_last": lambda: with_empty_mesh(scenes.instanced_scene(3, 2, 100), "last"),
    "empty_mesh_first": lambda: with_empty_mesh(scenes.cornell_scene(), "first"),
}


@pytest.mark.parametrize("name", list(SCENES))
def test_product_reference_exact_builder_equals_oracle(oracle, hiplib, name):
    sc = SCENES[name]()
    ref = oracle.build_scene(sc)
    ctx = host.Context(-1)  # JPT_DEVICE_HOST_ONLY
    ctx.build_scene(sc, capi.BUILD_REFERENCE_EXACT)
    for which, dt, want in ((capi.BUF_TRIANGLES, wire.TRIANGLE, ref.triangles), (capi.BUF_BVH_NODES, wire.BVH_NODE, ref.bvh_nodes),
                            (capi.BUF_INSTANCES, wire.BLAS_INSTANCE, ref.instances), (capi.BUF_TLAS_NODES, wire.TLAS_NODE, ref.tlas_nodes),
                            (capi.BUF_TRI_GEOMETRY, wire.TRI_GEOMETRY, ref.tri_geom), (capi.BUF_TRI_DATA, wire.TRI_DATA, ref.tri_data),
                            (capi.BUF_MATERIALS, wire.MATERIAL, ref.materials)):
        got = ctx.reference_buffer(which, dt)
        assert got.tobytes() == want.tobytes(), "buffer %d differs" % which
    ctx.close()


def _check_tree(nodes, tris, root):
    """Every triangle of the mesh is in exactly one leaf; boxes contain their triangles' vertices."""
    seen = []
    stack = [root]
    while stack:
        n = nodes[stack.pop()]
        if n["tri_count"] > 0:
            ids = range(int(n["first_tri_index"]), int(n["first_tri_index"] + n["tri_count"]))
            seen.extend(ids)
            v = tris["vertices"][list(ids)][..., :3].reshape(-1, 3)
            assert (v >= n["aabbMin"][:3] - 0).all() and (v <= n["aabbMax"][:3] + 0).all()
        else:
            for c in (int(n["left_child"]), int(n["right_child"])):
                ch = nodes[c]
                assert (ch["aabbMin"][:3] >= n["aabbMin"][:3]).all() and (ch["aabbMax"][:3] <= n["aabbMax"][:3]).all()
                stack.append(c)
    return sorted(seen)


def test_reference_builder_invariants_and_quirks(oracle):
    sc = scenes.demo_scene(8192)
    ref = oracle.build_scene(sc)
    nodes, tris = ref.bvh_nodes, ref.triangles
    first = 0
    for mesh, root in zip(sc.meshes, ref.roots):
        ids = _check_tree(nodes, tris, root)
        assert ids == list(range(first, first + mesh.n_tris))
        first += mesh.n_tris
    # pre-order numbering: left child = parent + 1 (bvh.cpp:114-115,180)
    internal = np.nonzero(nodes["tri_count"] == 0)[0]
    assert (nodes["left_child"][internal] == internal + 1).all()
    # default-box quirk (bvh.cpp:8-9 + vec.h:49): every box reaches 0 on y and z, and x_max >= FLT_MIN
    assert (nodes["aabbMin"][:, 1] <= 0).all() and (nodes["aabbMin"][:, 2] <= 0).all()
    assert (nodes["aabbMax"][:, 1] >= 0).all() and (nodes["aabbMax"][:, 2] >= 0).all()
    assert (nodes["aabbMax"][:, 0] >= np.finfo(np.float32).tiny).all()
    assert (nodes["aabbMin"][:, 3] == 1).all() and (nodes["aabbMax"][:, 3] == 1).all()
    # centroid.w = (1+1+1) * 0.33333333f (vec4 arithmetic touches w, vec.h:53-66)
    assert (tris["centroid"][:, 3] == np.float32(3.0) * np.float32(0.33333333)).all()


def test_off_origin_mesh_box_is_inflated_towards_origin(oracle):
    """SURVEY.md A-1 probe: a mesh in [10,18]^3 gets root min (10,0,0)."""
    m = scenes.box_mesh(8, 8, 8)
    for s in m.surfaces:
        s.vertices += np.float32(14.0)
    sc = scenes.Scene("off", [m], [scenes.Instance(0, scenes.transform12(), [0])], scenes.cornell_scene().materials,
                      scenes.cornell_scene().camera)
    ref = oracle.build_scene(sc)
    root = ref.bvh_nodes[ref.roots[0]]
    assert tuple(root["aabbMin"][:3]) == (10.0, 0.0, 0.0) and tuple(root["aabbMax"][:3]) == (18.0, 18.0, 18.0)


def test_tlas_structure(oracle):
    sc = scenes.instanced_scene(5, 2, 128)
    ref = oracle.build_scene(sc)
    n = len(sc.instances)
    t = ref.tlas_nodes
    assert len(t) == 2 * n                                   # bvh.cpp:267-316 emits 2N nodes
    leaves = t[1:n + 1]
    assert (leaves["leftRight"] == 0).all() and (leaves["blas"] == np.arange(n)).all()
    assert t[0].tobytes() == t[-1].tobytes()                 # root copied to slot 0 (bvh.cpp:316)
    # every instance reachable exactly once; child boxes inside parents
    seen, stack = [], [0]
    while stack:
        node = t[stack.pop()]
        if node["leftRight"] == 0:
            seen.append(int(node["blas"]))
            continue
        for c in (int(node["leftRight"]) & 0xFFFF, int(node["leftRight"]) >> 16):
            assert (t[c]["aabbMin"] >= node["aabbMin"]).all() and (t[c]["aabbMax"] <= node["aabbMax"]).all()
            stack.append(c)
    assert sorted(seen) == list(range(n))
    # single instance: the root is a copy of the leaf
    one = scenes.Scene("one", sc.meshes[:1], sc.instances[:1], sc.materials, sc.camera)
    r1 = oracle.build_scene(one)
    assert len(r1.tlas_nodes) == 2 and r1.tlas_nodes[0]["leftRight"] == 0


def test_instance_record(oracle):
    sc = scenes.cornell_scene()
    ref = oracle.build_scene(sc)
    for inst_desc, rec in zip(sc.instances, ref.instances):
        m = rec["transform"].reshape(4, 4).T               # column-major (utils.h:15-37)
        inv = rec["inverse_transform"].reshape(4, 4).T
        t = inst_desc.transform
        assert np.array_equal(m[:3, :3], t[:9].reshape(3, 3)) and np.array_equal(m[:3, 3], t[9:])
        assert np.allclose(m.astype(np.float64) @ inv.astype(np.float64), np.eye(4), atol=1e-5)
        assert list(rec["material"][:len(inst_desc.material_ids)]) == inst_desc.material_ids
        # world box = transformed corners of the (inflated) BLAS root box (bvh.h:90-115)
        root = ref.bvh_nodes[rec["blas_index"]]
        c = np.array([[root["aabbMax" if i & 1 else "aabbMin"][0], root["aabbMax" if i & 2 else "aabbMin"][1],
                       root["aabbMax" if i & 4 else "aabbMin"][2], 1.0] for i in range(8)])
        w = c @ m.T.astype(np.float64)
        assert np.allclose(rec["aabbMin"][:3], w[:, :3].min(axis=0), atol=1e-5)
        assert np.allclose(rec["aabbMax"][:3], w[:, :3].max(axis=0), atol=1e-5)


NTH_SRC = r"""
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
struct T { float key; int id; };
int main(int argc, char** argv) {
    int n = atoi(argv[1]), nth = atoi(argv[2]);
    std::vector<T> v(n);
    for (int i = 0; i < n; i++) { if (scanf("%f", &v[i].key) != 1) return 1; v[i].id = i; }
    std::nth_element(v.begin(), v.begin() + nth, v.end(), [](const T& a, const T& b) { return a.key < b.key; });
    for (int i = 0; i < n; i++) printf("%d\n", v[i].id);
    return 0;
}
"""


@pytest.fixture(scope="module")
def nth_exe():
    d = tempfile.mkdtemp()
    src, exe = os.path.join(d, "nth.cpp"), os.path.join(d, "nth")
    open(src, "w").write(NTH_SRC)
    subprocess.check_call(["g++", "-O1", "-o", exe, src])
    return exe


def _median_of_3_killer(n):
    # Musser's median-of-3 killer sequence: drives introselect towards its depth limit
    k = n // 2
    a = [0.0] * n
    for i in range(1, k + 1):
        if i % 2 == 1:
            a[i - 1] = float(i)
            a[i] = float(k + i)
        a[k + i - 1] = float(2 * i)
    return a


@pytest.mark.parametrize("case", ["random", "ties", "sorted", "reverse", "killer", "tiny"])
def test_nth_element_restatement_equals_libstdcxx(oracle, nth_exe, case):
    """oracle_bvh.c restates libstdc++'s introselect for the median fallback (bvh.cpp:170-177); the
    real std::nth_element of this toolchain is the known answer."""
    rng = np.random.RandomState(5)
    for n in ((2, 3, 4, 5, 7) if case == "tiny" else (33, 200, 1000, 4097)):
        if case in ("random", "tiny"):
            keys = rng.uniform(-1, 1, n)
        elif case == "ties":
            keys = rng.randint(0, 5, n).astype(float)
        elif case == "sorted":
            keys = np.arange(n, dtype=float)
        elif case == "reverse":
            keys = np.arange(n, dtype=float)[::-1]
        else:
            keys = np.array(_median_of_3_killer(n - n % 2) + ([0.5] if n % 2 else []))
        keys = keys.astype(np.float32)
        for nth in sorted({n // 2, 0, n - 1, n // 3}):
            out = subprocess.run([nth_exe, str(n), str(nth)], input="\n".join("%.9g" % k for k in keys), text=True,
                                 capture_output=True, check=True).stdout.split()
            want = [int(x) for x in out]
            tris = np.zeros(n, dtype=wire.TRIANGLE)
            tris["centroid"][:, 1] = keys          # axis 1
            tris["materialIndex"] = np.arange(n)
            oracle.lib().jpto_nth_element_centroid(tris.ctypes.data_as(C.c_void_p), 0, nth, n, 1)
            assert list(tris["materialIndex"]) == want, (case, n, nth)


def test_affine_inverse(oracle):
    rng = np.random.RandomState(6)
    for _ in range(50):
        t = np.concatenate([(np.eye(3) + 0.5 * rng.normal(size=(3, 3))).reshape(-1), rng.uniform(-5, 5, 3)]).astype(np.float32)
        out = np.zeros(12, dtype=np.float32)
        oracle.lib().jpto_affine_inverse(t.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        m = np.eye(4); m[:3, :3] = t[:9].reshape(3, 3); m[:3, 3] = t[9:]
        inv = np.eye(4); inv[:3, :3] = out[:9].reshape(3, 3); inv[:3, 3] = out[9:]
        assert np.allclose(m @ inv, np.eye(4), atol=2e-4)


def test_sah_builder_is_valid_and_better(oracle, hiplib):
    """The native builder's tree covers every triangle exactly once with conservative boxes and needs far
    fewer node expansions per ray than the reference's tree (whose boxes are inflated to the origin)."""
    sc = scenes.demo_scene(8192)
    ctx = host.Context(-1)
    ctx.build_scene(sc, capi.BUILD_SAH)
    nodes = ctx.reference_buffer(capi.BUF_BVH_NODES, wire.BVH_NODE)
    tris = ctx.reference_buffer(capi.BUF_TRIANGLES, wire.TRIANGLE)
    inst = ctx.reference_buffer(capi.BUF_INSTANCES, wire.BLAS_INSTANCE)
    roots = sorted(set(int(r) for r in inst["blas_index"]))
    covered = []
    for r in roots:
        covered += _check_tree(nodes, tris, r)
    assert sorted(covered) == list(range(len(tris)))
    leaves = nodes[nodes["tri_count"] > 0]
    assert leaves["tri_count"].max() <= 4
    # same multiset of triangles as the input (the builder only permutes)
    ref = oracle.build_scene(sc)
    key = lambda a: sorted(map(bytes, a["vertices"].reshape(len(a), -1)))
    assert key(tris) == key(ref.triangles)
    # traversal cost, measured with the oracle's counters on both trees
    w, h = 96, 54
    cam = scenes.camera_block(sc.camera, w, h)
    sah = oracle.RefLayoutScene(tris, ctx.reference_buffer(capi.BUF_TRI_GEOMETRY, wire.TRI_GEOMETRY),
                                ctx.reference_buffer(capi.BUF_TRI_DATA, wire.TRI_DATA), ref.materials, nodes, inst,
                                ctx.reference_buffer(capi.BUF_TLAS_NODES, wire.TLAS_NODE))
    a1, _, _, c_sah, _ = oracle.render(sah, cam, w, h, 4, 2, 1, wire.ACCUM_HDR_F32)
    a0, _, _, c_ref, _ = oracle.render(ref, cam, w, h, 4, 2, 1, wire.ACCUM_HDR_F32)
    assert np.array_equal(a0, a1)      # the image does not depend on the tree
    assert c_sah["rays"] == c_ref["rays"]
    assert c_sah["tri_tests"] * 3 < c_ref["tri_tests"]
    ctx.close()
