"""The upload boundary under hostile input (VERDICT r03 task 3; contract: include/jpt.h, jpt_scene_upload_reference_layout,
which replaces the six create_storage_buffer_uniform(get_*_buffer()) calls of path_tracing_camera.cpp:170-175): whatever
the six arrays hold, the answer is JPT_OK or a JPT_E_* code.  tests/tools/upload_mutations.py forks one child per case
(single and double field mutations of child words, counts, first indices, roots, `blas`, NaN / Inf boxes and matrices,
truncated arrays; both upload modes; jpt_scene_update_reference_tlas; the commit route with hostile vertices / indices /
transforms through the three builders) on host-only contexts: no signal, no hang > 5 s.  The second test runs the same
harness against the ASan + UBSan build of the host side (make -C gdpathtracing_amd/csrc asan).  CPU only."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, "tests", "tools", "upload_mutations.py")
ASAN_LIB = os.path.join(ROOT, "build", "asan", "libjpt_hip_asan.so")


def _run(n, seed, env_extra=None):
    env = dict(os.environ, PYTHONPATH=ROOT)
    env.pop("JPT_LIB", None)
    if env_extra:
        env.update(env_extra)
    p = subprocess.run([sys.executable, HARNESS, str(n), str(seed)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-4000:]
    return json.loads(p.stdout.strip().splitlines()[-1]), p.stderr


def test_tlas_cycle_is_an_error_not_a_crash():
    """a TLAS node whose leftRight names itself overflowed the host stack in Flattener::tlas_child."""
    from gdpathtracing_amd import capi, host, scenes
    from oracle import binding as ob
    ref = ob.build_scene(scenes.instanced_scene(4, 2, 64))
    left = int(ref.tlas_nodes["leftRight"][0]) & 0xffff
    t = ref.tlas_nodes.copy()
    t["leftRight"][left] = left | left << 16
    for as_given in (False, True):
        ctx = host.Context(-1)
        with pytest.raises(capi.JptError) as e:
            ctx.upload_reference_layout(ref.tri_geom, ref.tri_data, ref.materials, ref.bvh_nodes, ref.instances, t, as_given=as_given)
        assert "(-1)" in str(e.value) and "cycle" in str(e.value)        # JPT_E_INVALID
        # the context is still usable
        ctx.upload_reference_layout(ref.tri_geom, ref.tri_data, ref.materials, ref.bvh_nodes, ref.instances, ref.tlas_nodes,
                                    as_given=as_given)
        ctx.close()


def test_shared_subtrees_are_refused_not_walked_two_to_the_depth_times():
    """a DAG (both children of every node the same next node) is refused: the flatten would expand each node once (its memo), but
    the traversal on the device has no visited set and follows every path -- 2^60 box visits per ray that enters the box, a render
    that never ends.  Sharing that cannot multiply paths stays legal: several instances on one BLAS root, and a LEAF
    under two parents"""
    from gdpathtracing_amd import capi, host, scenes, wire
    from oracle import binding as ob
    ref = ob.build_scene(scenes.cornell_scene())
    depth = 60
    bvh = np.zeros(depth + 1, dtype=wire.BVH_NODE)
    bvh["aabbMin"][:, :3] = -1.0
    bvh["aabbMax"][:, :3] = 1.0
    for i in range(depth):
        bvh["left_child"][i] = bvh["right_child"][i] = i + 1
    bvh["first_tri_index"][depth] = 0
    bvh["tri_count"][depth] = 1
    inst = ref.instances[:1].copy()
    inst["blas_index"] = 0
    tlas = np.zeros(2, dtype=wire.TLAS_NODE)
    tlas["aabbMin"], tlas["aabbMax"] = -10.0, 10.0
    tlas["leftRight"][0] = 0
    tlas["blas"][0] = 0
    ctx = host.Context(-1)
    for as_given in (False, True):
        with pytest.raises(capi.JptError) as e:
            ctx.upload_reference_layout(ref.tri_geom, ref.tri_data, ref.materials, bvh, inst, tlas, as_given=as_given)   # (2^60 calls before the memo)
        assert "(-1)" in str(e.value) and "two parents" in str(e.value)      # JPT_E_INVALID
    # a LEAF under two parents multiplies nothing: walked as given (the reach rule does not cover it)
    leafy = np.zeros(2, dtype=wire.BVH_NODE)
    leafy["aabbMin"][:, :3] = -1.0
    leafy["aabbMax"][:, :3] = 1.0
    leafy["left_child"][0] = leafy["right_child"][0] = 1
    leafy["tri_count"][1] = 1
    ctx.upload_reference_layout(ref.tri_geom, ref.tri_data, ref.materials, leafy, inst, tlas)
    assert ctx.tree_kind() == capi.TREE_AS_GIVEN and "reachable twice" in ctx.upload_note()
    # two instances on one BLAS root: legal, and still the native tree
    two = np.concatenate([inst, inst])
    tl2 = np.zeros(4, dtype=wire.TLAS_NODE)
    tl2["aabbMin"], tl2["aabbMax"] = -10.0, 10.0
    tl2["leftRight"][0] = 1 | (2 << 16)
    tl2["blas"][1], tl2["blas"][2] = 0, 1
    good = np.zeros(1, dtype=wire.BVH_NODE)
    good["aabbMin"][:, :3] = -1.0
    good["aabbMax"][:, :3] = 1.0
    good["tri_count"][0] = 1
    ctx.upload_reference_layout(ref.tri_geom, ref.tri_data, ref.materials, good, two, tl2)
    # ... and a BVH cycle is refused
    bvh["right_child"][:depth] = 0
    bvh["left_child"][depth - 1] = 3
    for i in range(depth - 1):
        bvh["right_child"][i] = depth      # (a leaf under many parents: fine)
    with pytest.raises(capi.JptError) as e:
        ctx.upload_reference_layout(ref.tri_geom, ref.tri_data, ref.materials, bvh, inst, tlas)
    assert "cycle" in str(e.value) or "two parents" in str(e.value)
    ctx.close()


def test_oversized_leaf_chain_is_a_loop():
    """a leaf of many triangles becomes a chain of 64-triangle records without recursion; absurd counts are refused"""
    from gdpathtracing_amd import capi, host, scenes, wire
    from oracle import binding as ob
    ref = ob.build_scene(scenes.instanced_scene(1, 1, 3000))
    n = len(ref.tri_geom)
    bvh = np.zeros(1, dtype=wire.BVH_NODE)
    bvh["aabbMin"][:, :3] = -100.0
    bvh["aabbMax"][:, :3] = 100.0
    bvh["tri_count"][0] = n
    inst = ref.instances[:1].copy()
    inst["blas_index"] = 0
    tlas = np.zeros(2, dtype=wire.TLAS_NODE)
    tlas["aabbMin"], tlas["aabbMax"] = -1000.0, 1000.0
    ctx = host.Context(-1)
    ctx.upload_reference_layout(ref.tri_geom, ref.tri_data, ref.materials, bvh, inst, tlas, as_given=True)
    bvh["tri_count"][0] = 0xfffffff0
    with pytest.raises(capi.JptError):
        ctx.upload_reference_layout(ref.tri_geom, ref.tri_data, ref.materials, bvh, inst, tlas, as_given=True)
    ctx.close()


def test_two_thousand_mutations_never_crash_or_hang():
    tally, _ = _run(2100, 20261003)
    assert tally["cases"] == 2100
    assert tally["crashed"] == [] and tally["hung"] == [], tally
    assert tally["refused"] > 200 and tally["accepted"] > 200          # the mutations do reach both outcomes


def test_mutations_under_asan_and_ubsan():
    rt = "/opt/rocm/lib/llvm/lib/clang/22/lib/linux/libclang_rt.asan-x86_64.so"
    if not os.path.exists(rt):
        pytest.skip("no clang sanitizer runtime in this image")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "gdpathtracing_amd", "csrc"), "asan"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=1500)
    assert r.returncode == 0 and os.path.exists(ASAN_LIB), r.stdout[-3000:]
    tally, err = _run(2000, 4, {"JPT_LIB": ASAN_LIB, "LD_PRELOAD": rt, "ASAN_OPTIONS": "detect_leaks=0:abort_on_error=1",
                                "UBSAN_OPTIONS": "print_stacktrace=1"})
    assert tally["lib"] == ASAN_LIB and tally["cases"] == 2000
    assert tally["crashed"] == [] and tally["hung"] == [], (tally, err[-3000:])
    assert "runtime error" not in err and "AddressSanitizer" not in err, err[-3000:]
