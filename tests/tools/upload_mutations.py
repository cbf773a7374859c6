"""Mutation fuzz of the upload boundary (VERDICT r03 task 3): jpt_scene_upload_reference_layout and
jpt_scene_update_reference_tlas take six caller-made arrays (path_tracing_camera.cpp:170-175 hands over what
GeometryGroup3D::get_*_buffer() returns) and must answer ANY content with JPT_OK or a JPT_E_* code -- never a
signal, never a hang.  Every case runs in a forked child on a host-only context (no GPU): the child applies one or
two field mutations to a valid scene's arrays, uploads in one of the two upload modes and exits 0 (accepted), 1
(refused with an error code) -- anything else (a signal, exit code of a sanitizer report, a timeout) is a failure.

Run directly for a summary line (JSON), or through tests/test_upload_mutations.py, which also runs it against the
ASan + UBSan build of the host side (make -C gdpathtracing_amd/csrc asan)."""
import json
import os
import signal
import sys
import time

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

ARRAYS = ("tri_geom", "tri_data", "materials", "bvh_nodes", "instances", "tlas_nodes")
U32_SPECIALS = (0, 1, 2, 3, 0x7fffffff, 0x80000000, 0xffffffff, 0xfffffffe, 0xffff, 0x10000, 0x10001, 0xffff0000,
                1 << 25, (1 << 25) - 1, 64, 65, 4096, 32768, 65535, 65536)
F32_SPECIALS = (np.nan, np.inf, -np.inf, 0.0, -0.0, 3.4e38, -3.4e38, 1e-45, 1e30, -1e30)


def base_scenes():
    from gdpathtracing_amd import scenes
    from oracle import binding as ob
    out = []
    for sc in (scenes.instanced_scene(4, 2, 64), scenes.cornell_scene(), scenes.instanced_scene(2, 1, 300)):
        ref = ob.build_scene(sc)
        out.append({k: np.array(getattr(ref, k)).copy() for k in ARRAYS})
    return out


def _u32_value(rng, n_self, idx):
    """a replacement for an index / count word: random, special, or pointing at itself / its neighbours"""
    k = rng.integers(0, 6)
    if k == 0:
        return int(rng.integers(0, 1 << 32))
    if k == 1:
        return int(U32_SPECIALS[rng.integers(0, len(U32_SPECIALS))])
    if k == 2:
        return int(idx)                                   # itself
    if k == 3:
        return int(rng.integers(0, max(1, n_self)))       # any element in range (cycles, shared subtrees)
    if k == 4:
        return int(max(0, idx - int(rng.integers(0, 4)))) # an ancestor-ish slot
    return int(max(0, n_self + int(rng.integers(-1, 2))))  # just past / at the end


def mutate_once(rng, a):
    """one field mutation in place on the dict of arrays; returns a short description"""
    kind = rng.integers(0, 12)
    bvh, tlas, inst = a["bvh_nodes"], a["tlas_nodes"], a["instances"]
    if kind == 0 and len(bvh):                            # BVH child word
        i = int(rng.integers(0, len(bvh)))
        f = ("left_child", "right_child")[rng.integers(0, 2)]
        v = _u32_value(rng, len(bvh), i)
        bvh[f][i] = v
        return "bvh[%d].%s=%d" % (i, f, v)
    if kind == 1 and len(bvh):                            # counts / first indices
        i = int(rng.integers(0, len(bvh)))
        f = ("tri_count", "first_tri_index")[rng.integers(0, 2)]
        v = _u32_value(rng, len(a["tri_geom"]), i)
        bvh[f][i] = v
        return "bvh[%d].%s=%d" % (i, f, v)
    if kind == 2 and len(tlas):                           # TLAS child word (two 16-bit halves)
        i = int(rng.integers(0, len(tlas)))
        if rng.integers(0, 2):
            v = (_u32_value(rng, len(tlas), i) & 0xffff) | ((_u32_value(rng, len(tlas), i) & 0xffff) << 16)
        else:
            v = _u32_value(rng, len(tlas), i)
        tlas["leftRight"][i] = v
        return "tlas[%d].leftRight=0x%x" % (i, v)
    if kind == 3 and len(tlas):                           # TLAS leaf -> instance
        i = int(rng.integers(0, len(tlas)))
        v = _u32_value(rng, len(inst), i)
        tlas["blas"][i] = v
        return "tlas[%d].blas=%d" % (i, v)
    if kind == 4 and len(inst):                           # instance root
        i = int(rng.integers(0, len(inst)))
        v = _u32_value(rng, len(bvh), i)
        inst["blas_index"][i] = v
        return "inst[%d].blas_index=%d" % (i, v)
    if kind == 5 and len(inst):                           # instance material slots
        i = int(rng.integers(0, len(inst)))
        v = _u32_value(rng, len(a["materials"]), i)
        inst["material"][i, rng.integers(0, 3)] = v
        return "inst[%d].material=%d" % (i, v)
    if kind == 6:                                         # NaN / Inf / huge boxes
        name = ("bvh_nodes", "tlas_nodes", "instances")[rng.integers(0, 3)]
        arr = a[name]
        if len(arr):
            i = int(rng.integers(0, len(arr)))
            f = ("aabbMin", "aabbMax")[rng.integers(0, 2)]
            v = F32_SPECIALS[rng.integers(0, len(F32_SPECIALS))]
            arr[f][i, rng.integers(0, 3)] = v
            return "%s[%d].%s=%r" % (name, i, f, v)
    if kind == 7 and len(inst):                           # matrices
        i = int(rng.integers(0, len(inst)))
        f = ("transform", "inverse_transform")[rng.integers(0, 2)]
        v = F32_SPECIALS[rng.integers(0, len(F32_SPECIALS))]
        inst[f][i, rng.integers(0, 16)] = v
        return "inst[%d].%s=%r" % (i, f, v)
    if kind == 8:                                         # truncated (or emptied) arrays
        name = ARRAYS[rng.integers(0, len(ARRAYS))]
        n = len(a[name])
        keep = int(rng.integers(0, n + 1)) if rng.integers(0, 3) else 0
        a[name] = a[name][:keep].copy()
        if name in ("tri_geom", "tri_data"):              # the C ABI has ONE triangle count for the two arrays
            other = "tri_data" if name == "tri_geom" else "tri_geom"
            a[other] = a[other][:keep].copy()
        return "%s truncated %d->%d" % (name, n, keep)
    if kind == 9 and len(a["tri_data"]):                  # material slot of a triangle
        i = int(rng.integers(0, len(a["tri_data"])))
        v = _u32_value(rng, 3, i)
        a["tri_data"]["material_index"][i] = v
        return "tri_data[%d].material_index=%d" % (i, v)
    if kind == 10 and len(a["tri_geom"]):                 # vertices
        i = int(rng.integers(0, len(a["tri_geom"])))
        v = F32_SPECIALS[rng.integers(0, len(F32_SPECIALS))]
        a["tri_geom"]["vertices"][i, rng.integers(0, 3), rng.integers(0, 3)] = v
        return "tri_geom[%d]=%r" % (i, v)
    if len(a["materials"]):                               # texture index of a material
        i = int(rng.integers(0, len(a["materials"])))
        v = int(np.int32(np.uint32(_u32_value(rng, 4, i))))
        a["materials"]["albedo_texture_index"][i] = v
        return "materials[%d].albedo_texture_index=%d" % (i, v)
    return "none"


def _commit_child(rng):
    """route (ii): jpt_scene_add_mesh / add_instance / commit with hostile vertices, indices, transforms and material ids,
    through each of the three builders, then a moved instance + jpt_scene_update_tlas"""
    import copy
    from gdpathtracing_amd import capi, host, scenes
    sc = copy.deepcopy(COMMIT_SCENES[int(rng.integers(0, len(COMMIT_SCENES)))])
    for _ in range(1 if rng.integers(0, 3) else 2):
        kind = rng.integers(0, 6)
        mesh = sc.meshes[int(rng.integers(0, len(sc.meshes)))]
        surf = mesh.surfaces[int(rng.integers(0, len(mesh.surfaces)))]
        if kind == 0 and len(surf.vertices):
            k = int(rng.integers(1, 4))
            for _k in range(k):
                surf.vertices[rng.integers(0, len(surf.vertices)), rng.integers(0, 3)] = F32_SPECIALS[rng.integers(0, len(F32_SPECIALS))]
        elif kind == 1 and len(surf.indices):
            surf.indices[rng.integers(0, len(surf.indices))] = int(np.int32(np.uint32(_u32_value(rng, len(surf.vertices), 0))))
        elif kind == 2 and len(surf.indices):
            keep = int(rng.integers(0, len(surf.indices) + 1))
            surf.indices = surf.indices[:keep].copy()
        elif kind == 3:
            inst = sc.instances[int(rng.integers(0, len(sc.instances)))]
            inst.transform = np.array(inst.transform, dtype=np.float32)
            inst.transform[rng.integers(0, 12)] = F32_SPECIALS[rng.integers(0, len(F32_SPECIALS))]
        elif kind == 4:
            inst = sc.instances[int(rng.integers(0, len(sc.instances)))]
            inst.material_ids = [int(np.int32(np.uint32(_u32_value(rng, len(sc.materials), 0)))) for _ in range(int(rng.integers(0, 5)))]
        else:
            surf.vertices[:] = surf.vertices[0]            # every triangle degenerate and coincident
    ctx = host.Context(-1)
    builder = (capi.BUILD_SAH, capi.BUILD_REFERENCE_EXACT, capi.BUILD_SAH_WATERTIGHT)[int(rng.integers(0, 3))]
    try:
        ctx.build_scene(sc, builder)
        t = np.array(sc.instances[0].transform, dtype=np.float32)
        t[rng.integers(0, 12)] = F32_SPECIALS[rng.integers(0, len(F32_SPECIALS))]
        ctx.set_instance_transform(0, t)
        ctx.update_tlas()
    except capi.JptError:
        os._exit(1)
    os._exit(0)


def _debug_child(rng):
    """the audit entry points take raw records too: jpt_debug_quantize_nodes4 / jpt_debug_node_step4 (host restatement) on
    random bit patterns -- NaNs, infinities, denormals, huge and inverted boxes, child words of any value"""
    import ctypes as C
    from gdpathtracing_amd import capi
    L = capi.lib()
    n = int(rng.integers(1, 40))
    nodes = rng.integers(0, 1 << 32, size=(n, 32), dtype=np.uint64).astype(np.uint32)
    special = np.array([0x7fc00000, 0x7f800000, 0xff800000, 0x00000001, 0x80000000, 0x7f7fffff, 0xff7fffff, 0x3f800000, 0], dtype=np.uint32)
    mask = rng.uniform(size=nodes.shape) < 0.5
    nodes[mask] = special[rng.integers(0, len(special), size=int(mask.sum()))]
    if rng.integers(0, 2):       # half the time plausible boxes, so the step itself runs on sane planes with hostile rays
        f = nodes.view(np.float32)
        f[:, 0:12] = rng.uniform(-10, 0, size=(n, 12)).astype(np.float32)
        f[:, 16:28] = rng.uniform(0, 10, size=(n, 12)).astype(np.float32)
    out = np.zeros((n, 16), dtype=np.uint32)
    rc = L.jpt_debug_quantize_nodes4(nodes.ctypes.data_as(C.c_void_p), n, out.ctypes.data_as(C.c_void_p))
    m = int(rng.integers(1, 200))
    cases = rng.integers(0, 1 << 32, size=(m, 8), dtype=np.uint64).astype(np.uint32)
    cmask = rng.uniform(size=cases.shape) < 0.3
    cases[cmask] = special[rng.integers(0, len(special), size=int(cmask.sum()))]
    cases[:, 7] = rng.integers(0, n + (2 if rng.integers(0, 4) == 0 else 0), size=m)
    taken = np.zeros(m, dtype=np.uint8)
    rc2 = L.jpt_debug_node_step4(-1, nodes.ctypes.data_as(C.c_void_p), n, cases.ctypes.data_as(C.c_void_p), m, int(rng.integers(-3, 4)),
                                 taken.ctypes.data_as(C.c_void_p))
    os._exit(0 if (rc == 0 and rc2 == 0) else 1)


COMMIT_SCENES = []


def _child(case_rng_seed, base, as_given, update_route):
    """runs in the forked child; exit code 0 accepted, 1 refused"""
    from gdpathtracing_amd import capi, host
    rng = np.random.default_rng(case_rng_seed)
    if update_route == 2:
        _commit_child(rng)
    if update_route == 3:
        _debug_child(rng)
    a = {k: v.copy() for k, v in base.items()}
    ctx = host.Context(-1)
    if update_route:
        # a valid upload first, then a mutated instance level through jpt_scene_update_reference_tlas
        ctx.upload_reference_layout(*[a[k] for k in ARRAYS], as_given=as_given)
    n_mut = 1 if rng.integers(0, 3) else 2
    for _ in range(n_mut):
        mutate_once(rng, a)
    try:
        if update_route:
            ctx.update_reference_tlas(a["instances"], a["tlas_nodes"])
        else:
            ctx.upload_reference_layout(*[a[k] for k in ARRAYS], as_given=as_given)
    except capi.JptError:
        os._exit(1)
    os._exit(0)


def describe(seed, base_index, bases):
    rng = np.random.default_rng(seed)
    a = {k: v.copy() for k, v in bases[base_index].items()}
    n_mut = 1 if rng.integers(0, 3) else 2
    return "; ".join(mutate_once(rng, a) for _ in range(n_mut))


def run(n_cases=2000, seed=1, timeout_s=5.0, verbose=False):
    bases = base_scenes()
    from gdpathtracing_amd import capi, scenes
    if not COMMIT_SCENES:
        COMMIT_SCENES.extend([scenes.instanced_scene(4, 2, 64), scenes.cornell_scene()])
    capi.lib()                                            # load before forking: the children share the mapping
    master = np.random.default_rng(seed)
    tally = {"cases": 0, "accepted": 0, "refused": 0, "crashed": [], "hung": [], "lib": os.environ.get("JPT_LIB", "default")}
    t0 = time.time()
    for case in range(n_cases):
        case_seed = int(master.integers(0, 1 << 62))
        bi = case % len(bases)
        as_given = bool((case // len(bases)) & 1)
        update_route = {6: 1, 5: 2, 4: 3 if (case % 21) == 4 else 0}.get(case % 7, 0)   # 0 upload, 1 TLAS update, 2 commit route, 3 audit entries
        pid = os.fork()
        if pid == 0:
            try:
                _child(case_seed, bases[bi], as_given, update_route)
            except BaseException:                         # a Python-level failure in the child is a harness bug: report it
                import traceback
                traceback.print_exc()
                os._exit(3)
        deadline = time.time() + timeout_s
        status = None
        while True:
            got, st = os.waitpid(pid, os.WNOHANG)
            if got == pid:
                status = st
                break
            if time.time() > deadline:
                os.kill(pid, signal.SIGKILL)
                os.waitpid(pid, 0)
                break
            time.sleep(0.0005)
        tally["cases"] += 1
        what = {"case": case, "seed": case_seed, "base": bi, "as_given": as_given, "update_route": update_route}
        if status is None:
            what["mutation"] = describe(case_seed, bi, bases) if update_route < 2 else "commit / audit route"
            tally["hung"].append(what)
        elif os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0:
            tally["accepted"] += 1
        elif os.WIFEXITED(status) and os.WEXITSTATUS(status) == 1:
            tally["refused"] += 1
        else:
            what["status"] = ("signal %d" % os.WTERMSIG(status)) if os.WIFSIGNALED(status) else ("exit %d" % os.WEXITSTATUS(status))
            what["mutation"] = describe(case_seed, bi, bases) if update_route < 2 else "commit / audit route"
            tally["crashed"].append(what)
        if verbose and (case + 1) % 500 == 0:
            print("  %d cases, %.1f s" % (case + 1, time.time() - t0), file=sys.stderr)
    tally["seconds"] = round(time.time() - t0, 1)
    return tally


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    print(json.dumps(run(n, seed, verbose=True)))
