"""What would ONE world-space tree over all instanced triangles buy?  (An estimate before building it: the demo scene with its
four instances baked into a single mesh under an identity instance -- not the same image bit for bit, the same geometry --
against the scene as it is: event counts per ray and queued rates.)   gpurun -- python tools/flat_estimate.py"""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gdpathtracing_amd import capi, host, scenes

def baked(sc):
    vs, ns, us, idx, off = [], [], [], [], 0
    for inst in sc.instances:
        t = np.asarray(inst.transform, dtype=np.float64)
        basis = t[:9].reshape(3, 3)        # rows xx xy xz / yx yy yz / zx zy zz
        origin = t[9:12]
        for s in sc.meshes[inst.mesh].surfaces:
            v = np.asarray(s.vertices, dtype=np.float64) @ basis.T + origin
            n = np.asarray(s.normals, dtype=np.float64) @ basis.T
            vs.append(v); ns.append(n); us.append(s.uvs); idx.append(np.asarray(s.indices) + off); off += len(v)
    surf = scenes.Surface(np.concatenate(vs), np.concatenate(ns), np.concatenate(us), np.concatenate(idx))
    return scenes.Scene(sc.name + "_baked", [scenes.Mesh([surf])], [scenes.Instance(0, scenes.transform12(), [0])], sc.materials, sc.camera)

for cam_name in ("demo", "closeup"):
    base = scenes.demo_scene(51200)
    if cam_name == "closeup":
        base.camera = scenes.CameraDesc(scenes.transform12(None, (0.0, 0.0, 4.2)), fov_deg=75.0)
    for label, sc in (("two-level", base), ("baked", baked(base))):
        ctx = host.Context(0); ctx.build_scene(sc, capi.BUILD_SAH_WATERTIGHT); ctx.set_params(1920, 1080, 4, 0)
        ctx.set_camera(scenes.camera_block(sc.camera, 1920, 1080))
        ctx.render(8, 1, counted=True); st = ctx.stats(); ph = st["phase"]
        r = max(st["rays"] - st["sky_culled"], 1)
        for _ in range(8): ctx.accum_reset(); ctx.render(8, 1, asynchronous=True)
        ctx.sync(); best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            for _ in range(60): ctx.accum_reset(); ctx.render(8, 1, asynchronous=True)
            ctx.sync(); best = min(best, (time.perf_counter() - t0) / 60 * 1e6)
        print(cam_name, label, "rays", st["rays"], "per traced ray:", {k: round(st[k] / r, 2) for k in ("blas_expand", "tlas_expand", "tri_tests", "inst_visits")},
              "node_iters %d (%.1f lanes) leaf_phases %d (%.1f) inst_phases %d (%.1f)" % (ph[1], ph[2] / max(ph[1], 1), ph[3], ph[4] / max(ph[3], 1), ph[5], ph[6] / max(ph[5], 1)),
              "| %.1f us/step" % best, flush=True)
        ctx.close()
