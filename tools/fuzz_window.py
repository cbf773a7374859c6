"""Random cameras x strip partitions x accumulation modes x blocking / queued renders: the rows every rank renders must be
the rows of the whole-image render (accumulation buffer, display image, depth) and the ray counts must add up; the whole
image must equal a render with the sky cull (and with it the render window) switched off.   gpurun -- python tools/fuzz_window.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gdpathtracing_amd import capi, host, scenes, partition

if os.environ.get("FUZZ_NOCULL_CHILD"):
    pass

def look_at(eye, target, up=(0.0, 1.0, 0.0)):
    eye, target, up = (np.asarray(v, np.float64) for v in (eye, target, up))
    z = eye - target; z /= np.linalg.norm(z)
    x = np.cross(up, z); x /= np.linalg.norm(x)
    y = np.cross(z, x)
    return scenes.transform12(np.stack([x, y, z], axis=1), eye)

def render(sc, cam, w, h, b, frames, mode, rank=0, world=1, asynchronous=False):
    ctx = host.Context(0); ctx.build_scene(sc, capi.BUILD_SAH)
    if world > 1: ctx.set_partition(rank, world)
    ctx.set_params(w, h, b, mode); ctx.set_camera(cam)
    ctx.render(frames, 1, asynchronous=asynchronous); ctx.sync()
    out = ctx.read_accum(), ctx.read_ldr(), ctx.read_depth(), ctx.stats()["rays"]
    ctx.close()
    return out

rng = np.random.RandomState(int(os.environ.get("FUZZ_SEED", "5")))
sc = scenes.demo_scene(1500)
bad = 0
n = int(os.environ.get("FUZZ_N", "40"))
for it in range(n):
    eye = rng.uniform(-9, 9, 3); eye[2] = abs(eye[2]) + 0.5
    target = rng.uniform(-6, 6, 3) if it % 3 else rng.uniform(-1, 1, 3)
    fov = float(rng.choice([6.0, 20.0, 45.0, 79.5, 100.0]))
    sc.camera = scenes.CameraDesc(look_at(eye, target, up=(rng.uniform(-0.3, 0.3), 1.0, rng.uniform(-0.3, 0.3))), fov_deg=fov)
    w, h = int(rng.choice([97, 160, 200, 321])), int(rng.choice([45, 90, 117, 181]))
    b, frames, mode = int(rng.randint(0, 4)), int(rng.randint(1, 5)), int(rng.randint(0, 2))
    cam = scenes.camera_block(sc.camera, w, h)
    full = render(sc, cam, w, h, b, frames, mode, asynchronous=bool(it & 1))
    world = int(rng.choice([2, 3, 5, 8]))
    acc, ldr, dep, rays = np.zeros_like(full[0]), np.zeros_like(full[1]), np.zeros_like(full[2]), 0
    for r in range(world):
        a, l, d, n_r = render(sc, cam, w, h, b, frames, mode, r, world, asynchronous=bool(r & 1))
        rows = partition.rows_of_rank(h, r, world)
        acc[rows], ldr[rows], dep[rows] = a[rows], l[rows], d[rows]; rays += n_r
    # (ray statistics are those of blocking renders: compared separately below)
    oks = (np.array_equal(acc, full[0], equal_nan=True), np.array_equal(ldr, full[1]), np.array_equal(dep, full[2], equal_nan=True))
    ok = all(oks)
    if not ok: print("   accum / ldr / depth equal:", oks, "differing accum pixels", int((acc != full[0]).any(-1).sum()))
    if it % 4 == 0:
        want_rays = render(sc, cam, w, h, b, frames, mode)[3]
        got_rays = sum(render(sc, cam, w, h, b, frames, mode, r, world)[3] for r in range(world))
        if want_rays != got_rays: ok = False; print("   rays", got_rays, "!=", want_rays)
    print("case", it, "fov", fov, "%dx%d" % (w, h), "bounces", b, "frames", frames, "mode", mode, "world", world, "rays", full[3], "OK" if ok else "MISMATCH", flush=True)
    bad += 0 if ok else 1
print("window fuzz done, mismatches:", bad)
