"""Names the cause of the native route's residual disagreement with the reference walk when the camera is 500 000 units
away (VERDICT r03 task 7; tests/tools/far_probe.py counted 32 of 2 132 hit pixels).  CPU only.

The native route's hit rule is the oracle's JPTO_FLAG_REACH_ONLY mode: the closest triangle among those whose reference
LEAF box (and instance box) the ray passes at all (`d < 1e30`), exact ties decided by the reference's visiting order.  The
reference itself (main.glsl:288-299) keeps a child only while `d < hitInfo.t`.  With exact arithmetic the two agree: a box
contains its triangles, so its entry distance is <= the triangle's t, and a box that holds the closest hit is never culled.
In float32, 500 000 units out, distances are spaced 0.03 apart and a box's entry distance can come out ABOVE its own
triangle's t.  This script finds the pixels where the oracle's two modes differ, replays the reference walk for each primary
ray in float32 numpy with its history, and checks, for the triangle T the reach rule keeps and the reference does not,
whether some box on T's chain was culled with entry distance d >= hitInfo.t although T's own t is <= that hitInfo.t.

    python tests/tools/far_diag.py [distance]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import np_path
import np_restatement as npr
from gdpathtracing_amd import scenes, wire
from oracle import binding as ob

F = np.float32


def aabb(o, rD, bmin, bmax):       # main.glsl:259-268, float32, minNum / maxNum
    with np.errstate(all="ignore"):
        tx1, tx2 = (bmin[0] - o[0]) * rD[0], (bmax[0] - o[0]) * rD[0]
        tmin, tmax = np.fmin(tx1, tx2), np.fmax(tx1, tx2)
        ty1, ty2 = (bmin[1] - o[1]) * rD[1], (bmax[1] - o[1]) * rD[1]
        tmin, tmax = np.fmax(tmin, np.fmin(ty1, ty2)), np.fmin(tmax, np.fmax(ty1, ty2))
        tz1, tz2 = (bmin[2] - o[2]) * rD[2], (bmax[2] - o[2]) * rD[2]
        tmin, tmax = np.fmax(tmin, np.fmin(tz1, tz2)), np.fmin(tmax, np.fmax(tz1, tz2))
    return tmin if (tmax >= tmin and tmax > 0) else F(1e30)


def tri_t(o, d, g):                 # main.glsl:224-257 without the hitInfo.t test; None when rejected
    with np.errstate(all="ignore"):
        v0, v1, v2 = (g[k][:3].astype(F) for k in range(3))
        e1, e2 = v1 - v0, v2 - v0
        pvec = np_path._cross(d, e2)
        det = np_path._dot(e1, pvec)
        if abs(det) < F(1e-5):
            return None
        inv = F(1.0) / det
        tvec = o - v0
        u = np_path._dot(tvec, pvec) * inv
        if u < 0 or u > 1:
            return None
        qvec = np_path._cross(tvec, e1)
        v = np_path._dot(d, qvec) * inv
        if v < 0 or u + v > 1:
            return None
        t = np_path._dot(e2, qvec) * inv
        if t < 0:
            return None
        return F(t), F(u), F(v), bool(np_path._dot(np_path._cross(e1, e2), d) > 0)


def walk(ref, o, d, reach_only):
    """ray_trace_tlas / ray_trace_blas (main.glsl:270-350) in float32; returns the hit and, per BLAS node examined, the entry
    distance of its box and hitInfo.t at that moment"""
    tl, bv, geom = ref.tlas_nodes, ref.bvh_nodes, ref.tri_geom["vertices"]
    with np.errstate(all="ignore"):
        rD = F(1.0) / d
    hit_t, hit_tri, hit_inst = F(1e9), -1, -1
    extra = None    # u, v, front, local position, local out_dir, hitInfo.blas
    min_t, blas = F(1e9), 0
    seen = {}       # (inst, node) -> (d_box, hitInfo.t when its parent was expanded, kept)
    stack = [0]
    while stack:
        node = tl[stack.pop()]
        if node["leftRight"] == 0:
            i = int(node["blas"])
            inst = ref.instances[i]
            inv = inst["inverse_transform"].astype(F)
            lo, ld = np_path._mat_point(inv, o), np_path._mat_dir(inv, d)
            with np.errstate(all="ignore"):
                lrD = F(1.0) / ld
            bs = [int(inst["blas_index"])]
            while bs:
                ni = bs.pop()
                n = bv[ni]
                if n["tri_count"] > 0:
                    for k in range(int(n["tri_count"])):
                        ti = int(n["first_tri_index"]) + k
                        got = tri_t(lo, ld, geom[ti])
                        if got is not None and not got[0] > hit_t:
                            hit_t, hit_tri, hit_inst = got[0], ti, i
                            extra = [got[1], got[2], got[3], lo + got[0] * ld, -ld]
                    continue
                l, r = int(n["left_child"]), int(n["right_child"])
                d1 = aabb(lo, lrD, bv[l]["aabbMin"][:3], bv[l]["aabbMax"][:3])
                d2 = aabb(lo, lrD, bv[r]["aabbMin"][:3], bv[r]["aabbMax"][:3])
                if reach_only:
                    lv = d1 < F(1e30) if bv[l]["tri_count"] > 0 else True
                    rv = d2 < F(1e30) if bv[r]["tri_count"] > 0 else True
                else:
                    lv, rv = d1 < hit_t, d2 < hit_t
                seen[(i, l)] = (d1, hit_t, lv)
                seen[(i, r)] = (d2, hit_t, rv)
                if d1 < d2:
                    if rv: bs.append(r)
                    if lv: bs.append(l)
                else:
                    if lv: bs.append(l)
                    if rv: bs.append(r)
            if hit_t < min_t:        # main.glsl:324-327
                blas, min_t = i, hit_t
            continue
        l, r = int(node["leftRight"]) & 0xffff, int(node["leftRight"]) >> 16
        d1 = aabb(o, rD, tl[l]["aabbMin"], tl[l]["aabbMax"])
        d2 = aabb(o, rD, tl[r]["aabbMin"], tl[r]["aabbMax"])
        if reach_only:
            lv = d1 < F(1e30) if tl[l]["leftRight"] == 0 else True
            rv = d2 < F(1e30) if tl[r]["leftRight"] == 0 else True
        else:
            lv, rv = d1 < hit_t, d2 < hit_t
        if d1 < d2:
            if rv: stack.append(r)
            if lv: stack.append(l)
        else:
            if lv: stack.append(l)
            if rv: stack.append(r)
    return hit_t, hit_tri, hit_inst, seen, (extra + [blas] if extra else None)


def chain_to(ref, inst, tri):
    """the BLAS nodes from the instance's root down to the leaf that holds triangle `tri`"""
    bv = ref.bvh_nodes
    path, out = [], None

    def rec(ni):
        nonlocal out
        n = bv[ni]
        path.append(ni)
        if n["tri_count"] > 0:
            if int(n["first_tri_index"]) <= tri < int(n["first_tri_index"]) + int(n["tri_count"]):
                out = list(path)
        else:
            rec(int(n["left_child"]))
            if out is None:
                rec(int(n["right_child"]))
        path.pop()

    rec(int(ref.instances[inst]["blas_index"]))
    return out


def primary_rays(cam, w, h, px, py, frame):
    """main.glsl:405-421, as np_path.trace_frame makes them"""
    seed = npr.prng_seed(px, py, frame)
    seed, r = npr.pcg2d(seed)
    js, jc = np_path._sincos(F(6.2831853) * (r[:, 1] * F(0.25)))
    sx = (px.astype(F) + jc) / F(w) * F(2.0) - F(1.0)
    sy = (py.astype(F) + js) / F(h) * F(2.0) - F(1.0)
    m = cam["ivp"].reshape(-1).astype(F)
    nx, ny = sx, -sy
    ww = m[3] * nx + m[7] * ny + m[11] + m[15]
    world = np.stack([(m[0] * nx + m[4] * ny + m[8] + m[12]) / ww, (m[1] * nx + m[5] * ny + m[9] + m[13]) / ww,
                      (m[2] * nx + m[6] * ny + m[10] + m[14]) / ww], axis=-1)
    cpos = np.array([cam["position"].reshape(-1)[k] for k in range(3)], dtype=F)
    return np.broadcast_to(cpos, world.shape).astype(F), np_path._normalize(world - cpos[None, :]), seed


def main(dist=500000.0, n_bounces=2, w=192, h=108, frames=(1, 2), quiet=False):
    """returns (explained, exact ties, unexplained, rays compared)"""
    sc = scenes.demo_scene(1500)
    ref = ob.build_scene(sc)
    say = (lambda *a, **k: None) if quiet else print
    compared = 0
    fov = float(np.degrees(2.0 * np.arctan(3.2 / dist)))
    sc.camera = scenes.CameraDesc(scenes.transform12(None, (0.3, 0.2, dist)), fov_deg=fov)
    explained = unexplained = ties = 0
    for frame in frames:
        cam = scenes.camera_block(sc.camera, w, h)
        cam["frame_index"] = frame
        ys, xs = np.mgrid[0:h, 0:w]     # every pixel: both modes are replayed for each primary ray
        px, py = xs.reshape(-1).astype(np.int64), ys.reshape(-1).astype(np.int64)
        with np.errstate(all="ignore"):
            rays_o, rays_d, seed = primary_rays(cam, w, h, px, py, frame)
        for bounce in range(n_bounces + 1):
            keep, hv, n_diff = [], [], 0
            compared += len(px)
            for k in range(len(px)):
                o, d = rays_o[k], rays_d[k]
                t_ref, tri_ref, inst_ref, seen, ex = walk(ref, o, d, False)
                if tri_ref >= 0:
                    keep.append(k)
                    hv.append([tri_ref] + ex)
                t_rch, tri_rch, inst_rch, _, _ = walk(ref, o, d, True)
                if tri_ref == tri_rch and inst_ref == inst_rch:
                    continue
                n_diff += 1
                line = "  frame %d bounce %d px (%d,%d): reference keeps tri %d (inst %d) t=%.9g; reach rule keeps tri %d (inst %d) t=%.9g" % (
                    frame, bounce, px[k], py[k], tri_ref, inst_ref, t_ref, tri_rch, inst_rch, t_rch)
                if t_rch == t_ref:
                    ties += 1
                    say(line + "  -> an exact tie decided by order (the tie walk's business on the device)", flush=True)
                    continue
                # the reach rule keeps something strictly closer that the reference walk never tested: which box hid it?
                culprit = None
                for ni in chain_to(ref, inst_rch, tri_rch)[1:]:
                    s_ = seen.get((inst_rch, ni))
                    if s_ is not None and not s_[2]:
                        culprit = (ni, s_[0], s_[1])
                        break
                if culprit and culprit[1] >= culprit[2] and t_rch <= culprit[2]:
                    explained += 1
                    say(line + "\n      box of node %d on its chain: entry distance %.9g >= hitInfo.t %.9g when its parent was expanded, although its own "
                          "triangle's t is %.9g (entry - t = %.3g = %.1f ulp of t; |origin| = %.6g): culled by `d < hitInfo.t` (main.glsl:290-291)" % (
                              culprit[0], culprit[1], culprit[2], t_rch, float(culprit[1]) - float(t_rch),
                              (float(culprit[1]) - float(t_rch)) / float(np.spacing(F(t_rch))), float(np.abs(o).max())), flush=True)
                else:
                    unexplained += 1
                    say(line + "\n      NOT explained by a culled chain box: %r" % (culprit,), flush=True)
            say("frame %d bounce %d: %d rays, %d where the reference walk and the reach rule keep different triangles" % (
                frame, bounce, len(px), n_diff), flush=True)
            if bounce == n_bounces or not keep:
                break
            # the next rays of the REFERENCE's paths (main.glsl:378-397), with np_path's shading and sampling
            keep = np.array(keep, dtype=np.int64)
            with np.errstate(all="ignore"):
                tri = np.array([x[0] for x in hv], dtype=np.int64)
                u = np.array([x[1] for x in hv], dtype=F)
                v = np.array([x[2] for x in hv], dtype=F)
                front = np.array([x[3] for x in hv], dtype=bool)
                lpos = np.stack([x[4] for x in hv]).astype(F)
                lout = np.stack([x[5] for x in hv]).astype(F)
                blas = np.array([x[6] for x in hv], dtype=np.int64)
                s_ = np_path._shading(ref, tri, blas, lpos, lout, u, v, front)
                seed2, xi = npr.pcg2d(seed[keep])
                new_d = np_path._sample_brdf(s_, xi)
                new_o = s_["position"] + s_["normal"] * F(0.001)
                lambert_in = np_path._dot(s_["normal"], new_d)
                go = ~(lambert_in <= 0)
            px, py, seed = px[keep][go], py[keep][go], seed2[go]
            rays_o, rays_d = new_o[go].astype(F), new_d[go].astype(F)
    say("summary at distance %g: %d rays explained by a chain box whose float entry distance exceeds its own triangle's t and reaches "
        "hitInfo.t; %d exact ties; %d unexplained" % (dist, explained, ties, unexplained))
    return explained, ties, unexplained, compared


if __name__ == "__main__":
    main(float(sys.argv[1]) if len(sys.argv) > 1 else 500000.0)
