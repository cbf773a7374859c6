"""Conservativeness of the native walk's quantised box tests, tested DIRECTLY (not sampled through images).

The four-child record step (Traversal<.., W4>::node_step4, csrc/jpt_trace_core.h) tests a ray against child boxes whose
planes are quantised to 8 bits (csrc/jpt_nodeq.h), with reciprocals from v_rcp_f32 and `b = origin * rD - o * rD`.  Which hit
counts is decided by exact arithmetic elsewhere (Moller-Trumbore as written, the reach records), so the only thing these
tests may never do is DROP a child whose box holds a triangle that Moller-Trumbore accepts.  The property, per (record,
ray, triangle inside child k) triple with hitInfo.t at its tightest (= the triangle's own distance):

    intersectTriangle as written (main.glsl:224-257, float32) accepts at t
    AND the reference keeps the box of the triangle's leaf: intersectAABB as written (main.glsl:259-268)
        returns d < hitInfo.t (main.glsl:290-291)                                   ==>   the step keeps child k,

with hitInfo.t = 1e9 (nothing found yet) and with hitInfo.t = t itself (the tightest bound a walk can hold while the
triangle is still to be found).

The second premise is the reach rule (csrc/jpt_types.h): a hit the reference's own leaf-box test rejects is not a hit of the
reference (a "crack"), the default route declines it anyway (reach records), so the walk owes nothing for it.  With it the
property holds at EVERY distance tested (ray origins up to 1e6 mesh sizes away).  Without it -- the watertight route,
JPT_BUILD_SAH_WATERTIGHT: Moller-Trumbore alone decides -- it holds while the ray origin is within a few tens of mesh
sizes: farther out Moller-Trumbore's own rounding accepts rays that pass OUTSIDE the padded box (its error grows with the
distance, the builder's padding does not); the distance of the first such case is reported and recorded in DESIGN.md.

Cases are generated with numpy (seeded): random and adversarial -- axis-parallel and near-parallel directions, flat nodes
(scale = FLT_MIN on an axis), slivers, child planes exactly on grid steps, and ray origins up to 1e5 scene sizes away.
Child boxes are built the way the builder builds them: the exact float min/max of what they hold, padded by 2e-6 x the
largest coordinate magnitude of the mesh (SahBlasBuilder::prepare).  The step runs through the C ABI's audit entry
(jpt_debug_node_step4): on the GPU the very device function, on the CPU a host restatement with the reciprocals perturbed
by up to +-2 ulps.  Also here: quantize_node4's own contract (the quantised box contains the float box, and is at most a
step + slack larger per side)."""
import ctypes as C

import numpy as np
import pytest

from gdpathtracing_amd import capi

F = np.float32
EMPTY = np.int32(-2 ** 31)
NODE4 = np.dtype([("lo_x", "<f4", (4,)), ("lo_y", "<f4", (4,)), ("lo_z", "<f4", (4,)), ("child", "<i4", (4,)),
                  ("hi_x", "<f4", (4,)), ("hi_y", "<f4", (4,)), ("hi_z", "<f4", (4,)), ("_pad", "<u4", (4,))])
NODEQ = np.dtype([("origin", "<f4", (3,)), ("sx", "<f4"), ("sy", "<f4"), ("sz", "<f4"), ("lo", "<u4", (2,)),
                  ("lo_z", "<u4"), ("hi", "<u4", (3,)), ("child", "<i4", (4,))])
CASE = np.dtype([("o", "<f4", (3,)), ("d", "<f4", (3,)), ("t_max", "<f4"), ("node", "<u4")])
assert NODE4.itemsize == 128 and NODEQ.itemsize == 64 and CASE.itemsize == 32
SLACK = 1.0 / 256.0


def _dot(a, b):
    return a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1] + a[..., 2] * b[..., 2]


def _cross(a, b):
    return np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1], a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                     a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], axis=-1)


def moller_trumbore(o, d, v0, v1, v2):
    """intersectTriangle (main.glsl:224-257) in float32, one IEEE operation per operator, vectorised; hitInfo.t = 1e9.
    Returns (accepted, t)."""
    with np.errstate(all="ignore"):
        e1, e2 = v1 - v0, v2 - v0
        pvec = _cross(d, e2)
        det = _dot(e1, pvec)
        inv = F(1.0) / det
        tvec = o - v0
        u = _dot(tvec, pvec) * inv
        qvec = _cross(tvec, e1)
        v = _dot(d, qvec) * inv
        t = _dot(e2, qvec) * inv
        out = (np.abs(det) < F(1e-5)) | (u < 0) | (u > 1) | (v < 0) | (u + v > 1) | (t < 0) | (t > F(1e9))
        out |= ~np.isfinite(t)
    return ~out, t


def intersect_aabb(o, d, bmin, bmax):
    """intersectAABB (main.glsl:259-268) in float32 with ray.rD = 1.0 / ray.d, min / max ignoring a NaN operand; True = the
    box is entered (the function returns tmin, not 1e30)"""
    with np.errstate(all="ignore"):
        rD = F(1.0) / d
        t1, t2 = (bmin - o) * rD, (bmax - o) * rD
        lo, hi = np.fmin(t1, t2), np.fmax(t1, t2)
        tmin = np.fmax(np.fmax(lo[:, 0], lo[:, 1]), lo[:, 2])
        tmax = np.fmin(np.fmin(hi[:, 0], hi[:, 1]), hi[:, 2])
        return np.where((tmax >= tmin) & (tmax > 0), tmin, F(1e30))


def make_cases(rng, n, ratio, kind):
    """n (record, ray, triangle in child k) triples.  `ratio` = distance of the ray origin from the mesh in units of the
    mesh's size; `kind` picks the adversarial family.  Returns (nodes NODE4[n], cases CASE[n], slot[n], reachable[n], tight[n])
    of the triples whose triangle Moller-Trumbore accepts; reachable = the reference keeps the triangle's leaf box; tight =
    hitInfo.t is the triangle's own distance (else 1e9)."""
    M = F(10.0) ** rng.uniform(-1, 2, size=n).astype(F)                  # the mesh's coordinate magnitude: 0.1 .. 100
    centre = (rng.uniform(-1, 1, size=(n, 3)) * M[:, None] * 0.5).astype(F)
    size = (M * F(10.0) ** rng.uniform(-3.5, -0.3, size=n)).astype(F)     # triangle size: 3e-4 .. 0.5 of the mesh
    tri = (centre[:, None, :] + rng.uniform(-1, 1, size=(n, 3, 3)).astype(F) * size[:, None, None]).astype(F)
    if kind == "sliver":
        tri[:, 2] = (tri[:, 0] + (tri[:, 1] - tri[:, 0]) * F(0.5) + rng.uniform(-1, 1, size=(n, 3)).astype(F) * size[:, None] * F(1e-4)).astype(F)
    flat_axis = rng.randint(0, 3, size=n)
    if kind == "flat":                                                     # everything the node holds lies in one plane
        for a in range(3):
            m = flat_axis == a
            tri[m, :, a] = tri[m, 0:1, a]
    pad = (M * F(2e-6) + F(1e-30)).astype(F)                               # SahBlasBuilder::prepare
    # child k: the triangle's exact box, grown by a random amount (an ancestor's child holds more than this triangle)
    grow = (size[:, None] * F(10.0) ** rng.uniform(-3, 1.5, size=(n, 3)).astype(F) * (rng.uniform(size=(n, 3)) < 0.6)).astype(F)
    if kind == "flat":
        grow[np.arange(n), flat_axis] = 0
    c_lo = (tri.min(axis=1) - grow * rng.uniform(size=(n, 3)).astype(F)).astype(F)
    c_hi = (tri.max(axis=1) + grow * rng.uniform(size=(n, 3)).astype(F)).astype(F)
    # the other three children: boxes around the first, sometimes empty slots
    nodes = np.zeros(n, NODE4)
    slot = rng.randint(0, 4, size=n)
    spread = (np.maximum(c_hi - c_lo, size[:, None]) * F(10.0) ** rng.uniform(-1, 1.7, size=(n, 1)).astype(F)).astype(F)
    for k in range(4):
        olo = (c_lo + rng.uniform(-1, 1, size=(n, 3)).astype(F) * spread).astype(F)
        ohi = (olo + rng.uniform(0, 1, size=(n, 3)).astype(F) * spread).astype(F)
        if kind == "flat":
            olo[np.arange(n), flat_axis] = c_lo[np.arange(n), flat_axis]
            ohi[np.arange(n), flat_axis] = c_hi[np.arange(n), flat_axis]
        mine = slot == k
        lo = np.where(mine[:, None], c_lo, olo)
        hi = np.where(mine[:, None], c_hi, ohi)
        lo, hi = (lo - pad[:, None]).astype(F), (hi + pad[:, None]).astype(F)
        empty = (~mine) & (rng.uniform(size=n) < 0.25)
        for a, (fl, fh) in enumerate((("lo_x", "hi_x"), ("lo_y", "hi_y"), ("lo_z", "hi_z"))):
            nodes[fl][:, k] = np.where(empty, F(3.4e38), lo[:, a])
            nodes[fh][:, k] = np.where(empty, F(-3.4e38), hi[:, a])
        nodes["child"][:, k] = np.where(empty, EMPTY, k + 1)
    if kind == "grid":   # the child's planes exactly on steps of the node's grid (as far as float allows)
        nlo = np.stack([np.where(nodes["child"] != EMPTY, nodes[f], np.inf).min(axis=1) for f in ("lo_x", "lo_y", "lo_z")], axis=1).astype(F)
        nhi = np.stack([np.where(nodes["child"] != EMPTY, nodes[f], -np.inf).max(axis=1) for f in ("hi_x", "hi_y", "hi_z")], axis=1).astype(F)
        step = ((nhi - nlo) / F(254.0)).astype(F)
        for a, (fl, fh) in enumerate((("lo_x", "hi_x"), ("lo_y", "hi_y"), ("lo_z", "hi_z"))):
            cur_lo, cur_hi = nodes[fl][np.arange(n), slot], nodes[fh][np.arange(n), slot]
            ql = np.floor((cur_lo - nlo[:, a]) / np.maximum(step[:, a], F(1e-37)))
            qh = np.ceil((cur_hi - nlo[:, a]) / np.maximum(step[:, a], F(1e-37)))
            nodes[fl][np.arange(n), slot] = (nlo[:, a] + ql.astype(F) * step[:, a]).astype(F)
            nodes[fh][np.arange(n), slot] = np.maximum((nlo[:, a] + qh.astype(F) * step[:, a]).astype(F), cur_hi)
            nodes[fl][np.arange(n), slot] = np.minimum(nodes[fl][np.arange(n), slot], cur_lo)
    # the ray: towards a point of the triangle (interior, or near an edge / a vertex), from `ratio` mesh sizes away
    w = rng.dirichlet((1.0, 1.0, 1.0), size=n).astype(F)
    edge = rng.uniform(size=n) < 0.3
    w[edge, 0] *= F(1e-4)
    w = (w / w.sum(axis=1, keepdims=True)).astype(F)
    p = (tri * w[:, :, None]).sum(axis=1).astype(F)
    direction = rng.normal(size=(n, 3))
    direction /= np.linalg.norm(direction, axis=1, keepdims=True)
    dist = (M * F(ratio) * rng.uniform(0.3, 1.0, size=n).astype(F)).astype(F)
    o = (p - direction.astype(F) * dist[:, None]).astype(F)
    if kind in ("axis", "near_axis"):
        a = rng.randint(0, 3, size=n)
        two = rng.uniform(size=n) < 0.3
        for ax in range(3):
            m = (a != ax) if kind == "axis" else np.zeros(n, bool)
            keep_one = (a == ax)
            # axis-parallel: the origin shares the target's other two coordinates (one for the `two` cases stays free)
            share = m & ~(two & (((a + 1) % 3) == ax))
            o[share, ax] = p[share, ax]
            if kind == "near_axis":
                tiny = ~keep_one
                o[tiny, ax] = (p[tiny, ax] + (dist[tiny] * F(10.0) ** rng.uniform(-9, -5, size=tiny.sum()).astype(F) * rng.choice([-1, 1], size=tiny.sum()).astype(F))).astype(F)
    with np.errstate(all="ignore"):
        d = (p - o).astype(F)
        unit = rng.uniform(size=n) < 0.8                                  # the shader's rays are normalised; the local ray
        inv = F(1.0) / np.sqrt(_dot(d, d))                                # of a scaled instance is not
        d = np.where(unit[:, None], d * inv[:, None], d * (F(10.0) ** rng.uniform(-2, 2, size=(n, 1)).astype(F))).astype(F)
    hit, t = moller_trumbore(o, d, tri[:, 0], tri[:, 1], tri[:, 2])
    ok = hit & np.isfinite(d).all(axis=1) & (np.abs(d).max(axis=1) > 0)
    entry = intersect_aabb(o, d, c_lo, c_hi)         # the leaf's box: the exact min / max of what it holds, unpadded
    tight = rng.uniform(size=n) < 0.5                # hitInfo.t when the record is expanded: the triangle's own t, or 1e9
    t = np.where(tight, t, F(1e9)).astype(F)
    reachable = entry < t                            # `bool leftValid = d1 < hitInfo.t` (main.glsl:290)
    cases = np.zeros(n, CASE)
    cases["o"], cases["d"], cases["t_max"], cases["node"] = o, d, t, np.arange(n)
    keep = np.flatnonzero(ok)
    cases = cases[keep]
    cases["node"] = np.arange(len(keep))
    return nodes[keep], cases, slot[keep], reachable[keep], tight[keep]


def run_step(device, nodes, cases, ulps=0):
    L = capi.lib()
    nodes, cases = np.ascontiguousarray(nodes), np.ascontiguousarray(cases)
    out = np.zeros(len(cases), np.uint8)
    rc = L.jpt_debug_node_step4(device, nodes.ctypes.data_as(C.c_void_p), len(nodes), cases.ctypes.data_as(C.c_void_p), len(cases), ulps,
                                out.ctypes.data_as(C.c_void_p))
    assert rc == 0, L.jpt_debug_last_error()
    return out


def violations(device, rng, n, ratio, kind, ulps=0):
    """(lost although reachable, reachable triples, lost with hitInfo.t = 1e9 -- Moller-Trumbore alone as the premise --,
    triples with hitInfo.t = 1e9, the lost reachable ones)"""
    nodes, cases, slot, reachable, tight = make_cases(rng, n, ratio, kind)
    taken = run_step(device, nodes, cases, ulps)
    lost = (taken >> slot) & 1 == 0
    bad = lost & reachable
    return int(bad.sum()), int(reachable.sum()), int((lost & ~tight).sum()), int((~tight).sum()), (nodes[bad], cases[bad], slot[bad])


KINDS = ["random", "axis", "near_axis", "flat", "sliver", "grid"]
# the ray origin's distance from the mesh, in mesh sizes
RATIOS = (0.5, 3.0, 30.0, 300.0, 3e3, 3e4, 3e5, 1e6)
WATERTIGHT_SAFE_RATIO = 8.0   # Moller-Trumbore alone as the premise: asserted up to here, reported beyond


@pytest.mark.parametrize("kind", KINDS)
def test_host_mirror_keeps_every_reachable_child(hiplib, kind):
    rng = np.random.RandomState(11 + KINDS.index(kind))
    total = 0
    for ratio in RATIOS:
        for ulps in (-2, -1, 0, 1, 2):
            bad, n_reach, lost, n, (bn, bc, bs) = violations(capi_host_only(), rng, 12000, ratio, kind, ulps)
            total += n_reach
            assert bad == 0, (kind, ratio, ulps, bad, n_reach, bc[:2], bs[:2])
            if ratio <= WATERTIGHT_SAFE_RATIO:
                assert lost == 0, (kind, ratio, ulps, lost, n)
    assert total > 30000    # (most generated triples are hits, and reachable; slivers mostly fail `abs(det) < 1e-5`)


def capi_host_only():
    return -1   # JPT_DEVICE_HOST_ONLY


def test_quantised_boxes_contain_the_float_boxes(hiplib):
    """quantize_node4's contract: child k's quantised box [origin + (lo - slack) s, origin + (hi + slack) s] contains its
    float box and is larger by less than one step + slack per side; an empty slot has lo > hi on every axis.  Random,
    flat, degenerate (zero-size, huge, subnormal) nodes."""
    L = capi.lib()
    rng = np.random.RandomState(5)
    n = 200000
    nodes = np.zeros(n, NODE4)
    scale = (F(10.0) ** rng.uniform(-30, 30, size=(n, 1))).astype(F)
    scale[: n // 4] = (F(10.0) ** rng.uniform(-3, 3, size=(n // 4, 1))).astype(F)
    base = (rng.uniform(-1, 1, size=(n, 3)).astype(F) * scale * F(10.0) ** rng.uniform(0, 4, size=(n, 1)).astype(F)).astype(F)
    for k in range(4):
        lo = (base + rng.uniform(0, 1, size=(n, 3)).astype(F) * scale).astype(F)
        hi = (lo + rng.uniform(0, 1, size=(n, 3)).astype(F) * scale * (rng.uniform(size=(n, 3)) < 0.8)).astype(F)   # some flat axes
        empty = rng.uniform(size=n) < 0.2
        empty[rng.randint(0, n, size=n // 50)] = True
        if k == 0:
            empty[:] = False
        for a, (fl, fh) in enumerate((("lo_x", "hi_x"), ("lo_y", "hi_y"), ("lo_z", "hi_z"))):
            nodes[fl][:, k], nodes[fh][:, k] = lo[:, a], hi[:, a]
        nodes["child"][:, k] = np.where(empty, EMPTY, k + 1)
    q = np.zeros(n, NODEQ)
    assert L.jpt_debug_quantize_nodes4(nodes.ctypes.data_as(C.c_void_p), n, q.ctypes.data_as(C.c_void_p)) == 0
    origin = q["origin"].astype(np.float64)
    s = np.stack([q["sx"], q["sy"], q["sz"]], axis=1).astype(np.float64)
    lo_w = np.stack([q["lo"][:, 0], q["lo"][:, 1], q["lo_z"]], axis=1)
    hi_w = q["hi"]
    assert np.array_equal(q["child"], nodes["child"])
    assert (s >= 1.17549435e-38).all()
    for k in range(4):
        live = nodes["child"][:, k] != EMPTY
        ql = ((lo_w >> (8 * k)) & 255).astype(np.float64)
        qh = ((hi_w >> (8 * k)) & 255).astype(np.float64)
        assert (ql[~live] > qh[~live]).all()
        flo = np.stack([nodes["lo_x"][:, k], nodes["lo_y"][:, k], nodes["lo_z"][:, k]], axis=1).astype(np.float64)
        fhi = np.stack([nodes["hi_x"][:, k], nodes["hi_y"][:, k], nodes["hi_z"][:, k]], axis=1).astype(np.float64)
        box_lo = origin + (ql - SLACK) * s
        box_hi = origin + (qh + SLACK) * s
        # containment, in exact (float64) arithmetic up to 1e-12 of the magnitudes involved
        tol = 1e-12 * (np.abs(origin) + 256.0 * s)
        assert (box_lo[live] <= flo[live] + tol[live]).all()
        assert (box_hi[live] >= fhi[live] - tol[live]).all()
        # tightness: less than a step + slack beyond the float box (where the node is not flat on that axis)
        loose_lo = (flo - box_lo)[live]
        loose_hi = (box_hi - fhi)[live]
        bound = ((1.0 + SLACK) * s + tol)[live]
        assert (loose_lo <= bound).all() and (loose_hi <= bound).all()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", KINDS)
def test_device_step_keeps_every_reachable_child(hiplib, kind):
    """The same property on the GPU, through the device function the tracing kernels inline (v_rcp_f32 reciprocals):
    more than 1e7 triples in total over the six families, ray origins up to 1e6 mesh sizes away: 0 reachable children
    lost.  With Moller-Trumbore alone as the premise (the watertight route) 0 lost up to WATERTIGHT_SAFE_RATIO mesh sizes;
    the counts beyond are printed (DESIGN.md section 8 records where the first one appears)."""
    rng = np.random.RandomState(101 + KINDS.index(kind))
    total = 0
    watertight = {}
    for ratio in RATIOS:
        lost_all = n_all = 0
        for rep in range(5):
            bad, n_reach, lost, n, (bn, bc, bs) = violations(0, rng, 100000, ratio, kind)
            total += n_reach
            lost_all += lost
            n_all += n
            assert bad == 0, (kind, ratio, bad, n_reach, bc[:2], bs[:2])
        watertight[ratio] = (lost_all, n_all)
        if ratio <= WATERTIGHT_SAFE_RATIO:
            assert lost_all == 0, (kind, ratio, lost_all, n_all)
    print(kind, "reachable triples tested:", total, "| lost with Moller-Trumbore alone as the premise (lost, triples) by distance:", watertight)
    assert total > 250_000    # (slivers: most fail `abs(det) < 1e-5`; the six families together: 1.1e7 reachable triples)


@pytest.mark.gpu
def test_device_step_agrees_with_the_host_mirror(hiplib):
    """The host restatement is the same arithmetic: with exact reciprocals it keeps the same children as the device step
    on all but the few cases where v_rcp_f32's last bit decides."""
    rng = np.random.RandomState(7)
    nodes, cases, slot, _, _ = make_cases(rng, 200000, 10.0, "random")
    dev = run_step(0, nodes, cases)
    same = np.zeros(len(cases), bool)
    for ulps in (-1, 0, 1):
        same |= run_step(-1, nodes, cases, ulps) == dev
    print("device step == host mirror within 1 ulp of the reciprocals:", int(same.sum()), "of", len(cases))
    assert same.mean() > 0.999
