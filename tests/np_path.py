"""A second, independent statement of the WHOLE path (main.glsl:163-436 + brdfs.glsl) in vectorised float32 numpy, used
ONLY to cross-check the C oracle (tests/test_oracle_kats.py).  Test infrastructure.

What makes it independent of oracle/*.c: another language and another structure -- every pixel of the image advances
together as arrays, and the closest hit is found WITHOUT any tree: every triangle of every instance is tested against
every ray (what the oracle's JPTO_FLAG_NO_CULL mode must then reproduce bit for bit).  What it shares with the oracle
is only what has to be shared: the wire-format input arrays and the pinned float semantics of DESIGN.md section 2 (one
IEEE binary32 operation per + - * / sqrt in source order, minNum/maxNum, mix(a,b,t) = a*(1-t) + b*t, normalize(v) =
v * (1 / sqrt(dot)), the pinned sin/cos routine); numpy's float32 ufuncs are exactly such single operations.

Citations: M = main.glsl, B = brdfs.glsl under project/addons/jar_path_tracing/src/shaders/ of the reference.
Textures are not supported (the scenes this is run on have none).
"""
import numpy as np

import np_restatement as npr

F = np.float32
PI = F(3.141592653589793238462643)


def _dot(a, b):
    return a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1] + a[..., 2] * b[..., 2]


def _cross(a, b):
    return np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1], a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                     a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], axis=-1)


def _normalize(v):
    inv = F(1.0) / np.sqrt(_dot(v, v))
    return v * inv[..., None]


def _mix(a, b, t):
    return a * (F(1.0) - t) + b * t


def _sincos(x):
    """the pinned routine (oracle_pins.h p_sincos / jpt_device_math.h sincos_), arguments >= 0 here"""
    ax = np.abs(x)
    y = np.floor(ax * F(1.27323954473516))
    j = y.astype(np.int64)
    odd = (j & 1) == 1
    j = np.where(odd, j + 1, j)
    y = np.where(odd, y + F(1.0), y)
    j = j & 7
    r = ((ax - y * F(0.78515625)) - y * F(2.4187564849853515625e-4)) - y * F(3.77489497744594108e-8)
    z = r * r
    ps = ((F(-1.9515295891e-4) * z + F(8.3321608736e-3)) * z - F(1.6666654611e-1)) * z * r + r
    pc = ((F(2.443315711809948e-5) * z - F(1.388731625493765e-3)) * z + F(4.166664568298827e-2)) * z * z - F(0.5) * z + F(1.0)
    s = np.where(j == 0, ps, np.where(j == 2, pc, np.where(j == 4, -ps, -pc)))
    c = np.where(j == 0, pc, np.where(j == 2, -ps, np.where(j == 4, -pc, ps)))
    return np.where(x < 0, -s, s), c


def _mat_point(m, p):   # column-major mat4 * (p, 1), summed left to right (M:200)
    return np.stack([m[0] * p[..., 0] + m[4] * p[..., 1] + m[8] * p[..., 2] + m[12],
                     m[1] * p[..., 0] + m[5] * p[..., 1] + m[9] * p[..., 2] + m[13],
                     m[2] * p[..., 0] + m[6] * p[..., 1] + m[10] * p[..., 2] + m[14]], axis=-1)


def _mat_dir(m, d):     # mat4 * (d, 0)
    return np.stack([m[0] * d[..., 0] + m[4] * d[..., 1] + m[8] * d[..., 2], m[1] * d[..., 0] + m[5] * d[..., 1] + m[9] * d[..., 2],
                     m[2] * d[..., 0] + m[6] * d[..., 1] + m[10] * d[..., 2]], axis=-1)


def _leaf_triangles(bvh, root):
    """every triangle index under BLAS root `root` (tree used only as a list of triangle ranges)"""
    out, stack = [], [int(root)]
    while stack:
        n = bvh[stack.pop()]
        if n["tri_count"] > 0:
            out.extend(range(int(n["first_tri_index"]), int(n["first_tri_index"]) + int(n["tri_count"])))
        else:
            stack += [int(n["right_child"]), int(n["left_child"])]
    return out


def _closest_hit(ref, o, d):
    """M:224-257 on every triangle of every instance, M:316-327 per instance.  Returns per ray: t, tri, blas (hitInfo.blas),
    local position, local out_dir, u, v, front."""
    n = len(o)
    t_best = np.full(n, F(1e9), dtype=F)
    tri_best = np.zeros(n, dtype=np.int64)
    blas = np.zeros(n, dtype=np.int64)
    min_t = np.full(n, F(1e9), dtype=F)
    pos = np.zeros((n, 3), dtype=F)
    out_dir = np.zeros((n, 3), dtype=F)
    bu = np.zeros(n, dtype=F)
    bv = np.zeros(n, dtype=F)
    front = np.zeros(n, dtype=bool)
    geom = ref.tri_geom["vertices"]
    for i, inst in enumerate(ref.instances):
        inv = inst["inverse_transform"].astype(F)
        lo, ld = _mat_point(inv, o), _mat_dir(inv, d)
        for ti in _leaf_triangles(ref.bvh_nodes, inst["blas_index"]):
            v0, v1, v2 = (geom[ti][k][:3].astype(F) for k in range(3))
            e1, e2 = v1 - v0, v2 - v0
            pvec = _cross(ld, e2[None, :])
            det = _dot(e1[None, :], pvec)
            inv_det = F(1.0) / det
            tvec = lo - v0[None, :]
            u = _dot(tvec, pvec) * inv_det
            qvec = _cross(tvec, e1[None, :])
            v = _dot(ld, qvec) * inv_det
            t = _dot(e2[None, :], qvec) * inv_det
            ok = ~(np.abs(det) < F(1e-5)) & ~((u < 0) | (u > 1)) & ~((v < 0) | (u + v > 1)) & ~((t < 0) | (t > t_best))
            t_best = np.where(ok, t, t_best)
            tri_best = np.where(ok, ti, tri_best)
            pos = np.where(ok[:, None], lo + t[:, None] * ld, pos)
            out_dir = np.where(ok[:, None], -ld, out_dir)
            bu, bv = np.where(ok, u, bu), np.where(ok, v, bv)
            g = _cross(e1[None, :], e2[None, :])
            front = np.where(ok, _dot(g, ld) > 0, front)
        closer = t_best < min_t          # M:324-327
        blas = np.where(closer, i, blas)
        min_t = np.where(closer, t_best, min_t)
    return t_best, tri_best, blas, pos, out_dir, bu, bv, front


def _shading(ref, tri, blas, pos, out_dir, u, v, front):   # M:194-222
    td = ref.tri_data[tri]
    inst = ref.instances[blas]
    words = np.ascontiguousarray(ref.instances).view(np.uint32).reshape(-1)
    w = blas.astype(np.int64) * 44 + 41 + td["material_index"].astype(np.int64)     # b.materials[tri.materialIndex], unchecked
    mat_id = np.where(w < len(words), words[np.minimum(w, len(words) - 1)], 0)
    mat_id = np.where(mat_id >= len(ref.materials), 0, mat_id)
    mat = ref.materials[mat_id]
    assert (mat["albedo_texture_index"] < 0).all(), "np_path has no textures"
    m = np.moveaxis(inst["transform"].astype(F), -1, 0)      # m[k] = component k of every ray's matrix
    position = _mat_point(m, pos)
    od = _normalize(_mat_dir(m, out_dir))
    w0 = F(1.0) - u - v
    n0 = td["n0"].astype(F)
    n1, n2 = td["n1"][..., :3].astype(F), td["n2"][..., :3].astype(F)
    nrm = n0 * w0[:, None] + n1 * u[:, None] + n2 * v[:, None]
    nrm = _normalize(_mat_dir(m, nrm))
    nrm = np.where(front[:, None], nrm, -nrm)
    lambert_out = _dot(nrm, od)
    em = mat["emission"].astype(F)
    emission = em[:, :3] * np.fmax(F(0.0), em[:, 3])[:, None]
    albedo = mat["albedo"].astype(F)[:, :3]
    metal = mat["metallic"].astype(F)
    f0 = _mix(F(0.02), albedo, metal[:, None])
    diffuse = albedo - albedo * metal[:, None]
    rough = np.fmax(F(0.006), mat["roughness"].astype(F))
    return dict(position=position, out_dir=od, normal=nrm, lambert_out=lambert_out, emission=emission, f0=f0, diffuse=diffuse, rough=rough)


def _schlick_factor(c):
    f = F(1.0) - c
    f2 = f * f
    return f2 * f2 * f


def _brdf(s, l):   # B:10-38
    ndl, ndv = _dot(s["normal"], l), s["lambert_out"]
    h = _normalize(l + s["out_dir"])
    hdv = _dot(h, s["out_dir"])
    f90 = (hdv * hdv) * (F(2.0) * s["rough"]) + F(0.5)
    dfres = _mix(F(1.0), f90, _schlick_factor(ndv)) * _mix(F(1.0), f90, _schlick_factor(ndl))
    r = dfres[:, None] * s["diffuse"]
    hdn = _dot(h, s["normal"])
    a2 = s["rough"] * s["rough"]
    den = hdn * (a2 - F(1.0)) + F(1.0)
    dist = a2 / (den * den)
    masking = ndl * np.sqrt((ndv - a2 * ndv) * ndv + a2)
    shadowing = ndv * np.sqrt((ndl - a2 * ndl) * ndl + a2)
    geo = F(0.5) / (masking + shadowing)
    ff = _schlick_factor(np.fmax(F(0.0), hdv))
    spec = _mix(s["f0"], F(1.0), ff[:, None])
    r = r + (dist * geo)[:, None] * spec
    r = r / PI
    return np.where((np.fmin(ndl, ndv) < 0)[:, None], F(0.0), r)


def _diffuse_prob(s):
    return np.fmin(F(0.5), _dot(s["diffuse"], np.array([0.2126, 0.7152, 0.0722], dtype=F)[None, :]))


def _sample_brdf(s, xi):   # B:112-128
    n = s["normal"]
    sign = np.where(n[:, 2] > 0, F(1.0), F(-1.0))
    a = F(-1.0) / (sign + n[:, 2])
    b = n[:, 0] * n[:, 1] * a
    c0 = np.stack([F(1.0) + sign * n[:, 0] * n[:, 0] * a, sign * b, -sign * n[:, 0]], axis=-1)
    c1 = np.stack([b, sign + n[:, 1] * n[:, 1] * a, -n[:, 1]], axis=-1)
    c2 = n
    pd = _diffuse_prob(s)
    x0, x1 = xi[:, 0], xi[:, 1]
    # diffuse branch: cosine hemisphere (B:95-101)
    xd = x0 / pd
    sp, cp = _sincos((F(2.0) * PI) * xd)
    radius = np.sqrt(x1)
    local_d = np.stack([radius * cp, radius * sp, np.sqrt(F(1.0) - radius * radius)], axis=-1)
    # specular branch: VNDF (B:40-54, :69-72)
    xs = (x0 - pd) / (F(1.0) - pd)
    view = np.stack([_dot(c0, s["out_dir"]), _dot(c1, s["out_dir"]), _dot(c2, s["out_dir"])], axis=-1)
    rg = s["rough"]
    tv = _normalize(np.stack([view[:, 0] * rg, view[:, 1] * rg, view[:, 2]], axis=-1))
    sp, cp = _sincos((F(2.0) * PI) * xs)
    z = F(1.0) - x1 * (F(1.0) + tv[:, 2])
    st = np.sqrt(np.fmax(F(0.0), F(1.0) - z * z))
    sm = np.stack([st * cp, st * sp, z], axis=-1) + tv
    hv = _normalize(np.stack([sm[:, 0] * rg, sm[:, 1] * rg, sm[:, 2]], axis=-1))
    k = F(2.0) * _dot(hv, view)
    local_s = -(view - hv * k[:, None])
    local = np.where((x0 < pd)[:, None], local_d, local_s)
    return c0 * local[:, 0:1] + c1 * local[:, 1:2] + c2 * local[:, 2:3]


def _density(s, l):   # B:130-138 with :56-81, :103-105
    pd = _diffuse_prob(s)
    h = _normalize(l + s["out_dir"])
    hdv = _dot(h, s["out_dir"])
    hdn = _dot(h, s["normal"])
    ndv = s["lambert_out"]
    a2 = s["rough"] * s["rough"]
    ia2 = F(1.0) - a2
    den = ndv + np.sqrt(a2 + ia2 * ndv * ndv)
    d_vis = np.fmax(F(0.0), hdv) * (F(2.0) / PI) / den
    msq = F(1.0) - ia2 * hdn * hdn
    vndf = np.where(hdn < 0, F(0.0), d_vis * a2 / (msq * msq))
    spec = vndf / (F(4.0) * hdv)
    diff = np.fmax(F(0.0), _dot(s["normal"], l)) / PI
    return _mix(spec, diff, pd)


def trace_frame(ref, cam, width, height, max_bounces):
    """One dispatch of main.glsl (M:404-436): float radiance [H, W, 3] and reversed-Z depth [H, W], float32."""
    with np.errstate(all="ignore"):
        ys, xs = np.mgrid[0:height, 0:width]
        px, py = xs.reshape(-1), ys.reshape(-1)
        n = len(px)
        seed = npr.prng_seed(px, py, int(cam["frame_index"]))
        seed, r = npr.pcg2d(seed)
        js, jc = _sincos(F(6.2831853) * (r[:, 1] * F(0.25)))                 # box_muller keeps theta only (M:183-187)
        sx = (px.astype(F) + jc) / F(width) * F(2.0) - F(1.0)
        sy = (py.astype(F) + js) / F(height) * F(2.0) - F(1.0)
        nx, ny = sx, -sy
        m = cam["ivp"].astype(F)
        wx = m[0] * nx + m[4] * ny + m[8] + m[12]
        wy = m[1] * nx + m[5] * ny + m[9] + m[13]
        wz = m[2] * nx + m[6] * ny + m[10] + m[14]
        ww = m[3] * nx + m[7] * ny + m[11] + m[15]
        world = np.stack([wx / ww, wy / ww, wz / ww], axis=-1)
        cpos = np.array([cam["position"][k] for k in range(3)], dtype=F)
        o = np.broadcast_to(cpos, (n, 3)).astype(F)
        d = _normalize(world - cpos[None, :])
        far, near = F(cam["far"]), F(cam["near"])
        depth = np.full(n, far, dtype=F)
        radiance = np.zeros((n, 3), dtype=F)
        throughput = np.ones((n, 3), dtype=F)
        alive = np.ones(n, dtype=bool)
        for i in range(max_bounces + 1):                                          # M:377
            t, tri, blas, lpos, lout, u, v, front = _closest_hit(ref, o, d)
            hit = t < F(1e9)
            tsky = F(0.5) * (d[:, 1] + F(1.0))
            sky = np.stack([_mix(F(0.95), F(0.9), tsky) * F(1.0), _mix(F(0.95), F(0.94), tsky) * F(1.0), _mix(F(0.95), F(1.0), tsky) * F(1.0)], axis=-1)
            s = _shading(ref, tri, blas, lpos, lout, u, v, front)
            emission = np.where(hit[:, None], s["emission"], sky)
            radiance = np.where(alive[:, None], radiance + throughput * emission, radiance)
            alive = alive & hit
            if i == 0:
                diff = s["position"] - o
                depth = np.where(alive, np.sqrt(_dot(diff, diff)), depth)
            new_o = s["position"] + s["normal"] * F(0.001)
            seed2, xi = npr.pcg2d(seed)
            seed = np.where(alive[:, None], seed2, seed)
            new_d = _sample_brdf(s, xi)
            dens = _density(s, new_d)
            lambert_in = _dot(s["normal"], new_d)
            o = np.where(alive[:, None], new_o, o)
            d = np.where(alive[:, None], new_d, d)
            alive = alive & ~(lambert_in <= 0)
            f = (_brdf(s, new_d) * lambert_in[:, None]) / dens[:, None]
            throughput = np.where(alive[:, None], throughput * f, throughput)
        depth = far / (far - near) * (F(1.0) - near / depth)
        return radiance.reshape(height, width, 3), depth.reshape(height, width)
