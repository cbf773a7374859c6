"""Independent numpy restatement of pieces of the reference shaders, used ONLY to cross-check the C
oracle (tests).  Written from main.glsl / brdfs.glsl / progressive_rendering.glsl directly, vectorised and
mostly in float64, so it shares no code and no rounding behaviour with oracle/*.c: integer pieces must
agree exactly, float pieces to a tolerance.

Citations: M = main.glsl, B = brdfs.glsl, P = progressive_rendering.glsl under
project/addons/jar_path_tracing/src/shaders/ of the reference.
"""
import numpy as np

U32 = np.uint32


def prng_seed(px, py, frame):  # M:176-181
    with np.errstate(over="ignore"):
        s = np.stack([np.asarray(px, dtype=U32), np.asarray(py, dtype=U32)], axis=-1)
        s = s * U32(0x9E3779B9) + U32(frame)
        s = s ^ (s >> U32(16))
        return s * U32(0x9E3779B9)


def pcg2d(seed):  # M:163-174 ; returns (new_seed, floats)
    with np.errstate(over="ignore"):
        s = np.array(seed, dtype=U32, copy=True)
        s = U32(1664525) * s + U32(1013904223)
        s[..., 0] += U32(1664525) * s[..., 1]
        s[..., 1] += U32(1664525) * s[..., 0]
        s = s ^ (s >> U32(16))
        s[..., 0] += U32(1664525) * s[..., 1]
        s[..., 1] += U32(1664525) * s[..., 0]
        s = s ^ (s >> U32(16))
        f = s.astype(np.float32) * np.float32(2.32830643654e-10)
        return s, f


def primary_ray(ivp_colmajor, cam_pos, width, height, px, py, frame):  # M:405-421, float64
    seed = prng_seed(px, py, frame)
    seed, r = pcg2d(seed)
    theta = 6.2831853 * (r[..., 1].astype(np.float64) * 0.25)          # M:183-187 (R is discarded)
    jitter = np.stack([np.cos(theta), np.sin(theta)], axis=-1)
    pos = np.stack([np.asarray(px, dtype=np.float64), np.asarray(py, dtype=np.float64)], axis=-1)
    screen = (pos + jitter) / np.array([width, height], dtype=np.float64) * 2.0 - 1.0
    ndc = np.stack([screen[..., 0], -screen[..., 1], np.ones_like(screen[..., 0]), np.ones_like(screen[..., 0])], axis=-1)
    m = np.asarray(ivp_colmajor, dtype=np.float64).reshape(4, 4).T       # column-major -> matrix
    world = ndc @ m.T
    world = world[..., :3] / world[..., 3:4]
    d = world - np.asarray(cam_pos, dtype=np.float64)[:3]
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    return d, seed


def intersect_aabb(o, rD, bmin, bmax):  # M:259-268, float32 like the shader
    o, rD, bmin, bmax = (np.asarray(a, dtype=np.float32) for a in (o, rD, bmin, bmax))
    with np.errstate(invalid="ignore", over="ignore"):
        t1 = (bmin - o) * rD
        t2 = (bmax - o) * rD
        tmin = np.fmax.reduce(np.fmin(t1, t2), axis=-1)
        tmax = np.fmin.reduce(np.fmax(t1, t2), axis=-1)
    return np.where((tmax >= tmin) & (tmax > 0), tmin, np.float32(1e30))


def intersect_triangle(o, d, v0, v1, v2, t_max):  # M:224-257, float64; returns (hit, t, u, v, front)
    o, d, v0, v1, v2 = (np.asarray(a, dtype=np.float64) for a in (o, d, v0, v1, v2))
    e1, e2 = v1 - v0, v2 - v0
    pvec = np.cross(d, e2)
    det = np.dot(e1, pvec)
    if abs(det) < 1e-5:
        return False, 0, 0, 0, False
    inv = 1.0 / det
    tvec = o - v0
    u = np.dot(tvec, pvec) * inv
    if u < 0 or u > 1:
        return False, 0, 0, 0, False
    qvec = np.cross(tvec, e1)
    v = np.dot(d, qvec) * inv
    if v < 0 or u + v > 1:
        return False, 0, 0, 0, False
    t = np.dot(e2, qvec) * inv
    if t < 0 or t > t_max:
        return False, 0, 0, 0, False
    return True, t, u, v, bool(np.dot(np.cross(e1, e2), d) > 0)


def _schlick(f0, f90, c):  # B:3-8
    f = (1.0 - c) ** 5
    return f0 * (1 - f) + f90 * f


def brdf(n, v, lambert_out, diffuse_albedo, f0, rough, l):  # B:10-38, float64
    n, v, l = (np.asarray(a, dtype=np.float64) for a in (n, v, l))
    ndl, ndv = float(n @ l), float(lambert_out)
    if min(ndl, ndv) < 0:
        return np.zeros(3)
    h = (l + v) / np.linalg.norm(l + v)
    hdv = float(h @ v)
    f90 = hdv * hdv * 2 * rough + 0.5
    dfres = _schlick(1.0, f90, ndv) * _schlick(1.0, f90, ndl)
    out = dfres * np.asarray(diffuse_albedo, dtype=np.float64)
    hdn = float(h @ n)
    a2 = rough * rough
    den = hdn * (a2 - 1) + 1                      # un-squared n.h, as the reference (B:27)
    D = a2 / (den * den)
    mask = ndl * np.sqrt((ndv - a2 * ndv) * ndv + a2)
    shad = ndv * np.sqrt((ndl - a2 * ndl) * ndl + a2)
    G = 0.5 / (mask + shad)
    F = _schlick(np.asarray(f0, dtype=np.float64), 1.0, max(0.0, hdv))
    return (out + D * G * F) / np.pi


def diffuse_prob(diffuse_albedo):  # B:107-110
    return min(0.5, float(np.dot(diffuse_albedo, [0.2126, 0.7152, 0.0722])))


def brdf_density(n, v, lambert_out, diffuse_albedo, rough, l):  # B:130-138 with :56-81,:103-105
    n, v, l = (np.asarray(a, dtype=np.float64) for a in (n, v, l))
    p = diffuse_prob(diffuse_albedo)
    h = (l + v) / np.linalg.norm(l + v)
    hdv, hdn = float(h @ v), float(h @ n)
    if hdn < 0:
        vndf = 0.0
    else:
        a2 = rough * rough
        ia = 1 - a2
        ndv = float(lambert_out)
        den = ndv + np.sqrt(a2 + ia * ndv * ndv)
        dvis = max(0.0, hdv) * (2 / np.pi) / den
        m = 1 - ia * hdn * hdn
        vndf = dvis * a2 / (m * m)
    spec = vndf / (4 * hdv)
    diff = max(0.0, float(n @ l)) / np.pi
    return spec * (1 - p) + diff * p


def shading_space(n):  # B:83-93 -> columns
    s = 1.0 if n[2] > 0 else -1.0
    a = -1.0 / (s + n[2])
    b = n[0] * n[1] * a
    return (np.array([1 + s * n[0] * n[0] * a, s * b, -s * n[0]]), np.array([b, s + n[1] * n[1] * a, -n[1]]),
            np.asarray(n, dtype=np.float64))


def sample_brdf(n, v, diffuse_albedo, rough, xi):  # B:112-128, float64
    c0, c1, c2 = shading_space(np.asarray(n, dtype=np.float64))
    M = np.stack([c0, c1, c2], axis=1)
    p = diffuse_prob(diffuse_albedo)
    x0, x1 = float(xi[0]), float(xi[1])
    if x0 < p:
        x0 /= p
        phi = 2 * np.pi * x0
        r = np.sqrt(x1)
        z = np.sqrt(max(0.0, 1 - r * r))
        local = np.array([r * np.cos(phi), r * np.sin(phi), z])
    else:
        x0 = (x0 - p) / (1 - p)
        view = M.T @ np.asarray(v, dtype=np.float64)
        tv = np.array([view[0] * rough, view[1] * rough, view[2]])
        tv /= np.linalg.norm(tv)
        phi = 2 * np.pi * x0
        z = 1 - x1 * (1 + tv[2])
        st = np.sqrt(max(0.0, 1 - z * z))
        hs = np.array([st * np.cos(phi), st * np.sin(phi), z])
        hv = hs + tv
        h = np.array([hv[0] * rough, hv[1] * rough, hv[2]])
        h /= np.linalg.norm(h)
        local = -(view - 2 * np.dot(h, view) * h)
    return M @ local


def aces(x):  # P:19-26
    x = np.asarray(x, dtype=np.float64)
    return np.clip((x * (2.51 * x + 0.03)) / (x * (2.43 * x + 0.59) + 0.14), 0.0, 1.0)


def sky(d):  # M:189-192
    t = 0.5 * (np.asarray(d, dtype=np.float64)[..., 1] + 1.0)
    lo = np.array([0.95, 0.95, 0.95])
    hi = np.array([0.9, 0.94, 1.0])
    return lo * (1 - t[..., None]) + hi * t[..., None]


def temporal_reproject(delta_colmajor, frame_count, screen_u8, depth, fb1, fb2):
    """temporal_reprojection.glsl:30-72 (T) vectorised in float32, one operation per numpy call so the rounding
    sequence is the shader's.  Returns (new_screen u8, written history f32, which) with which = 1 or 2 for the
    history image written."""
    f = np.float32
    H, W = depth.shape
    m = np.asarray(delta_colmajor, dtype=f)
    prev, which = (fb1, 2) if frame_count % 2 == 0 else (fb2, 1)          # T:46,61,67
    cur = screen_u8[..., :3].astype(f) / f(255.0)                          # T:35
    rep = cur.copy()
    ys, xs = np.mgrid[0:H, 0:W]
    with np.errstate(all="ignore"):
        nx = (xs.astype(f) + f(0.5)) / f(W) * f(2.0) - f(1.0)             # T:38-43
        ny = (ys.astype(f) + f(0.5)) / f(H) * f(-2.0) + f(1.0)
        if frame_count > 0:
            row = lambda r: ((m[r] * nx + m[4 + r] * ny) + m[8 + r] * depth) + m[12 + r] * f(1.0)   # T:50
            cw = row(3)
            cx, cy, cz = row(0) / cw, row(1) / cw, row(2) / cw             # T:51
            u = (cx + f(1.0)) * f(0.5)                                     # T:53-56
            v = (f(1.0) - cy) * f(0.5)

            def to_int(a):                                                 # T:57, pinned: NaN -> 0, saturating
                a = np.where(np.isnan(a), f(0.0), a)
                return np.clip(np.trunc(a.astype(np.float64)), -2.0 ** 31, 2.0 ** 31 - 1).astype(np.int64)
            px, py = to_int(u * f(W)), to_int(v * f(H))
            ok = (px >= 0) & (px < W) & (py >= 0) & (py < H)               # T:59
            pxc, pyc = np.clip(px, 0, W - 1), np.clip(py, 0, H - 1)
            ok &= np.abs(depth[pyc, pxc] - cz) < f(0.1)
            rep[ok] = prev[pyc, pxc][..., :3][ok]                          # T:60
    blended = cur * (f(1.0) - f(0.75)) + rep * f(0.75)                     # T:64 mix()
    hist = np.concatenate([blended, np.ones((H, W, 1), f)], axis=-1)       # T:66
    x = blended
    a, b, c, d, e = f(2.51), f(0.03), f(2.43), f(0.59), f(0.14)
    col = np.clip((x * (a * x + b)) / (x * (c * x + d) + e), f(0.0), f(1.0))  # T:19-27
    out = np.zeros((H, W, 4), np.uint8)
    out[..., :3] = np.floor(col * f(255.0) + f(0.5)).astype(np.uint8)
    out[..., 3] = 255
    return out, hist, which


def sample_texture(tex, u, v, layer, mode):
    """texture(sampler2DArray, vec3(u, v, layer)) after the Vulkan texel-addressing rules (the pin written down in
    oracle/oracle_trace.c), in float64: mode bit 0 = repeat (else clamp-to-edge), bit 1 = linear (else nearest).
    tex: [layers, res, res, 4] uint8, UNORM, no sRGB decode.  Returns rgb in [0, 1]."""
    res = tex.shape[1]
    repeat, linear = bool(mode & 1), bool(mode & 2)

    def wrap(i):
        if i != i:
            return 0
        if not repeat:                 # clamp-to-edge: saturate, then convert (so +-inf are the edge texels)
            return int(min(max(i, 0.0), float(res - 1)))
        return 0 if abs(i) >= 2.0 ** 30 else int(i) % res

    def texel(ix, iy):
        return tex[layer, iy, ix, :3].astype(np.float64) / 255.0

    x, y = float(np.float32(u) * np.float32(res)), float(np.float32(v) * np.float32(res))   # one float32 multiply, as pinned
    if not linear:
        return texel(wrap(np.floor(x)), wrap(np.floor(y)))
    xs, ys = x - 0.5, y - 0.5
    fx, fy = np.floor(xs), np.floor(ys)
    a, b = xs - fx, ys - fy
    a, b = (0.0 if a != a else a), (0.0 if b != b else b)      # inf - inf: weight 0 (pinned)
    x0, x1, y0, y1 = wrap(fx), wrap(fx + 1), wrap(fy), wrap(fy + 1)
    r0 = texel(x0, y0) * (1 - a) + texel(x1, y0) * a
    r1 = texel(x0, y1) * (1 - a) + texel(x1, y1) * a
    return r0 * (1 - b) + r1 * b
