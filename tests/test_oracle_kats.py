"""Known-answer tests that pin the C oracle piece by piece against an independent numpy restatement
(tests/np_restatement.py) and against analytic answers.  CPU only."""
import ctypes as C

import numpy as np
import pytest

import np_restatement as npr
from gdpathtracing_amd import scenes, wire


def _f3(*v):
    return (C.c_float * 3)(*v)


def test_struct_sizes_match_reference(oracle):
    # 144 / 48 / 32 / 176 / 48 / 80 / 64 / 160
    assert [d.itemsize for d in (wire.TRIANGLE, wire.BVH_NODE, wire.TLAS_NODE, wire.BLAS_INSTANCE, wire.TRI_GEOMETRY,
                                 wire.TRI_DATA, wire.MATERIAL, wire.CAMERA)] == [144, 48, 32, 176, 48, 80, 64, 160]


def test_prng_seed_and_pcg2d_bit_exact(oracle):
    L = oracle.lib()
    rng = np.random.RandomState(0)
    px = rng.randint(0, 4096, size=200).astype(np.uint32)
    py = rng.randint(0, 4096, size=200).astype(np.uint32)
    for frame in (0, 1, 7, 123456789, 0xFFFFFFFF):
        want = npr.prng_seed(px, py, frame)
        for i in range(len(px)):
            got = (C.c_uint32 * 2)()
            L.jpto_prng_seed(int(px[i]), int(py[i]), frame, got)
            assert (got[0], got[1]) == (int(want[i, 0]), int(want[i, 1]))
            s2, f2 = npr.pcg2d(want[i])
            out = (C.c_float * 2)()
            L.jpto_pcg2d(got, out)
            assert (got[0], got[1]) == (int(s2[0]), int(s2[1]))
            assert np.float32(out[0]) == f2[0] and np.float32(out[1]) == f2[1]


def test_pcg2d_known_vector_and_unit_interval(oracle):
    L = oracle.lib()
    s = (C.c_uint32 * 2)(0, 0)
    out = (C.c_float * 2)()
    L.jpto_pcg2d(s, out)
    # hand-derived from the published PCG2D recipe (jcgt 9(3) 2020) on seed (0,0)
    x = y = 1013904223
    x = (x + 1664525 * y) & 0xFFFFFFFF
    y = (y + 1664525 * x) & 0xFFFFFFFF
    x ^= x >> 16
    y ^= y >> 16
    x = (x + 1664525 * y) & 0xFFFFFFFF
    y = (y + 1664525 * x) & 0xFFFFFFFF
    x ^= x >> 16
    y ^= y >> 16
    assert (s[0], s[1]) == (x, y)
    # u32 -> f32 rounds to nearest, so the top of the range returns exactly 1.0
    assert np.float32(0xFFFFFFFF) * np.float32(2.32830643654e-10) == np.float32(1.0)


def test_sincos_accuracy_and_symmetry(oracle):
    L = oracle.lib()
    xs = np.concatenate([np.linspace(0, 2 * np.pi, 4001), [0.0, np.pi / 2, np.pi, 1.5 * np.pi, 6.2831855]]).astype(np.float32)
    s, c = C.c_float(), C.c_float()
    err = 0.0
    for x in xs:
        L.jpto_sincos(float(x), C.byref(s), C.byref(c))
        err = max(err, abs(s.value - np.sin(np.float64(x))), abs(c.value - np.cos(np.float64(x))))
        assert abs(s.value * s.value + c.value * c.value - 1.0) < 5e-7
    assert err < 2.5e-7  # ~2 ulp at 1.0
    L.jpto_sincos(0.0, C.byref(s), C.byref(c))
    assert (s.value, c.value) == (0.0, 1.0)


def test_intersect_aabb_against_numpy_and_analytic(oracle):
    L = oracle.lib()
    rng = np.random.RandomState(1)
    for _ in range(500):
        o = rng.uniform(-3, 3, 3).astype(np.float32)
        d = rng.normal(size=3).astype(np.float32)
        d /= np.linalg.norm(d)
        rD = (np.float32(1.0) / d).astype(np.float32)
        a, b = rng.uniform(-2, 2, 3), rng.uniform(-2, 2, 3)
        bmin, bmax = np.minimum(a, b).astype(np.float32), np.maximum(a, b).astype(np.float32)
        got = L.jpto_intersect_aabb(_f3(*o), _f3(*rD), _f3(*bmin), _f3(*bmax))
        assert np.float32(got) == npr.intersect_aabb(o, rD, bmin, bmax)
    # analytic: ray along +x from (-5,0,0) into the unit box enters at t = 4
    assert L.jpto_intersect_aabb(_f3(-5, 0, 0), _f3(1, np.inf, np.inf), _f3(-1, -1, -1), _f3(1, 1, 1)) == 4.0
    # origin inside: tmin negative but returned (main.glsl:267)
    assert L.jpto_intersect_aabb(_f3(0, 0, 0), _f3(1, np.inf, np.inf), _f3(-1, -1, -1), _f3(1, 1, 1)) == -1.0
    # box behind the ray -> miss sentinel
    assert L.jpto_intersect_aabb(_f3(5, 0, 0), _f3(1, np.inf, np.inf), _f3(-1, -1, -1), _f3(1, 1, 1)) == np.float32(1e30)
    # flat box with the ray in its plane: both y slabs give 0 * inf = NaN, minNum/maxNum ignore them
    assert L.jpto_intersect_aabb(_f3(-5, 1, 0), _f3(1, np.inf, np.inf), _f3(-1, 1, -1), _f3(1, 1, 1)) == 4.0
    # ray in the top face plane of a thick box: (-inf, NaN) -> tmax = -inf -> miss (pinned, not "robust")
    assert L.jpto_intersect_aabb(_f3(-5, 1, 0), _f3(1, np.inf, np.inf), _f3(-1, -1, -1), _f3(1, 1, 1)) == np.float32(1e30)


def test_intersect_triangle_table(oracle):
    L = oracle.lib()
    v0, v1, v2 = (0, 0, 0), (1, 0, 0), (0, 1, 0)
    tuv = (C.c_float * 3)()
    front = C.c_int()

    def call(o, d, tmax=1e9):
        return L.jpto_intersect_triangle(_f3(*o), _f3(*d), _f3(*v0), _f3(*v1), _f3(*v2), tmax, tuv, C.byref(front))

    assert call((0.25, 0.25, 1), (0, 0, -1)) == 1 and tuple(tuv) == (1.0, 0.25, 0.25)
    assert front.value == 0          # cross(e1,e2) = +z, d = -z: dot < 0 -> not "front" (main.glsl:254-255)
    assert call((0.25, 0.25, -1), (0, 0, 1)) == 1 and front.value == 1
    assert call((0.75, 0.75, 1), (0, 0, -1)) == 0     # u + v > 1
    assert call((-0.1, 0.2, 1), (0, 0, -1)) == 0      # u < 0
    assert call((0.25, 0.25, 1), (0, 0, 1)) == 0      # t < 0
    assert call((0.25, 0.25, 1), (1, 0, 0)) == 0      # parallel: |det| < 1e-5
    assert call((0.25, 0.25, 1), (0, 0, -1), tmax=0.5) == 0   # t > hitInfo.t
    assert call((0.25, 0.25, 1), (0, 0, -1), tmax=1.0) == 1   # t == hitInfo.t is accepted (main.glsl:247)
    rng = np.random.RandomState(2)
    n_hit = 0
    for _ in range(300):
        tri = rng.uniform(-1, 1, (3, 3)).astype(np.float32)
        o = rng.uniform(-2, 2, 3).astype(np.float32)
        tgt = tri.mean(axis=0) + rng.normal(scale=0.4, size=3)
        d = (tgt - o).astype(np.float32)
        d /= np.linalg.norm(d)
        got = L.jpto_intersect_triangle(_f3(*o), _f3(*d), _f3(*tri[0]), _f3(*tri[1]), _f3(*tri[2]), 1e9, tuv, C.byref(front))
        hit, t, u, v, fr = npr.intersect_triangle(o, d, tri[0], tri[1], tri[2], 1e9)
        if hit and min(u, v, 1 - u - v) > 1e-4 and t > 1e-4:
            assert got == 1
            assert np.allclose(tuple(tuv), (t, u, v), rtol=2e-4, atol=2e-5)
            assert bool(front.value) == fr
            n_hit += 1
    assert n_hit > 50


def test_primary_ray_matches_float64_math(oracle):
    L = oracle.lib()
    sc = scenes.cornell_scene()
    w, h = 320, 180
    cam = scenes.camera_block(sc.camera, w, h, frame_index=3)
    camc = np.ascontiguousarray(cam).reshape(1)
    o, d, seed = (C.c_float * 3)(), (C.c_float * 3)(), (C.c_uint32 * 2)()
    for (px, py) in [(0, 0), (319, 179), (160, 90), (17, 101), (300, 5)]:
        L.jpto_primary_ray(camc.ctypes.data_as(C.c_void_p), w, h, px, py, o, d, seed)
        want_d, want_seed = npr.primary_ray(cam["ivp"], cam["position"], w, h, px, py, 3)
        assert (seed[0], seed[1]) == (int(want_seed[0]), int(want_seed[1]))
        assert np.allclose(tuple(d), want_d, atol=3e-6)
        assert tuple(o) == tuple(np.float32(cam["position"][:3]))
    # the centre pixel looks down -z (camera basis = identity, demo.tscn:53)
    L.jpto_primary_ray(camc.ctypes.data_as(C.c_void_p), w, h, 160, 90, o, d, seed)
    assert d[2] < -0.99


def _shading(oracle, n, v, albedo, f0, rough):
    s = oracle.Shading()
    n = np.asarray(n, dtype=np.float64) / np.linalg.norm(n)
    v = np.asarray(v, dtype=np.float64) / np.linalg.norm(v)
    s.normal[:] = n
    s.out_dir[:] = v
    s.lambert_out = float(np.float32(n @ v))
    s.diffuse_albedo[:] = albedo
    s.fresnel_0[:] = f0
    s.roughness = rough
    return s, n, v


@pytest.mark.parametrize("rough,albedo,f0", [(0.6, (1, 1, 1), (0.02, 0.02, 0.02)), (0.16, (0, 0, 0), (1, 1, 1)),
                                              (1.0, (0.42, 1, 0.13), (0.02, 0.02, 0.02)), (0.006, (0.8, 0.8, 0.8), (0.02,) * 3)])
def test_brdf_pdf_and_sampling_match_float64(oracle, rough, albedo, f0):
    L = oracle.lib()
    rng = np.random.RandomState(3)
    for _ in range(100):
        n = rng.normal(size=3)
        v = n / np.linalg.norm(n) + 0.8 * rng.normal(size=3)
        if np.dot(n, v) <= 0.05:
            continue
        s, n, v = _shading(oracle, n, v, albedo, f0, rough)
        xi = rng.uniform(size=2).astype(np.float32)
        out = (C.c_float * 3)()
        L.jpto_sample_brdf(C.byref(s), (C.c_float * 2)(*xi), out)
        want_l = npr.sample_brdf(n, v, albedo, rough, xi)
        l = np.array(tuple(out), dtype=np.float64)
        tol = 2e-3 if rough < 0.01 else 5e-5
        assert np.allclose(l, want_l, atol=tol)
        assert abs(np.linalg.norm(l) - 1) < 1e-4
        pdf = L.jpto_brdf_density(C.byref(s), _f3(*l))
        f = (C.c_float * 3)()
        L.jpto_brdf(C.byref(s), _f3(*l), f)
        lf = np.float32(l)
        want_pdf = npr.brdf_density(n, v, s.lambert_out, albedo, rough, lf)
        want_f = npr.brdf(n, v, s.lambert_out, albedo, f0, rough, lf)
        if rough > 0.01:
            assert np.isclose(pdf, want_pdf, rtol=2e-3, atol=1e-6)
            assert np.allclose(tuple(f), want_f, rtol=2e-3, atol=1e-6)


def test_pdf_integrates_to_one(oracle):
    """The sampling density (brdfs.glsl:130-138) integrates to ~1 over the sphere (the VNDF lobe can
    leave the upper hemisphere, so the integral runs over all directions)."""
    L = oracle.lib()
    s, n, v = _shading(oracle, (0.2, 0.1, 1.0), (0.3, -0.2, 1.0), (0.5, 0.5, 0.5), (0.04, 0.04, 0.04), 0.5)
    rng = np.random.RandomState(4)
    d = rng.normal(size=(200000, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    tot = 0.0
    for l in d[:60000]:
        p = L.jpto_brdf_density(C.byref(s), _f3(*l))
        if np.isfinite(p) and p > 0:
            tot += p
    est = tot / 60000 * 4 * np.pi
    assert 0.93 < est < 1.07


def test_unorm8_and_aces(oracle):
    L = oracle.lib()
    assert [L.jpto_unorm8(x) for x in (-1.0, 0.0, 0.5, 1.0, 2.0, float("nan"), 0.999)] == [0, 0, 128, 255, 255, 0, 255]
    assert L.jpto_unorm8(127.49 / 255) == 127 and L.jpto_unorm8(127.51 / 255) == 128
    for x in ([0.0, 0.18, 1.0], [4.0, 0.5, 0.01], [100.0, 1e-6, 2.5]):
        out = (C.c_float * 3)()
        L.jpto_aces(_f3(*x), out)
        assert np.allclose(tuple(out), npr.aces(x), atol=2e-6)


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_sampler_modes_match_the_numpy_restatement(oracle, mode):
    """texture(textureArray, ...) in the four sampler modes (nearest / linear x clamp-to-edge / repeat, jpt.h JPT_SAMPLER_*):
    the oracle against an independent float64 restatement of the Vulkan addressing rules, inside and far outside
    [0, 1], plus the exact cases (texel centres reproduce the texel in every mode; repeat has period 1)."""
    L = oracle.lib()
    rng = np.random.RandomState(5)
    tex = rng.randint(0, 256, size=(2, 8, 8, 4)).astype(np.uint8)
    ref = oracle.RefLayoutScene(*(np.zeros(0, d) for d in (wire.TRIANGLE, wire.TRI_GEOMETRY, wire.TRI_DATA, wire.MATERIAL,
                                                            wire.BVH_NODE, wire.BLAS_INSTANCE, wire.TLAS_NODE)), textures=tex)
    view = ref.view(mode)

    def sample(u, v, layer):
        out = (C.c_float * 3)()
        L.jpto_sample_texture(C.byref(view), float(u), float(v), layer, out)
        return np.array(out[:], dtype=np.float64)

    for u, v in rng.uniform(-2.5, 3.5, size=(300, 2)):
        for layer in (0, 1):
            assert np.allclose(sample(u, v, layer), npr.sample_texture(tex, u, v, layer, mode), rtol=0, atol=2e-6), (u, v, mode)
    for ix in range(8):
        for iy in range(8):
            want = tex[1, iy, ix, :3].astype(np.float32) / np.float32(255)
            got = sample((ix + 0.5) / 8.0, (iy + 0.5) / 8.0, 1)
            assert np.array_equal(got.astype(np.float32), want), (ix, iy, mode)
    if mode & 1:   # repeat: whole texture periods are exact in float for these coordinates
        for u, v in ((0.3125, 0.6875), (0.03125, 0.96875)):
            assert np.array_equal(sample(u, v, 0), sample(u + 2.0, v - 1.0, 0))
    else:          # clamp-to-edge: everything beyond the edge texel centre is the edge texel
        assert np.array_equal(sample(-3.0, 0.5, 0), sample(0.01, 0.5, 0)) and np.array_equal(sample(7.0, 0.5, 0), sample(0.99, 0.5, 0))
        # ... however far: the clamp saturates before the integer conversion (Vulkan clamp-to-edge), so huge and infinite
        # coordinates address the last / first texel, not texel 0
        for big in (1e9, 3e38, float("inf")):
            assert np.array_equal(sample(big, 0.5, 0), sample(0.99, 0.5, 0)), big
            assert np.array_equal(sample(0.5, big, 1), sample(0.5, 0.99, 1)), big
            assert np.array_equal(sample(-big, 0.5, 0), sample(0.01, 0.5, 0)), big
            assert np.allclose(sample(big, 0.5, 0), npr.sample_texture(tex, big, 0.5, 0, mode), rtol=0, atol=2e-6)
    # NaN coordinates address texel 0 with weight 0 in every mode
    nan = float("nan")
    assert np.array_equal(sample(nan, nan, 0), tex[0, 0, 0, :3].astype(np.float32).astype(np.float64) / 255.0) or \
        np.allclose(sample(nan, nan, 0), tex[0, 0, 0, :3] / 255.0, rtol=0, atol=1e-7)


def test_unorm8_decode_without_division_is_exact():
    """csrc/jpt_device_math.h::from_unorm8 computes q / 255.0f as y = q * c, r = fma(-255, y, q), fma(r, c, y) with
    c = RN(1 / 255).  Emulated here in exact rational arithmetic with a correct round-to-nearest-even to binary32: equal to
    the IEEE quotient (the pinned semantics, oracle_pins.h) for every q in 0..255 -- the only inputs there are."""
    from fractions import Fraction

    def rn32(x):
        f = np.float32(float(x))
        cands = [np.nextafter(f, np.float32(-np.inf)), f, np.nextafter(f, np.float32(np.inf))]
        return np.float32(min(cands, key=lambda c: (abs(Fraction(float(c)) - x), int(np.float32(c).view(np.uint32)) & 1)))

    c = np.float32(1.0) / np.float32(255.0)
    for q in range(256):
        want = np.float32(q) / np.float32(255.0)
        assert want == rn32(Fraction(q, 255))
        y = rn32(Fraction(q) * Fraction(float(c)))
        r = rn32(Fraction(q) - 255 * Fraction(float(y)))
        assert rn32(Fraction(float(r)) * Fraction(float(c)) + Fraction(float(y))) == want, q


@pytest.mark.parametrize("frame", [1, 7])
def test_whole_path_numpy_float32_restatement_equals_the_oracle(oracle, frame):
    """tests/np_path.py: the whole path (jittered primary ray, closest hit, shading fetch, BRDF sample / pdf / value, 5
    segments) restated in vectorised float32 numpy WITHOUT any tree -- every triangle of every instance against every
    ray -- on the 36-triangle, 4-instance Cornell scene at 32x32.  The oracle's tree-independent mode
    (JPTO_FLAG_NO_CULL) must give the same float radiance and depth bit for bit; the oracle's normal walk of the
    reference tree must give that image too (no crack at this size)."""
    import np_path
    sc = scenes.cornell_scene()
    sc.camera = scenes.CameraDesc(scenes.transform12(None, (0.0, 0.0, 4.2)), fov_deg=75.0)   # at the box opening: every pixel traces
    w = h = 32
    cam = scenes.camera_block(sc.camera, w, h).copy()
    cam["frame_index"] = frame
    ref = oracle.build_scene(sc)
    rad, depth = np_path.trace_frame(ref, cam, w, h, 4)
    want, want_depth, cnt = oracle.trace_frame(ref, cam, w, h, 4, flags=1)
    walk, walk_depth, _ = oracle.trace_frame(ref, cam, w, h, 4)
    assert cnt["rays"] > 2.5 * w * h                      # paths really bounce
    bad = np.argwhere((rad != want[..., :3]).any(axis=-1))
    assert len(bad) == 0, "pixels %s differ: numpy %s oracle %s" % (bad[:4].tolist(), rad[tuple(bad[0])], want[tuple(bad[0])][:3])
    assert np.array_equal(depth, want_depth)
    assert np.array_equal(walk, want) and np.array_equal(walk_depth, want_depth)
