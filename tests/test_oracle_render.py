"""End-to-end checks of the oracle's tracer: analytic scenes, BVH-independence, committed fixtures.  CPU only."""
import os

import numpy as np
import pytest

import make_golden
import np_restatement as npr
from gdpathtracing_amd import scenes, wire

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _empty_scene():
    base = scenes.cornell_scene()
    return scenes.Scene("empty", [], [], base.materials, base.camera)


def test_empty_scene_is_exactly_the_sky(oracle):
    """No geometry: every pixel = sampleSky(d) of its jittered primary ray (main.glsl:189-192,366-367)."""
    sc = _empty_scene()
    w, h = 48, 27
    cam = scenes.camera_block(sc.camera, w, h)
    ref = oracle.build_scene(sc)
    rad, depth, cnt = oracle.trace_frame(ref, dict_to_cam(cam, 7), w, h, 4)
    assert cnt["rays"] == w * h and cnt["shaded_hits"] == 0
    ys, xs = np.mgrid[0:h, 0:w]
    d, _ = npr.primary_ray(cam["ivp"], cam["position"], w, h, xs.reshape(-1), ys.reshape(-1), 7)
    want = npr.sky(d).reshape(h, w, 3)
    assert np.allclose(rad[..., :3], want, atol=2e-6)
    assert (rad[..., 3] == 1).all()
    # depth on a miss: far/(far-near) * (1 - near/far) (main.glsl:373,432)
    far, near = np.float32(1000.0), np.float32(0.01)
    assert np.allclose(depth, far / (far - near) * (1 - near / far), rtol=1e-6)


def dict_to_cam(cam, frame_index):
    c = cam.copy()
    c["frame_index"] = frame_index
    return c


def test_black_emissive_quad_radiance_is_emission(oracle):
    """An emissive quad filling the view, max_bounces = 0 (one loop iteration): radiance = emission on
    every pixel (main.glsl:380), depth = reversed-Z of the first-hit distance."""
    quad = scenes.plane_mesh(40.0)
    mats = np.stack([scenes.material(), scenes.material(albedo=(0, 0, 0), emission=(0.2, 0.5, 0.7), energy=1.0)])
    # plane faces +y; rotate it to face the camera (+z): basis maps y -> z
    t = scenes.transform12([[1, 0, 0], [0, 0, -1], [0, 1, 0]], (0, 0, 0))
    sc = scenes.Scene("emit", [quad], [scenes.Instance(0, t, [1])], mats, scenes.cornell_scene().camera)
    w, h = 32, 18
    ref = oracle.build_scene(sc)
    rad, depth, cnt = oracle.trace_frame(ref, scenes.camera_block(sc.camera, w, h, 1), w, h, 0)
    assert cnt["shaded_hits"] == w * h and cnt["rays"] == w * h
    assert np.allclose(rad[..., :3], np.float32([0.2, 0.5, 0.7]), atol=1e-6)
    # first-hit distance -> reversed-Z depth (main.glsl:382-383,432)
    ys, xs = np.mgrid[0:h, 0:w]
    cam = scenes.camera_block(sc.camera, w, h, 1)
    d, _ = npr.primary_ray(cam["ivp"], cam["position"], w, h, xs.reshape(-1), ys.reshape(-1), 1)
    dist = (9.7694 / -d[:, 2]).reshape(h, w)
    want = 1000.0 / (1000.0 - 0.01) * (1 - 0.01 / dist)
    assert np.allclose(depth, want, atol=1e-6)


@pytest.mark.parametrize("name", ["cornell", "demo"])
def test_closest_hit_does_not_depend_on_bvh_culling(oracle, name):
    """NO_CULL makes every box test pass (all triangles of all instances are tested): the image must equal
    the normally culled traversal -> the reference BVH never hides an accepted triangle on these inputs."""
    sc = scenes.cornell_scene() if name == "cornell" else scenes.demo_scene(800)
    w, h = (64, 36) if name == "cornell" else (48, 27)
    cam = scenes.camera_block(sc.camera, w, h)
    ref = oracle.build_scene(sc)
    a, _, _, c0, _ = oracle.render(ref, cam, w, h, 3, 2, 1, wire.ACCUM_HDR_F32)
    b, _, _, c1, _ = oracle.render(ref, cam, w, h, 3, 2, 1, wire.ACCUM_HDR_F32, flags=1)
    assert c1["tri_tests"] > c0["tri_tests"]
    assert np.array_equal(a, b)


def test_threads_and_frame_split_do_not_change_the_result(oracle):
    sc = scenes.cornell_scene()
    w, h = 40, 40
    cam = scenes.camera_block(sc.camera, w, h)
    ref = oracle.build_scene(sc)
    a, la, da, _, _ = oracle.render(ref, cam, w, h, 2, 3, 1, wire.ACCUM_REF_LDR8, n_threads=1)
    b, lb, db, _, _ = oracle.render(ref, cam, w, h, 2, 3, 1, wire.ACCUM_REF_LDR8, n_threads=5)
    assert np.array_equal(a, b) and np.array_equal(la, lb) and np.array_equal(da, db)


def test_ldr8_accumulation_is_sum_of_quantised_frames(oracle):
    """REF_LDR8 = sum over frames of quantise8(clamp01(radiance)), alpha = 1, and the
    display image = unorm8(ACES(sum / n))."""
    sc = scenes.cornell_scene()
    w, h = 32, 32
    cam = scenes.camera_block(sc.camera, w, h)
    ref = oracle.build_scene(sc)
    n = 3
    accum, ldr, _, _, _ = oracle.render(ref, cam, w, h, 2, n, 4, wire.ACCUM_REF_LDR8)
    total = np.zeros((h, w, 3), dtype=np.float32)
    for f in range(n):
        rad, _, _ = oracle.trace_frame(ref, dict_to_cam(cam, 4 + f), w, h, 2)
        q = np.floor(np.clip(rad[..., :3], 0, 1) * np.float32(255) + np.float32(0.5)).astype(np.uint8)
        cur = q.astype(np.float32) / np.float32(255)
        total = cur if f == 0 else (cur + total)
    assert np.array_equal(accum[..., :3], total) and (accum[..., 3] == 1).all()
    want_ldr = np.floor(npr.aces(total.astype(np.float64) / n) * 255 + 0.5)
    assert np.abs(ldr[..., :3].astype(np.int32) - want_ldr).max() <= 1
    assert (ldr[..., 3] == 255).all()


def test_texture_lookup_nearest_clamp(oracle):
    quad = scenes.plane_mesh(40.0)
    tex = scenes.checker_texture(16, 4)
    mats = np.stack([scenes.material(), scenes.material(albedo=(1, 1, 1), emission=(0, 0, 0), texture=0)])
    t = scenes.transform12([[1, 0, 0], [0, 0, -1], [0, 1, 0]], (0, 0, 0))
    sc = scenes.Scene("tex", [quad], [scenes.Instance(0, t, [1])], mats, scenes.cornell_scene().camera, textures=tex)
    w, h = 24, 24
    ref = oracle.build_scene(sc)
    a, _, _, _, _ = oracle.render(ref, scenes.camera_block(sc.camera, w, h), w, h, 1, 1, 1, wire.ACCUM_HDR_F32)
    sc2 = scenes.Scene("notex", [quad], [scenes.Instance(0, t, [0])], mats, sc.camera)
    b, _, _, _, _ = oracle.render(oracle.build_scene(sc2), scenes.camera_block(sc.camera, w, h), w, h, 1, 1, 1, wire.ACCUM_HDR_F32)
    assert not np.array_equal(a, b) and np.isfinite(a).all()


@pytest.mark.parametrize("name", list(make_golden.CASES))
def test_committed_fixtures(oracle, name):
    """tests/golden/*.npz (self-generated, see tests/make_golden.py): the oracle still reproduces them."""
    want = np.load(os.path.join(GOLDEN, name + ".npz"))
    _, got = make_golden.render_case(name)
    for k in ("accum", "ldr", "depth"):
        assert np.array_equal(got[k], want[k]), k
    assert int(got["rays"]) == int(want["rays"]) and int(got["n_nodes"]) == int(want["n_nodes"])


def test_debug_steps_mode_counts_the_primary_rays_triangle_tests(oracle):
    """main.glsl built with DEBUG_STEPS (main.glsl:4,358-361,423-427): the frame is clamp(hitInfo.steps / 256) per pixel --
    steps = intersectTriangle calls of the PRIMARY ray --, one ray per pixel, depth = far.  The per-pixel counts add up to
    the frame's triangle-test counter, and a scene with nothing in it gives black (0 tests)."""
    sc = scenes.cornell_scene()
    w, h = 48, 32
    cam = scenes.camera_block(sc.camera, w, h)
    ref = oracle.build_scene(sc)
    rad, depth, cnt = oracle.trace_frame(ref, cam, w, h, 4, flags=4)
    steps = rad[..., 0] * np.float32(256)
    assert np.array_equal(steps, np.round(steps)) and np.array_equal(rad[..., 0], rad[..., 1]) and np.array_equal(rad[..., 0], rad[..., 2])
    assert steps.max() < 256 and int(steps.sum()) == cnt["tri_tests"] > 0
    assert cnt["rays"] == w * h and cnt["shaded_hits"] == 0                 # one ray per pixel, nothing shaded
    far, near = np.float32(cam["far"]), np.float32(cam["near"])
    assert np.all(depth == far / (far - near) * (np.float32(1) - near / far))   # depth stays camera.far
    import copy
    empty = copy.deepcopy(sc)
    empty.instances = []
    rad0, _, cnt0 = oracle.trace_frame(oracle.build_scene(empty), cam, w, h, 4, flags=4)
    assert not rad0[..., :3].any() and cnt0["tri_tests"] == 0
