"""The C++ host layer (include/jpt_host.hpp: GeometryGroup3D / ProgressiveRendering / PathTracingCamera with
the reference's names and call order) driven by tests/cpp/host_demo.cpp, checked against the oracle."""
import os
import subprocess

import numpy as np
import pytest

from gdpathtracing_amd import capi, host, scenes, wire

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host_demo(hiplib, tmp_path_factory):
    d = tmp_path_factory.mktemp("cpp")
    exe = str(d / "host_demo")
    libdir = os.path.dirname(capi.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "host_demo.cpp"), "-o", exe, "-L", libdir, "-ljpt_hip",
                           "-Wl,-rpath," + libdir])
    return exe, str(d)


def test_cpp_geometry_group_emits_the_reference_buffers(oracle, host_demo):
    exe, d = host_demo
    sc = scenes.demo_scene(3000)
    path = os.path.join(d, "s.jpts")
    scenes.write_scene_file(sc, path)
    out = subprocess.run([exe, "buffers", path, os.path.join(d, "b")], capture_output=True, text=True, check=True).stdout
    assert "tris %d blas 4" % sc.n_unique_tris in out
    ref = oracle.build_scene(sc)
    for k, want in enumerate((ref.tri_geom, ref.tri_data, None, ref.bvh_nodes, ref.instances, ref.tlas_nodes)):
        got = open(os.path.join(d, "b_buf%d.bin" % k), "rb").read()
        if want is not None:
            assert got == want.tobytes(), "get_*_buffer %d" % k
    # materials: only the default + the override materials in use, in first-use order (geometry_group3d.cpp:186-202)
    mats = np.frombuffer(open(os.path.join(d, "b_buf2.bin"), "rb").read(), dtype=wire.MATERIAL)
    used = [0] + sorted({m for i in sc.instances for m in i.material_ids if m > 0})
    assert len(mats) == len(used)
    for got, idx in zip(mats, used):
        assert got.tobytes() == sc.materials[idx].tobytes()
    # Camera::set_camera_transform in float agrees with the float64 helper to float precision
    cam = np.frombuffer(open(os.path.join(d, "b_camera.bin"), "rb").read(), dtype=wire.CAMERA)[0]
    want = scenes.camera_block(sc.camera, 64, 36)
    assert np.allclose(cam["vp"], want["vp"], rtol=1e-5, atol=1e-6) and np.allclose(cam["ivp"], want["ivp"], rtol=2e-4, atol=1e-4)  # near/far = 1e-5: ill-conditioned
    assert tuple(cam["position"]) == tuple(want["position"]) and (cam["near"], cam["far"]) == (want["near"], want["far"])


def test_cpp_update_transforms_equals_a_rebuild(oracle, host_demo):
    """GeometryGroup3D::update_transforms (moving nodes, no rebuild): instance records and TLAS equal the oracle
    builder's arrays for the moved scene."""
    import copy
    exe, d = host_demo
    sc = scenes.instanced_scene(n_side=4, n_unique=2, tris_per_mesh=64)
    path = os.path.join(d, "m.jpts")
    scenes.write_scene_file(sc, path)
    out = subprocess.run([exe, "moved", path, os.path.join(d, "m")], capture_output=True, text=True, check=True).stdout
    assert "moved %d then 0, bvh unchanged 1" % (len(sc.instances) - 1) in out
    moved = copy.deepcopy(sc)
    for i in range(1, len(moved.instances)):
        t = moved.instances[i].transform.astype(np.float32).copy()
        t[9] = np.float32(t[9]) + np.float32(0.25) * np.float32(i)
        t[11] = np.float32(t[11]) - np.float32(0.125) * np.float32(i)
        moved.instances[i].transform = t
    ref = oracle.build_scene(moved)
    inst = np.frombuffer(open(os.path.join(d, "m_buf4.bin"), "rb").read(), dtype=wire.BLAS_INSTANCE)
    tlas = np.frombuffer(open(os.path.join(d, "m_buf5.bin"), "rb").read(), dtype=wire.TLAS_NODE)
    for f in ("transform", "inverse_transform", "aabbMin", "aabbMax", "blas_index"):
        assert np.array_equal(inst[f], ref.instances[f]), f
    for f in ("aabbMin", "aabbMax", "leftRight"):
        assert np.array_equal(tlas[f], ref.tlas_nodes[f]), f


@pytest.mark.gpu
@pytest.mark.parametrize("builder,mode", [(capi.BUILD_REFERENCE_EXACT, wire.ACCUM_REF_LDR8), (capi.BUILD_SAH, wire.ACCUM_HDR_F32)])
def test_cpp_path_tracing_camera_renders_like_the_oracle(oracle, host_demo, builder, mode):
    """PathTracingCamera::render() called once per frame (frame_index pre-incremented, accumulation restarted by
    the first frame's camera-moved test): same image as the oracle fed the same 160-byte camera block."""
    exe, d = host_demo
    sc = scenes.cornell_scene()
    path = os.path.join(d, "c.jpts")
    scenes.write_scene_file(sc, path)
    w, h, frames = 96, 64, 3
    out = subprocess.run([exe, "render", path, os.path.join(d, "r"), str(w), str(h), str(frames), str(builder), str(mode)],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "frame_index %d, frame_count %d" % (frames, frames) in out.stdout
    cam = np.frombuffer(open(os.path.join(d, "r_camera.bin"), "rb").read(), dtype=wire.CAMERA)[0]
    got = np.frombuffer(open(os.path.join(d, "r_accum.bin"), "rb").read(), dtype=np.float32).reshape(h, w, 4)
    got_ldr = np.frombuffer(open(os.path.join(d, "r_ldr.bin"), "rb").read(), dtype=np.uint8).reshape(h, w, 4)
    # the C++ layer's material table holds the materials in use; same ids for this scene
    ref = oracle.build_scene(sc)
    want, want_ldr, _, _, _ = oracle.render(ref, cam, w, h, 4, frames, 1, mode)
    assert np.array_equal(got, want)
    assert np.array_equal(got_ldr, want_ldr)


@pytest.mark.gpu
def test_cpp_camera_overlapped_frames(oracle, host_demo):
    """PathTracingCamera::render_overlapped(): frames queued with jpt_render_async + split read-back, image k-1
    returned by call k; after flush() the accumulation and the last image equal the stalling loop's."""
    exe, d = host_demo
    sc = scenes.cornell_scene()
    path = os.path.join(d, "c.jpts")
    scenes.write_scene_file(sc, path)
    w, h, frames = 96, 64, 5
    out = subprocess.run([exe, "render", path, os.path.join(d, "ov"), str(w), str(h), str(frames), str(capi.BUILD_REFERENCE_EXACT),
                          str(wire.ACCUM_REF_LDR8), "0", "1"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    cam = np.frombuffer(open(os.path.join(d, "ov_camera.bin"), "rb").read(), dtype=wire.CAMERA)[0]
    got = np.frombuffer(open(os.path.join(d, "ov_accum.bin"), "rb").read(), dtype=np.float32).reshape(h, w, 4)
    got_ldr = np.frombuffer(open(os.path.join(d, "ov_ldr.bin"), "rb").read(), dtype=np.uint8).reshape(h, w, 4)
    ref = oracle.build_scene(sc)
    want, want_ldr, _, _, _ = oracle.render(ref, cam, w, h, 4, frames, 1, wire.ACCUM_REF_LDR8)
    assert np.array_equal(got, want) and np.array_equal(got_ldr, want_ldr)


def _check_other_modes(oracle, pre, sc, w, h, frames, denoise):
    """Replays the dumped per-frame Camera blocks (and temporal RenderParameters) through the oracle."""
    ref = oracle.build_scene(sc)
    fb = [np.zeros((h, w, 4), np.float32), np.zeros((h, w, 4), np.float32)]
    screen = written = None
    for f in range(frames):
        cam = np.frombuffer(open("%s_camera_f%d.bin" % (pre, f), "rb").read(), dtype=wire.CAMERA)[0]
        assert int(cam["frame_index"]) == f + 1
        rad, depth, _ = oracle.trace_frame(ref, cam, w, h, 4)
        screen = oracle.screen_rgba8(rad)
        if denoise == 1:
            tp = np.frombuffer(open("%s_tp_f%d.bin" % (pre, f), "rb").read(), dtype=wire.TEMPORAL_PARAMS)[0]
            assert int(tp["frame_count"]) == f + 2 and (int(tp["width"]), int(tp["height"])) == (w, h)
            assert tuple(tp["deltaMatrix"][[3, 7, 11, 15]]) == (0.0, 0.0, 0.0, 1.0)   # the Projection -> Transform3D truncation
            oracle.temporal_reproject(tp, screen, depth, fb[0], fb[1])
            written = fb[1] if int(tp["frame_count"]) % 2 == 0 else fb[0]
    got_ldr = np.frombuffer(open(pre + "_ldr.bin", "rb").read(), dtype=np.uint8).reshape(h, w, 4)
    assert np.array_equal(got_ldr, screen)
    if denoise == 1:
        got = np.frombuffer(open(pre + "_accum.bin", "rb").read(), dtype=np.float32).reshape(h, w, 4)
        assert np.array_equal(got, written)
        # the delta matrix of the last step agrees with the float64 helper (it is an input, not a parity claim)
        import copy
        c0, c1 = copy.deepcopy(sc.camera), copy.deepcopy(sc.camera)
        c0.transform = sc.camera.transform.copy(); c1.transform = sc.camera.transform.copy()
        c0.transform[9] += np.float32(0.05) * 2; c0.transform[10] += np.float32(0.01) * 3
        c1.transform[9] += np.float32(0.05) * 3; c1.transform[10] += np.float32(0.01) * 6
        want = scenes.temporal_delta(scenes.view_projection(c0, w, h), scenes.view_projection(c1, w, h))
        assert np.allclose(tp["deltaMatrix"], want, rtol=5e-3, atol=5e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("denoise", [1, 2])
def test_cpp_camera_other_denoising_modes(oracle, host_demo, denoise):
    """PathTracingCamera::render() with denoising_mode TEMPORAL_REPROJECTION (1) / NONE (2) and a camera that moves
    every frame: the C++ layer's Camera blocks and temporal RenderParameters, fed to the oracle, give the same images."""
    exe, d = host_demo
    sc = scenes.cornell_scene()
    path = os.path.join(d, "c.jpts")
    scenes.write_scene_file(sc, path)
    w, h, frames = 80, 48, 4
    pre = os.path.join(d, "dn%d" % denoise)
    out = subprocess.run([exe, "render", path, pre, str(w), str(h), str(frames), str(capi.BUILD_REFERENCE_EXACT),
                          str(wire.ACCUM_REF_LDR8), str(denoise)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    _check_other_modes(oracle, pre, sc, w, h, frames, denoise)


@pytest.fixture(scope="module")
def gdcs_test(hiplib, tmp_path_factory):
    d = tmp_path_factory.mktemp("gdcs")
    exe = str(d / "gdcs_adapter_test")
    libdir = os.path.dirname(capi.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "gdcs_adapter_test.cpp"), "-o", exe, "-L", libdir, "-ljpt_hip",
                           "-Wl,-rpath," + libdir])
    return exe, str(d)


def test_gdcs_adapter_compiles(gdcs_test):
    """The ComputeShader-shaped adapter (include/jpt_gdcs_adapter.hpp) and its driver build on the CPU box."""
    assert os.path.exists(gdcs_test[0])


@pytest.mark.gpu
def test_gdcs_shaped_adapter_replays_the_reference_call_sequence(oracle, gdcs_test):
    """The eleven ComputeShader methods the reference calls, in the reference's order, through the
    adapter: three frames of render(); image identical to the oracle's for the same camera block."""
    exe, d = gdcs_test
    sc = scenes.cornell_scene()
    path = os.path.join(d, "c.jpts")
    scenes.write_scene_file(sc, path)
    w, h, frames = 80, 48, 3
    out = subprocess.run([exe, path, os.path.join(d, "g"), str(w), str(h), str(frames)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "frame_count %d" % frames in out.stdout
    assert "tree %d" % capi.TREE_NATIVE_REACH in out.stdout     # the adapter's upload is walked on the native tree
    cam = np.frombuffer(open(os.path.join(d, "g_camera.bin"), "rb").read(), dtype=wire.CAMERA)[0]
    got = np.frombuffer(open(os.path.join(d, "g_accum.bin"), "rb").read(), dtype=np.float32).reshape(h, w, 4)
    got_ldr = np.frombuffer(open(os.path.join(d, "g_ldr.bin"), "rb").read(), dtype=np.uint8).reshape(h, w, 4)
    ref = oracle.build_scene(sc)
    want, want_ldr, _, _, _ = oracle.render(ref, cam, w, h, 4, frames, 1, wire.ACCUM_REF_LDR8)
    assert np.array_equal(got, want) and np.array_equal(got_ldr, want_ldr)


@pytest.mark.gpu
def test_gdcs_adapter_when_the_first_camera_transform_is_the_identity(oracle, gdcs_test):
    """ProgressiveRendering's previous_transform starts as the identity (progressive_rendering.h:44), so a camera node left
    at the origin never "moves": the reference's first frame already has frame_count = 2 (progressive_rendering.cpp:53-60),
    the shader adds the zero-filled frameBuffer and divides by 2 (progressive_rendering.glsl:33-39), and every later
    frame divides by one more than the frames summed.  The adapter hands the library the frame_count the caller computed."""
    exe, d = gdcs_test
    sc = scenes.cornell_scene()
    path = os.path.join(d, "c.jpts")
    scenes.write_scene_file(sc, path)
    w, h, frames = 80, 48, 3
    out = subprocess.run([exe, path, os.path.join(d, "gi"), str(w), str(h), str(frames), "0", "1"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "frame_count %d" % (frames + 1) in out.stdout
    cam = np.frombuffer(open(os.path.join(d, "gi_camera.bin"), "rb").read(), dtype=wire.CAMERA)[0]
    got = np.frombuffer(open(os.path.join(d, "gi_accum.bin"), "rb").read(), dtype=np.float32).reshape(h, w, 4)
    got_ldr = np.frombuffer(open(os.path.join(d, "gi_ldr.bin"), "rb").read(), dtype=np.uint8).reshape(h, w, 4)
    ref = oracle.build_scene(sc)
    want, _, _, _, _ = oracle.render(ref, cam, w, h, 4, frames, 1, wire.ACCUM_REF_LDR8)
    assert np.array_equal(got, want)          # the sums are those of `frames` frames (0 + f1 + f2 + f3) ...
    # ... and the screen is ACES(sum / (frames + 1)), through the oracle's own ACES and rgba8 conversion
    import ctypes as C
    L = oracle.lib()
    L.jpto_aces.argtypes = [C.c_void_p, C.c_void_p]
    L.jpto_aces.restype = None
    L.jpto_unorm8.argtypes = [C.c_float]
    L.jpto_unorm8.restype = C.c_uint8
    avg = (want[..., :3] / np.float32(frames + 1)).astype(np.float32)
    want_ldr = np.zeros((h, w, 4), np.uint8)
    want_ldr[..., 3] = 255
    col = np.zeros(3, np.float32)
    for y in range(h):
        for x in range(w):
            px = np.ascontiguousarray(avg[y, x])
            L.jpto_aces(px.ctypes.data_as(C.c_void_p), col.ctypes.data_as(C.c_void_p))
            for k in range(3):
                want_ldr[y, x, k] = L.jpto_unorm8(float(col[k]))
    assert np.array_equal(got_ldr, want_ldr)


@pytest.mark.gpu
def test_gdcs_adapter_debug_steps_define(oracle, gdcs_test):
    """`new ComputeShader("...main.glsl", rd, {"#define DEBUG_STEPS"})`: the shader's own debug build through the adapter --
    the uploaded arrays are then walked node for node, so the counts are the reference tree's: equal to the oracle's
    DEBUG_STEPS render (progressive mode, three frames) bit for bit."""
    exe, d = gdcs_test
    sc = scenes.cornell_scene()
    path = os.path.join(d, "c.jpts")
    scenes.write_scene_file(sc, path)
    w, h, frames = 80, 48, 3
    out = subprocess.run([exe, path, os.path.join(d, "gs"), str(w), str(h), str(frames), "0", "0", "1"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "tree %d" % capi.TREE_AS_GIVEN in out.stdout
    cam = np.frombuffer(open(os.path.join(d, "gs_camera.bin"), "rb").read(), dtype=wire.CAMERA)[0]
    got = np.frombuffer(open(os.path.join(d, "gs_accum.bin"), "rb").read(), dtype=np.float32).reshape(h, w, 4)
    got_ldr = np.frombuffer(open(os.path.join(d, "gs_ldr.bin"), "rb").read(), dtype=np.uint8).reshape(h, w, 4)
    ref = oracle.build_scene(sc)
    want, want_ldr, _, _, _ = oracle.render(ref, cam, w, h, 4, frames, 1, wire.ACCUM_REF_LDR8, flags=4)
    assert np.array_equal(got, want) and np.array_equal(got_ldr, want_ldr)
    assert got[..., :3].max() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("denoise", [1, 2])
def test_gdcs_adapter_other_denoising_modes(oracle, gdcs_test, denoise):
    """The adapter with a third ComputeShader on temporal_reprojection.glsl (TemporalReprojection::init/render call
    for call), and with no post-processing shader at all (denoising_mode NONE: the raw rgba8 frame)."""
    exe, d = gdcs_test
    sc = scenes.cornell_scene()
    path = os.path.join(d, "c.jpts")
    scenes.write_scene_file(sc, path)
    w, h, frames = 80, 48, 4
    pre = os.path.join(d, "gd%d" % denoise)
    out = subprocess.run([exe, path, pre, str(w), str(h), str(frames), str(denoise)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    _check_other_modes(oracle, pre, sc, w, h, frames, denoise)


@pytest.mark.gpu
@pytest.mark.parametrize("refit", [0, 1])
def test_cpp_moving_nodes_on_the_gpu(host_demo, refit):
    """GeometryGroup3D::update_transforms between PathTracingCamera::render() calls -- host rebuild (0) or device
    refit (1): after three animation steps the screen equals a fresh context's render of the moved scene."""
    import copy
    exe, d = host_demo
    sc = scenes.instanced_scene(n_side=4, n_unique=2, tris_per_mesh=64)
    path = os.path.join(d, "a.jpts")
    scenes.write_scene_file(sc, path)
    w, h, steps = 160, 90, 3
    pre = os.path.join(d, "a%d" % refit)
    out = subprocess.run([exe, "animate", path, pre, str(w), str(h), str(steps), str(refit)], capture_output=True, text=True, check=True).stdout
    assert "animated %d steps, %d moves, frame_index %d" % (steps, steps * (len(sc.instances) - 1), steps) in out
    got = np.frombuffer(open(pre + "_ldr.bin", "rb").read(), dtype=np.uint8).reshape(h, w, 4)
    moved = copy.deepcopy(sc)
    for i in range(1, len(moved.instances)):
        t = moved.instances[i].transform.astype(np.float32).copy()
        for _ in range(steps):
            t[9] = np.float32(t[9]) + np.float32(0.25) * np.float32(i)
            t[11] = np.float32(t[11]) - np.float32(0.125) * np.float32(i)
        moved.instances[i].transform = t
    ctx = host.Context(0)
    ctx.build_scene(moved, capi.BUILD_SAH)
    ctx.set_params(w, h, 4, wire.ACCUM_REF_LDR8)
    ctx.set_camera(np.frombuffer(open(pre + "_camera.bin", "rb").read(), dtype=wire.CAMERA)[0])   # the C++ layer's float camera block
    ctx.set_denoising_mode(capi.DENOISE_NONE)
    ctx.render(1, steps)
    want = ctx.read_ldr()
    ctx.close()
    assert np.array_equal(got, want)


@pytest.mark.gpu
def test_cpp_multi_device_camera_renders_like_the_oracle(oracle, host_demo):
    """PathTracingCameraMulti (include/jpt_host.hpp over jpt_multi_*): three devices' worth of contexts in one process, the
    scene built once and shared, rows gathered to rank 0 peer-to-peer -- the image equals the oracle's bit for bit."""
    exe, d = host_demo
    sc = scenes.cornell_scene()
    path = os.path.join(d, "c.jpts")
    scenes.write_scene_file(sc, path)
    w, h, frames = 96, 70, 3
    out = subprocess.run([exe, "multi", path, os.path.join(d, "mu"), str(w), str(h), str(frames), str(capi.BUILD_SAH),
                          str(wire.ACCUM_REF_LDR8), "3"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "on 3 devices, frame_index %d, frame_count %d" % (frames, frames) in out.stdout
    cam = np.frombuffer(open(os.path.join(d, "mu_camera.bin"), "rb").read(), dtype=wire.CAMERA)[0]
    got = np.frombuffer(open(os.path.join(d, "mu_accum.bin"), "rb").read(), dtype=np.float32).reshape(h, w, 4)
    got_ldr = np.frombuffer(open(os.path.join(d, "mu_ldr.bin"), "rb").read(), dtype=np.uint8).reshape(h, w, 4)
    want, want_ldr, _, _, _ = oracle.render(oracle.build_scene(sc), cam, w, h, 4, frames, 1, wire.ACCUM_REF_LDR8)
    assert np.array_equal(got, want) and np.array_equal(got_ldr, want_ldr)


# ---- scene ingest without Godot: include/jpt_host.hpp load_obj / load_mtl / resize_rgba8 -----------------------------

_OBJ = """# an open cube of three surfaces plus a quad without normals or uvs (public Cornell-box numbers)
mtllib box.mtl
v 5 5 -5
v 5 -5 -5
v 5 5 5
v 5 -5 5
v -5 5 -5
v -5 -5 -5
v -5 5 5
v -5 -5 5
v -2 -4.99 -2
v 2 -4.99 -2
v 2 -4.99 2
v -2 -4.99 2
vn 0 -1 0
vn 0 0 -1
vn 1 0 0
vn 0 1 0
vn 0 0 1
vt 0.625 0.5
vt 0.375 0.5
vt 0.125 0.25
vt 0.875 0.75
usemtl white
f 1/1/1 3/2/1 7/3/1 5/4/1
f 8/2/3 6/3/3 5/4/3 7/1/3
f 6/1/4 8/2/4 4/3/4 2/4/4
usemtl green
f 6/1/5 2/2/5 1/3/5 5/4/5
usemtl red
f 4/1/2 8/2/2 7/3/2 3/4/2
usemtl lamp
f 9 12 11 10
f -4 -1 -2
"""
_MTL = """newmtl white
Ns 250.0
Kd 0.8 0.8 0.8
map_Kd checker.ppm
newmtl green
Pr 0.35
Ns 10
Kd 0.03 0.77 0.06
Pm 0.25
newmtl red
Ns 2
Kd 0.8 0.064 0.019
newmtl lamp
Kd 1 1 1
Ke 6.0 4.5 3.0
"""


def _write_obj_inputs(d):
    open(os.path.join(d, "box.obj"), "w").write(_OBJ)
    open(os.path.join(d, "box.mtl"), "w").write(_MTL)
    tex = scenes.checker_texture(32, 4)[0]                       # 32 x 32, resized to the array's 64 x 64 by the loader
    with open(os.path.join(d, "checker.ppm"), "wb") as f:
        f.write(b"P6\n32 32\n255\n" + tex[..., :3].tobytes())
    return tex


def _python_side_of_the_obj_scene(tex):
    from gdpathtracing_amd import objio
    mesh = objio.load_obj(_OBJ)
    recs, maps = objio.load_mtl(_MTL)
    assert maps == ["checker.ppm"] and len(mesh.surfaces) == 4
    mats = np.stack([scenes.material(albedo=(0.5, 0.5, 0.5), roughness=0.5)] + [recs[k] for k in ("white", "green", "red", "lamp")])
    layers = objio.pack_texture_array([tex], 64)
    sc = scenes.Scene("obj", [mesh], [scenes.Instance(0, scenes.transform12(None, (0, 0, 0)), [1, 2, 3, 4])], mats,
                      scenes.CameraDesc(scenes.transform12(None, (0, 0, 9.7694)), fov_deg=79.5), textures=layers)
    return sc, mats, layers


def test_cpp_obj_and_mtl_ingest_equals_the_python_mirror(oracle, host_demo):
    """load_obj + load_mtl + resize_rgba8 of include/jpt_host.hpp (product code) fed through GeometryGroup3D::build on a
    host-only context: the triangle buffers, the GpuMaterial table and the texture layer equal what the Python mirror
    (objio.py) and the oracle's builder make of the same text -- surfaces per usemtl, face normals where the file has
    none, negative indices, Ke above 1 split into colour and energy, roughness from Pr or Ns, map_Kd resized."""
    exe, d = host_demo
    tex = _write_obj_inputs(d)
    out = subprocess.run([exe, "obj", os.path.join(d, "box.obj"), os.path.join(d, "o"), "64", "36", "1", os.path.join(d, "box.mtl"), "host"],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "obj: 4 surfaces, 4 materials in the library, 1 texture layers" in out.stdout
    sc, mats, layers = _python_side_of_the_obj_scene(tex)
    ref = oracle.build_scene(sc)
    # a mesh of four surfaces: the fourth surface's material slot does not exist in the 176-byte record (SURVEY A-6 / A-24)
    assert open(os.path.join(d, "o_buf0.bin"), "rb").read() == ref.tri_geom.tobytes()
    assert open(os.path.join(d, "o_buf1.bin"), "rb").read() == ref.tri_data.tobytes()
    got_mats = np.frombuffer(open(os.path.join(d, "o_buf2.bin"), "rb").read(), dtype=wire.MATERIAL)
    assert got_mats.tobytes() == mats.tobytes()
    assert float(got_mats[4]["emission"][3]) == 6.0 and np.allclose(got_mats[4]["emission"][:3], [1.0, 0.75, 0.5])
    assert np.isclose(got_mats[1]["roughness"], np.sqrt(2.0 / 252.0)) and got_mats[2]["roughness"] == np.float32(0.35)
    assert open(os.path.join(d, "o_tex0.bin"), "rb").read() == layers[0].tobytes()


@pytest.mark.gpu
def test_cpp_obj_scene_renders_like_the_oracle(oracle, host_demo):
    """The same files rendered by host_demo without Python in the loop (OBJ + MTL + PPM -> GeometryGroup3D ->
    PathTracingCamera::render x 3): accumulation buffer and screen equal the oracle's render of the Python-side scene."""
    exe, d = host_demo
    tex = _write_obj_inputs(d)
    w, h, frames = 96, 54, 3
    out = subprocess.run([exe, "obj", os.path.join(d, "box.obj"), os.path.join(d, "og"), str(w), str(h), str(frames), os.path.join(d, "box.mtl")],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    sc, _, _ = _python_side_of_the_obj_scene(tex)
    cam = np.frombuffer(open(os.path.join(d, "og_camera.bin"), "rb").read(), dtype=wire.CAMERA)[0]
    got = np.frombuffer(open(os.path.join(d, "og_accum.bin"), "rb").read(), dtype=np.float32).reshape(h, w, 4)
    got_ldr = np.frombuffer(open(os.path.join(d, "og_ldr.bin"), "rb").read(), dtype=np.uint8).reshape(h, w, 4)
    want, want_ldr, _, _, _ = oracle.render(oracle.build_scene(sc), cam, w, h, 4, frames, 1, wire.ACCUM_REF_LDR8)
    assert np.array_equal(got, want) and np.array_equal(got_ldr, want_ldr)
