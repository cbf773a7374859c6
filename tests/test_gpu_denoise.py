"""GPU parity of the three post-processing modes of PathTracingCamera (path_tracing_camera.cpp:207-225) through the
C ABI: progressive rendering is covered by test_gpu_parity.py; here NONE (main.glsl's own rgba8 store) and
TEMPORAL_REPROJECTION (temporal_reprojection.glsl) against the oracle: trace_frame -> screen_rgba8 -> temporal_reproject."""
import copy

import numpy as np
import pytest

from gdpathtracing_amd import capi, host, scenes, wire

pytestmark = pytest.mark.gpu

KERNELS = [capi.KERNEL_WAVEFRONT, capi.KERNEL_REFERENCE_LAYOUT]


def _cam_with_index(sc, w, h, frame_index):
    return scenes.camera_block(sc.camera, w, h, frame_index=frame_index)


@pytest.mark.parametrize("kernel", KERNELS)
def test_none_mode_shows_the_frame_itself(oracle, hiplib, kernel):
    sc = scenes.cornell_scene()
    w, h = 96, 64
    ref = oracle.build_scene(sc)
    ctx = host.Context(0)
    try:
        ctx.set_kernel(kernel)
        ctx.upload_reference_layout(ref.tri_geom, ref.tri_data, ref.materials, ref.bvh_nodes, ref.instances, ref.tlas_nodes)
        ctx.set_params(w, h, 3, wire.ACCUM_REF_LDR8)
        ctx.set_denoising_mode(capi.DENOISE_NONE)
        ctx.set_camera(scenes.camera_block(sc.camera, w, h))
        for frame in (1, 2, 7):
            ctx.render(1, frame)
            rad, depth, _ = oracle.trace_frame(ref, _cam_with_index(sc, w, h, frame), w, h, 3)
            assert np.array_equal(ctx.read_ldr(), oracle.screen_rgba8(rad))
            assert np.array_equal(ctx.read_depth(), depth)
        # several frames in one call: the screen is the last one's
        ctx.render(3, 10)
        rad, _, _ = oracle.trace_frame(ref, _cam_with_index(sc, w, h, 12), w, h, 3)
        assert np.array_equal(ctx.read_ldr(), oracle.screen_rgba8(rad))
        # and back to progressive: ACES of the mean again, accumulation restarted
        ctx.set_denoising_mode(capi.DENOISE_PROGRESSIVE)
        ctx.render(2, 1)
        want, want_ldr, _, _, _ = oracle.render(ref, scenes.camera_block(sc.camera, w, h), w, h, 3, 2, 1, wire.ACCUM_REF_LDR8)
        assert np.array_equal(ctx.read_accum(), want) and np.array_equal(ctx.read_ldr(), want_ldr)
    finally:
        ctx.close()


@pytest.mark.parametrize("kernel", [capi.KERNEL_WAVEFRONT, capi.KERNEL_REFERENCE_LAYOUT])
@pytest.mark.parametrize("builder", [capi.BUILD_REFERENCE_EXACT])
def test_temporal_reprojection_sequence(oracle, hiplib, builder, kernel):
    """Six frames of PathTracingCamera::render in TEMPORAL_REPROJECTION mode with a camera that moves every frame:
    screen and history image equal the oracle's, bit for bit, every frame."""
    sc = scenes.demo_scene(800)
    w, h, bounces = 128, 72, 2
    ref = oracle.build_scene(sc)
    cam = host.PathTracingCamera(host.GeometryGroup3D(sc, builder), 0, max_bounces=bounces)
    cam.ctx.set_kernel(kernel)
    cam.denoising_mode = cam.TEMPORAL_REPROJECTION
    cam.init(w, h)
    fb = [np.zeros((h, w, 4), np.float32), np.zeros((h, w, 4), np.float32)]   # frameBuffer1, frameBuffer2
    o_temporal = host.TemporalReprojection(w, h)
    try:
        for frame in range(1, 7):
            if frame > 1:
                sc.camera.transform = sc.camera.transform.copy()
                sc.camera.transform[9:12] += np.array([0.05, 0.01 * frame, -0.03], np.float32)
            got_screen = cam.render()
            got_hist = cam.ctx.read_accum()
            # oracle: main.glsl, its rgba8 store, then the temporal dispatch with the same RenderParameters
            rad, depth, _ = oracle.trace_frame(ref, _cam_with_index(sc, w, h, frame), w, h, bounces)
            screen = oracle.screen_rgba8(rad)
            tp = o_temporal.render(scenes.view_projection(sc.camera, w, h))
            assert int(tp["frame_count"]) == frame + 1
            oracle.temporal_reproject(tp, screen, depth, fb[0], fb[1])
            written = fb[1] if (frame + 1) % 2 == 0 else fb[0]
            assert np.array_equal(got_screen, screen), "screen, frame %d" % frame
            assert np.array_equal(got_hist, written), "history, frame %d" % frame
        # history really is used: some pixel differs from the no-history blend
        assert np.abs(written[..., :3] - (screen[..., :3] / 255.0)).max() > 0
    finally:
        cam.ctx.close()


def test_temporal_mode_call_order_errors(hiplib):
    sc = scenes.cornell_scene()
    ctx = host.Context(0)
    try:
        ctx.build_scene(sc, capi.BUILD_SAH)
        ctx.set_params(32, 32, 1)
        ctx.set_camera(scenes.camera_block(sc.camera, 32, 32))
        with pytest.raises(capi.JptError, match="unknown denoising mode"):
            ctx.set_denoising_mode(5)
        ctx.set_denoising_mode(capi.DENOISE_TEMPORAL)
        with pytest.raises(capi.JptError, match="jpt_set_temporal_params"):
            ctx.render(1, 1)
        t = host.TemporalReprojection(16, 32)
        ctx.set_temporal_params(t.render(scenes.view_projection(sc.camera, 32, 32)))
        with pytest.raises(capi.JptError, match="width/height"):
            ctx.render(1, 1)
        t = host.TemporalReprojection(32, 32)
        ctx.set_temporal_params(t.render(scenes.view_projection(sc.camera, 32, 32)))
        with pytest.raises(capi.JptError, match="one frame per call"):
            ctx.render(2, 1)
        with pytest.raises(capi.JptError, match="no temporal pass"):
            ctx.read_accum()
        ctx.render(1, 1)
        assert ctx.read_accum().shape == (32, 32, 4)
        ctx.set_partition(0, 2)
        with pytest.raises(capi.JptError, match="whole image"):
            ctx.render(1, 2)
    finally:
        ctx.close()
