"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs."""
import numpy as np
import pytest

from gdpathtracing_amd import capi, host, scenes, wire

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a = a[..., :3].astype(np.float64)
    b = b[..., :3].astype(np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def _render_hip(sc, cam, w, h, bounces, frames, mode, builder=None, ref=None, first=1, kernel=capi.KERNEL_WAVEFRONT):
    ctx = host.Context(0)
    try:
        ctx.set_kernel(kernel)
        if ref is not None:
            ctx.upload_reference_layout(ref.tri_geom, ref.tri_data, ref.materials, ref.bvh_nodes, ref.instances,
                                        ref.tlas_nodes, ref.textures)
        else:
            ctx.build_scene(sc, builder)
        ctx.set_params(w, h, bounces, mode)
        ctx.set_camera(cam)
        ctx.render(frames, first)
        return ctx.read_accum(), ctx.read_ldr(), ctx.read_depth()
    finally:
        ctx.close()


KERNELS = [capi.KERNEL_WAVEFRONT, capi.KERNEL_REFERENCE_LAYOUT, capi.KERNEL_WAVEFRONT_V1]


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("mode", [wire.ACCUM_REF_LDR8, wire.ACCUM_HDR_F32])
def test_c1_cornell_reference_layout_bit_exact(oracle, hiplib, mode, kernel):
    """Config C1: Cornell box 256x256, 1 spp, 2 bounces; drop-in route (reference-layout upload)."""
    sc = scenes.cornell_scene()
    w = h = 256
    cam = scenes.camera_block(sc.camera, w, h)
    ref = oracle.build_scene(sc)
    want, want_ldr, want_depth, _, _ = oracle.render(ref, cam, w, h, 2, 1, 1, mode)
    got, got_ldr, got_depth = _render_hip(sc, cam, w, h, 2, 1, mode, ref=ref, kernel=kernel)
    assert rel_l2(got, want) <= 1e-4        # north-star tolerance
    assert np.array_equal(got, want)        # and in fact bit-identical
    assert np.array_equal(got_ldr, want_ldr)
    assert np.array_equal(got_depth, want_depth)


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("builder", [capi.BUILD_REFERENCE_EXACT, capi.BUILD_SAH])
def test_demo_scene_multi_frame(oracle, hiplib, builder, kernel):
    sc = scenes.demo_scene(5000)
    w, h = 192, 108
    cam = scenes.camera_block(sc.camera, w, h)
    ref = oracle.build_scene(sc)
    want, want_ldr, _, _, _ = oracle.render(ref, cam, w, h, 4, 4, 1, wire.ACCUM_REF_LDR8)
    got, got_ldr, _ = _render_hip(sc, cam, w, h, 4, 4, wire.ACCUM_REF_LDR8, builder=builder, kernel=kernel)
    ndiff = int((got != want).any(axis=-1).sum())
    print("builder", builder, "kernel", kernel, "differing pixels", ndiff, "rel_l2", rel_l2(got, want))
    assert rel_l2(got, want) <= 1e-4
    if builder == capi.BUILD_REFERENCE_EXACT:
        assert ndiff == 0  # same tree, same visit order: bit-identical
        assert np.array_equal(got_ldr, want_ldr)
