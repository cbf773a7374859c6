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


# ---- moving instances: TLAS update without a rebuild -------------------------------------

def _moved(scene, moves):
    import copy
    out = copy.deepcopy(scene)
    for i, t in moves.items():
        out.instances[i].transform = np.asarray(t, dtype=np.float32)
    return out


def _moves_for(sc, seed, n):
    rng = np.random.RandomState(seed)
    return {int(i): scenes.transform12(scenes.rot_y(rng.uniform(0, 360)) * rng.uniform(0.6, 1.4),
                                       np.asarray(sc.instances[int(i)].transform[9:12]) + rng.uniform(-1.0, 1.0, size=3))
            for i in rng.choice(np.arange(2, len(sc.instances)), size=n, replace=False)}


@pytest.mark.parametrize("kernel", [capi.KERNEL_WAVEFRONT, capi.KERNEL_REFERENCE_LAYOUT])
@pytest.mark.parametrize("builder", [capi.BUILD_REFERENCE_EXACT, capi.BUILD_SAH])
def test_tlas_update_renders_like_a_fresh_build(oracle, hiplib, builder, kernel):
    """An animation step: render, move instances, jpt_scene_update_tlas, render again.  The second image must equal
    the oracle's image of the moved scene (reference-exact builder: bit for bit) and a fresh context's image."""
    sc = scenes.instanced_scene(n_side=6, n_unique=3, tris_per_mesh=128)
    w, h, bounces, frames = 160, 96, 3, 2
    cam = scenes.camera_block(sc.camera, w, h)
    moves = _moves_for(sc, 21, 9)
    ctx = host.Context(0)
    try:
        ctx.set_kernel(kernel)
        ctx.build_scene(sc, builder)
        ctx.set_params(w, h, bounces, wire.ACCUM_REF_LDR8)
        ctx.set_camera(cam)
        ctx.render(frames, 1)
        before = ctx.read_accum()
        for i, t in moves.items():
            ctx.set_instance_transform(i, t)
        ctx.update_tlas()
        ctx.accum_reset()
        ctx.render(frames, 1)
        after, after_depth = ctx.read_accum(), ctx.read_depth()
    finally:
        ctx.close()
    fresh, _, fresh_depth = _render_hip(_moved(sc, moves), cam, w, h, bounces, frames, wire.ACCUM_REF_LDR8, builder=builder, kernel=kernel)
    assert np.array_equal(after, fresh) and np.array_equal(after_depth, fresh_depth)
    assert not np.array_equal(after, before)
    ref = oracle.build_scene(_moved(sc, moves))
    want, _, want_depth, _, _ = oracle.render(ref, cam, w, h, bounces, frames, 1, wire.ACCUM_REF_LDR8)
    assert rel_l2(after, want) <= 1e-4
    if builder == capi.BUILD_REFERENCE_EXACT:
        assert np.array_equal(after, want) and np.array_equal(after_depth, want_depth)


def test_reference_tlas_update_route(oracle, hiplib):
    """Route (i): the addon keeps its own builder and hands over new BLASInstance[] + TLASNode[] only."""
    sc = scenes.instanced_scene(n_side=5, n_unique=2, tris_per_mesh=96)
    w, h = 128, 80
    cam = scenes.camera_block(sc.camera, w, h)
    moves = _moves_for(sc, 4, 6)
    r0, r1 = oracle.build_scene(sc), oracle.build_scene(_moved(sc, moves))
    ctx = host.Context(0)
    try:
        ctx.upload_reference_layout(r0.tri_geom, r0.tri_data, r0.materials, r0.bvh_nodes, r0.instances, r0.tlas_nodes)
        ctx.set_params(w, h, 3, wire.ACCUM_REF_LDR8)
        ctx.set_camera(cam)
        ctx.render(1, 1)
        ctx.update_reference_tlas(r1.instances, r1.tlas_nodes)
        ctx.accum_reset()
        ctx.render(2, 5)
        got, got_ldr = ctx.read_accum(), ctx.read_ldr()
    finally:
        ctx.close()
    want, want_ldr, _, _, _ = oracle.render(r1, cam, w, h, 3, 2, 5, wire.ACCUM_REF_LDR8)
    assert np.array_equal(got, want) and np.array_equal(got_ldr, want_ldr)
