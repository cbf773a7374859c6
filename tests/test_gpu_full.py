"""GPU parity at scale: committed fixtures, the BASELINE configs at reduced size against the oracle, exact
event-counter equality, and size-independent properties at the full C3 size (1920x1080, 8 spp, 4 bounces).
Everything goes through the C ABI (host.Context)."""
import os

import numpy as np
import pytest

import make_golden
from gdpathtracing_amd import capi, host, partition, scenes, wire

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KERNELS = [capi.KERNEL_WAVEFRONT, capi.KERNEL_REFERENCE_LAYOUT]


def rel_l2(a, b):
    a = a[..., :3].astype(np.float64)
    b = b[..., :3].astype(np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def upload_ctx(ref, cam, w, h, bounces, mode, as_given=False):
    """route (i): the reference-layout arrays (here: the oracle's builder's, i.e. what GeometryGroup3D::build emits)"""
    ctx = host.Context(0)
    ctx.upload_reference_layout(ref.tri_geom, ref.tri_data, ref.materials, ref.bvh_nodes, ref.instances, ref.tlas_nodes,
                                ref.textures, as_given=as_given)
    assert ctx.tree_kind() == (capi.TREE_AS_GIVEN if as_given else capi.TREE_NATIVE_REACH), ctx.upload_note()
    ctx.set_params(w, h, bounces, mode)
    ctx.set_camera(cam)
    return ctx


def make_ctx(sc, w, h, bounces, mode, builder=capi.BUILD_SAH, kernel=capi.KERNEL_WAVEFRONT, rank=0, world=1):
    ctx = host.Context(0)
    ctx.set_kernel(kernel)
    ctx.build_scene(sc, builder)
    ctx.set_partition(rank, world)
    ctx.set_params(w, h, bounces, mode)
    ctx.set_camera(scenes.camera_block(sc.camera, w, h))
    return ctx


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("builder", [capi.BUILD_REFERENCE_EXACT, capi.BUILD_SAH])
@pytest.mark.parametrize("name", list(make_golden.CASES))
def test_committed_fixtures(hiplib, name, builder, kernel):
    mk, w, h, b, f, first, mode = make_golden.CASES[name]
    want = np.load(os.path.join(GOLDEN, name + ".npz"))
    ctx = make_ctx(mk(), w, h, b, mode, builder, kernel)
    ctx.render(f, first)
    got, ldr, depth = ctx.read_accum(), ctx.read_ldr(), ctx.read_depth()
    assert rel_l2(got, want["accum"]) <= 1e-4           # north-star tolerance on the accumulated buffer
    if builder == capi.BUILD_REFERENCE_EXACT:           # same tree, same order -> bit-identical
        assert np.array_equal(got, want["accum"]) and np.array_equal(ldr, want["ldr"]) and np.array_equal(depth, want["depth"])
    assert ctx.stats()["rays"] == int(want["rays"]) or kernel == capi.KERNEL_REFERENCE_LAYOUT
    ctx.close()


@pytest.mark.parametrize("kernel", KERNELS)
def test_event_counters_equal_the_oracles(oracle, hiplib, kernel):
    """On the reference-exact tree the kernels expand the same nodes and test the same triangles as
    main.glsl's traversal restated by the oracle: all six counters match exactly."""
    sc = scenes.demo_scene(4096)
    w, h = 128, 72
    ref = oracle.build_scene(sc)
    _, _, _, want, _ = oracle.render(ref, scenes.camera_block(sc.camera, w, h), w, h, 4, 2, 1, wire.ACCUM_REF_LDR8)
    ctx = make_ctx(sc, w, h, 4, wire.ACCUM_REF_LDR8, capi.BUILD_REFERENCE_EXACT, kernel)
    ctx.render(2, 1, counted=True)
    st = ctx.stats()
    for k in ("rays", "blas_expand", "tri_tests", "tlas_expand", "inst_visits", "shaded_hits"):
        assert st[k] == want[k], k
    ctx.close()


@pytest.mark.parametrize("mode", [wire.ACCUM_REF_LDR8, wire.ACCUM_HDR_F32])
def test_c2_demo_scene_reduced(oracle, hiplib, mode):
    """Config C2's scene (51 200-tri character mesh x2 + Cornell cube + light), 3 bounces, 4 spp, at 320x180."""
    sc = scenes.demo_scene(51200)
    w, h = 320, 180
    ref = oracle.build_scene(sc)
    want, want_ldr, want_depth, cnt, _ = oracle.render(ref, scenes.camera_block(sc.camera, w, h), w, h, 3, 4, 1, mode)
    ctx = make_ctx(sc, w, h, 3, mode)
    ctx.render(4, 1)
    got = ctx.read_accum()
    ndiff = int((got != want).any(axis=-1).sum())
    print("C2 mode", mode, "differing pixels", ndiff, "rel_l2", rel_l2(got, want))
    assert rel_l2(got, want) <= 1e-4
    assert ctx.stats()["rays"] == cnt["rays"]
    assert np.array_equal(ctx.read_depth(), want_depth)
    ctx.close()


def test_c4_instanced_scene_reduced(oracle, hiplib):
    """Config C4: 1024 instances of 8 meshes x 1024 tris (1.05 M instanced triangles, TLAS of 2048 nodes)."""
    sc = scenes.instanced_scene(32, 8, 1024)
    assert len(sc.instances) == 1026 and sc.n_instanced_tris > 1_000_000
    w, h = 240, 135
    ref = oracle.build_scene(sc)
    want, _, _, cnt, _ = oracle.render(ref, scenes.camera_block(sc.camera, w, h), w, h, 4, 2, 1, wire.ACCUM_REF_LDR8)
    assert cnt["stack_overflow"] == 0
    for builder in (capi.BUILD_SAH, capi.BUILD_REFERENCE_EXACT):
        ctx = make_ctx(sc, w, h, 4, wire.ACCUM_REF_LDR8, builder)
        ctx.render(2, 1)
        got = ctx.read_accum()
        print("C4 builder", builder, "differing pixels", int((got != want).any(axis=-1).sum()), "rel_l2", rel_l2(got, want))
        assert rel_l2(got, want) <= 1e-4
        ctx.close()


@pytest.mark.parametrize("sampler", [capi.SAMPLER_NEAREST_CLAMP, capi.SAMPLER_NEAREST_REPEAT, capi.SAMPLER_LINEAR_CLAMP, capi.SAMPLER_LINEAR_REPEAT])
@pytest.mark.parametrize("uv_scale", [1.0, 1e9])
def test_texture_scene(oracle, hiplib, sampler, uv_scale):
    """A textured quad whose uv run from -1.3 to 2.6 (so clamp and repeat differ) with a random two-layer texture array,
    in the four sampler modes: every kernel and builder equals the oracle bit for bit.  uv_scale 1e9: coordinates far past
    2^30 texels (clamp-to-edge saturates to the edge texel before the integer conversion; repeat addresses texel 0)."""
    quad = scenes.plane_mesh(40.0)
    quad.surfaces[0].uvs = ((quad.surfaces[0].uvs * np.float32(3.9) - np.float32(1.3)) * np.float32(uv_scale)).astype(np.float32)
    mats = np.stack([scenes.material(), scenes.material(albedo=(1, 1, 1), texture=1)])
    t = scenes.transform12([[1, 0, 0], [0, 0, -1], [0, 1, 0]], (0, 0, 0))
    tex = np.concatenate([scenes.checker_texture(16, 4), np.random.RandomState(3).randint(0, 256, size=(1, 16, 16, 4)).astype(np.uint8)])
    sc = scenes.Scene("tex", [quad], [scenes.Instance(0, t, [1])], mats, scenes.cornell_scene().camera, textures=tex)
    w, h = 64, 36
    want, _, _, _, _ = oracle.render(oracle.build_scene(sc), scenes.camera_block(sc.camera, w, h), w, h, 2, 2, 1, wire.ACCUM_HDR_F32,
                                     sampler_mode=sampler)
    base, _, _, _, _ = oracle.render(oracle.build_scene(sc), scenes.camera_block(sc.camera, w, h), w, h, 2, 2, 1, wire.ACCUM_HDR_F32)
    assert sampler == 0 or uv_scale != 1.0 or not np.array_equal(want, base)   # the modes really differ on this scene
    for kernel in KERNELS:
        for builder in (capi.BUILD_REFERENCE_EXACT, capi.BUILD_SAH):
            ctx = host.Context(0)
            ctx.set_kernel(kernel)
            ctx.build_scene(sc, builder)
            ctx.set_params(w, h, 2, wire.ACCUM_HDR_F32, sampler)
            ctx.set_camera(scenes.camera_block(sc.camera, w, h))
            ctx.render(2, 1)
            assert np.array_equal(ctx.read_accum(), want), (kernel, builder)
            ctx.close()


def test_empty_scene_and_call_order_errors(hiplib):
    base = scenes.cornell_scene()
    sc = scenes.Scene("empty", [], [], base.materials, base.camera)
    ctx = host.Context(0)
    with pytest.raises(capi.JptError, match="no scene"):
        ctx.render(1, 1)
    ctx.build_scene(sc, capi.BUILD_SAH)
    with pytest.raises(capi.JptError, match="jpt_set_params"):
        ctx.render(1, 1)
    ctx.set_params(32, 16, 4, wire.ACCUM_HDR_F32)
    with pytest.raises(capi.JptError, match="jpt_set_camera"):
        ctx.render(1, 1)
    ctx.set_camera(scenes.camera_block(sc.camera, 32, 16))
    ctx.render(1, 1)
    a = ctx.read_accum()
    assert ctx.stats()["rays"] == 32 * 16 and (a[..., :3] > 0.89).all() and (a[..., :3] <= 1.0).all()
    ctx.render(0, 1)   # zero frames is a no-op
    assert np.array_equal(ctx.read_accum(), a)
    ctx.close()


# ---- full-size properties (C3: 1920x1080, 8 spp, 4 bounces, 51 200-tri demo scene) -----------------

W, H, SPP, B = 1920, 1080, 8, 4


@pytest.fixture(scope="module")
def c3():
    sc = scenes.demo_scene(51200)
    ctx = make_ctx(sc, W, H, B, wire.ACCUM_REF_LDR8)
    ctx.render(SPP, 1)
    out = dict(sc=sc, accum=ctx.read_accum(), ldr=ctx.read_ldr(), depth=ctx.read_depth(), rays=ctx.stats()["rays"])
    ctx.close()
    return out


def test_c3_is_deterministic_and_frame_splittable(hiplib, c3):
    ctx = make_ctx(c3["sc"], W, H, B, wire.ACCUM_REF_LDR8)
    ctx.render(3, 1)
    ctx.render(5, 4)              # continues the accumulation: frames 4..8
    assert np.array_equal(ctx.read_accum(), c3["accum"])
    assert np.array_equal(ctx.read_ldr(), c3["ldr"])
    ctx.accum_reset()
    ctx.render(SPP, 1)
    assert np.array_equal(ctx.read_accum(), c3["accum"])
    ctx.close()


def test_c3_full_size_equals_the_oracle(oracle, hiplib):
    """BASELINE config C3 at its full size, reference-exact tree, against the oracle itself (all host cores, a few
    seconds): 2 073 600 pixels x 8 frames bit for bit -- accumulation buffer, display image, depth, ray count."""
    sc = scenes.demo_scene(51200)
    ref = oracle.build_scene(sc)
    cam = scenes.camera_block(sc.camera, W, H)
    want, want_ldr, want_depth, cnt, _ = oracle.render(ref, cam, W, H, B, SPP, 1, wire.ACCUM_REF_LDR8)
    ctx = make_ctx(sc, W, H, B, wire.ACCUM_REF_LDR8, capi.BUILD_REFERENCE_EXACT)
    ctx.render(SPP, 1)
    assert np.array_equal(ctx.read_accum(), want)
    assert np.array_equal(ctx.read_ldr(), want_ldr) and np.array_equal(ctx.read_depth(), want_depth)
    assert ctx.stats()["rays"] == cnt["rays"]
    ctx.close()
    # The drop-in route at the same size: the reference-layout arrays uploaded as the addon would
    # (path_tracing_camera.cpp:170-175), walked on the native tree the upload builds, reach records from the uploaded
    # leaf / TLAS-leaf boxes: the same image, bit for bit.
    up = upload_ctx(ref, cam, W, H, B, wire.ACCUM_REF_LDR8)
    up.render(SPP, 1)
    got = up.read_accum()
    print("C3 via reference-layout upload: differing pixels", int((got != want).any(axis=-1).sum()), "rel_l2", rel_l2(got, want),
          "render ms", up.stats()["last_render_ms"])
    assert np.array_equal(got, want)
    assert np.array_equal(up.read_ldr(), want_ldr) and np.array_equal(up.read_depth(), want_depth)
    assert up.stats()["rays"] == cnt["rays"]
    up.close()
    # ... and the BENCHMARKED configuration itself, named here against the oracle directly (it was only
    # checked GPU against GPU and by bench.py's parity object): jpt_scene_commit(JPT_BUILD_SAH), blocking and queued renders
    for asynchronous in (False, True):
        sah = make_ctx(sc, W, H, B, wire.ACCUM_REF_LDR8, capi.BUILD_SAH)
        assert sah.tree_kind() == capi.TREE_NATIVE_REACH and sah.ties_exact() == (True, "")
        if asynchronous:
            for _ in range(3):                       # the timed region's shape: resets and queued renders back to back
                sah.accum_reset()
                sah.render(SPP, 1, asynchronous=True)
            sah.sync()
        else:
            sah.render(SPP, 1)
        got = sah.read_accum()
        assert np.array_equal(got, want), int((got != want).any(axis=-1).sum())
        assert np.array_equal(sah.read_ldr(), want_ldr) and np.array_equal(sah.read_depth(), want_depth)
        sah.close()


@pytest.mark.parametrize("config", ["C2", "C4"])
def test_other_baseline_configs_at_full_size(oracle, hiplib, config):
    """BASELINE configs C2 (demo scene, 1280x720, 4 spp, 3 bounces) and C4 (1 024 instances x 1 024 triangles,
    1920x1080, 8 spp, 4 bounces) at full size, reference-exact tree, bit for bit against the oracle."""
    if config == "C2":
        sc, w, h, spp, b = scenes.demo_scene(51200), 1280, 720, 4, 3
    else:
        sc, w, h, spp, b = scenes.instanced_scene(), 1920, 1080, 8, 4
    ref = oracle.build_scene(sc)
    cam = scenes.camera_block(sc.camera, w, h)
    want, want_ldr, want_depth, cnt, _ = oracle.render(ref, cam, w, h, b, spp, 1, wire.ACCUM_REF_LDR8)
    ctx = make_ctx(sc, w, h, b, wire.ACCUM_REF_LDR8, capi.BUILD_REFERENCE_EXACT)
    ctx.render(spp, 1)
    assert np.array_equal(ctx.read_accum(), want)
    assert np.array_equal(ctx.read_ldr(), want_ldr) and np.array_equal(ctx.read_depth(), want_depth)
    assert ctx.stats()["rays"] == cnt["rays"]
    # The native tree (JPT_BUILD_SAH: with reach records) on the same full-size input.  North-star tolerance: relative L2
    # <= 1e-4 against the reference tree's image -- asserted as is.  The reach records make the native route answer
    # like the reference even where the REFERENCE tree lets a ray through a crack of its own boxes (DESIGN.md section
    # 8; C2 has one such ray, at pixel (688, 551)), so the two images are expected to be identical.
    fast = make_ctx(sc, w, h, b, wire.ACCUM_REF_LDR8, capi.BUILD_SAH)
    fast.render(spp, 1)
    got = fast.read_accum()
    ys, xs = np.nonzero((got != want).any(axis=-1))
    err = rel_l2(got, want)
    print(config, "native tree + reach records: differing pixels", len(ys), "rel_l2", err, "set aside", fast.stats()["set_aside"])
    assert err <= 1e-4
    # Identical, C4 (1 024 instances, 25.9 M rays) included: beside the cracks its image depends on the ORDER of the
    # reference's walk -- exact distance ties between different triangles -- and those few dozen vertices are decided on the
    # reference's own trees (jpt_tie_walk.h; one pixel differed before)
    assert len(ys) == 0, "pixels %s differ from the reference tree's image" % list(zip(xs.tolist(), ys.tolist()))[:8]
    assert fast.stats()["rays"] == cnt["rays"] or config != "C2"
    fast.close()
    # The same through the drop-in route (reference-layout upload -> native tree + reach records from the uploaded boxes)
    up = upload_ctx(ref, cam, w, h, b, wire.ACCUM_REF_LDR8)
    up.render(spp, 1)
    got = up.read_accum()
    ys, xs = np.nonzero((got != want).any(axis=-1))
    print(config, "reference-layout upload on the native tree: differing pixels", len(ys), "rel_l2", rel_l2(got, want))
    assert len(ys) == 0, "pixels %s differ from the reference tree's image" % list(zip(xs.tolist(), ys.tolist()))[:8]
    up.close()
    # The native builder ALONE (JPT_BUILD_SAH_WATERTIGHT) does not reproduce the cracks: it differs from the reference
    # in those pixels and only there, and its value is the tree-independent answer -- the oracle with every box
    # test passing (JPTO_FLAG_NO_CULL, all triangles tested), run on that pixel's row.  One such pixel weighs about 1e-4
    # at these sizes (C2: relative L2 1.08e-4 > 1e-4), which is why this mode is not the default.
    fast = make_ctx(sc, w, h, b, wire.ACCUM_REF_LDR8, capi.BUILD_SAH_WATERTIGHT)
    fast.render(spp, 1)
    got = fast.read_accum()
    ys, xs = np.nonzero((got != want).any(axis=-1))
    err = rel_l2(got, want)
    print(config, "native tree, watertight: differing pixels", list(zip(xs.tolist(), ys.tolist())), "rel_l2", err)
    assert len(ys) <= 2      # (this mode is not held to the 1e-4 bar: it is characterised pixel by pixel below)
    if len(ys) and sc.n_instanced_tris <= 200_000:
        for y, x in zip(ys.tolist(), xs.tolist()):
            acc = np.zeros(3, np.float32)
            for f in range(spp):
                c = cam.copy()
                c["frame_index"] = 1 + f
                rad, _, _ = oracle.trace_frame(ref, c, w, h, b, flags=1, rows=(y, y + 1))
                q = oracle.screen_rgba8(rad[y:y + 1, x:x + 1])[0, 0, :3].astype(np.float32) / np.float32(255)
                acc = q if f == 0 else acc + q
            assert np.array_equal(acc, got[y, x, :3]), "pixel (%d, %d): the native tree is not the tree-independent answer" % (y, x)
    else:
        assert err <= 1e-4
    ctx.close(); fast.close()


def test_c3_both_kernels_agree(hiplib, c3):
    ctx = make_ctx(c3["sc"], W, H, B, wire.ACCUM_REF_LDR8, kernel=capi.KERNEL_REFERENCE_LAYOUT)
    ctx.render(SPP, 1)
    assert np.array_equal(ctx.read_accum(), c3["accum"]) and np.array_equal(ctx.read_depth(), c3["depth"])
    ctx.close()


def test_c3_native_tree_vs_reference_tree(hiplib, c3):
    """Full size, both trees on the GPU.  The reference-exact tree is the oracle's tree (the kernels visit it
    node for node, see test_event_counters_equal_the_oracles), so this is the full-size parity check of the
    fast route (the benchmarked one: JPT_BUILD_SAH, native tree + reach records): the accumulated buffers are equal bit
    for bit, including the pixel (678, 827) where the reference tree lets the ray of frame 4 through a crack of its own
    boxes -- and the native builder alone (JPT_BUILD_SAH_WATERTIGHT) differs in exactly that pixel."""
    ctx = make_ctx(c3["sc"], W, H, B, wire.ACCUM_REF_LDR8, builder=capi.BUILD_REFERENCE_EXACT)
    ctx.render(SPP, 1)
    want = ctx.read_accum()
    rays_ref = ctx.stats()["rays"]
    ctx.close()
    ndiff = int((want != c3["accum"]).any(axis=-1).sum())
    err = rel_l2(c3["accum"], want)
    print("C3 full size: differing pixels", ndiff, "rel_l2", err, "rays", c3["rays"], "vs", rays_ref)
    assert err <= 1e-4
    assert ndiff == 0 and c3["rays"] == rays_ref
    ctx = make_ctx(c3["sc"], W, H, B, wire.ACCUM_REF_LDR8, builder=capi.BUILD_SAH_WATERTIGHT)
    ctx.render(SPP, 1)
    tight = ctx.read_accum()
    ctx.close()
    ys, xs = np.nonzero((tight != want).any(axis=-1))
    print("C3 full size, watertight builder: differing pixels", list(zip(xs.tolist(), ys.tolist())), "rel_l2", rel_l2(tight, want))
    assert list(zip(xs.tolist(), ys.tolist())) == [(678, 827)]


def test_c3_sky_rows_are_analytic(oracle, hiplib, c3):
    """Rows that see only sky: accum = sum over frames of quantise8(sampleSky(d)) -- checked with the oracle
    on a band of full-width rows (cheap: every path is one ray)."""
    import np_restatement as npr
    band = slice(0, 16)
    cam = scenes.camera_block(c3["sc"].camera, W, H)
    ys, xs = np.mgrid[0:16, 0:W]
    total = np.zeros((16, W, 3), dtype=np.float32)
    for f in range(1, SPP + 1):
        d, _ = npr.primary_ray(cam["ivp"], cam["position"], W, H, xs.reshape(-1), ys.reshape(-1), f)
        s = npr.sky(d).reshape(16, W, 3)
        q = np.floor(np.clip(s, 0, 1) * 255 + 0.5)
        total = total + (q / 255).astype(np.float32)
    got = c3["accum"][band, :, :3]
    # float64 sky vs float32 kernel can flip a quantisation step in a few frames of rare pixels
    assert np.abs(got - total).max() <= 3.01 / 255
    assert (np.abs(got - total) > 1e-6).mean() < 0.02


def test_c3_screen_partition_is_bit_identical(hiplib, c3):
    """Two contexts render the strips of rank 0 and rank 1 of 2; the gathered pieces assembled on the
    device reproduce the single-context image exactly."""
    import torch
    pieces, ldr_pieces = [], []
    ctxs = []
    for r in range(2):
        ctx = make_ctx(c3["sc"], W, H, B, wire.ACCUM_REF_LDR8, rank=r, world=2)
        ctx.render(SPP, 1)
        assert ctx.local_rows() == len(partition.rows_of_rank(H, r, 2))
        ptr, nbytes = ctx.device_accum()

        class V:
            __cuda_array_interface__ = {"shape": (nbytes // 4,), "typestr": "<f4", "data": (ptr, False), "version": 2}
        pieces.append(torch.as_tensor(V(), device="cuda:0").clone())
        lptr, lbytes = ctx.device_ldr()

        class L:
            __cuda_array_interface__ = {"shape": (lbytes // 4,), "typestr": "<i4", "data": (lptr, False), "version": 2}
        ldr_pieces.append(torch.as_tensor(L(), device="cuda:0").clone())
        # the host-side read of a partial context returns its rows in place
        part = ctx.read_accum()
        rows = partition.rows_of_rank(H, r, 2)
        assert np.array_equal(part[rows], c3["accum"][rows])
        ctxs.append(ctx)
    # the display rows alone (4 bytes per pixel): every rank's rgba8 rows are final
    gathered_ldr = torch.stack(ldr_pieces).contiguous()
    torch.cuda.synchronize()
    ctxs[0].assemble_ldr_from_ranks(gathered_ldr.data_ptr(), 2)
    assert np.array_equal(ctxs[0].read_ldr(), c3["ldr"])
    ctxs[0]._ck(ctxs[0]._lib.jpt_readback_ldr_begin(ctxs[0].h), "jpt_readback_ldr_begin")
    split = np.zeros((H, W, 4), np.uint8)
    ctxs[0]._ck(ctxs[0]._lib.jpt_readback_ldr_end(ctxs[0].h, split.ctypes.data), "jpt_readback_ldr_end")
    assert np.array_equal(split, c3["ldr"])
    gathered = torch.stack(pieces).contiguous()
    torch.cuda.synchronize()
    ctxs[0].assemble_from_ranks(gathered.data_ptr(), 2)
    assert np.array_equal(ctxs[0].read_accum(), c3["accum"])
    assert np.array_equal(ctxs[0].read_ldr(), c3["ldr"])
    # and the host mirror of the assembly agrees
    host_img = partition.assemble(gathered.cpu().numpy().reshape(2, -1, W, 4), H, 2)
    assert np.array_equal(host_img, c3["accum"])
    for c in ctxs:
        c.close()


def test_c3_ldr8_bounds_and_hdr_mode(hiplib, c3):
    a = c3["accum"]
    assert (a[..., 3] == 1).all() and (a[..., :3] >= 0).all() and (a[..., :3] <= SPP + 1e-4).all()
    assert c3["rays"] >= W * H * SPP and c3["rays"] <= W * H * SPP * (B + 1)
    ctx = make_ctx(c3["sc"], W, H, B, wire.ACCUM_HDR_F32)
    ctx.render(SPP, 1)
    hdr = ctx.read_accum()
    assert ctx.stats()["rays"] == c3["rays"]          # same paths, only the accumulation differs
    assert np.isfinite(hdr).all() and (hdr[..., :3] >= 0).all()
    dark = hdr[..., :3].max(axis=-1) <= 0.99 * 1.0    # pixels that never clip agree to quantisation error
    assert np.abs(hdr[..., :3] - a[..., :3])[dark].max() <= SPP * 0.51 / 255 + 1e-5
    ctx.close()


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("w,h", [(1, 1), (7, 5), (33, 17), (250, 131), (64, 9)])
def test_ragged_resolutions(oracle, hiplib, w, h, kernel):
    """Widths/heights that are not multiples of the 8x8 tile (padding lanes) or of the 8-row strip."""
    sc = scenes.cornell_scene()
    ref = oracle.build_scene(sc)
    want, want_ldr, want_depth, cnt, _ = oracle.render(ref, scenes.camera_block(sc.camera, w, h), w, h, 3, 2, 1, wire.ACCUM_REF_LDR8)
    ctx = make_ctx(sc, w, h, 3, wire.ACCUM_REF_LDR8, capi.BUILD_REFERENCE_EXACT, kernel)
    ctx.render(2, 1)
    assert np.array_equal(ctx.read_accum(), want)
    assert np.array_equal(ctx.read_ldr(), want_ldr)
    assert np.array_equal(ctx.read_depth(), want_depth)
    if kernel != capi.KERNEL_REFERENCE_LAYOUT:
        assert ctx.stats()["rays"] == cnt["rays"]
    ctx.close()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_ragged_partition_matches_single_context(hiplib, world):
    """Strip partition with a height that leaves ranks unequal (and one partial strip)."""
    sc = scenes.cornell_scene()
    w, h = 96, 45
    full = make_ctx(sc, w, h, 2, wire.ACCUM_REF_LDR8)
    full.render(2, 1)
    want = full.read_accum()
    full.close()
    got = np.zeros_like(want)
    rows_seen = 0
    for r in range(world):
        ctx = make_ctx(sc, w, h, 2, wire.ACCUM_REF_LDR8, rank=r, world=world)
        ctx.render(2, 1)
        rows = partition.rows_of_rank(h, r, world)
        assert ctx.local_rows() == len(rows)
        part = ctx.read_accum()
        got[rows] = part[rows]
        rows_seen += len(rows)
        ctx.close()
    assert rows_seen == h and np.array_equal(got, want)


@pytest.mark.parametrize("view", ["far", "narrow", "behind"])
def test_render_window_under_a_partition(hiplib, view):
    """The render window (tile-aligned bounding rectangle of the sky cull's screen rectangles) under a 5-way strip
    partition: with the scene far away, in a narrow lens or behind the camera the window holds a few tile rows, so some
    ranks own none of them (their window is one culled tile) -- rows, depth and ray counts still add up to the whole
    image's."""
    from tests.test_gpu_parity import _look_at
    sc = scenes.demo_scene(1500)
    sc.camera = {"far": scenes.CameraDesc(scenes.transform12(None, (0.3, 0.2, 70.0)), fov_deg=30.0),
                 "narrow": scenes.CameraDesc(_look_at((8.0, 1.0, 8.0), (2.9, 2.0, 0.0)), fov_deg=8.0),
                 "behind": scenes.CameraDesc(_look_at((0.0, 0.0, 9.0), (0.0, 0.0, 20.0)), fov_deg=79.5)}[view]
    w, h, world = 200, 117, 5
    full = make_ctx(sc, w, h, 3, wire.ACCUM_REF_LDR8)
    full.render(3, 1)
    want, want_depth, want_rays = full.read_accum(), full.read_depth(), full.stats()["rays"]
    full.close()
    got, got_depth, rays = np.zeros_like(want), np.zeros_like(want_depth), 0
    for r in range(world):
        ctx = make_ctx(sc, w, h, 3, wire.ACCUM_REF_LDR8, rank=r, world=world)
        ctx.render(3, 1)
        rows = partition.rows_of_rank(h, r, world)
        got[rows] = ctx.read_accum()[rows]
        got_depth[rows] = ctx.read_depth()[rows]
        rays += ctx.stats()["rays"]
        ctx.close()
    assert np.array_equal(got, want) and np.array_equal(got_depth, want_depth) and rays == want_rays


def test_c5_like_config_reduced(oracle, hiplib):
    """Config C5's depth (6 bounces, 16 spp) at 160x90 on one GPU."""
    sc = scenes.demo_scene(51200)
    w, h = 160, 90
    ref = oracle.build_scene(sc)
    want, _, _, cnt, _ = oracle.render(ref, scenes.camera_block(sc.camera, w, h), w, h, 6, 16, 1, wire.ACCUM_REF_LDR8)
    ctx = make_ctx(sc, w, h, 6, wire.ACCUM_REF_LDR8)
    ctx.render(16, 1)
    got = ctx.read_accum()
    assert rel_l2(got, want) <= 1e-4 and ctx.stats()["rays"] == cnt["rays"]
    assert int((got != want).any(axis=-1).sum()) == 0
    ctx.close()


def test_c5_full_size_on_one_gpu(oracle, hiplib):
    """BASELINE config C5 at its full size -- 3840x2160, 16 spp, 6 bounces: 132.7 M paths in flight, a 15 GB wavefront
    workspace -- on ONE context, through size-independent properties: every rank of an 8-way screen partition renders
    its strips to exactly the rows the whole-image render has (checked for two of the eight ranks, one of them with the
    ragged last strip); rows that see only sky equal the oracle's rows bit for bit; a band through the middle of the
    cube equals the oracle walking the REFERENCE tree bit for bit (16 frames x 6 bounces of the full-resolution
    camera); the frames can be split over two calls; the ray count lies between its bounds."""
    sc = scenes.demo_scene(51200)
    w, h, spp, b = 3840, 2160, 16, 6
    ctx = make_ctx(sc, w, h, b, wire.ACCUM_REF_LDR8)
    ctx.render(spp, 1)
    full, full_ldr, rays = ctx.read_accum(), ctx.read_ldr(), ctx.stats()["rays"]
    assert w * h * spp <= rays <= w * h * spp * (b + 1)
    assert (full[..., 3] == 1).all() and (full[..., :3] >= 0).all() and (full[..., :3] <= spp + 1e-4).all()
    ctx.accum_reset()
    ctx.render(5, 1)
    ctx.render(11, 6)
    assert np.array_equal(ctx.read_accum(), full) and np.array_equal(ctx.read_ldr(), full_ldr)
    ctx.close()
    for r in (3, 7):
        part = make_ctx(sc, w, h, b, wire.ACCUM_REF_LDR8, rank=r, world=8)
        part.render(spp, 1)
        rows = partition.rows_of_rank(h, r, 8)
        assert part.local_rows() == len(rows)
        assert np.array_equal(part.read_accum()[rows], full[rows]) and np.array_equal(part.read_ldr()[rows], full_ldr[rows])
        part.close()
    ref = oracle.build_scene(sc)
    cam = scenes.camera_block(sc.camera, w, h)
    for band in ((0, 8), (1076, 1084)):   # sky rows; rows through the cube, both characters and the light
        acc = None
        for f in range(spp):
            c = cam.copy()
            c["frame_index"] = 1 + f
            rad, _, _ = oracle.trace_frame(ref, c, w, h, b, rows=band)
            q = oracle.screen_rgba8(rad[band[0]:band[1]])[..., :3].astype(np.float32) / np.float32(255)
            acc = q if acc is None else q + acc
        assert np.array_equal(full[band[0]:band[1], :, :3], acc), band


def test_unique_triangle_scene_reduced(oracle, hiplib):
    """S-unique (one BLAS of unique triangles filling the cube, seen from the box opening) at 20 000 triangles, 256x144,
    2 spp, 4 bounces: both builders and both kernels against the oracle."""
    sc = scenes.unique_scene(20000)
    w, h = 256, 144
    ref = oracle.build_scene(sc)
    want, want_ldr, want_depth, cnt, _ = oracle.render(ref, scenes.camera_block(sc.camera, w, h), w, h, 4, 2, 1, wire.ACCUM_REF_LDR8)
    for builder in (capi.BUILD_SAH, capi.BUILD_REFERENCE_EXACT):
        for kernel in KERNELS:
            ctx = make_ctx(sc, w, h, 4, wire.ACCUM_REF_LDR8, builder, kernel)
            ctx.render(2, 1)
            assert np.array_equal(ctx.read_accum(), want), (builder, kernel)
            assert np.array_equal(ctx.read_depth(), want_depth)
            assert kernel != capi.KERNEL_WAVEFRONT or ctx.stats()["rays"] == cnt["rays"]
            ctx.close()


def test_unique_triangle_scene_full_size(oracle, hiplib):
    """S-unique at 1 000 012 unique triangles (~110 MB of flattened records: past L2, inside the Infinity Cache),
    1920x1080, 2 spp, 4 bounces.  Properties: a 2-way partition reproduces the whole image; the audit kernel agrees bit for bit (both decide
    exact distance ties on the reference's own trees, kept beside the native ones); a band of rows equals the oracle walking the reference tree
    of the same million triangles; every pixel's first ray hits (the camera sits in the box opening), so the depth
    image holds no `far`."""
    sc = scenes.unique_scene()
    assert sc.n_unique_tris > 1_000_000
    w, h, spp, b = 1920, 1080, 2, 4
    ctx = make_ctx(sc, w, h, b, wire.ACCUM_REF_LDR8)
    ctx.render(spp, 1)
    full, depth, rays = ctx.read_accum(), ctx.read_depth(), ctx.stats()["rays"]
    aside = ctx.stats()["set_aside"]
    ctx.set_kernel(capi.KERNEL_REFERENCE_LAYOUT)
    ctx.accum_reset()
    ctx.render(spp, 1)
    audit = ctx.read_accum()
    ndiff = int((audit != full).any(axis=-1).sum())
    print("unique scene: set aside", aside, "| pixels where the audit kernel (native tree alone) differs", ndiff, "rel_l2", rel_l2(audit, full))
    assert ndiff == 0       # (both kernels decide exact ties on the reference's own trees: the oracle comparison is below)
    ctx.close()
    assert rays > 2.2 * w * h * spp                     # paths go on bouncing inside the box
    cam = scenes.camera_block(sc.camera, w, h)
    assert (depth < 0.9999).all()                       # a miss stores far / (far - near) * (1 - near / far) = 1 (main.glsl:432)
    for r in range(2):
        part = make_ctx(sc, w, h, b, wire.ACCUM_REF_LDR8, rank=r, world=2)
        part.render(spp, 1)
        rows = partition.rows_of_rank(h, r, 2)
        assert np.array_equal(part.read_accum()[rows], full[rows])
        part.close()
    ref = oracle.build_scene(sc)
    band = (536, 544)
    acc = None
    for f in range(spp):
        c = cam.copy()
        c["frame_index"] = 1 + f
        rad, _, _ = oracle.trace_frame(ref, c, w, h, b, rows=band)
        q = oracle.screen_rgba8(rad[band[0]:band[1]])[..., :3].astype(np.float32) / np.float32(255)
        acc = q if acc is None else q + acc
    got = full[band[0]:band[1], :, :3]
    print("unique scene, rows 536..543: differing pixels", int((got != acc).any(axis=-1).sum()), "rel_l2", rel_l2(got, acc))
    assert rel_l2(got, acc) <= 1e-4


def test_frame_batching_under_a_workspace_budget(hiplib, monkeypatch):
    """With a small workspace budget a many-frame render is split into batches in frame order: same image, same
    ray count as the unbatched render; queued (asynchronous) renders queue their batches and give the image of one
    blocking render of all the frames."""
    import subprocess, sys, json
    code = r'''
import sys, json, numpy as np
sys.path.insert(0, %r)
from gdpathtracing_amd import capi, host, scenes
sc = scenes.cornell_scene(); w, h = 160, 96
ctx = host.Context(0); ctx.build_scene(sc, capi.BUILD_SAH); ctx.set_params(w, h, 3, 0); ctx.set_camera(scenes.camera_block(sc.camera, w, h))
ctx.render(7, 1)
a = ctx.read_accum(); rays = ctx.stats()["rays"]
ctx.accum_reset(); ctx.render(7, 1, asynchronous=True); ctx.render(3, 8, asynchronous=True); ctx.sync()   # queued renders queue their batches
ctx2 = host.Context(0); ctx2.build_scene(sc, capi.BUILD_SAH); ctx2.set_params(w, h, 3, 0); ctx2.set_camera(scenes.camera_block(sc.camera, w, h))
ctx2.render(10, 1)
assert np.array_equal(ctx.read_accum(), ctx2.read_accum()) and np.array_equal(ctx.read_ldr(), ctx2.read_ldr())
print(json.dumps(dict(sum=float(a.sum()), rays=rays, crc=int(np.frombuffer(a.tobytes(), dtype=np.uint32).sum() %% (1 << 32)))))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for budget in ("16384", "1"):   # 1 MiB: two frames per batch at this size
        env = dict(os.environ, JPT_WORKSPACE_BUDGET_MB=budget)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert outs[0] == outs[1] and outs[0]["rays"] > 160 * 96 * 7


def test_set_aside_buffer_overflow_is_counted_not_silent(hiplib):
    """Hits the reference's traversal cannot reach are set aside and finished exactly; the buffer for them holds 1/64 of the
    paths.  With the capacity forced to 0 every such hit is instead shaded as found -- the image is then the watertight
    builder's -- and jpt_stats says how many: nothing is dropped silently, and the default capacity drops none.
    Config C2 (1280x720, 4 spp, 3 bounces), whose reference tree has a crack at pixel (688, 551)."""
    import subprocess, sys, json
    code = r'''
import sys, json, numpy as np
sys.path.insert(0, %r)
from gdpathtracing_amd import capi, host, scenes
sc = scenes.demo_scene(51200); w, h = 1280, 720
cam = scenes.camera_block(sc.camera, w, h)
img = {}
for name, builder in (("reach", capi.BUILD_SAH), ("watertight", capi.BUILD_SAH_WATERTIGHT)):
    ctx = host.Context(0); ctx.build_scene(sc, builder); ctx.set_params(w, h, 3, 0); ctx.set_camera(cam)
    ctx.render(4, 1)
    img[name] = ctx.read_accum(); st = ctx.stats()
    if name == "reach": aside, dropped = st["set_aside"], st["set_aside_dropped"]
    ctx.close()
print(json.dumps(dict(aside=aside, dropped=dropped, differing=int((img["reach"] != img["watertight"]).any(axis=-1).sum()))))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for cap in (None, "0"):
        env = dict(os.environ)
        if cap is not None:
            env["JPT_SET_ASIDE_CAP"] = cap
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        res[cap] = json.loads(r.stdout.strip().splitlines()[-1])
    print("C2 set-aside hits:", res)
    assert res[None]["aside"] >= 1 and res[None]["dropped"] == 0
    assert res[None]["differing"] >= 1                     # honoured, the crack changes the image ...
    assert res["0"]["dropped"] == res["0"]["aside"] >= 1   # ... dropped, it is counted ...
    assert res["0"]["differing"] == 0                      # ... and the image is the watertight builder's


def test_split_readback_overlaps_the_next_render(hiplib):
    """jpt_readback_ldr_begin/end: the copy of frame N's screen image is queued behind frame N, frame N+1 is
    queued behind the copy; `end` returns frame N's image exactly."""
    sc = scenes.cornell_scene()
    w, h = 128, 72
    ctx = make_ctx(sc, w, h, 2, wire.ACCUM_REF_LDR8)
    ctx.render(1, 1)
    want1 = ctx.read_ldr()
    ctx.render(1, 2)
    want2 = ctx.read_ldr()
    ctx.accum_reset()
    ctx.render(1, 1, asynchronous=True)
    ctx.readback_ldr_begin()
    with pytest.raises(capi.JptError, match="already in flight"):
        ctx.readback_ldr_begin()
    ctx.render(1, 2, asynchronous=True)          # queued behind the copy
    got1 = ctx.readback_ldr_end()
    with pytest.raises(capi.JptError, match="no read-back"):
        ctx.readback_ldr_end()
    assert np.array_equal(got1, want1)
    assert np.array_equal(ctx.read_ldr(), want2)
    ctx.close()


@pytest.mark.parametrize("groups", [2, 3, 4])
def test_frame_groups_on_concurrent_streams(oracle, hiplib, monkeypatch, groups):
    """Large renders split their frames into groups that run the pipeline concurrently on helper streams
    (launch_wf2_render); paths never cross groups and the accumulation reads the frames in order, so the result is the
    serial one bit for bit -- accumulation buffer, display image, depth (written by the last frame's group) and the
    ray count (summed over the groups)."""
    sc = scenes.demo_scene(3000)
    w, h, bounces, frames = 200, 120, 3, 7          # 7 frames: uneven groups
    ref = oracle.build_scene(sc)
    cam = scenes.camera_block(sc.camera, w, h)
    want, want_ldr, want_depth, cnt, _ = oracle.render(ref, cam, w, h, bounces, frames, 3, wire.ACCUM_REF_LDR8)
    monkeypatch.setenv("JPT_GROUPS", str(groups))
    ctx = make_ctx(sc, w, h, bounces, wire.ACCUM_REF_LDR8, capi.BUILD_REFERENCE_EXACT)
    ctx.render(frames, 3)
    assert np.array_equal(ctx.read_accum(), want) and np.array_equal(ctx.read_ldr(), want_ldr)
    assert np.array_equal(ctx.read_depth(), want_depth)
    assert ctx.stats()["rays"] == cnt["rays"]
    # a second render continues the accumulation (frame_count carries over) exactly like the serial path
    ctx.render(2, 10)
    monkeypatch.setenv("JPT_GROUPS", "1")
    solo = make_ctx(sc, w, h, bounces, wire.ACCUM_REF_LDR8, capi.BUILD_REFERENCE_EXACT)
    solo.render(frames, 3)
    solo.render(2, 10)
    assert np.array_equal(ctx.read_accum(), solo.read_accum()) and np.array_equal(ctx.read_ldr(), solo.read_ldr())
    ctx.close(); solo.close()


def test_asynchronous_renders_pipeline_in_order(oracle, hiplib):
    """jpt_render_async: consecutive asynchronous renders run their path kernels on alternating helper streams and
    workspaces while the accumulation kernels stay in call order (chained through the context's stream).  Five queued renders (continuing one
    accumulation, then a reset, then a blocking render in between) give exactly what the same calls give one by one."""
    sc = scenes.demo_scene(2500)
    w, h, bounces = 176, 100, 3
    cam = scenes.camera_block(sc.camera, w, h)
    calls = [(2, 1), (1, 3), (3, 4), (2, 7), (1, 9)]

    def run(asynchronous, priorities=None):
        ctx = make_ctx(sc, w, h, bounces, wire.ACCUM_REF_LDR8, capi.BUILD_REFERENCE_EXACT)
        outs = []
        for k, (n, first) in enumerate(calls):
            if priorities:      # jpt_set_stream_priority between queued renders: slot streams are drained and re-made
                ctx.set_stream_priority(priorities[k % len(priorities)])
            if k == 3:
                ctx.accum_reset()
            ctx.render(n, first, asynchronous=asynchronous and k != 2)   # call 2 is blocking in both runs
            if k in (1, 4):
                outs.append((ctx.read_accum(), ctx.read_ldr(), ctx.read_depth()))
        ctx.close()
        return outs

    a, b = run(True), run(False)
    c = run(True, [capi.STREAM_PRIORITY_NORMAL, capi.STREAM_PRIORITY_LOW, capi.STREAM_PRIORITY_HIGH, capi.STREAM_PRIORITY_DEFAULT])
    for (x, y, z) in zip(a, b, c):
        for u, v, t in zip(x, y, z):
            assert np.array_equal(u, v) and np.array_equal(u, t)
    ref = oracle.build_scene(sc)
    want, want_ldr, _, _, _ = oracle.render(ref, cam, w, h, bounces, 3, 7, wire.ACCUM_REF_LDR8)   # frames 7, 8, 9 after the reset
    assert np.array_equal(a[1][0], want) and np.array_equal(a[1][1], want_ldr)


def test_one_render_in_flight_at_a_time_then_a_burst(oracle, hiplib):
    """A host that queues one render, waits, queues the next: from the third such render on the library launches them like
    blocking renders (frame groups, full-width launches: jpt_capi.cpp, `lone_async`), and goes back to the pipelined form as soon
    as a render finds work in flight.  Eight renders continuing one accumulation -- five one at a time (sync, read-back or the
    split read-back in between), then a burst of three -- leave what the same frames leave through jpt_render, and that is the
    oracle's image."""
    sc = scenes.demo_scene(2500)
    w, h, bounces = 200, 120, 3
    cam = scenes.camera_block(sc.camera, w, h)
    calls = [(2, 1), (1, 3), (2, 4), (3, 6), (1, 9), (2, 10), (2, 12), (1, 14)]

    def run(asynchronous):
        ctx = make_ctx(sc, w, h, bounces, wire.ACCUM_REF_LDR8, capi.BUILD_REFERENCE_EXACT)
        mid = None
        for k, (n, first) in enumerate(calls):
            ctx.render(n, first, asynchronous=asynchronous)
            if k < 5:                                   # one at a time
                if k == 1:
                    ctx.read_ldr()
                elif k == 3:
                    ctx.readback_ldr_begin()
                    ctx.readback_ldr_end()
                else:
                    ctx.sync()
            if k == 4:
                mid = ctx.read_accum()
        out = (mid, ctx.read_accum(), ctx.read_ldr(), ctx.read_depth())
        ctx.close()
        return out

    a, b = run(True), run(False)
    for u, v in zip(a, b):
        assert np.array_equal(u, v)
    ref = oracle.build_scene(sc)
    want, want_ldr, _, _, _ = oracle.render(ref, cam, w, h, bounces, 14, 1, wire.ACCUM_REF_LDR8)
    assert np.array_equal(a[1], want) and np.array_equal(a[2], want_ldr)


def test_memory_policy_caps_the_workspaces_and_keeps_the_image(hiplib):
    """jpt_set_memory_policy: queued renders with six, four, two and one workspace in flight, and with a budget per workspace
    that forces batches of frames, leave the same accumulation buffer; jpt_get_workspace_bytes shows the cap taking
    effect (workspaces past it are freed by the call)."""
    sc = scenes.demo_scene(2500)
    w, h, bounces, spp = 320, 180, 3, 6
    cam = scenes.camera_block(sc.camera, w, h)

    def run(slots, budget):
        ctx = make_ctx(sc, w, h, bounces, wire.ACCUM_REF_LDR8, capi.BUILD_SAH)
        if slots or budget:
            ctx.set_memory_policy(slots, budget)
        for k in range(6):
            ctx.render(spp, 1 + spp * k, asynchronous=True)
        ctx.sync()
        out, used = ctx.read_accum(), ctx.workspace_bytes()
        ctx.set_memory_policy(1, budget)
        after = ctx.workspace_bytes()
        ctx.render(spp, 1 + spp * 6, asynchronous=True)       # still renders after the cap moved
        ctx.sync()
        ctx.close()
        return out, used, after

    full, used4, after4 = run(4, 0)
    one_ws = used4 // 4
    assert used4 == 4 * one_ws and after4 == one_ws            # four equal workspaces; the cap to one freed three
    ruled, used_rule, _ = run(0, 0)                             # the library's rule: four, or six where six slot streams run side by side
    assert used_rule in (4 * one_ws, 6 * one_ws) and np.array_equal(full, ruled)
    six, used6, _ = run(6, 0)
    assert used6 == 6 * one_ws and np.array_equal(full, six)
    two, used2, _ = run(2, 0)
    one, used1, _ = run(1, 0)
    assert used2 == 2 * one_ws and used1 == one_ws
    batched, used_b, _ = run(4, one_ws // 3)                    # two of six frames fit the budget: three batches per render
    assert used_b <= 4 * (one_ws // 3)
    for other in (two, one, batched):
        assert np.array_equal(full, other)
    with pytest.raises(RuntimeError):
        make_ctx(sc, w, h, bounces, wire.ACCUM_REF_LDR8, capi.BUILD_SAH).set_memory_policy(9, 0)
    # a budget below ONE frame's workspace is refused by the render (JPT_E_LIMIT), nothing is allocated past it;
    # the audit kernel, which needs no workspace, still renders
    ctx = make_ctx(sc, w, h, bounces, wire.ACCUM_REF_LDR8, capi.BUILD_SAH)
    ctx.set_memory_policy(0, one_ws // (spp * 2))
    for asynchronous in (False, True):
        with pytest.raises(capi.JptError) as e:
            ctx.render(spp, 1, asynchronous=asynchronous)
        assert "(-3)" in str(e.value) and "budget" in str(e.value)
    assert ctx.workspace_bytes() == 0
    ctx.set_kernel(capi.KERNEL_REFERENCE_LAYOUT)
    ctx.render(spp, 1)
    ctx.set_kernel(capi.KERNEL_WAVEFRONT)
    ctx.set_memory_policy(0, 0)
    ctx.render(spp, 1)
    ctx.close()


def test_foreign_work_on_the_context_stream_sees_each_queued_render(hiplib):
    """jpt_get_stream: work a framework queues on the context's stream between asynchronous renders (bench.py's
    gather) is ordered after the render before it and before the render after it -- although the renders' kernels,
    the accumulation included, run on the library's own streams."""
    import torch
    sc = scenes.demo_scene(2500)
    w, h, bounces = 320, 180, 3
    ctx = make_ctx(sc, w, h, bounces, wire.ACCUM_HDR_F32)
    s = ctx.get_stream()
    assert s != 0
    stream = torch.cuda.ExternalStream(s, device=torch.device("cuda", 0))
    ptr, nbytes = ctx.device_accum()

    class _View:
        __cuda_array_interface__ = {"shape": (nbytes // 4,), "typestr": "<f4", "data": (ptr, False), "version": 2}

    accum = torch.as_tensor(_View(), device=torch.device("cuda", 0))
    snaps = []
    with torch.cuda.stream(stream):
        for k in range(6):
            ctx.accum_reset()
            ctx.render(2, 1 + 2 * k, asynchronous=True)
            snaps.append(accum.clone())          # queued on the context's stream right behind render k
        stream.synchronize()
    ctx.sync()
    solo = make_ctx(sc, w, h, bounces, wire.ACCUM_HDR_F32)
    for k in range(6):
        solo.accum_reset()
        solo.render(2, 1 + 2 * k)
        assert np.array_equal(snaps[k].cpu().numpy().reshape(h, w, 4), solo.read_accum()), k
    # replacing the stream and restoring it: the getter follows
    other = torch.cuda.Stream()
    ctx.set_stream(other.cuda_stream)
    assert ctx.get_stream() == other.cuda_stream
    ctx.set_stream(None)
    assert ctx.get_stream() == s
    ctx.close(); solo.close()


def test_c4_device_refit_at_full_size(hiplib):
    """The instanced scene of config C4 (1 026 instances, a TLAS ten records deep) at 1920x1080: every instance moves,
    jpt_scene_refit_tlas, queued renders; sums, depth and ray count equal a fresh commit of the moved scene.  Then the
    host route (jpt_scene_update_tlas) on top of the refitted state gives the same image again."""
    sc = scenes.instanced_scene()
    w, h, bounces, frames = 1920, 1080, 4, 2
    cam = scenes.camera_block(sc.camera, w, h)
    rng = np.random.RandomState(11)
    ctx = make_ctx(sc, w, h, bounces, wire.ACCUM_REF_LDR8)
    ctx.render(1, 1, asynchronous=True)
    import copy
    moved = copy.deepcopy(sc)
    for step in range(2):
        for i in range(2, len(moved.instances)):
            t = np.asarray(moved.instances[i].transform, dtype=np.float32).copy()
            t[9:12] += rng.uniform(-0.4, 0.4, size=3).astype(np.float32)
            moved.instances[i].transform = t
        ctx.refit_tlas(np.stack([np.asarray(i.transform, dtype=np.float32) for i in moved.instances]))
        ctx.accum_reset()
        ctx.render(frames, 1, asynchronous=True)
    got, got_depth = ctx.read_accum(), ctx.read_depth()
    fresh = make_ctx(moved, w, h, bounces, wire.ACCUM_REF_LDR8)
    fresh.render(frames, 1, counted=True)
    want, want_depth, want_rays = fresh.read_accum(), fresh.read_depth(), fresh.stats()["rays"]
    fresh.close()
    assert np.array_equal(got, want) and np.array_equal(got_depth, want_depth)
    ctx.accum_reset()
    ctx.render(frames, 1, counted=True)          # counted renders run serially on the context's stream: same records
    assert ctx.stats()["rays"] == want_rays and np.array_equal(ctx.read_accum(), want)
    ctx.update_tlas()                            # host rebuild from the same transforms: new topology, same image
    ctx.accum_reset()
    ctx.render(frames, 1)
    assert np.array_equal(ctx.read_accum(), want) and np.array_equal(ctx.read_depth(), want_depth)
    ctx.close()


@pytest.mark.parametrize("launcher", ["torchrun", "multi"])
def test_bench_self_launch_two_ranks_on_one_gpu(hiplib, launcher):
    """`python3 bench.py --gpus 2` with WORLD_SIZE unset (VERDICT r03 weak 3b / task 2): the default launcher starts
    torch.distributed.run itself as a child process (gloo here: two ranks share the box's one GPU), `--launcher multi` runs
    the one-process jpt_multi route.  Either way: rc 0, ONE JSON line, two ranks seen, both assembled images (C3's size and
    C5's) bit-identical to one context's."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["JPT_BENCH_BACKEND"] = "gloo"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1", "--tris", "5000"]
    if launcher == "multi":
        cmd += ["--launcher", "multi"]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["multi_gpu"]["ranks_seen"] == 2
    assert d["verified_bit_identical_to_one_context"] is True
    assert d["multi_gpu"]["c5"]["verified_bit_identical_to_one_context"] is True
    if launcher == "multi":
        assert d["multi_gpu"]["gather_plan"] == {"peer_copies": 1, "distinct_streams": 1, "own_piece_copies": 0}


def test_bench_line_holds_the_contract_at_one_gpu(hiplib):
    """The one JSON line of `python bench.py` at N = 1 (a small scene, few steps): the contract's fields, the `roofline` object with the
    binding resource on top and the HBM pair beneath it (fractions are fractions or null -- the counter passes are committed for the
    profiled workloads only), `cpu_baseline`, `parity` against the oracle on the image the timed region left behind, exit status 0."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "JPT_BENCH_BACKEND", "JPT_BENCH_FORCE_DIST")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "6", "--warmup", "2", "--tris", "5000", "--width", "640", "--height", "360", "--project-ranks", "2",
           "--preheat-ms", "20"]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["unit"] == "Mrays/s" and d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"]
    assert d["config"]["preheat_ms"] == 20.0 and d["config"]["preheat_steps"] >= 10
    assert abs(d["value"] - d["config"]["rays_per_step"] / d["ms_per_step"] / 1e3) <= 1e-3 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "lane_frac", "traffic", "kernel", "kernel_ms", "hbm", "binding"):
        assert k in r, k
    for k in ("achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes", "cache_served_achieved", "cache_served_frac"):
        assert k in r["hbm"], k
    for f in (r["frac"], r["lane_frac"], r["hbm"]["frac"]):
        assert f is None or 0.0 <= f <= 1.0
    assert r["kernel_ms"] > 0.0 and r["hbm"]["algorithmic_bytes"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "Mrays/s" and "sample" in c
    assert d["parity"]["ok"] is True and d["parity"]["differing_pixels"] == 0
    assert d["dropin"]["differing_pixels_vs_commit_route"] == 0
    assert d["projected_scaling"]["measured"] is False and d["projected_scaling"]["c3"]["ranks"]["2"]["rays_sum_equals_whole"] is True


def test_bench_multi_gpu_flow_with_one_rank_of_rccl(hiplib):
    """VERDICT r05 task 2b: the first time RCCL is loaded beside the library.  bench.py's N > 1 flow -- a torch.distributed process
    group on the `nccl` backend (= RCCL), every step's dist.gather of the context's float4 rows through the ExternalStream the bench
    orders its work on, jpt_assemble_from_ranks, the per-rank phase times (all_gather), the C5 leg -- with a world of ONE rank
    (JPT_BENCH_FORCE_DIST=1), which is what a one-GPU box can run of it: RCCL's communicator and streams come up beside the six
    high-priority slot streams and GPU_MAX_HW_QUEUES=6, and the assembled images are one context's, bit for bit."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "JPT_BENCH_BACKEND")}
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        env["MASTER_PORT"] = str(sk.getsockname()[1])
    env["MASTER_ADDR"] = "127.0.0.1"
    env["JPT_BENCH_FORCE_DIST"] = "1"
    env.pop("GPU_MAX_HW_QUEUES", None)   # (bench.py's own default: six)
    env.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)   # (... and its own setdefault, as on the driver's torchrun form)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "1", "--tris", "5000", "--no-cpu-baseline", "--preheat-ms", "0"]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    m = d["multi_gpu"]
    assert m["backend"] == "nccl" and m["ranks_seen"] == 1 and m["rccl_version"]
    assert d["verified_bit_identical_to_one_context"] is True and m["c5"]["verified_bit_identical_to_one_context"] is True
    r0 = m["ranks"][0]
    assert r0["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and r0["GPU_MAX_HW_QUEUES"] == "6" and r0["renders_in_flight"] in (4, 6)
    assert len(m["c3_phase_ms_per_rank"]) == 1 and len(m["c3_phase_ms_per_rank"][0]) == 3


@pytest.mark.parametrize("switch", ["JPT_TAIL=1 JPT_TAIL_ROUNDS=2 JPT_TAIL_LANES=8", "JPT_GROUPS=3", "JPT_GROUPS=2 JPT_PIPE_SLOTS=2"])
def test_alternative_tracing_launches_are_bit_identical(hiplib, switch):
    """The environment switches (include/jpt.h) are read once per process, so the alternative launches run the parity tests in a
    child process: JPT_TAIL=1 with eager thresholds (a wave walks its last rays with all its lanes, coop_walk: what large scenes get
    by default for their few very long rays), JPT_GROUPS=3 / 2 (blocking renders split into frame groups of unequal size: every
    group's paths live in a block of their own, which wf2_accumulate must find) and two renders in flight instead of four or six.
    Same images bit for bit as the default launches: a subset of the parity suite, against the oracle."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for kv in switch.split():
        k, v = kv.split("=")
        env[k] = v
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_parity.py"), os.path.join(root, "tests", "test_fuzz.py"),
                        os.path.abspath(__file__), "-m", "gpu", "-x", "-q", "-k",
                        "c1_cornell or demo_scene_multi_frame or coincident or tie_between or duplicated or native_tree or walk_length_statistics"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-3000:]
    assert " passed" in p.stdout and "failed" not in p.stdout


def test_walk_length_statistics_of_a_counting_render(hiplib):
    """jpt_stats.walk_steps_max / walk_steps_hist (ABI 4): the record steps every traced ray took, from a counting render.  Every
    ray a kernel walks is in exactly one bucket (rays - sky_culled of them), the longest walk lies in the last non-empty
    bucket, and the same render with the long-walk hand-over made eager (another process: the switch is read once) leaves the
    image alone -- covered by test_alternative_tracing_launches_are_bit_identical -- while this one pins the counters."""
    sc = scenes.demo_scene(5000)
    ctx = make_ctx(sc, 320, 180, 3, wire.ACCUM_REF_LDR8, capi.BUILD_SAH)
    ctx.render(2, 1, counted=True)
    st = ctx.stats()
    hist = st["walk_steps_hist"]
    walked = st["rays"] - st["sky_culled"]
    # (the later segments of a set-aside path are traced by wf2_finish, which keeps no walk statistics: a handful at most)
    assert walked - 8 * (st["set_aside"] + 1) <= sum(hist) <= walked
    last = max(k for k, v in enumerate(hist) if v)
    lo = 0 if last == 0 else 16 * 4 ** (last - 1)
    assert lo <= st["walk_steps_max"] < 16 * 4 ** last
    assert st["walk_steps_max"] >= 1
    # every record step is counted once: internal records + leaf turns + instance entries (a leaf turn tests one or two triangles)
    assert st["walk_steps_max"] <= st["blas_expand"] + st["tlas_expand"] + st["tri_tests"] + st["inst_visits"]
    ctx.close()


def test_depth_image_is_optional(hiplib):
    """jpt_set_outputs (ABI 5; SURVEY.md section 7: the r32f depth image has one reader, temporal reprojection).  With the depth
    output off the kernels keep no first-hit distances and write no depth image: the accumulation and the display image are the
    same bytes, jpt_read_depth_f32 says JPT_E_STATE; switched on again the next render writes it; the temporal mode produces it
    whatever the switch says.  Blocking, queued and audit-kernel renders."""
    sc = scenes.demo_scene(5000)
    w, h = 320, 184
    for kernel in KERNELS:
        ctx = make_ctx(sc, w, h, 3, wire.ACCUM_REF_LDR8, capi.BUILD_SAH, kernel)
        ctx.render(3, 1)
        want, want_ldr, want_depth = ctx.read_accum(), ctx.read_ldr(), ctx.read_depth()
        ctx.set_outputs(depth=False)
        for asynchronous in (False, True):
            ctx.accum_reset()
            ctx.render(3, 1, asynchronous=asynchronous)
            assert np.array_equal(ctx.read_accum(), want) and np.array_equal(ctx.read_ldr(), want_ldr)
            with pytest.raises(capi.JptError) as e:
                ctx.read_depth()
            assert "(-4)" in str(e.value) and "switched off" in str(e.value)      # JPT_E_STATE
        ctx.set_outputs(depth=True)
        with pytest.raises(capi.JptError):
            ctx.read_depth()                      # on again, but no render has written it yet
        ctx.accum_reset()
        ctx.render(3, 1)
        assert np.array_equal(ctx.read_depth(), want_depth) and np.array_equal(ctx.read_accum(), want)
        ctx.close()


def test_rank_shares_add_up_to_the_whole_image(hiplib):
    """bench.py's `projected_scaling` times every rank's share of a render on one device (VERDICT r04 task 2): the eight shares of an
    8-way partition are the whole image's work -- their ray segments sum to the whole render's, every counter does, and their rows
    are the whole image's rows."""
    sc = scenes.demo_scene(5000)
    w, h, spp, bounces = 480, 270, 2, 3
    whole = make_ctx(sc, w, h, bounces, wire.ACCUM_REF_LDR8)
    whole.render(spp, 1, counted=True)
    want, st = whole.read_accum(), whole.stats()
    whole.close()
    keys = ("rays", "blas_expand", "tri_tests", "tlas_expand", "inst_visits", "shaded_hits", "sky_culled", "zero_throughput")
    for world in (2, 8):
        tot = dict.fromkeys(keys, 0)
        got = np.zeros_like(want)
        ctx = make_ctx(sc, w, h, bounces, wire.ACCUM_REF_LDR8)
        for r in range(world):
            ctx.set_partition(r, world)
            ctx.set_params(w, h, bounces, wire.ACCUM_REF_LDR8)
            ctx.set_camera(scenes.camera_block(sc.camera, w, h))
            ctx.render(spp, 1, counted=True)
            s = ctx.stats()
            for k in keys:
                tot[k] += s[k]
            rows = partition.rows_of_rank(h, r, world)
            got[rows] = ctx.read_accum()[rows]
        ctx.close()
        assert tot == {k: st[k] for k in keys}
        assert np.array_equal(got, want)


def test_zero_throughput_vertices_are_counted(hiplib):
    """jpt_stats.zero_throughput (ABI 5): path vertices that go on with a throughput of exactly (0, 0, 0) -- brdf() returned 0 because
    n.v < 0 at a hit whose interpolated normal faces away from the viewer (brdfs.glsl:14) -- counted by a counting render."""
    sc = scenes.demo_scene(5000)
    w, h, spp, bounces = 320, 180, 2, 4
    ctx = make_ctx(sc, w, h, bounces, wire.ACCUM_REF_LDR8)
    ctx.render(spp, 1, counted=True)
    st = ctx.stats()
    ctx.close()
    assert 0 <= st["zero_throughput"] <= st["shaded_hits"]
    # such vertices exist on a smooth-shaded mesh seen at grazing angles, and they are few
    assert st["zero_throughput"] < 0.05 * st["rays"]


@pytest.mark.gpu
@pytest.mark.parametrize("queues,slots", [("6", 6), ("4", 4)])
def test_renders_in_flight_follow_the_hardware_queues(hiplib, queues, slots):
    """Six renders in flight where six of the library's streams get a hardware queue each (GPU_MAX_HW_QUEUES >= 6 at the process's
    first HIP call), four where they would have to share the default pool of four -- measured by the library before its first queued
    render (jpt_capi.cpp, six_queues_probe), reported by jpt_renders_in_flight.  Either way a queue of renders leaves the image of
    the same renders made one at a time.  The variable is read once per process: a child process per setting."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from gdpathtracing_amd import capi, host, scenes
sc = scenes.demo_scene(3000)
w, h = 256, 144
cam = scenes.camera_block(sc.camera, w, h)
def ctx_for():
    c = host.Context(0); c.build_scene(sc, capi.BUILD_SAH); c.set_params(w, h, 3, capi.ACCUM_REF_LDR8); c.set_camera(cam); return c
a = ctx_for()
assert a.renders_in_flight() == 0
for k in range(9): a.render(2, 1 + 2 * k, asynchronous=True)
a.sync()
b = ctx_for()
for k in range(9): b.render(2, 1 + 2 * k)
assert np.array_equal(a.read_accum(), b.read_accum()) and np.array_equal(a.read_ldr(), b.read_ldr())
print("slots", a.renders_in_flight())
''' % root
    env = dict(os.environ)
    env["GPU_MAX_HW_QUEUES"] = queues
    env.pop("JPT_PIPE_SLOTS", None)
    p = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    assert ("slots %d" % slots) in p.stdout, p.stdout
