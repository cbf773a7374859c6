#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz.

PARITY UNPINNED: the reference has no golden vectors and cannot be executed here, so these fixtures are
produced by the repository's own C oracle (oracle/).  They pin the oracle against drift between rounds
and give the GPU tests committed expected outputs; they are NOT outputs of the reference.

    python tests/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gdpathtracing_amd import scenes, wire  # noqa: E402
from oracle import binding as ob  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

CASES = {
    # name: (scene factory, W, H, bounces, frames, first_frame, accum_mode)
    "cornell_64x64_b2_f2_ldr8": (scenes.cornell_scene, 64, 64, 2, 2, 1, wire.ACCUM_REF_LDR8),
    "cornell_64x64_b4_f3_hdr": (scenes.cornell_scene, 64, 64, 4, 3, 5, wire.ACCUM_HDR_F32),
    "demo2048_96x54_b4_f2_ldr8": (lambda: scenes.demo_scene(2048), 96, 54, 4, 2, 1, wire.ACCUM_REF_LDR8),
    "inst36_80x45_b3_f2_ldr8": (lambda: scenes.instanced_scene(6, 3, 128), 80, 45, 3, 2, 1, wire.ACCUM_REF_LDR8),
}


def render_case(name):
    mk, w, h, b, f, first, mode = CASES[name]
    sc = mk()
    ref = ob.build_scene(sc)
    cam = scenes.camera_block(sc.camera, w, h)
    accum, ldr, depth, cnt, _ = ob.render(ref, cam, w, h, b, f, first, mode)
    return sc, dict(accum=accum, ldr=ldr, depth=depth, rays=np.int64(cnt["rays"]), tri_tests=np.int64(cnt["tri_tests"]),
                    n_nodes=np.int64(len(ref.bvh_nodes)), n_tlas=np.int64(len(ref.tlas_nodes)))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    for name in CASES:
        _, d = render_case(name)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
        print(name, d["accum"].shape, int(d["rays"]))
