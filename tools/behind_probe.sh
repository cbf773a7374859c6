#!/bin/bash
# how many record steps a distance stored with each stack entry would cull (counting build with -DJPT_COUNT_BEHIND)
cd "$GRAFT_REPO_ROOT"
make -s -C gdpathtracing_amd/csrc -j8 OUT=/tmp/libjpt_behind.so OBJDIR=/tmp/obj_behind EXTRA="-DJPT_COUNT_BEHIND" > /tmp/build_behind.log 2>&1 || tail -3 /tmp/build_behind.log
export JPT_LIB=/tmp/libjpt_behind.so
python - <<'PY'
import sys
sys.path.insert(0, '.')
from gdpathtracing_amd import capi, host, scenes
def run(name, sc):
    ctx = host.Context(0); ctx.build_scene(sc, capi.BUILD_SAH); ctx.set_params(1920, 1080, 4, capi.ACCUM_REF_LDR8)
    ctx.set_camera(scenes.camera_block(sc.camera, 1920, 1080))
    ctx.render(8, 1, counted=True)
    st = ctx.stats()
    steps = st["blas_expand"] + st["tlas_expand"]
    h = st["walk_steps_hist"]
    print("%s record steps %d  behind the hit %d (%.1f %%)  no usable child %d (%.1f %%)  leaves %d instance entries %d" % (
        name, steps, h[6], 100.0 * h[6] / steps, h[7], 100.0 * h[7] / steps, st["tri_tests"], st["inst_visits"]))
    ctx.close()
sc = scenes.demo_scene(51200); run("demo", sc)
sc.camera = scenes.CameraDesc(scenes.transform12(None, (0.0, 0.0, 4.2)), fov_deg=75.0); run("closeup", sc)
run("C4", scenes.instanced_scene())
run("unique", scenes.unique_scene(1000000))
PY
