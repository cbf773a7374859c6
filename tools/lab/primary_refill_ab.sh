#!/bin/bash
# the primary launch alone (max_bounces 0) with different refill thresholds: do whole chunks stay together better?
cd "$GRAFT_REPO_ROOT"
for ri in 24 8 16 32 40 48 56 64; do
  export JPT_REFILL_IDLE=$ri
  python - $ri <<'PY'
import sys
sys.path.insert(0, '.')
from gdpathtracing_amd import capi, host, scenes
out = []
def run(name, sc):
    ctx = host.Context(0); ctx.build_scene(sc, capi.BUILD_SAH)
    ctx.set_params(1920, 1080, 0, capi.ACCUM_REF_LDR8); ctx.set_camera(scenes.camera_block(sc.camera, 1920, 1080))
    ctx.render(8, 1, counted=True); st = ctx.stats(); ph = st["phase"]
    ctx.set_kernel_timing(True)
    best = 1e9
    for _ in range(5):
        ctx.accum_reset(); ctx.render(8, 1); best = min(best, ctx.stats()["last_primary_ms"])
    out.append("%s %.1f us rounds %d lanes %.1f" % (name, best * 1e3, ph[0], ph[2] / max(ph[1], 1)))
    ctx.close()
sc = scenes.demo_scene(51200); run("demo", sc)
sc.camera = scenes.CameraDesc(scenes.transform12(None, (0.0, 0.0, 4.2)), fov_deg=75.0); run("closeup", sc)
run("C4", scenes.instanced_scene())
print("refill_idle=%s: " % sys.argv[1] + " | ".join(out), flush=True)
PY
done
