#!/bin/bash
# exact event counters and phase statistics of one C3 render for one or more library builds: tools/counters.sh <label>:<lib.so or -> ...
cd "$GRAFT_REPO_ROOT"
for spec in "$@"; do
  label="${spec%%:*}"; lib="${spec#*:}"
  if [ "$lib" != "-" ]; then export JPT_LIB="$lib"; else unset JPT_LIB; fi
  python - "$label" <<'PY'
import sys, json
sys.path.insert(0, '.')
from gdpathtracing_amd import capi, host, scenes
label = sys.argv[1]
for cam_name in ("demo", "closeup"):
    sc = scenes.demo_scene(51200)
    if cam_name == "closeup":
        sc.camera = scenes.CameraDesc(scenes.transform12(None, (0.0, 0.0, 4.2)), fov_deg=75.0)
    ctx = host.Context(0); ctx.build_scene(sc, capi.BUILD_SAH); ctx.set_params(1920, 1080, 4, capi.ACCUM_REF_LDR8)
    ctx.set_camera(scenes.camera_block(sc.camera, 1920, 1080))
    ctx.render(8, 1, counted=True)
    st = ctx.stats()
    ph = st["phase"]
    print(label, cam_name, {k: st[k] for k in ("rays", "blas_expand", "tri_tests", "tlas_expand", "inst_visits", "shaded_hits", "sky_culled", "zero_throughput") if k in st},
          "rounds %d node_iters %d lanes/iter %.1f leaf_phases %d lanes %.1f inst_phases %d lanes %.1f" % (
              ph[0], ph[1], ph[2] / max(ph[1], 1), ph[3], ph[4] / max(ph[3], 1), ph[5], ph[6] / max(ph[5], 1)))
    ctx.close()
PY
done
