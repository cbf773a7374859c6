#!/bin/bash
# path ids with the frame running fastest + a wave's 64 consecutive ids (every frame's sample of 8 pixels) against (a) the same
# numbering with one frame's sample of an 8x8 tile per wave (-DJPT_PRIMARY_SAMPLES_TOGETHER=0) and (b) the library of the
# commit before (gdpathtracing_amd/csrc/libjpt_head_ab.so, built by hand from `git archive HEAD`: [frame][slot] ids, tile per wave)
cd "$GRAFT_REPO_ROOT"
make -s -C gdpathtracing_amd/csrc -j8 OUT=/tmp/libjpt_tile.so OBJDIR=/tmp/obj_tile EXTRA="-DJPT_PRIMARY_SAMPLES_TOGETHER=0" > /tmp/build_tile.log 2>&1 || tail -3 /tmp/build_tile.log
cp gdpathtracing_amd/csrc/libjpt_head_ab.so /tmp/libjpt_head.so
python -m pytest tests/test_gpu_parity.py tests/test_fuzz.py -m gpu -x -q 2>&1 | tail -2
JPT_LIB=/tmp/libjpt_tile.so python -m pytest tests/test_gpu_parity.py tests/test_fuzz.py -m gpu -x -q 2>&1 | tail -2
for v in default tile head; do
  if [ $v = default ]; then unset JPT_LIB; else export JPT_LIB=/tmp/libjpt_$v.so; fi
  python - $v <<'PY'
import sys
sys.path.insert(0, '.')
from gdpathtracing_amd import capi, host, scenes
def run(name, sc):
    ctx = host.Context(0); ctx.build_scene(sc, capi.BUILD_SAH)
    ctx.set_params(1920, 1080, 0, capi.ACCUM_REF_LDR8); ctx.set_camera(scenes.camera_block(sc.camera, 1920, 1080))
    ctx.render(8, 1, counted=True); st = ctx.stats(); ph = st["phase"]
    ctx.set_kernel_timing(True)
    best = 1e9
    for _ in range(5):
        ctx.accum_reset(); ctx.render(8, 1); best = min(best, ctx.stats()["last_primary_ms"])
    print(sys.argv[1], name, "primary launch %.1f us; rounds %d record-step turns %d lanes %.1f leaf turns %d lanes %.1f instance turns %d lanes %.1f" % (
        best * 1e3, ph[0], ph[1], ph[2] / max(ph[1], 1), ph[3], ph[4] / max(ph[3], 1), ph[5], ph[6] / max(ph[5], 1)), flush=True)
    ctx.close()
sc = scenes.demo_scene(51200); run("demo", sc)
sc.camera = scenes.CameraDesc(scenes.transform12(None, (0.0, 0.0, 4.2)), fov_deg=75.0); run("closeup", sc)
run("C4", scenes.instanced_scene())
run("unique", scenes.unique_scene(1000000))
PY
done
rate() { python tools/rate.py "$@" 2>&1 | grep -o "[0-9.]* us/step\|rror.*" | tail -1; }
for rep in 1 2 3; do for v in default tile head; do
  if [ $v = default ]; then unset JPT_LIB; else export JPT_LIB=/tmp/libjpt_$v.so; fi
  echo "$v: C3 $(rate 1920 1080 8 150) | closeup $(RATE_CLOSEUP=1 rate 1920 1080 8 40) | C4 $(RATE_SCENE=instanced rate 1920 1080 8 40) | unique $(RATE_SCENE=unique rate 1920 1080 8 12) | C3 blocking $(RATE_BLOCKING=1 rate 1920 1080 8 40) | C2 $(rate 1280 720 4 200) | 1080p x1 $(rate 1920 1080 1 300) | 4K x16 $(rate 3840 2160 16 12)"
done; done
