#!/bin/bash
# Bench lines for the other BASELINE configs (not the headline): C2, C4, HDR accumulation, reference-exact tree.
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/configs
n=0
# (the whole JSON line of every config is kept: gpurun_out/configs/NN.json -> profiles/<round>/configs/)
run() { label="$1"; shift; n=$((n+1)); python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-closeup --no-dropin --project-ranks 0 "$@" 2>&1 | grep '^{' | tee gpurun_out/configs/$(printf %02d $n).json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label |', d['value'], 'Mrays/s |', d['ms_per_step'], 'ms/step | rays/step', d['config']['rays_per_step'], '| build_s', d['config']['scene_build_s'])"; }
run "C3 (headline)" 
run "C3 hdr accumulation" --accum hdr
run "C2 1280x720 4spp 3b" --width 1280 --height 720 --spp 4 --bounces 3
run "C4 1024 inst x 1024 tris" --scene inst
run "C5-size on 1 GPU 3840x2160 16spp 6b" --width 3840 --height 2160 --spp 16 --bounces 6
run "C3 scene, close-up camera (every pixel hits)" --camera closeup
run "1 frame per render (interactive use)" --spp 1
run "C3 via reference-layout upload (route i, native tree + reach records from the uploaded boxes)" --route upload
run "C2 via reference-layout upload" --route upload --width 1280 --height 720 --spp 4 --bounces 3
run "C4 via reference-layout upload" --route upload --scene inst
run "C3 reference-layout upload walked as given (audit)" --route upload-given
run "C3 reference-exact tree" --builder exact
run "C3 audit kernel" --kernel ref
run "S-unique: 1 M unique triangles in one BLAS, close-up camera, 1920x1080 8spp 4b" --scene unique
