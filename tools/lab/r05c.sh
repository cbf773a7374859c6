#!/bin/bash
# round 5, call c: one launch per bounce (wf2_bounce, JPT_FUSE_BOUNCE) against the separate shade / trace launches, same box:
# a GPU's share of C3 under 2 / 4 / 8-way partitions, one 1-spp frame, C2's size, C3 itself -- queued and blocking
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c
mkdir -p $O
JPT_FUSE_BOUNCE=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_fuzz.py tests/test_gpu_full.py -m gpu -x -q -k "c1_cornell or demo_scene_multi_frame or coincident or tie_between or duplicated or native_tree or c3_full_size or other_baseline or texture_scene or set_aside" > $O/fuse_parity.log 2>&1; echo "fuse parity rc $?"; tail -3 $O/fuse_parity.log
r() { echo -n "$1: "; shift; env "$@" 2>&1 | grep -o "[0-9.]* us/step"; }
for rep in 1 2; do
for f in 0 1; do
  r "fuse$f C3/8 queued" JPT_FUSE_BOUNCE=$f python tools/rate.py 1920 1080 8 200 8
  r "fuse$f C3/4 queued" JPT_FUSE_BOUNCE=$f python tools/rate.py 1920 1080 8 200 4
  r "fuse$f C3/2 queued" JPT_FUSE_BOUNCE=$f python tools/rate.py 1920 1080 8 100 2
  r "fuse$f C3 queued" JPT_FUSE_BOUNCE=$f python tools/rate.py 1920 1080 8 100
  r "fuse$f C3 blocking" JPT_FUSE_BOUNCE=$f RATE_BLOCKING=1 python tools/rate.py 1920 1080 8 40
  r "fuse$f C3/8 blocking" JPT_FUSE_BOUNCE=$f RATE_BLOCKING=1 python tools/rate.py 1920 1080 8 100 8
  r "fuse$f 1080p x1 queued" JPT_FUSE_BOUNCE=$f python tools/rate.py 1920 1080 1 200
  r "fuse$f 1080p x1 blocking" JPT_FUSE_BOUNCE=$f RATE_BLOCKING=1 python tools/rate.py 1920 1080 1 100
  r "fuse$f 720p x4 queued" JPT_FUSE_BOUNCE=$f python tools/rate.py 1280 720 4 200
  r "fuse$f 720p x4 blocking" JPT_FUSE_BOUNCE=$f RATE_BLOCKING=1 python tools/rate.py 1280 720 4 100
  r "fuse$f closeup queued" JPT_FUSE_BOUNCE=$f RATE_CLOSEUP=1 python tools/rate.py 1920 1080 8 40
  r "fuse$f closeup/8 queued" JPT_FUSE_BOUNCE=$f RATE_CLOSEUP=1 python tools/rate.py 1920 1080 8 100 8
  r "fuse$f 4K x16 /8 queued" JPT_FUSE_BOUNCE=$f python tools/rate.py 3840 2160 16 40 8
done
done > $O/fuse_rates.txt 2>&1
cat $O/fuse_rates.txt
# ---- the pooled launches again, every turn asking for exactly four loads (does the prefetch overlap now?) ----
JPT_TRACE_REGROUP=2 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_fuzz.py tests/test_gpu_full.py -m gpu -x -q -k "c1_cornell or demo_scene_multi_frame or coincident or tie_between or duplicated or native_tree or c3_full_size" > $O/pool_parity.log 2>&1; echo "pool parity rc $?"; tail -2 $O/pool_parity.log
for rep in 1 2; do
  r "base C3" python tools/rate.py 1920 1080 8 100
  r "base closeup" env RATE_CLOSEUP=1 python tools/rate.py 1920 1080 8 40
  for mp in 1 32 48 65; do
    r "pool mp$mp C3" JPT_TRACE_REGROUP=2 JPT_POOL_MIN_PREFETCH=$mp python tools/rate.py 1920 1080 8 100
    r "pool mp$mp closeup" JPT_TRACE_REGROUP=2 JPT_POOL_MIN_PREFETCH=$mp RATE_CLOSEUP=1 python tools/rate.py 1920 1080 8 40
  done
done > $O/pool_rates.txt 2>&1
cat $O/pool_rates.txt
JPT_TRACE_REGROUP=2 DIAG_OUT=gpurun_out/diag_pool bash tools/diag.sh > $O/diag_pool.log 2>&1; cp gpurun_out/diag_pool/sq.json $O/pool_sq.json 2>/dev/null
JPT_TRACE_REGROUP=2 DIAG_OUT=gpurun_out/diag_pool_closeup bash tools/diag.sh --camera closeup > $O/diag_pool_closeup.log 2>&1; cp gpurun_out/diag_pool_closeup/sq.json $O/pool_closeup_sq.json 2>/dev/null
DIAG_OUT=gpurun_out/diag_base_closeup bash tools/diag.sh --camera closeup > $O/diag_base_closeup.log 2>&1; cp gpurun_out/diag_base_closeup/sq.json $O/base_closeup_sq.json 2>/dev/null
python3 - <<'PY'
import json
for f in ("pool_sq", "pool_closeup_sq", "base_closeup_sq"):
    try:
        d = json.load(open("gpurun_out/r05c/%s.json" % f))
        for k, v in d.items():
            if "trace" in k: print(f, k, json.dumps(v)[:1500])
    except Exception as e: print(f, e)
PY
# the siblings-side-by-side node order on the scene that leaves the Infinity Cache (every miss fills a 128-byte line: fetch_calib)
for no in 0 1; do echo -n "unique4m node_order $no: "; JPT_NODE_ORDER=$no RATE_SCENE=unique RATE_TRIS=4000000 RATE_CLOSEUP=1 python tools/rate.py 1920 1080 8 12 2>&1 | grep -o "[0-9.]* us/step"; done | tee $O/node_order_unique4m.txt
rm -rf gpurun_out/diag* gpurun_out/prof gpurun_out/fetch_calib gpurun_out/configs gpurun_out/round
du -sh gpurun_out
