#!/bin/bash
# round 5, call b: the pooled tracing launches (JPT_TRACE_REGROUP=2) -- parity subset, rates against the default launches on one box
# (pool sizes, prefetch thresholds), lane counters, SQ counters; the finish-misses variants (rates + fabric traffic)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05b
mkdir -p $O
L=$PWD/gdpathtracing_amd
JPT_TRACE_REGROUP=2 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_fuzz.py tests/test_gpu_full.py -m gpu -x -q -k "c1_cornell or demo_scene_multi_frame or coincident or tie_between or duplicated or native_tree or c3_full_size or other_baseline" > $O/pool_parity.log 2>&1; echo "pool parity rc $?"; tail -3 $O/pool_parity.log
JPT_TRACE_REGROUP=2 bash tools/counters.sh pool:- 2>&1 | grep -v amdgpu.ids > $O/counters_pool.txt; cat $O/counters_pool.txt
bash tools/counters.sh base:- 2>&1 | grep -v amdgpu.ids > $O/counters_base.txt; cat $O/counters_base.txt
rate() { echo -n "$1: "; shift; env "$@" python tools/rate.py 1920 1080 8 100 2>&1 | grep -o "[0-9.]* us/step"; }
ratec() { echo -n "$1 closeup: "; shift; env "$@" RATE_CLOSEUP=1 python tools/rate.py 1920 1080 8 40 2>&1 | grep -o "[0-9.]* us/step"; }
for rep in 1 2; do
  rate base X=1; ratec base X=1
  for mp in 1 32 48 65; do rate "pool192 mp$mp" JPT_TRACE_REGROUP=2 JPT_POOL_MIN_PREFETCH=$mp; ratec "pool192 mp$mp" JPT_TRACE_REGROUP=2 JPT_POOL_MIN_PREFETCH=$mp; done
  for P in 128 256; do rate "pool$P" JPT_TRACE_REGROUP=2 JPT_LIB=$L/libjpt_pool$P.so; ratec "pool$P" JPT_TRACE_REGROUP=2 JPT_LIB=$L/libjpt_pool$P.so; done
  rate "pool192 s8" JPT_TRACE_REGROUP=2 JPT_LIB=$L/libjpt_pool_s8.so; ratec "pool192 s8" JPT_TRACE_REGROUP=2 JPT_LIB=$L/libjpt_pool_s8.so
  for wv in 2 4 8; do rate "pool192 waves$wv" JPT_TRACE_REGROUP=2 JPT_RG_WAVES=$wv; ratec "pool192 waves$wv" JPT_TRACE_REGROUP=2 JPT_RG_WAVES=$wv; done
  rate fm1 JPT_LIB=$L/libjpt_fm1.so; ratec fm1 JPT_LIB=$L/libjpt_fm1.so
  rate fm2 JPT_LIB=$L/libjpt_fm2.so; ratec fm2 JPT_LIB=$L/libjpt_fm2.so
done > $O/rates.txt 2>&1
cat $O/rates.txt
echo -n "base blocking: "; RATE_BLOCKING=1 python tools/rate.py 1920 1080 8 40 2>&1 | grep -o "[0-9.]* us/step"
echo -n "pool blocking: "; JPT_TRACE_REGROUP=2 RATE_BLOCKING=1 python tools/rate.py 1920 1080 8 40 2>&1 | grep -o "[0-9.]* us/step"
# SQ counters of the pooled launches, serial (diag.sh) -> summary; then the kernel trace
JPT_TRACE_REGROUP=2 DIAG_OUT=gpurun_out/diag_pool bash tools/diag.sh > $O/diag_pool.log 2>&1; cp gpurun_out/diag_pool/sq.json $O/pool_sq.json 2>/dev/null
JPT_TRACE_REGROUP=2 DIAG_OUT=gpurun_out/diag_pool_closeup bash tools/diag.sh --camera closeup > $O/diag_pool_closeup.log 2>&1; cp gpurun_out/diag_pool_closeup/sq.json $O/pool_closeup_sq.json 2>/dev/null
python3 - <<'PY'
import json
for f in ("gpurun_out/r05b/pool_sq.json", "gpurun_out/r05b/pool_closeup_sq.json"):
    try:
        d = json.load(open(f))
        for k, v in d.items():
            if "trace" in k: print(f, k, {x: v.get(x) for x in ("valu_issue_frac", "lane_utilisation", "wait_frac", "us", "insts_valu", "insts_salu", "waves_per_simd")})
    except Exception as e: print(f, e)
PY
# fabric traffic of the finish-misses variants (pmc.sh with JPT_LIB)
for v in fm1 fm2; do
  JPT_LIB=$L/libjpt_$v.so bash tools/pmc.sh > $O/pmc_$v.log 2>&1
  python3 tools/summarize_prof.py gpurun_out/prof $O $v > $O/${v}_summary.txt 2>&1
done
bash tools/pmc.sh > $O/pmc_base.log 2>&1; python3 tools/summarize_prof.py gpurun_out/prof $O base > $O/base_summary.txt 2>&1
python3 - <<'PY'
import json
for v in ("base", "fm1", "fm2"):
    try:
        d = json.load(open("gpurun_out/r05b/%s_pmc.json" % v)); tot = 0
        for k, x in d.items():
            if k.startswith("_"): continue
            mb = x["hbm_bytes_per_launch"] / 1e6; n = (x.get("time") or {}).get("calls", 0) / 7.0
            print(v, k, "%.1f MB/launch" % mb, "avg_us", (x.get("time") or {}).get("avg_us"))
    except Exception as e: print(v, e)
PY
bash tools/fetch_calib.sh > $O/fetch_calib.log 2>&1; cp gpurun_out/fetch_calib/summary.* $O/ 2>/dev/null; cat $O/summary.txt
