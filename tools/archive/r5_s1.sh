#!/usr/bin/env bash
# round-5 session 1: the new stand-in (v5), the single-workgroup NMS, the one-launch CBFuse and the oracle cache meet the hardware:
# (1) the -m gpu suite exactly as the driver runs it, (2) smoke, (3) the default bench line, (4) detector per-op tables
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r5_s1
mkdir -p "$OUT"
echo "=== 1. pytest tests/ -x -q -m gpu (one process)"
t0=$(date +%s)
( timeout 1400 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider --durations=15 > "$OUT/pytest.log" 2>&1; echo "exit $?" >> "$OUT/pytest.log" )
echo "($(( $(date +%s) - t0 )) s)"; grep -v "Warning\|warnings.warn\|^$\|_create_method\|amdgpu.ids" "$OUT/pytest.log" | tail -40 | cut -c1-400
ls gpurun_out/oracle_cache_misses 2>/dev/null
echo "=== 2. smoke"
( timeout 300 python3 -c 'import __graft_entry__ as e; e.smoke()' > "$OUT/smoke.txt" 2>&1; echo "rc=$?" >> "$OUT/smoke.txt" )
grep "rc=\|smoke OK" "$OUT/smoke.txt" | cut -c1-200
echo "=== 3. default bench line"
t0=$(date +%s)
( OMNI_BENCH_WATCHDOG=300 timeout 1000 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "exit $?" >> "$OUT/bench.err" )
echo "($(( $(date +%s) - t0 )) s)"; grep -v "^  File\|^Thread\|Warning" "$OUT/bench.err" | tail -8 | cut -c1-300
python - "$OUT/bench.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(d["value"], d["ms_per_step"], d["dtype"], d["config"].get("mean_crops_per_screenshot"), r["achieved"], r["frac"], r["gemm_ms_per_step"], r["profiled_step_ms"], r["non_gemm_share"])
    print(r["kernel_family_ms_per_step"])
    for row in r.get("per_kernel", []): print("  ", row)
    print("cpu_baseline", d.get("cpu_baseline"))
    print("extra", json.dumps(d.get("extra"))[:3000])
except Exception as e:
    print("no bench line", e)
PY
echo "=== 4. detector per-op tables"
for cfg in "1 640" "8 640" "1 native"; do
  set -- $cfg
  ( BATCH=$1 IMGSZ=$2 timeout 200 python tools/profile_plan.py > "$OUT/detector_per_op_b$1_$2.txt" 2>&1; echo "b$1 $2 exit $?" )
  grep -v Warning "$OUT/detector_per_op_b$1_$2.txt" | grep "precision\|non-conv" | cut -c1-300
done
ls -la "$OUT" | head -30
