#!/usr/bin/env bash
# round-4 closing session at the commit that becomes HEAD: (1) the -m gpu suite exactly as the driver runs it (one process, -x),
# (2) smoke, (3) the default bench line (what the driver records: extras + CPU baseline), (4) the same command at the driver's K = 20
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r4_final
mkdir -p "$OUT"
git rev-parse HEAD 2>/dev/null | head -1
echo "=== 1. pytest tests/ -x -q -m gpu -p no:cacheprovider (one process)"
t0=$(date +%s)
( timeout 1500 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider --durations=8 > "$OUT/pytest.log" 2>&1; echo "exit $?" >> "$OUT/pytest.log" )
echo "($(( $(date +%s) - t0 )) s)"; grep -v "Warning\|warnings.warn\|^$\|_create_method\|amdgpu.ids" "$OUT/pytest.log" | tail -14 | cut -c1-200
echo "=== 2. smoke"
( timeout 300 python3 -c 'import __graft_entry__ as e; e.smoke()' > "$OUT/smoke.txt" 2>&1; echo "rc=$?" >> "$OUT/smoke.txt" )
grep "rc=\|smoke OK" "$OUT/smoke.txt" | cut -c1-160
echo "=== 3. default bench line"
t0=$(date +%s)
( OMNI_BENCH_WATCHDOG=200 timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "exit $?" >> "$OUT/bench.err" )
echo "($(( $(date +%s) - t0 )) s)"; grep -v "^  File\|^Thread\|Warning" "$OUT/bench.err" | tail -3 | cut -c1-200
python - "$OUT/bench.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(d["value"], d["ms_per_step"], r["achieved"], r["frac"], r["gemm_ms_per_step"], r["profiled_step_ms"], r["non_gemm_share"], d["config"]["hbm_peak_allocated_gb"])
    print("cpu_baseline", json.dumps(d.get("cpu_baseline"))[:600])
except Exception as e:
    print("no bench line", e)
PY
echo "=== 4. K = 20 (the driver's step count)"
( OMNI_BENCH_WATCHDOG=200 timeout 600 python bench.py --steps 20 --warmup 2 --no-extra --no-cpu-baseline > "$OUT/bench_k20.json" 2> "$OUT/bench_k20.err"; echo "exit $?" )
python - "$OUT/bench_k20.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["steps"])
except Exception as e:
    print("no bench line", e)
PY
