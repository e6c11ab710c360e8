#!/usr/bin/env bash
# round-5 session 4: the -m gpu suite as the driver runs it + smoke at the candidate commit (session 3's suite stopped at a test this
# session's commit repaired), then the default bench line
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r5_s4
mkdir -p "$OUT"
t0=$(date +%s)
( timeout 1400 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider --durations=12 > "$OUT/pytest.log" 2>&1; echo "exit $?" >> "$OUT/pytest.log" )
echo "($(( $(date +%s) - t0 )) s)"; grep -v "Warning\|warnings.warn\|^$\|_create_method\|amdgpu.ids" "$OUT/pytest.log" | tail -22 | cut -c1-300
ls gpurun_out/oracle_cache_misses 2>/dev/null
( timeout 300 python3 -c 'import __graft_entry__ as e; e.smoke()' > "$OUT/smoke.txt" 2>&1; echo "rc=$?" >> "$OUT/smoke.txt" )
grep "rc=\|smoke OK" "$OUT/smoke.txt" | cut -c1-200
( OMNI_BENCH_WATCHDOG=400 timeout 1200 python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "exit $?" >> "$OUT/bench.err" )
python - "$OUT/bench.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(d["value"], d["ms_per_step"], r["achieved"], r["frac"], r["gemm_ms_per_step"], r["profiled_step_ms"], r["non_gemm_share"])
    print("cpu_baseline", d.get("cpu_baseline"))
    print({k: (v.get("value"), v.get("ms_per_step")) for k, v in d["extra"].items() if isinstance(v, dict)})
except Exception as e:
    print("no bench line", e)
PY
