#!/usr/bin/env bash
# round-5 closing session (second, at the final commit): the driver's two commands (suite, smoke), the driver's bench command, the default bench line
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r5_s10
mkdir -p "$OUT"
t0=$(date +%s)
( timeout 1400 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider --durations=10 > "$OUT/pytest.log" 2>&1; echo "exit $?" >> "$OUT/pytest.log" )
echo "($(( $(date +%s) - t0 )) s)"; grep -v "Warning\|warnings.warn\|^$\|_create_method\|amdgpu.ids" "$OUT/pytest.log" | tail -16 | cut -c1-300
ls gpurun_out/oracle_cache_misses 2>/dev/null
( timeout 300 python3 -c 'import __graft_entry__ as e; e.smoke()' > "$OUT/smoke.txt" 2>&1; echo "rc=$?" >> "$OUT/smoke.txt" )
grep "rc=\|smoke OK" "$OUT/smoke.txt" | cut -c1-200
( OMNI_BENCH_WATCHDOG=400 timeout 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_cmd.json" 2> "$OUT/bench_driver_cmd.err"; echo "exit $?" >> "$OUT/bench_driver_cmd.err" )
( OMNI_BENCH_WATCHDOG=400 timeout 1200 python3 bench.py --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "exit $?" >> "$OUT/bench.err" )
for f in bench_driver_cmd bench; do
python - "$OUT/$f.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(sys.argv[1], d["value"], d["ms_per_step"], d["steps"], r["achieved"], r["frac"], r["gemm_ms_per_step"], r["profiled_step_ms"], r["non_gemm_share"])
    print("  cpu_baseline", (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("seconds_per_screenshot"))
    print("  ", {k: (v.get("value"), v.get("ms_per_step")) for k, v in d["extra"].items() if isinstance(v, dict)})
except Exception as e:
    print("no bench line", e)
PY
done
