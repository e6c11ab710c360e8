#!/usr/bin/env bash
# round-6 session 11: the driver's suite at the new defaults (exact-row remainder graphs, row-patch first patch embedding; split-K
# combine opt-in, off), smoke, the driver's bench command
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s11
mkdir -p "$OUT"
rm -f gpurun_out/parity_counters.jsonl
t0=$(date +%s)
echo "=== 1. GPU suite"
( timeout 1300 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider --durations=8 > "$OUT/pytest.log" 2>&1; echo "exit $?" >> "$OUT/pytest.log" )
echo "($(( $(date +%s) - t0 )) s)"; grep -v "Warning\|warnings.warn\|^$\|_create_method\|amdgpu.ids" "$OUT/pytest.log" | tail -16 | cut -c1-500
cp gpurun_out/parity_counters.jsonl "$OUT/parity_counters.jsonl" 2>/dev/null
ls gpurun_out/oracle_cache_misses 2>/dev/null
echo "=== 2. smoke"
( timeout 300 python3 -c 'import __graft_entry__ as e; e.smoke()' > "$OUT/smoke.txt" 2>&1; echo "rc=$?" >> "$OUT/smoke.txt" )
grep "rc=\|smoke OK" "$OUT/smoke.txt" | cut -c1-200
echo "=== 3. bench: driver command"
( OMNI_BENCH_WATCHDOG=400 timeout 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_cmd.json" 2> "$OUT/bench_driver_cmd.err"; echo "exit $?" >> "$OUT/bench_driver_cmd.err" )
python3 - "$OUT/bench_driver_cmd.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(d["value"], d["ms_per_step"], d["steps"], "TF/s", r["achieved"], r["frac"], "gemm", r["gemm_ms_per_step"], "sum", r["profiled_step_ms"], "non-gemm", r["non_gemm_share"], "hbm", d["config"].get("hbm_peak_allocated_gb"))
    print("   wall", d["config"].get("step_wall_ms"))
    print("   scan", (d["config"].get("parity_scan") or {}).get("source"), "cpu_baseline", (d.get("cpu_baseline") or {}).get("value"))
    print("   ", {k: (v.get("value"), v.get("ms_per_step")) for k, v in (d.get("extra") or {}).items() if isinstance(v, dict)})
    for k in r["per_kernel"]: print("   ", k)
except Exception as e:
    print(sys.argv[1], "no bench line", e)
PY
echo "total $(( $(date +%s) - t0 )) s"
