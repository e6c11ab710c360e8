#!/usr/bin/env bash
# round-6 session 1: (1) the driver's suite at the parity-closure commit (every benched crop compared, margin 2e-4 with the f64 referee,
# range guard, ABI 3, decode plans in multiples of 32), (2) smoke, (3) GPU-vs-oracle scan over seeds 0..109 at HEAD with the tuning
# table on (provenance recorded), (4) the GEMM feed-vs-power experiment (tools/gemm_exp.py), (5) default bench line, (6) caption per-op
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s1
mkdir -p "$OUT"
t0=$(date +%s)
echo "=== 1. GPU suite"
( timeout 1500 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider --durations=12 > "$OUT/pytest.log" 2>&1; echo "exit $?" >> "$OUT/pytest.log" )
echo "($(( $(date +%s) - t0 )) s)"; grep -v "Warning\|warnings.warn\|^$\|_create_method\|amdgpu.ids" "$OUT/pytest.log" | tail -24 | cut -c1-400
cp gpurun_out/parity_counters.jsonl "$OUT/parity_counters.jsonl" 2>/dev/null
ls gpurun_out/oracle_cache_misses 2>/dev/null
echo "=== 2. smoke"
( timeout 300 python3 -c 'import __graft_entry__ as e; e.smoke()' > "$OUT/smoke.txt" 2>&1; echo "rc=$?" >> "$OUT/smoke.txt" )
grep "rc=\|smoke OK" "$OUT/smoke.txt" | cut -c1-300
echo "=== 3. GPU vs oracle scan at HEAD"
( timeout 500 python tools/scan_gpu_vs_oracle.py device > "$OUT/scan_gpu_vs_oracle.json" 2> "$OUT/scan.err"; echo "exit $?" )
tail -c 2500 "$OUT/scan_gpu_vs_oracle.json"; echo
echo "=== 4. GEMM feed-vs-power experiment"
( timeout 900 python tools/gemm_exp.py run 4 > "$OUT/gemm_exp.jsonl" 2> "$OUT/gemm_exp.err"; echo "exit $?" )
python3 - "$OUT/gemm_exp.jsonl" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    try:
        d = json.loads(l)
    except ValueError:
        continue
    if "cases" not in d:
        print(d); continue
    for k, c in d["cases"].items():
        print(d["variant"], k, c["ms_per_launch_events_20"], c["ms_per_launch_sustained"], c["algorithmic_tflops_sustained"], c["power_w_mean"], c["sclk_mhz_mean"])
PY
echo "=== 5. bench (default line, no CPU baseline)"
( OMNI_BENCH_WATCHDOG=400 timeout 900 python3 bench.py --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "exit $?" >> "$OUT/bench.err" )
python3 - "$OUT/bench.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(d["value"], d["ms_per_step"], d["steps"], r["achieved"], r["frac"], r["gemm_ms_per_step"], r["profiled_step_ms"], r["non_gemm_share"])
    print("  step_wall_ms", d["config"].get("step_wall_ms"))
    print("  families", r.get("kernel_family_ms_per_step"))
    print("  ", {k: (v.get("value"), v.get("ms_per_step")) for k, v in d["extra"].items() if isinstance(v, dict)})
except Exception as e:
    print("no bench line", e)
PY
tail -3 "$OUT/bench.err" | cut -c1-300
echo "=== 6. caption per-op"
( timeout 300 python tools/caption_profile.py 128 768 2 > "$OUT/caption_per_op.json" 2> "$OUT/caption_per_op.txt"; echo "exit $?" )
grep -v Warning "$OUT/caption_per_op.txt" | head -44 | cut -c1-200
echo "total $(( $(date +%s) - t0 )) s"
