#!/usr/bin/env bash
# round-6 session 3: the driver's suite with the regenerated oracle cache (v3 caption stand-in), smoke, the driver's bench command and the
# default line (GC settle, one-upload record packing, decode plans in multiples of 32), GPU-vs-oracle scan with the exchange classification
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s3
mkdir -p "$OUT"
rm -f gpurun_out/parity_counters.jsonl
t0=$(date +%s)
echo "=== 1. GPU suite"
( timeout 1300 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider --durations=12 > "$OUT/pytest.log" 2>&1; echo "exit $?" >> "$OUT/pytest.log" )
echo "($(( $(date +%s) - t0 )) s)"; grep -v "Warning\|warnings.warn\|^$\|_create_method\|amdgpu.ids" "$OUT/pytest.log" | tail -24 | cut -c1-500
cp gpurun_out/parity_counters.jsonl "$OUT/parity_counters.jsonl" 2>/dev/null
ls gpurun_out/oracle_cache_misses 2>/dev/null
echo "=== 2. smoke"
( timeout 300 python3 -c 'import __graft_entry__ as e; e.smoke()' > "$OUT/smoke.txt" 2>&1; echo "rc=$?" >> "$OUT/smoke.txt" )
grep "rc=\|smoke OK" "$OUT/smoke.txt" | cut -c1-200
echo "=== 3. bench: driver command, then default"
show() {
python3 - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], d["steps"], "TF/s", r["achieved"], r["frac"], "gemm", r["gemm_ms_per_step"], "sum", r["profiled_step_ms"], "non-gemm", r["non_gemm_share"])
    print("   wall", d["config"].get("step_wall_ms"))
    print("   gc", d["config"].get("python_gc"), "scan", (d["config"].get("parity_scan") or {}).get("source"))
    print("   cpu_baseline", (d.get("cpu_baseline") or {}).get("value"))
    print("   ", {k: (v.get("value"), v.get("ms_per_step")) for k, v in (d.get("extra") or {}).items() if isinstance(v, dict)})
except Exception as e:
    print(sys.argv[1], "no bench line", e)
PY
}
( OMNI_BENCH_WATCHDOG=400 timeout 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_cmd.json" 2> "$OUT/bench_driver_cmd.err"; echo "exit $?" >> "$OUT/bench_driver_cmd.err" )
show "$OUT/bench_driver_cmd.json"
( OMNI_BENCH_WATCHDOG=400 timeout 900 python3 bench.py --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "exit $?" >> "$OUT/bench.err" )
show "$OUT/bench.json"
echo "=== 4. GPU vs oracle scan"
( timeout 500 python tools/scan_gpu_vs_oracle.py device > "$OUT/scan_gpu_vs_oracle.json" 2> "$OUT/scan.err"; echo "exit $?" )
python3 -c "
import json,sys
d=json.load(open('$OUT/scan_gpu_vs_oracle.json')); print({k:v for k,v in d.items() if k not in ('frames_not_identical','definition')}); print([ (r['seed'], r.get('identical_as_sets'), r['boxes_identical']) for r in d['frames_not_identical']])"
echo "total $(( $(date +%s) - t0 )) s"
