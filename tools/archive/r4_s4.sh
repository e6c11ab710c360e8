#!/usr/bin/env bash
# round-4 session 4 (dress rehearsal of the closing session at the adopted composition): (1) the -m gpu suite exactly as the driver runs
# it, (2) smoke, (3) the default bench line (extras + CPU baseline), (4) configs[0] on the reference's demo image: MI355X facade vs ONE
# full pass of the reference-equivalent CPU pipeline, (5) rocprofv3 kernel trace of the bench command
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r4_s4
mkdir -p "$OUT"
echo "=== 1. pytest tests/ -x -q -m gpu (one process)"
t0=$(date +%s)
( timeout 1500 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider --durations=12 > "$OUT/pytest.log" 2>&1; echo "exit $?" >> "$OUT/pytest.log" )
echo "($(( $(date +%s) - t0 )) s)"; grep -v "Warning\|warnings.warn\|^$\|_create_method\|amdgpu.ids" "$OUT/pytest.log" | tail -24 | cut -c1-300
echo "=== 2. smoke"
( timeout 300 python3 -c 'import __graft_entry__ as e; e.smoke()' > "$OUT/smoke.txt" 2>&1; echo "rc=$?" >> "$OUT/smoke.txt" )
grep "rc=\|smoke OK" "$OUT/smoke.txt" | cut -c1-160
echo "=== 3. default bench line"
t0=$(date +%s)
( OMNI_BENCH_WATCHDOG=200 timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "exit $?" >> "$OUT/bench.err" )
echo "($(( $(date +%s) - t0 )) s)"; grep -v "^  File\|^Thread\|Warning" "$OUT/bench.err" | tail -6 | cut -c1-300
python - "$OUT/bench.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(d["value"], d["ms_per_step"], d["config"].get("mean_crops_per_screenshot"), r["achieved"], r["frac"], r["gemm_ms_per_step"], r["profiled_step_ms"], r["non_gemm_share"])
    print(r["kernel_family_ms_per_step"])
    print("cpu_baseline", d.get("cpu_baseline"))
    print("extra", json.dumps(d.get("extra"))[:2500])
except Exception as e:
    print("no bench line", e)
PY
echo "=== 4. configs[0]: demo_image.jpg through the facade on the MI355X and ONE full CPU pass"
( timeout 900 python tools/configs0.py > "$OUT/configs0.json" 2> "$OUT/configs0.err"; echo "exit $?" )
tail -c 1800 "$OUT/configs0.json"; echo; grep -v Warning "$OUT/configs0.err" | tail -3 | cut -c1-300
echo "=== 5. kernel trace of the bench command"
( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extra > "$OUT/stats_bench.json" 2> "$OUT/stats.err"; echo "exit $?" )
f=$(find "$OUT/stats" -name "*kernel_stats.csv" | head -1); echo "$f"; head -20 "$f" | cut -c1-200
cp "$f" "$OUT/kernel_stats.csv" 2>/dev/null
find "$OUT/stats" -name "*.csv" -size +4M -delete; find "$OUT/stats" -name "*.db" -delete
ls -la "$OUT" | head -30
