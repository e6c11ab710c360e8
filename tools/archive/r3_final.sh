#!/usr/bin/env bash
# Round-3 closing GPU session at the final commit: (1) the -m gpu suite exactly as the driver runs it (one process, -x), (2) smoke,
# (3) the default bench line (what the driver records, extras and CPU baseline included), (4) rocprofv3 kernel trace of the bench
# command, (5) the two PMC passes (FETCH_SIZE, WRITE_SIZE — separate runs, kernel trace only) over one 128-crop caption encode + decode.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r3final
mkdir -p "$OUT"
python tools/make_weights.py --ensure detector > /dev/null 2>&1
python tools/make_weights.py --ensure caption > /dev/null 2>&1
echo "=== 1. pytest tests/ -x -q -m gpu (one process)"
t0=$(date +%s)
( timeout 1200 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider --durations=10 > "$OUT/pytest.log" 2>&1; echo "exit $?" >> "$OUT/pytest.log" )
echo "($(( $(date +%s) - t0 )) s)"; grep -v "Warning\|warnings.warn\|^$\|_create_method\|amdgpu.ids" "$OUT/pytest.log" | tail -22 | cut -c1-300
echo "=== 2. smoke"
( timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | cut -c1-400 )
echo "=== 3. default bench line"
t0=$(date +%s)
( OMNI_BENCH_WATCHDOG=200 timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "exit $?" >> "$OUT/bench.err" )
echo "($(( $(date +%s) - t0 )) s)"; grep -v "^  File\|^Thread" "$OUT/bench.err" | tail -4 | cut -c1-300
python - "$OUT/bench.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(d["value"], d["ms_per_step"], d["config"].get("mean_crops_per_screenshot"), r["achieved"], r["frac"], r["gemm_ms_per_step"], r["profiled_step_ms"], r["non_gemm_share"])
    print(r["kernel_family_ms_per_step"])
    print("cpu_baseline", d.get("cpu_baseline"))
    print("extra", json.dumps(d.get("extra"))[:1500])
except Exception as e:
    print("no bench line", e)
PY
echo "=== 4. kernel trace of the bench command"
( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extra > "$OUT/stats_bench.json" 2> "$OUT/stats.err"; echo "exit $?" )
f=$(find "$OUT/stats" -name "*kernel_stats.csv" | head -1); echo "$f"; head -16 "$f" | cut -c1-200
cp "$f" "$OUT/kernel_stats.csv" 2>/dev/null
find "$OUT/stats" -name "*.csv" -size +4M -delete; find "$OUT/stats" -name "*.db" -delete
echo "=== 5. PMC passes over one 128-crop caption plan (2 encode passes + 21 decode steps per process)"
for c in FETCH_SIZE WRITE_SIZE; do
  ( timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/pmc_$c" -- python tools/caption_profile.py 128 768 1 > "$OUT/pmc_$c.json" 2> "$OUT/pmc_$c.err"; echo "$c exit $?" )
  python tools/pmc_summary.py "$OUT/pmc_$c" > "$OUT/pmc_summary_$c.json" 2>/dev/null
  find "$OUT/pmc_$c" -name "*.csv" -size +4M -delete; find "$OUT/pmc_$c" -name "*.db" -delete
done
ls -la "$OUT" | head -30
