#!/usr/bin/env bash
# Round-2 closing GPU session: default bench line first (what the driver records), whole GPU suite, smoke, opt-in path test + A/B, kernel stats.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r2final
mkdir -p "$OUT"
python tools/make_weights.py --ensure detector > /dev/null 2>&1
python tools/make_weights.py --ensure caption > /dev/null 2>&1
echo "=== default bench line"
( OMNI_BENCH_WATCHDOG=120 timeout 400 python bench.py --steps 5 --warmup 2 > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "exit $?" >> "$OUT/bench.err" )
grep -v "^  File\|^Thread" "$OUT/bench.err" | tail -8 | cut -c1-300
echo "=== pytest -m gpu"
( timeout 800 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 > "$OUT/pytest.log" 2>&1; echo "exit $?" >> "$OUT/pytest.log" )
grep -v "Warning\|warnings.warn\|^$\|_create_method" "$OUT/pytest.log" | tail -40 | cut -c1-300
echo "=== smoke"
( timeout 150 python __graft_entry__.py --smoke 2>&1 | tail -3 | cut -c1-600 )
echo "=== opt-in: attention kernels write format B"
( OMNI_ATTN_SPLIT_OUT=1 timeout 200 python -m pytest tests/test_gpu_b_caption_model.py -q -p no:cacheprovider -k "r64 or r768" 2>&1 | tail -3 | cut -c1-300 )
for v in "OMNI_ATTN_SPLIT_OUT=1"; do
  tag=$(echo "$v" | tr ' =' '__')
  ( env $v OMNI_BENCH_WATCHDOG=60 timeout 100 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > "$OUT/ab_$tag.json" 2> "$OUT/ab_$tag.err"; echo "$v -> exit $?" )
  python - "$OUT/ab_$tag.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("   ", d["value"], "screenshots/s", d["ms_per_step"], "ms/step; gemm", d["roofline"].get("gemm_ms_per_step"), "achieved", d["roofline"].get("achieved"))
except Exception as e:
    print("    no line:", e)
PY
done
echo "=== kernel stats"
( timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > "$OUT/stats_bench.json" 2> "$OUT/stats.err"; echo "exit $?" )
find "$OUT" -name "*.csv" -size +6M -delete
ls "$OUT"
