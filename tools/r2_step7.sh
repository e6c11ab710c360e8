#!/usr/bin/env bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r2s7
mkdir -p "$OUT"
python tools/make_weights.py --ensure detector > /dev/null 2>&1
python tools/make_weights.py --ensure caption > /dev/null 2>&1
echo "=== device glue, 3 steps"
( OMNI_DEVICE_GLUE=1 OMNI_BENCH_WATCHDOG=40 timeout 100 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra 2>&1 | grep -v "^  File\|^Thread\|amdgpu.ids" | cut -c1-300 | tail -6 )
echo "=== default bench line"
( OMNI_BENCH_WATCHDOG=90 timeout 420 python bench.py --steps 5 --warmup 2 > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "exit $?" >> "$OUT/bench.err" )
grep -v "^  File\|^Thread" "$OUT/bench.err" | tail -12 | cut -c1-300
echo "=== kernel stats"
( timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > "$OUT/stats_bench.json" 2> "$OUT/stats.err"; echo "exit $?" )
find "$OUT" -name "*.csv" -size +6M -delete
ls "$OUT" "$OUT"/stats/* | head -20
