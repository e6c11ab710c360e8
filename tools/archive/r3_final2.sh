#!/usr/bin/env bash
# Round-3 closing GPU session, second part: the one-process -m gpu suite after the NaN-row fix (the first closing run aborted in
# test_gpu_g: padding rows of the merged decode plan held recycled NaN memory -> all-NaN logits -> token id 0x7fffffff -> out-of-range
# embedding gather), then the kernel trace of the SYNCHRONOUS bench (--no-pipeline: one stream, kernel durations do not overlap) for the
# agreement check against bench.py's event timings.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r3final2
mkdir -p "$OUT"
python tools/make_weights.py --ensure detector > /dev/null 2>&1
python tools/make_weights.py --ensure caption > /dev/null 2>&1
echo "=== 1. pytest tests/ -x -q -m gpu (one process)"
t0=$(date +%s)
( timeout 900 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider --durations=8 > "$OUT/pytest.log" 2>&1; echo "exit $?" >> "$OUT/pytest.log" )
echo "($(( $(date +%s) - t0 )) s)"; grep -v "Warning\|warnings.warn\|^$\|_create_method\|amdgpu.ids\|^  File" "$OUT/pytest.log" | tail -16 | cut -c1-300
echo "=== 2. kernel trace of the synchronous bench"
( timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python bench.py --no-pipeline --steps 2 --warmup 1 --no-cpu-baseline --no-extra > "$OUT/stats_bench.json" 2> "$OUT/stats.err"; echo "exit $?" )
f=$(find "$OUT/stats" -name "*kernel_stats.csv" | head -1); echo "$f"; head -8 "$f" | cut -c1-200
cp "$f" "$OUT/kernel_stats_sync.csv" 2>/dev/null
find "$OUT/stats" -name "*.csv" -size +4M -delete; find "$OUT/stats" -name "*.db" -delete
