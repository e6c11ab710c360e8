#!/usr/bin/env bash
# Round-3 GPU session 4: rocprofv3 kernel trace of the bench command (per-kernel durations), 128x128 tiles on the short-K GEMM shapes,
# the model-level C entry points on hardware.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r3s4
mkdir -p "$OUT"
python tools/make_weights.py --ensure detector > /dev/null 2>&1
python tools/make_weights.py --ensure caption > /dev/null 2>&1
echo "=== 1. kernel trace of the bench command"
( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > "$OUT/stats_bench.json" 2> "$OUT/stats.err"; echo "exit $?" )
f=$(find "$OUT/stats" -name "*kernel_stats.csv" | head -1); echo "$f"; head -40 "$f" | cut -c1-230
cp "$f" "$OUT/kernel_stats.csv" 2>/dev/null
find "$OUT/stats" -name "*.csv" -size +4M -delete; find "$OUT/stats" -name "*.db" -delete
echo "=== 2. short-K GEMM shapes: 128x128 tiles (two blocks per CU) vs the default"
( VARIANTS="dma,dma:128x128" SHAPES="s0.fc1,s0.fc2,s0.qkv,s1.fc1,s1.qkv,s2.proj" timeout 300 python tools/gemm_bench.py > "$OUT/gemm_bench_short_k.txt" 2>&1; echo "exit $?" )
grep -v "Warn\|warn\|amdgpu.ids" "$OUT/gemm_bench_short_k.txt" | cut -c1-140
echo "=== 3. model-level C entry points"
( timeout 300 python -m pytest tests/test_gpu_i_model_capi.py -q -m gpu -p no:cacheprovider -x -s > "$OUT/capi.log" 2>&1; echo "exit $?" >> "$OUT/capi.log" )
grep -v "Warning\|warnings.warn\|^$\|_create_method\|amdgpu.ids" "$OUT/capi.log" | tail -6 | cut -c1-600
