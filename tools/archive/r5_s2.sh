#!/usr/bin/env bash
# round-5 session 2: (1) the two stream-parity tests + the NMS kernel checks with the restructured single-workgroup NMS, (2) measured
# identity of the device's final boxes / elements with the oracle's over seeds 0..109 (oracle finals computed in the CPU container),
# (3) detector per-op table at batch 1 (NMS v2), (4) products-per-MAC ablation of the LDS-DMA GEMM, (5) per-layer caption profile
# with the 256x256 and the 256x128 GEMM tile, (6) rocprofv3 kernel trace of the bench command
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r5_s2
mkdir -p "$OUT"
echo "=== 1. stream parity + NMS checks"
( timeout 600 python3 -m pytest tests/test_gpu_k_stream_parity.py tests/test_gpu_a_kernels.py -x -q -m gpu -p no:cacheprovider -k "stream or nms or decode or pool or bad_pointers" > "$OUT/pytest_subset.log" 2>&1; echo "exit $?" >> "$OUT/pytest_subset.log" )
grep -v "Warning\|warnings.warn\|^$\|_create_method\|amdgpu.ids" "$OUT/pytest_subset.log" | tail -12 | cut -c1-600
echo "=== 2. GPU vs oracle over seeds 0..109"
( timeout 400 python tools/scan_gpu_vs_oracle.py device > "$OUT/scan_gpu_vs_oracle.json" 2> "$OUT/scan.err"; echo "exit $?" )
tail -c 1500 "$OUT/scan_gpu_vs_oracle.json"; echo
echo "=== 3. detector per-op, batch 1 / 8"
for cfg in "1 640" "8 640"; do
  set -- $cfg
  ( BATCH=$1 IMGSZ=$2 timeout 200 python tools/profile_plan.py > "$OUT/detector_per_op_b$1_$2.txt" 2>&1; echo "b$1 $2 exit $?" )
  grep -v Warning "$OUT/detector_per_op_b$1_$2.txt" | grep "precision\|non-conv" | cut -c1-300
done
echo "=== 4. products-per-MAC ablation"
( timeout 1500 python tools/gemm_products_ablation.py > "$OUT/products_ablation.jsonl" 2> "$OUT/products.err"; echo "exit $?" )
cut -c1-400 "$OUT/products_ablation.jsonl"
echo "=== 5. caption per-op profile, 256x256 vs 256x128 GEMM tile"
( timeout 300 python tools/caption_profile.py 128 768 2 > "$OUT/caption_per_op.json" 2> "$OUT/caption_per_op.txt"; echo "exit $?" )
grep -v Warning "$OUT/caption_per_op.txt" | head -30 | cut -c1-200
( OMNI_GEMM_TILE=256x128 timeout 300 python tools/caption_profile.py 128 768 2 > "$OUT/caption_per_op_tile256x128.json" 2> "$OUT/caption_per_op_tile256x128.txt"; echo "exit $?" )
grep -v Warning "$OUT/caption_per_op_tile256x128.txt" | head -16 | cut -c1-200
echo "=== 6. kernel trace of the bench command"
( timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extra > "$OUT/stats_bench.json" 2> "$OUT/stats.err"; echo "exit $?" )
f=$(find "$OUT/stats" -name "*kernel_stats.csv" | head -1); echo "$f"; head -16 "$f" | cut -c1-220
cp "$f" "$OUT/kernel_stats.csv" 2>/dev/null
find "$OUT/stats" -name "*.csv" -size +4M -delete; find "$OUT/stats" -name "*.db" -delete
ls -la "$OUT" | head -30
