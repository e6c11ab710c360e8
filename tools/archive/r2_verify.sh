#!/usr/bin/env bash
# Round-2 short verification call: the benched-path parity test (all discrepancies reported), then the quick detector tests.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r2v
mkdir -p "$OUT"
echo "=== bench path parity"
( timeout 390 python -m pytest tests/test_gpu_d_pipeline.py -q -p no:cacheprovider -k bench_path -s > "$OUT/bench_path.log" 2>&1; echo "exit $?" >> "$OUT/bench_path.log" )
grep -v "Warning\|warnings.warn\|^$\|_create_method" "$OUT/bench_path.log" | tail -45 | cut -c1-1200
echo "=== detector tests"
( timeout 140 python -m pytest tests/test_gpu_c_detector.py -q -p no:cacheprovider -k "full_width_boxes_640 or half_width or native_resolution" -s > "$OUT/det.log" 2>&1; echo "exit $?" >> "$OUT/det.log" )
grep -v "Warning\|warnings.warn\|^$\|_create_method" "$OUT/det.log" | tail -30 | cut -c1-900
