#!/usr/bin/env bash
# Round-2 last short GPU call (half / quarter width only: no full-width blob to generate).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r2v2
mkdir -p "$OUT"
( timeout 150 python -m pytest tests/test_gpu_c_detector.py -q -p no:cacheprovider -k "half_width or tiled or f16" > "$OUT/det.log" 2>&1; echo "exit $?" >> "$OUT/det.log" )
grep -v "Warning\|warnings.warn\|^$\|_create_method\|^tests/test_gpu" "$OUT/det.log" | tail -25 | cut -c1-1500
( timeout 110 python __graft_entry__.py --smoke > "$OUT/smoke.log" 2>&1; echo "exit $?" >> "$OUT/smoke.log" )
tail -4 "$OUT/smoke.log" | cut -c1-1200
