#!/usr/bin/env bash
# round-6 session 4: the LZ77 + dynamic-Huffman device PNG (OMNI_OP_PNG_DEFLATE i5 = 1) on the MI355X: overlay / PNG GPU tests, size and
# time against Pillow and against the fixed-Huffman variant
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s4
mkdir -p "$OUT"
( timeout 900 python3 -m pytest tests/test_gpu_f_overlay_png.py -x -q -m gpu -p no:cacheprovider --durations=8 > "$OUT/pytest_png.log" 2>&1; echo "exit $?" >> "$OUT/pytest_png.log" )
grep -v "Warning\|warnings.warn\|^$\|_create_method\|amdgpu.ids" "$OUT/pytest_png.log" | tail -16 | cut -c1-400
( timeout 300 python3 tools/annotate_bench.py --iters 5 > "$OUT/annotate_bench.json" 2> "$OUT/annotate_bench.err"; echo "exit $?" )
cat "$OUT/annotate_bench.json" | cut -c1-2000; tail -3 "$OUT/annotate_bench.err" | cut -c1-300
