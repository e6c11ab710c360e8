#!/usr/bin/env bash
# Round-3 GPU session 8: the 4-wave K16 GEMM (two blocks per CU) — correctness + micro-benchmark against the 8-wave 256x256 kernel.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r3s8
mkdir -p "$OUT"
echo "=== 1. kernel tests"
( timeout 300 python -m pytest tests/test_gpu_a_kernels.py -q -m gpu -p no:cacheprovider -x -k "gemm" > "$OUT/kern.log" 2>&1; echo "exit $?" >> "$OUT/kern.log" )
grep "passed\|failed\|exit\|Error" "$OUT/kern.log" | tail -3
echo "=== 2. gemm bench"
( VARIANTS="dma,dma:256x128k16,dma,dma:256x128k16" timeout 500 python tools/gemm_bench.py > "$OUT/gemm_bench.txt" 2> "$OUT/gemm_bench.err"; echo "exit $?" )
cat "$OUT/gemm_bench.txt"
