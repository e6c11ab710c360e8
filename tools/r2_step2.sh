#!/usr/bin/env bash
# Round-2 second GPU session: parity of the pre-split LDS-DMA GEMM + producers, captioner parity on the new path, GEMM A/B.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r2s2
mkdir -p "$OUT"
step() { echo "=== $1" | tee -a "$OUT/log.txt"; shift; ( "$@" ) >>"$OUT/log.txt" 2>&1; echo "    exit $?" | tee -a "$OUT/log.txt"; }
step "pytest gemm_dma + conv" timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm_dma or conv_igemm or mfma"
step "gemm_bench" env VARIANTS="split:128x128,dma,dma:256x128,dma:128x128" timeout 400 python tools/gemm_bench.py
step "pytest caption" timeout 900 python -m pytest tests/test_gpu_caption.py -x -q -k "kernels or r64 or r768"
tail -5 "$OUT/log.txt"
