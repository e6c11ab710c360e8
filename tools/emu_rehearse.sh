#!/usr/bin/env bash
# Rehearse `-m gpu` test files WITHOUT a GPU: the product's device path runs on the host emulation of its own kernels (tests/emu,
# DESIGN.md 4a).  GPU sizes, so minutes to hours per file on 8 cores (≈2.4-4 G emulated f16 multiply-adds per second); the
# 768x768 bench-path test of test_gpu_d_pipeline.py (≈350 crops x 425 GFLOP) is out of reach — deselect it with -k.
#   tools/emu_rehearse.sh tests/test_gpu_a_kernels.py [pytest args]
# Output: /tmp/rehearse/<file>.log.  OMNI_EMU_THREADS=n limits the host threads the emulation uses.
set -u
cd "$(dirname "$0")/.."
mkdir -p /tmp/rehearse
f="$1"; shift
n=$(basename "$f" .py)
OMNI_EMU=1 timeout "${REHEARSE_LIMIT:-14400}" python -m pytest "$f" -m gpu -q --durations=0 -p no:cacheprovider "$@" > "/tmp/rehearse/$n.log" 2>&1
echo "exit $?" >> "/tmp/rehearse/$n.log"
grep -v "Warn\|warn" "/tmp/rehearse/$n.log" | tail -15
