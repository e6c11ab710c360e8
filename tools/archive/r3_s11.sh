#!/usr/bin/env bash
# Round-3 GPU session 11 (short): the bench-path parity test on the tie-free bench frames (synth.BENCH_SEEDS: all 8 frames element for
# element against the oracle's own list) and the kernel tests after the window-attention / channel-apply kernels were restored.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r3s11
mkdir -p "$OUT"
python tools/make_weights.py --ensure detector > /dev/null 2>&1
python tools/make_weights.py --ensure caption > /dev/null 2>&1
for f in tests/test_gpu_a_kernels.py tests/test_gpu_z_bench_path.py; do
  n=$(basename "$f" .py)
  t0=$(date +%s)
  ( timeout 600 python -m pytest "$f" -q -m gpu -p no:cacheprovider -x -s > "$OUT/$n.log" 2>&1; echo "exit $?" >> "$OUT/$n.log" )
  echo "--- $n ($(( $(date +%s) - t0 )) s)"; grep "passed\|failed\|skipped\|^exit\|Error\|^{\|problems" "$OUT/$n.log" | tail -6 | cut -c1-1500
done
