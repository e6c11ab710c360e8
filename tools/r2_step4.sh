#!/usr/bin/env bash
# Round-2 fourth GPU session: whole GPU suite on the new paths, e2e bench, kernel-time profile.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r2s4
mkdir -p "$OUT"
step() { echo "=== $1" | tee -a "$OUT/log.txt"; shift; ( "$@" ) >>"$OUT/log.txt" 2>&1; echo "    exit $?" | tee -a "$OUT/log.txt"; }
step "rocm-smi" bash -c "rocm-smi --showpower --showmaxpower --showclocks --showperflevel 2>&1 | head -60"
step "bench e2e" timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline
step "kernel stats" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline
step "pytest gpu" timeout 2400 python -m pytest tests -m gpu -q -x
find "$OUT" -name "*.csv" -size +8M -delete
tail -30 "$OUT/log.txt"
