#!/usr/bin/env bash
# Round-2 opening GPU session: everything that was built after the last GPU minute of round 1 gets measured in ONE
# gpurun call (≈30 GPU-min of the round's 90: parity ≈5, GEMM A/B ≈2, three benches ≈6, four PMC passes ≈12, stats + stream ≈5).  Usage:
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/r2_gpu_pack.sh'
# Outputs land in gpurun_out/r2/ (copy what should be judged into profiles/).  Every step has its own timeout and the
# script never aborts on a failing step, so one bad experiment cannot eat the session.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r2
mkdir -p "$OUT"
step() { echo "=== $1" | tee -a "$OUT/log.txt"; shift; ( "$@" ) >>"$OUT/log.txt" 2>&1; echo "    exit $?" | tee -a "$OUT/log.txt"; }

# 1. parity first: the N-partition tile order is the default now (bit-identical by construction, never run on a GPU)
step "pytest gpu kernels+caption (new tile order)" timeout 600 python -m pytest tests -m gpu -x -q

# 2. GEMM micro-benchmark A/B: tile order, residency budget, 256x128 tile
step "gemm_bench A/B" env VARIANTS="split:128x128:2+OMNI_XCD_NSPLIT=0,split:128x128:2,split:128x128:2+OMNI_XCD_L2_BUDGET_KB=1280,split:128x128:4,split:128x128:4+OMNI_XCD_NSPLIT=0,split:128x128:5,split:128x128:6" \
  timeout 300 python tools/gemm_bench.py

# 3. headline A/B (same process settings except the knob)
step "bench e2e, round-1 order" env OMNI_XCD_NSPLIT=0 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline
cp "$OUT/log.txt" "$OUT/log_after_bench0.txt"
step "bench e2e, N partition (default)" timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline
step "bench e2e, default but round-1 dwconv kernel" env OMNI_DWCONV_STRIP=0 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline

# 4. PMC evidence for the dominant kernel (separate passes, counters only: no sys/hip trace domains with --pmc)
for ctr in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
  tag=$(echo "$ctr" | tr ' ' '_')
  step "pmc $ctr" timeout 420 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$OUT/pmc_$tag" -- \
    python bench.py --steps 1 --warmup 0 --no-cpu-baseline
done
step "pmc FETCH_SIZE, round-1 order" env OMNI_XCD_NSPLIT=0 timeout 420 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv \
  -d "$OUT/pmc_FETCH_SIZE_nsplit0" -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline

# 5. kernel-time summary of the default configuration
step "kernel stats" timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- \
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline
step "stream bench (configs[3] stand-in, 1 GPU, 64x64 crops)" timeout 420 python tools/stream_bench.py --items 48 --caption-res 64
for d in "$OUT"/pmc_*; do                        # one summary per pass (the two FETCH_SIZE passes must not be merged)
  [ -d "$d" ] && python tools/pmc_summary.py "$d" > "$d.json" 2>>"$OUT/log.txt"
done
find "$OUT" -name "*.csv" -size +8M -delete      # merged-back budget is 64 MiB: keep summaries, drop raw traces
ls -la "$OUT" | tee -a "$OUT/log.txt"
