#!/usr/bin/env bash
# round-4 session 5: (1) GEMM tile A/B in the bench — do smaller-register tiles (256x128: 124 VGPRs, 128x128) let the other lane's
# HBM-bound kernels share a CU with GEMM blocks?  (2) PMC passes over one 128-crop caption plan at the round-4 composition:
# FETCH_SIZE, WRITE_SIZE (traffic), SQ counters (MFMA-pipe busy), GRBM_GUI_ACTIVE (clock) — separate runs, kernel trace only
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4_s5
mkdir -p $O
run_bench() {  # tag, env...
  tag=$1; shift
  ( env "$@" OMNI_BENCH_WATCHDOG=120 timeout 240 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra ${EXTRA:-} > "$O/bench_$tag.json" 2> "$O/bench_$tag.err"; echo "$tag exit $?" )
  python - "$O/bench_$tag.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("   ", d["value"], "screenshots/s", d["ms_per_step"], "ms/step; gemm", r.get("gemm_ms_per_step"), "profiled sum", r.get("profiled_step_ms"), "non-gemm share", r.get("non_gemm_share"))
except Exception as e:
    print("    no line:", e)
PY
}
echo "=== 1. GEMM tile A/B (K = 6)"
run_bench default A=1
run_bench tile256x128 OMNI_GEMM_TILE=256x128
run_bench tile128x128 OMNI_GEMM_TILE=128x128
EXTRA="--lanes 3" run_bench tile256x128_lanes3 OMNI_GEMM_TILE=256x128
echo "=== 2. PMC passes"
for c in FETCH_SIZE WRITE_SIZE; do
  ( timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$O/pmc_$c" -- python tools/caption_profile.py 128 768 1 > "$O/pmc_$c.json" 2> "$O/pmc_$c.err"; echo "$c exit $?" )
  python tools/pmc_summary.py "$O/pmc_$c" > "$O/pmc_summary_$c.json" 2>/dev/null
  find "$O/pmc_$c" -name "*.csv" -size +4M -delete; find "$O/pmc_$c" -name "*.db" -delete
done
i=0
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  ( timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$O/pmc_sq$i" -- python tools/caption_profile.py 128 768 1 > "$O/pmc_sq$i.json" 2> "$O/pmc_sq$i.err"; echo "sq pass $i exit $?" )
  python tools/pmc_summary.py "$O/pmc_sq$i" > "$O/pmc_summary_sq$i.json" 2>/dev/null
  find "$O/pmc_sq$i" -name "*.csv" -size +4M -delete; find "$O/pmc_sq$i" -name "*.db" -delete
done
ls -la $O | head -40
