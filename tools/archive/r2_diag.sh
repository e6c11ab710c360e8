#!/usr/bin/env bash
# Round-2 GEMM diagnosis in one short gpurun call: ablations of the default split kernel (which side bounds it: the
# MFMA+LDS side or the global->LDS side), the opt-in variants, and PMC passes on the micro-benchmark (not the e2e bench).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r2diag
mkdir -p "$OUT"
step() { echo "=== $1" | tee -a "$OUT/log.txt"; shift; ( "$@" ) >>"$OUT/log.txt" 2>&1; echo "    exit $?" | tee -a "$OUT/log.txt"; }
export SHAPES="s2.fc1,s2.fc2,s2.qkv,s0.fc1,enc.fc1"
step "gemm_bench ablations" env VARIANTS="split:128x128:2,split:128x128:2+OMNI_SPLIT_ABL=1,split:128x128:2+OMNI_SPLIT_ABL=4,split:128x128:2+OMNI_SPLIT_ABL=2,split:128x128:2+OMNI_SPLIT_ABL=3,split:128x128:6,split:128x128:4,split:128x128:5,split:128x128:2+OMNI_XCD_NSPLIT=0" \
  timeout 400 python tools/gemm_bench.py
export SHAPES="s2.fc1,s2.fc2"
export VARIANTS="split:128x128:2"
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAVES"; do
  tag=$(echo "$ctr" | tr ' ' '_' | cut -c1-60)
  step "pmc $ctr" timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$OUT/pmc_$tag" -- python tools/gemm_bench.py
done
for d in "$OUT"/pmc_*; do
  [ -d "$d" ] && python tools/pmc_summary.py "$d" > "$d.json" 2>>"$OUT/log.txt"
done
find "$OUT" -name "*.csv" -size +8M -delete
ls -laR "$OUT" | head -80 | tee -a "$OUT/log.txt"
