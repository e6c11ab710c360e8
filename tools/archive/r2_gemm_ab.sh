#!/usr/bin/env bash
# Round-2 third GPU session: schedule A/B + ablations + PMC of the LDS-DMA GEMM, unconditional detector parity.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r2s3
mkdir -p "$OUT"
step() { echo "=== $1" | tee -a "$OUT/log.txt"; shift; ( "$@" ) >>"$OUT/log.txt" 2>&1; echo "    exit $?" | tee -a "$OUT/log.txt"; }
export SHAPES="s2.qkv,s2.fc2n,s0.fc1"
step "gemm_bench schedule A/B + ablations" env VARIANTS="dma,dma+OMNI_GEMM_VAR=1,dma+OMNI_GEMM_ABL=1,dma+OMNI_GEMM_ABL=2,dma+OMNI_GEMM_ABL=3,dma+OMNI_GEMM_ABL=4,dma:256x128+OMNI_GEMM_VAR=1,dma:128x128+OMNI_GEMM_VAR=1,dma:128x128+OMNI_GEMM_ABL=3" \
  timeout 400 python tools/gemm_bench.py
export SHAPES="s2.qkv,s2.fc2n"
for v in "dma" "dma+OMNI_GEMM_VAR=1"; do
  tagv=$(echo "$v" | tr '+=' '__')
  for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
             "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_WAVES GRBM_GUI_ACTIVE"; do
    tag=$(echo "$ctr" | tr ' ' '_' | cut -c1-40)
    step "pmc $v $ctr" env VARIANTS="$v" timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$OUT/pmc_${tagv}_$tag" -- python tools/gemm_bench.py
  done
done
for d in "$OUT"/pmc_*; do
  [ -d "$d" ] && python tools/pmc_summary.py "$d" > "$d.json" 2>>"$OUT/log.txt"
done
find "$OUT" -name "*.csv" -size +8M -delete
step "pytest detector (unconditional parity)" timeout 1200 python -m pytest tests/test_gpu_c_detector.py -x -q
tail -5 "$OUT/log.txt"
