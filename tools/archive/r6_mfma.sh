#!/usr/bin/env bash
# round-6 MFMA-utilisation session (evidence only, no code under test): MFMA-pipe utilisation and sustained clock of the caption kernels from two PMC passes
# (SQ counters; GRBM_GUI_ACTIVE) over one 128-crop plan — counters collected with --kernel-trace only, in their own runs
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_mfma
mkdir -p "$OUT"
i=0
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  ( timeout 240 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$OUT/pmc_sq$i" -- python tools/caption_profile.py 128 768 1 > "$OUT/pmc_sq$i.json" 2> "$OUT/pmc_sq$i.err"; echo "sq pass $i exit $?" )
  python tools/pmc_summary.py "$OUT/pmc_sq$i" > "$OUT/pmc_summary_sq$i.json" 2>/dev/null
  find "$OUT/pmc_sq$i" -name "*.csv" -size +4M -delete; find "$OUT/pmc_sq$i" -name "*.db" -delete
done
python tools/pmc_mfma.py "$OUT/pmc_summary_sq1.json" "$OUT/pmc_summary_sq2.json" > "$OUT/mfma_utilisation.json" 2>/dev/null
head -c 1800 "$OUT/mfma_utilisation.json"
