#!/usr/bin/env bash
# Round-3: MFMA-pipe utilisation and sustained clock per kernel of one 128-crop caption plan (two PMC passes, kernel trace only).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r3pmc
mkdir -p "$OUT"
python tools/make_weights.py --ensure caption > /dev/null 2>&1
i=0
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  ( timeout 200 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$OUT/pmc_$i" -- python tools/caption_profile.py 128 768 1 > "$OUT/pmc_$i.json" 2> "$OUT/pmc_$i.err"; echo "pass $i exit $?" )
  python tools/pmc_summary.py "$OUT/pmc_$i" > "$OUT/pmc_summary_$i.json" 2>/dev/null
  find "$OUT/pmc_$i" -name "*.csv" -size +4M -delete; find "$OUT/pmc_$i" -name "*.db" -delete
done
python - "$OUT" <<'PY'
import json, sys
a = json.load(open(sys.argv[1] + "/pmc_summary_1.json"))
for k, v in list(a["derived"].items())[:12]:
    print(k[:80], v)
PY
