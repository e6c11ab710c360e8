#!/usr/bin/env bash
# round-5 session 8: (1) the driver's two commands once more on another fresh lease (flakiness check; the oracle model is now built once
# per process), (2) HBM traffic of the GEMM family from two PMC passes (FETCH_SIZE, WRITE_SIZE — separate rocprofv3 runs, kernel trace
# only besides the counter) over one 128-crop caption plan
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r5_s8
mkdir -p "$OUT"
t0=$(date +%s)
( timeout 1400 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider --durations=8 > "$OUT/pytest.log" 2>&1; echo "exit $?" >> "$OUT/pytest.log" )
echo "($(( $(date +%s) - t0 )) s)"; grep -v "Warning\|warnings.warn\|^$\|_create_method\|amdgpu.ids" "$OUT/pytest.log" | tail -12 | cut -c1-300
( timeout 300 python3 -c 'import __graft_entry__ as e; e.smoke()' > "$OUT/smoke.txt" 2>&1; echo "rc=$?" >> "$OUT/smoke.txt" )
grep "rc=\|smoke OK" "$OUT/smoke.txt" | cut -c1-160
for c in FETCH_SIZE WRITE_SIZE; do
  ( timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/pmc_$c" -- python tools/caption_profile.py 128 768 1 > "$OUT/pmc_$c.json" 2> "$OUT/pmc_$c.err"; echo "$c exit $?" )
  python tools/pmc_summary.py "$OUT/pmc_$c" > "$OUT/pmc_summary_$c.json" 2>/dev/null
  find "$OUT/pmc_$c" -name "*.csv" -size +4M -delete; find "$OUT/pmc_$c" -name "*.db" -delete
done
python tools/pmc_traffic.py "$OUT/pmc_summary_FETCH_SIZE.json" "$OUT/pmc_summary_WRITE_SIZE.json" 256 2654945280 > "$OUT/pmc_traffic.json" 2>/dev/null
python -c "
import json; d=json.load(open('$OUT/pmc_traffic.json')); print({k: d[k] for k in ('gemm_fetch_bytes_per_crop','gemm_write_bytes_per_crop','algorithmic_gemm_bytes_per_crop','ratio','gemm_bytes_per_launch')})"
