#!/usr/bin/env bash
# round-5 session 5: where does a batch-1 detector graph replay spend its 4.4 ms?  rocprofv3 kernel trace of `bench.py --mode detect`
# (kernel durations and the gaps between consecutive graph nodes), batch 1 and batch 8
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r5_s5
mkdir -p "$OUT"
for B in 1 8; do
  ( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/det_b$B" -- python bench.py --mode detect --batch $B --steps 100 --warmup 20 --no-extra --no-cpu-baseline > "$OUT/det_b$B.json" 2> "$OUT/det_b$B.err"; echo "b$B exit $?" )
  tail -c 600 "$OUT/det_b$B.json"; echo
  f=$(find "$OUT/det_b$B" -name "*kernel_stats.csv" | head -1); cp "$f" "$OUT/det_b${B}_kernel_stats.csv"; head -14 "$f" | cut -c1-200
  t=$(find "$OUT/det_b$B" -name "*kernel_trace.csv" | head -1)
  python tools/hwq_gaps.py "$t" > "$OUT/det_b${B}_gaps.json" 2>/dev/null || true
  python - "$t" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 20 replays: busy time vs span
n = len(rows)
tail = rows[n // 2:]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tail)
span = int(tail[-1]["End_Timestamp"]) - int(tail[0]["Start_Timestamp"])
gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(tail, tail[1:])]
gaps_small = [g for g in gaps if g < 100000]
print("kernels", len(tail), "busy ms", busy / 1e6, "span ms", span / 1e6, "mean gap us (gaps < 100 us)", sum(gaps_small) / max(len(gaps_small), 1) / 1e3,
      "mean kernel us", busy / len(tail) / 1e3)
PY
  find "$OUT/det_b$B" -name "*.csv" -size +3M -delete; find "$OUT/det_b$B" -name "*.db" -delete
done
ls -la "$OUT"
