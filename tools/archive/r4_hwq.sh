#!/usr/bin/env bash
# round-4 (after the closing session; measurement only): the hardware-queue cliff of round 3 — one encode lane + the decode stream at
# GPU_MAX_HW_QUEUES=8 ran 1034 instead of 709 ms per step.  Kernel traces of --lanes 1 at 4 and 8 queues (K = 3) and at 2 lanes / 8
# queues, per-queue gap summary (tools/hwq_gaps.py); does it still exist with the round-4 composition?
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r4_hwq
mkdir -p "$OUT"
for cfg in "1 4" "1 8" "2 8"; do
  set -- $cfg
  tag=lanes$1_hwq$2
  ( GPU_MAX_HW_QUEUES=$2 OMNI_BENCH_WATCHDOG=120 timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace_$tag" -- \
      python bench.py --steps 3 --warmup 2 --lanes $1 --no-cpu-baseline --no-extra > "$OUT/bench_$tag.json" 2> "$OUT/bench_$tag.err"; echo "$tag exit $?" )
  python - "$OUT/bench_$tag.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("   ", d["value"], "screenshots/s", d["ms_per_step"], "ms/step (under the tracer)", d["config"].get("hw_queues"))
except Exception as e:
    print("    no line:", e)
PY
  f=$(find "$OUT/trace_$tag" -name "*kernel_trace.csv" | head -1)
  python tools/hwq_gaps.py "$f" > "$OUT/hwq_gaps_$tag.json" 2> "$OUT/hwq_gaps_$tag.err"; head -c 1200 "$OUT/hwq_gaps_$tag.json"; echo
  find "$OUT/trace_$tag" -name "*.csv" -size +4M -delete; find "$OUT/trace_$tag" -name "*.db" -delete
done
