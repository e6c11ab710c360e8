#!/usr/bin/env bash
# Round-2 measurement session: default bench line (with extras + cpu baseline), kernel-trace stats and PMC traffic of the same command.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r2s6
mkdir -p "$OUT"
step() { echo "=== $1" | tee -a "$OUT/log.txt"; shift; ( "$@" ) >>"$OUT/log.txt" 2>&1; echo "    exit $?" | tee -a "$OUT/log.txt"; }
python tools/make_weights.py --ensure detector > /dev/null 2>&1
python tools/make_weights.py --ensure caption > /dev/null 2>&1
( OMNI_BENCH_WATCHDOG=120 timeout 600 python bench.py --steps 5 --warmup 2 > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "exit $?" >> "$OUT/bench.err" )
tail -12 "$OUT/bench.err" | cut -c1-300
step "kernel stats" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra
for ctr in "FETCH_SIZE" "WRITE_SIZE"; do
  step "pmc $ctr" timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$OUT/pmc_$ctr" -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra
done
for d in "$OUT"/pmc_*; do [ -d "$d" ] && python tools/pmc_summary.py "$d" > "$d.json" 2>>"$OUT/log.txt"; done
find "$OUT" -name "*.csv" -size +6M -delete
ls -la "$OUT" "$OUT"/stats/* 2>/dev/null | head -30
