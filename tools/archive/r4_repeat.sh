#!/usr/bin/env bash
# round-4, after the closing session: the driver's two commands once more on another fresh lease at the same frozen product code
# (flakiness check of the suite itself: tie-sensitive comparisons, CPU-oracle timing, first-launch canary)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r4_repeat
mkdir -p "$OUT"
t0=$(date +%s)
( timeout 1500 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider > "$OUT/pytest.log" 2>&1; echo "exit $?" >> "$OUT/pytest.log" )
echo "($(( $(date +%s) - t0 )) s)"; tail -3 "$OUT/pytest.log" | cut -c1-200
( timeout 300 python3 -c 'import __graft_entry__ as e; e.smoke()' > "$OUT/smoke.txt" 2>&1; echo "rc=$?" >> "$OUT/smoke.txt" )
grep "rc=\|smoke OK" "$OUT/smoke.txt" | cut -c1-120
