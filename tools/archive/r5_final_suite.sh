#!/usr/bin/env bash
# round-5 last session: the driver's two commands at HEAD (after the tiled-768 test's threshold and the plan-byte accounting changed)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r5_final
mkdir -p "$OUT"
t0=$(date +%s)
( timeout 1400 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider --durations=8 > "$OUT/pytest.log" 2>&1; echo "exit $?" >> "$OUT/pytest.log" )
echo "($(( $(date +%s) - t0 )) s)"; grep -v "Warning\|warnings.warn\|^$\|_create_method\|amdgpu.ids" "$OUT/pytest.log" | tail -12 | cut -c1-300
ls gpurun_out/oracle_cache_misses 2>/dev/null
( timeout 300 python3 -c 'import __graft_entry__ as e; e.smoke()' > "$OUT/smoke.txt" 2>&1; echo "rc=$?" >> "$OUT/smoke.txt" )
grep "rc=\|smoke OK" "$OUT/smoke.txt" | cut -c1-160
