#!/usr/bin/env bash
# round-6 final check at HEAD (host-side changes after the closing session: GC thaw on plan eviction, rendezvous port holder): the
# driver's two commands and its bench command once more
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_final
mkdir -p "$OUT"
t0=$(date +%s)
( timeout 1300 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider --durations=0 > "$OUT/pytest.log" 2>&1; echo "exit $?" >> "$OUT/pytest.log" )
echo "($(( $(date +%s) - t0 )) s)"; grep -v "Warning\|warnings.warn\|^$\|_create_method\|amdgpu.ids\|^tests/test_gpu\|^    " "$OUT/pytest.log" | tail -6 | cut -c1-200
( timeout 300 python3 -c 'import __graft_entry__ as e; e.smoke()' > "$OUT/smoke.txt" 2>&1; echo "rc=$?" >> "$OUT/smoke.txt" )
grep "rc=\|smoke OK" "$OUT/smoke.txt" | cut -c1-120
( OMNI_BENCH_WATCHDOG=400 timeout 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_cmd.json" 2> "$OUT/bench_driver_cmd.err"; echo "exit $?" >> "$OUT/bench_driver_cmd.err" )
python3 - "$OUT/bench_driver_cmd.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print(d["value"], d["ms_per_step"], d["steps"], "TF/s", r["achieved"], r["frac"], "traffic", r.get("traffic"), r.get("traffic_source", "")[:40])
print("   wall", d["config"].get("step_wall_ms"))
print("   scan", d["config"].get("parity_scan"))
print("   cpu_baseline", (d.get("cpu_baseline") or {}).get("value"))
PY
echo "total $(( $(date +%s) - t0 )) s"
