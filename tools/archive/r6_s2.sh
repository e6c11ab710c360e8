#!/usr/bin/env bash
# round-6 session 2: (1) caption micro-batch capacity sweep of the benched composition (does a working set that fits the 256 MB Infinity
# Cache pay?), (2) the +40 ms step: python GC on / off / frozen over the driver's 20 steps
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s2
mkdir -p "$OUT"
show() {
python3 - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], "gemm", r["gemm_ms_per_step"], "sum", r["profiled_step_ms"], "TF/s", r["achieved"],
          "mb", d["config"].get("caption_micro_batch"), "gc", d["config"].get("python_gc"))
    print("   wall", d["config"].get("step_wall_ms"))
    print("   fam", {k: round(v, 1) for k, v in (r.get("kernel_family_ms_per_step") or {}).items() if v > 2})
except Exception as e:
    print(sys.argv[1], "no bench line", e)
PY
}
for mb in 128 64 32 96; do
  ( OMNI_BENCH_WATCHDOG=400 timeout 600 python3 bench.py --micro-batch $mb --steps 3 --warmup 2 --no-cpu-baseline --no-extra > "$OUT/bench_mb$mb.json" 2> "$OUT/bench_mb$mb.err"; echo "mb$mb exit $?" )
  show "$OUT/bench_mb$mb.json"
done
for gcm in on off freeze; do
  ( OMNI_BENCH_GC=$gcm OMNI_BENCH_WATCHDOG=400 timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > "$OUT/bench_gc_$gcm.json" 2> "$OUT/bench_gc_$gcm.err"; echo "gc $gcm exit $?" )
  show "$OUT/bench_gc_$gcm.json"
  grep "timed region" "$OUT/bench_gc_$gcm.err" | cut -c1-200
done
