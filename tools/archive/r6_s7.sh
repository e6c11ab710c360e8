#!/usr/bin/env bash
# round-6 session 7: ping-pong with early release (SCHED 3: the phase's second barrier in front of the last product) vs SCHED 2
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s7
mkdir -p "$OUT"
( timeout 900 python tools/gemm_exp.py sched 4 > "$OUT/gemm_sched.jsonl" 2> "$OUT/gemm_sched.err"; echo "exit $?" )
python3 - "$OUT/gemm_sched.jsonl" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    try: d = json.loads(l)
    except ValueError: continue
    if "cases" not in d: print(d); continue
    for k, c in d["cases"].items():
        print(d["variant"], k, c["ms_per_launch_events_20"], c["ms_per_launch_sustained"], c["algorithmic_tflops_sustained"], c["power_w_mean"])
PY
for sch in 2 3 2 3; do
  ( OMNI_GEMM_SCHED=$sch OMNI_BENCH_WATCHDOG=400 timeout 600 python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extra > "$OUT/bench_s${sch}_$RANDOM.json" 2>> "$OUT/bench.err"; echo "bench sched $sch exit $?" )
done
for f in "$OUT"/bench_s*.json; do
python3 - "$f" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], "TF/s", r["achieved"], r["frac"], "gemm", r["gemm_ms_per_step"], "sum", r["profiled_step_ms"])
PY
done
