#!/usr/bin/env bash
# round-6 session 5: ping-pong schedule (SCHED 2) of gemm_dma_kernel vs the shipping schedule: GEMM micro-benchmark under power sampling
# (interleaved twice), the GEMM kernel check on hardware under both, then the bench with and without it
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s5
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
for sch in 3; do
  ( timeout 600 python3 -m pytest tests/test_gpu_a_kernels.py -x -q -m gpu -p no:cacheprovider -k "gemm_dma or schedules" > "$OUT/pytest_gemm_s$sch.log" 2>&1; echo "sched $sch pytest exit $?" )
  tail -2 "$OUT/pytest_gemm_s$sch.log" | cut -c1-200
done
show() {
python3 - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], "TF/s", r["achieved"], r["frac"], "gemm", r["gemm_ms_per_step"], "sum", r["profiled_step_ms"])
    print("   fam", {k: round(v, 1) for k, v in (r.get("kernel_family_ms_per_step") or {}).items() if v > 10})
    print("   per_kernel", [(k.get("shape"), k.get("tflops")) for k in (r.get("per_kernel") or [])[:5]])
except Exception as e:
    print(sys.argv[1], "no bench line", e)
PY
}
for sch in 1 2 3 1 2 3; do
  ( OMNI_GEMM_SCHED=$sch OMNI_BENCH_WATCHDOG=400 timeout 600 python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extra > "$OUT/bench_s${sch}_$RANDOM.json" 2>> "$OUT/bench.err"; echo "bench sched $sch exit $?" )
done
for f in "$OUT"/bench_s*.json; do show "$f"; done
