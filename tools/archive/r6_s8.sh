#!/usr/bin/env bash
# round-6 session 8: exact-row encode twins (florence.py::_CaptionPlans.encode_rows) — the new GPU test, the benched-path parity test
# (345 crops, the 89-row remainder now an exact-row graph in a 128-row plan set's buffers), then the bench A/B within one lease:
# OMNI_EXACT_ROWS=0 (padded 96-row bucket, as before) vs the default, interleaved twice
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s8
mkdir -p "$OUT"
rm -f gpurun_out/parity_counters.jsonl
t0=$(date +%s)
echo "=== 1. tests"
( timeout 900 python3 -m pytest tests/test_gpu_b_caption_model.py::test_exact_row_encode_twin_r768 tests/test_gpu_z_bench_path.py -x -q -m gpu -p no:cacheprovider --durations=6 > "$OUT/pytest.log" 2>&1; echo "exit $?" >> "$OUT/pytest.log" )
echo "($(( $(date +%s) - t0 )) s)"; grep -v "Warning\|warnings.warn\|^$\|_create_method\|amdgpu.ids" "$OUT/pytest.log" | tail -14 | cut -c1-600
cp gpurun_out/parity_counters.jsonl "$OUT/parity_counters.jsonl" 2>/dev/null
show() {
python3 - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], d["steps"], "TF/s", r["achieved"], r["frac"], "gemm", r["gemm_ms_per_step"], "sum", r["profiled_step_ms"], "hbm", d["config"].get("hbm_peak_allocated_gb"))
    print("   wall", d["config"].get("step_wall_ms"))
    print("   parts", {k: v.get("encode_gemm_ms") for k, v in r["parts"].items()})
except Exception as e:
    print(sys.argv[1], "no bench line", e)
PY
}
echo "=== 2. bench A/B"
for rep in 1 2; do
for ex in 0 1; do
( OMNI_EXACT_ROWS=$ex OMNI_BENCH_WATCHDOG=400 timeout 600 python3 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-extra > "$OUT/bench_exact${ex}_$rep.json" 2> "$OUT/bench_exact${ex}_$rep.err"; echo "exit $?" >> "$OUT/bench_exact${ex}_$rep.err" )
show "$OUT/bench_exact${ex}_$rep.json"
done
done
echo "total $(( $(date +%s) - t0 )) s"
