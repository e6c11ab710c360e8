#!/usr/bin/env bash
# round-6 final check at HEAD (host-side changes only since the second closing session): the driver's suite, smoke, the driver's bench
# command, then three repetitions of the tests that exercise the exact-row twins under varying crop counts (stream, tiled 4K, benched
# path, device hand-off) — the first half of the round found a stream-ordering race only by repeating
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_final2
mkdir -p "$OUT"
rm -f gpurun_out/parity_counters.jsonl
t0=$(date +%s)
( timeout 1300 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider --durations=6 > "$OUT/pytest.log" 2>&1; echo "exit $?" >> "$OUT/pytest.log" )
echo "($(( $(date +%s) - t0 )) s)"; grep -v "Warning\|warnings.warn\|^$\|_create_method\|amdgpu.ids\|^tests/test_gpu\|^    " "$OUT/pytest.log" | tail -12 | cut -c1-300
cp gpurun_out/parity_counters.jsonl "$OUT/parity_counters.jsonl" 2>/dev/null
( timeout 300 python3 -c 'import __graft_entry__ as e; e.smoke()' > "$OUT/smoke.txt" 2>&1; echo "rc=$?" >> "$OUT/smoke.txt" )
grep "rc=\|smoke OK" "$OUT/smoke.txt" | cut -c1-160
( OMNI_BENCH_WATCHDOG=400 timeout 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_cmd.json" 2> "$OUT/bench_driver_cmd.err"; echo "exit $?" >> "$OUT/bench_driver_cmd.err" )
python3 - "$OUT/bench_driver_cmd.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(d["value"], d["ms_per_step"], d["steps"], "TF/s", r["achieved"], r["frac"], "gemm", r["gemm_ms_per_step"], "sum", r["profiled_step_ms"], "hbm", d["config"].get("hbm_peak_allocated_gb"), "traffic", r.get("traffic"))
    print("   wall", d["config"].get("step_wall_ms"))
    print("   scan", (d["config"].get("parity_scan") or {}).get("source"), "cpu_baseline", (d.get("cpu_baseline") or {}).get("value"))
except Exception as e:
    print(sys.argv[1], "no bench line", e)
PY
echo "=== stress"
for rep in 1 2 3; do
( timeout 600 python3 -m pytest tests/test_gpu_k_stream_parity.py tests/test_gpu_d_pipeline.py tests/test_gpu_g_device_handoff.py tests/test_gpu_z_bench_path.py::test_bench_path_parity_batch8_full_width_r768 tests/test_gpu_b_caption_model.py::test_exact_row_encode_twin_r768 -x -q -m gpu -p no:cacheprovider > "$OUT/stress_$rep.log" 2>&1; echo "exit $?" >> "$OUT/stress_$rep.log" )
tail -2 "$OUT/stress_$rep.log" | cut -c1-200
done
echo "total $(( $(date +%s) - t0 )) s"
