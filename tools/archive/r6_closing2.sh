#!/usr/bin/env bash
# round-6 SECOND closing session at the final code of the round (exact-row twins, row-patch patch embedding): the driver's suite and smoke, the
# driver's bench command, the default line, the GPU-vs-oracle scan (provenance of the committed kernel sources), rocprofv3 kernel stats of the
# synchronous bench, two PMC passes (FETCH_SIZE / WRITE_SIZE) over one 128-crop caption plan
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_closing2
mkdir -p "$OUT"
rm -f gpurun_out/parity_counters.jsonl
t0=$(date +%s)
( timeout 1300 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider --durations=10 > "$OUT/pytest.log" 2>&1; echo "exit $?" >> "$OUT/pytest.log" )
echo "($(( $(date +%s) - t0 )) s)"; grep -v "Warning\|warnings.warn\|^$\|_create_method\|amdgpu.ids\|^tests/test_gpu" "$OUT/pytest.log" | tail -16 | cut -c1-300
cp gpurun_out/parity_counters.jsonl "$OUT/parity_counters.jsonl" 2>/dev/null
ls gpurun_out/oracle_cache_misses 2>/dev/null
( timeout 300 python3 -c 'import __graft_entry__ as e; e.smoke()' > "$OUT/smoke.txt" 2>&1; echo "rc=$?" >> "$OUT/smoke.txt" )
grep "rc=\|smoke OK" "$OUT/smoke.txt" | cut -c1-200
show() {
python3 - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], d["steps"], "TF/s", r["achieved"], r["frac"], "gemm", r["gemm_ms_per_step"], "sum", r["profiled_step_ms"], "non-gemm", r["non_gemm_share"], "traffic", r.get("traffic"))
    print("   wall", d["config"].get("step_wall_ms"))
    print("   cpu_baseline", (d.get("cpu_baseline") or {}).get("value"), "scan", (d["config"].get("parity_scan") or {}).get("source"))
    print("   ", {k: (v.get("value"), v.get("ms_per_step")) for k, v in (d.get("extra") or {}).items() if isinstance(v, dict)})
except Exception as e:
    print(sys.argv[1], "no bench line", e)
PY
}
( OMNI_BENCH_WATCHDOG=400 timeout 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_cmd.json" 2> "$OUT/bench_driver_cmd.err"; echo "exit $?" >> "$OUT/bench_driver_cmd.err" )
show "$OUT/bench_driver_cmd.json"
( OMNI_BENCH_WATCHDOG=400 timeout 900 python3 bench.py --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "exit $?" >> "$OUT/bench.err" )
show "$OUT/bench.json"
echo "=== scan"
( timeout 500 python tools/scan_gpu_vs_oracle.py device > "$OUT/scan_gpu_vs_oracle.json" 2> "$OUT/scan.err"; echo "exit $?" )
python3 -c "
import json
d=json.load(open('$OUT/scan_gpu_vs_oracle.json')); print({k:v for k,v in d.items() if k not in ('frames_not_identical','definition')}); print([(r['seed'], r.get('identical_as_sets'), r['boxes_identical']) for r in d['frames_not_identical']])"
echo "=== kernel stats of the synchronous bench command"
( timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extra --no-pipeline > "$OUT/stats_bench.json" 2> "$OUT/stats.err"; echo "exit $?" )
f=$(find "$OUT/stats" -name "*kernel_stats.csv" | head -1); echo "$f"; head -14 "$f" | cut -c1-200
cp "$f" "$OUT/kernel_stats_sync.csv" 2>/dev/null
find "$OUT/stats" -name "*.csv" -size +4M -delete; find "$OUT/stats" -name "*.db" -delete
show "$OUT/stats_bench.json"
echo "=== PMC traffic"
for c in FETCH_SIZE WRITE_SIZE; do
  ( timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/pmc_$c" -- python tools/caption_profile.py 128 768 1 > "$OUT/pmc_$c.json" 2> "$OUT/pmc_$c.err"; echo "$c exit $?" )
  python tools/pmc_summary.py "$OUT/pmc_$c" > "$OUT/pmc_summary_$c.json" 2>/dev/null
  find "$OUT/pmc_$c" -name "*.csv" -size +4M -delete; find "$OUT/pmc_$c" -name "*.db" -delete
done
python tools/pmc_traffic.py "$OUT/pmc_summary_FETCH_SIZE.json" "$OUT/pmc_summary_WRITE_SIZE.json" 256 2654945280 > "$OUT/pmc_traffic.json" 2>/dev/null
python3 -c "
import json; d=json.load(open('$OUT/pmc_traffic.json')); print({k: d[k] for k in ('gemm_fetch_bytes_per_crop','gemm_write_bytes_per_crop','algorithmic_gemm_bytes_per_crop','ratio','gemm_bytes_per_launch')})"
echo "=== annotate tail"
( timeout 300 python3 tools/annotate_bench.py --iters 5 > "$OUT/annotate_bench.json" 2>/dev/null; echo "exit $?" ); cut -c1-900 "$OUT/annotate_bench.json"
echo "total $(( $(date +%s) - t0 )) s"
