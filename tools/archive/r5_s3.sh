#!/usr/bin/env bash
# round-5 session 3 (dress rehearsal of the driver's round-end run at the candidate commit): (1) the -m gpu suite as the driver runs it,
# (2) smoke, (3) the driver's exact bench command, (4) the default bench line (extras + measured whole-screenshot CPU baseline),
# (5) f16 token-match rate printed, (6) configs[0] on the reference's demo image with the v5 stand-in
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r5_s3
mkdir -p "$OUT"
echo "=== 1. pytest tests/ -x -q -m gpu (one process)"
t0=$(date +%s)
( timeout 1400 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider --durations=12 > "$OUT/pytest.log" 2>&1; echo "exit $?" >> "$OUT/pytest.log" )
echo "($(( $(date +%s) - t0 )) s)"; grep -v "Warning\|warnings.warn\|^$\|_create_method\|amdgpu.ids" "$OUT/pytest.log" | tail -22 | cut -c1-300
ls gpurun_out/oracle_cache_misses 2>/dev/null
echo "=== 2. smoke"
( timeout 300 python3 -c 'import __graft_entry__ as e; e.smoke()' > "$OUT/smoke.txt" 2>&1; echo "rc=$?" >> "$OUT/smoke.txt" )
grep "rc=\|smoke OK" "$OUT/smoke.txt" | cut -c1-200
echo "=== 3. the driver's bench command"
t0=$(date +%s)
( OMNI_BENCH_WATCHDOG=400 timeout 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_cmd.json" 2> "$OUT/bench_driver_cmd.err"; echo "exit $?" >> "$OUT/bench_driver_cmd.err" )
echo "($(( $(date +%s) - t0 )) s)"; grep -v "^  File\|^Thread\|Warning" "$OUT/bench_driver_cmd.err" | tail -4 | cut -c1-300
python - "$OUT/bench_driver_cmd.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(d["value"], d["ms_per_step"], d["dtype"], d["config"].get("mean_crops_per_screenshot"), r["achieved"], r["frac"], r["gemm_ms_per_step"], r["profiled_step_ms"], r["non_gemm_share"])
    print(r["kernel_family_ms_per_step"])
    print("cpu_baseline", d.get("cpu_baseline"))
    print("parity_scan", d["config"].get("parity_scan"))
    print("extra", json.dumps(d.get("extra"))[:2500])
except Exception as e:
    print("no bench line", e)
PY
echo "=== 5. f16 token-match rate"
( timeout 200 python3 -m pytest tests/test_gpu_b_caption_model.py -q -m gpu -p no:cacheprovider -s -k f16_token_match > "$OUT/f16_rate.txt" 2>&1; echo "exit $?" )
grep "tokens_match" "$OUT/f16_rate.txt" | cut -c1-300
echo "=== 6. configs[0] on demo_image.jpg"
( timeout 900 python tools/configs0.py > "$OUT/configs0.json" 2> "$OUT/configs0.err"; echo "exit $?" )
tail -c 1500 "$OUT/configs0.json"; echo; grep -v Warning "$OUT/configs0.err" | tail -3 | cut -c1-300
ls -la "$OUT" | head -30
