#!/usr/bin/env bash
# round-6 session 13: decode-step autotune again (the first tool replayed the stateful step plan 120 times without a reset: memory
# fault, session 12), then the bench A/B of the table and the token-exact tests with it
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s13
mkdir -p "$OUT"
t0=$(date +%s)
echo "=== 1. decode autotune"
( timeout 420 python3 tools/decode_autotune.py --write 128 160 192 224 256 288 320 352 384 > "$OUT/decode_autotune.json" 2> "$OUT/decode_autotune.err"; echo "exit $?" )
python3 - "$OUT/decode_autotune.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    for r, v in d["rows"].items():
        print(r, v["graph_ms_heuristic"], "->", v["graph_ms_tuned"], {k: (p["choice"], p["heuristic_us"], p["tuned_us"]) for k, p in v["picked"].items()})
    print(len(d["choices"]), "choices")
except Exception as e:
    print("no autotune output", e)
PY
tail -3 "$OUT/decode_autotune.err" | cut -c1-300
cp omniparser_amd/decode_tuning_gfx950.json "$OUT/" 2>/dev/null
echo "($(( $(date +%s) - t0 )) s)"
[ -f omniparser_amd/decode_tuning_gfx950.json ] || { echo "no table: stop"; exit 0; }
echo "=== 2. bench A/B of the decode table"
showe() {
python3 - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], "TF/s", r["achieved"], r["frac"], "gemm", r["gemm_ms_per_step"], "sum", r["profiled_step_ms"], "decode", r["parts"].get("decode_rows352"))
except Exception as e:
    print(sys.argv[1], "no bench line", e)
PY
}
for arm in 0 1 0 1; do
tag="dt${arm}_$(date +%s)"
( OMNI_DECODE_TUNING=$arm OMNI_BENCH_WATCHDOG=400 timeout 600 python3 bench.py --gpus 1 --steps 10 --warmup 5 --no-cpu-baseline --no-extra > "$OUT/bench_$tag.json" 2> "$OUT/bench_$tag.err"; echo "exit $?" >> "$OUT/bench_$tag.err" )
showe "$OUT/bench_$tag.json"
done
echo "=== 3. tests with the table"
( timeout 900 python3 -m pytest tests/test_gpu_z_bench_path.py tests/test_gpu_k_stream_parity.py tests/test_gpu_d_pipeline.py -x -q -m gpu -p no:cacheprovider --durations=5 > "$OUT/pytest.log" 2>&1; echo "exit $?" >> "$OUT/pytest.log" )
grep -v "Warning\|warnings.warn\|^$\|_create_method\|amdgpu.ids" "$OUT/pytest.log" | tail -10 | cut -c1-400
echo "total $(( $(date +%s) - t0 )) s"
