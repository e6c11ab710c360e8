#!/usr/bin/env bash
# round-6 session 12: decode-step autotune (tools/decode_autotune.py --write), the mixed-resolution stream with the ladder / exact twin
# policy, the tests that exercise both (stream parity, benched path), bench A/B of the decode table
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s12
mkdir -p "$OUT"
t0=$(date +%s)
echo "=== 1. decode autotune"
( timeout 600 python3 tools/decode_autotune.py --write > "$OUT/decode_autotune.json" 2> "$OUT/decode_autotune.err"; echo "exit $?" )
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
echo "($(( $(date +%s) - t0 )) s)"
echo "=== 2. stream bench (r768, as bench.py's extra runs it)"
for rep in 1 2; do
( timeout 300 python3 tools/stream_bench.py --items 24 --caption-res 768 > "$OUT/stream_r768_$rep.json" 2> "$OUT/stream_r768_$rep.err"; echo "exit $?" )
tail -1 "$OUT/stream_r768_$rep.json" | cut -c1-400
done
echo "=== 3. tests"
( timeout 900 python3 -m pytest tests/test_gpu_k_stream_parity.py tests/test_gpu_z_bench_path.py tests/test_gpu_b_caption_model.py -x -q -m gpu -p no:cacheprovider --durations=5 > "$OUT/pytest.log" 2>&1; echo "exit $?" >> "$OUT/pytest.log" )
grep -v "Warning\|warnings.warn\|^$\|_create_method\|amdgpu.ids" "$OUT/pytest.log" | tail -10 | cut -c1-400
echo "=== 4. bench A/B of the decode table"
showe() {
python3 - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], "TF/s", r["achieved"], r["frac"], "gemm", r["gemm_ms_per_step"], "sum", r["profiled_step_ms"], "decode", r["parts"].get("decode_rows352"), {k: v.get("encode_gemm_ms") for k, v in r["parts"].items() if k.startswith("caption")})
except Exception as e:
    print(sys.argv[1], "no bench line", e)
PY
}
for arm in 0 1 0 1; do
tag="dt${arm}_$(date +%s)"
( OMNI_DECODE_TUNING=$arm OMNI_BENCH_WATCHDOG=400 timeout 600 python3 bench.py --gpus 1 --steps 10 --warmup 5 --no-cpu-baseline --no-extra > "$OUT/bench_$tag.json" 2> "$OUT/bench_$tag.err"; echo "exit $?" >> "$OUT/bench_$tag.err" )
showe "$OUT/bench_$tag.json"
done
echo "total $(( $(date +%s) - t0 )) s"
