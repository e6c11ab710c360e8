#!/usr/bin/env bash
# Round-3 GPU session 5: merged decode, MFMA channel-attention apply, window attention v3 — tests, per-op profile, bench, A/B of the merge.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r3s5
mkdir -p "$OUT"
python tools/make_weights.py --ensure detector > /dev/null 2>&1
python tools/make_weights.py --ensure caption > /dev/null 2>&1
echo "=== 1. tests"
for f in tests/test_gpu_b_caption_model.py tests/test_gpu_g_device_handoff.py tests/test_gpu_i_model_capi.py tests/test_gpu_d_pipeline.py; do
  n=$(basename "$f" .py)
  ( timeout 600 python -m pytest "$f" -q -m gpu -p no:cacheprovider -x -s > "$OUT/$n.log" 2>&1; echo "exit $?" >> "$OUT/$n.log" )
  echo "--- $n"; grep -v "Warning\|warnings.warn\|^$\|_create_method\|^tests/test_gpu\|amdgpu.ids" "$OUT/$n.log" | grep "passed\|failed\|Error\|tokens_match\|exit" | tail -6 | cut -c1-600
done
echo "=== 2. per-op profile"
( timeout 300 python tools/caption_profile.py 128 768 2 > "$OUT/prof_default.json" 2> "$OUT/prof_default.txt"; echo "exit $?" )
grep "attn_rows\|chan_attn\|dwconv3_ln\|^---" "$OUT/prof_default.txt" | cut -c1-150
echo "=== 3. bench (default) and with per-micro-batch decode"
for v in "" "OMNI_MERGED_DECODE=0"; do
  tag=${v:-default}
  ( env $v OMNI_BENCH_WATCHDOG=120 timeout 420 python bench.py --steps 5 --warmup 2 --no-extra --no-cpu-baseline > "$OUT/bench_$tag.json" 2> "$OUT/bench_$tag.err"; echo "exit $?" >> "$OUT/bench_$tag.err" )
  tail -1 "$OUT/bench_$tag.err"; python - "$OUT/bench_$tag.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["roofline"]["kernel_family_ms_per_step"], d["roofline"]["achieved"], d["roofline"].get("parts", {}).keys())
except Exception as e:
    print("no bench line", e)
PY
done
