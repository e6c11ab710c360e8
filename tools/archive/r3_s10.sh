#!/usr/bin/env bash
# Round-3 GPU session 10: the fused stage-0 FFN kernel (OMNI_OP_MLP_FUSED) — kernel test, caption tests, per-op profile with and
# without it; A/B of the window-attention / channel-apply kernels against their session-3 versions (OMNI_AB, temporary); the new
# tests (tiled 4K captions, pipelined stream inside the bench-path test, stage_ms on the batch route); bench with 4 vs 8 HW queues.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r3s10
mkdir -p "$OUT"
python tools/make_weights.py --ensure detector > /dev/null 2>&1
python tools/make_weights.py --ensure caption > /dev/null 2>&1
echo "=== 1. tests"
for f in tests/test_gpu_a_kernels.py tests/test_gpu_b_caption_model.py tests/test_gpu_d_pipeline.py tests/test_gpu_g_device_handoff.py tests/test_gpu_h_service.py tests/test_gpu_z_bench_path.py; do
  n=$(basename "$f" .py)
  t0=$(date +%s)
  ( timeout 900 python -m pytest "$f" -q -m gpu -p no:cacheprovider -x -s > "$OUT/$n.log" 2>&1; echo "exit $?" >> "$OUT/$n.log" )
  echo "--- $n ($(( $(date +%s) - t0 )) s)"; grep "passed\|failed\|skipped\|^exit\|Error\|^{" "$OUT/$n.log" | tail -5 | cut -c1-700
done
echo "=== 2. per-op profile: default, two-launch FFN, session-3 attention kernels"
for v in "default:" "nomlp:nomlp" "ab:"; do
  tag=${v%%:*}; flags=${v#*:}
  ab=""; [ "$tag" = "ab" ] && ab="win_s3,chan_valu"
  ( OMNI_AB="$ab" timeout 300 python tools/caption_profile.py 128 768 2 $flags > "$OUT/prof_$tag.json" 2> "$OUT/prof_$tag.txt"; echo "exit $?" )
  echo "--- $tag"; grep "^---\|mlp_fused\|4718592\|attn_rows\|chan_attn" "$OUT/prof_$tag.txt" | cut -c1-150 | head -24
done
echo "=== 3. bench (pipelined default): 4 HW queues (runtime default) vs 8"
for v in "" "GPU_MAX_HW_QUEUES=8"; do
  tag=${v:-default}
  ( env $v OMNI_BENCH_WATCHDOG=120 timeout 420 python bench.py --steps 6 --warmup 2 --no-extra --no-cpu-baseline > "$OUT/bench_$tag.json" 2> "$OUT/bench_$tag.err"; echo "exit $?" >> "$OUT/bench_$tag.err" )
  tail -1 "$OUT/bench_$tag.err"; python - "$OUT/bench_$tag.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(d["value"], d["ms_per_step"], d["config"].get("steps_pipelined"), r["achieved"], r["gemm_ms_per_step"], r["profiled_step_ms"], r["kernel_family_ms_per_step"])
except Exception as e:
    print("no bench line", e)
PY
done
