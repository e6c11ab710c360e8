#!/usr/bin/env bash
# Round-3 GPU session 2: the rewritten non-GEMM caption kernels (strip dwconv+LN, latency-oriented window attention, MFMA channel
# attention, four-wave cross decode attention), device hand-off as the default — per-op profile new vs round-2 composition, tests, bench.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r3s2
mkdir -p "$OUT"
python tools/make_weights.py --ensure detector > /dev/null 2>&1
python tools/make_weights.py --ensure caption > /dev/null 2>&1
echo "=== 1. per-op profile, 128-row plan at 768x768: default"
( timeout 300 python tools/caption_profile.py 128 768 2 > "$OUT/prof_default.json" 2> "$OUT/prof_default.txt"; echo "exit $?" )
grep -v "Warn\|warn" "$OUT/prof_default.txt" | head -60 | cut -c1-160
echo "=== 1b. same, round-2 composition (old window / channel / decode attention kernels, separate dwconv + LN, f32 attention output)"
( OMNI_WINDOW_ATTN=1 OMNI_CHAN_ATTN=1 OMNI_DECODE_ATTN=1 OMNI_FUSE_DWLN=0 OMNI_ATTN_SPLIT_OUT=0 timeout 300 python tools/caption_profile.py 128 768 2 > "$OUT/prof_r2.json" 2> "$OUT/prof_r2.txt"; echo "exit $?" )
grep -v "Warn\|warn" "$OUT/prof_r2.txt" | grep -v "conv_igemm" | head -40 | cut -c1-160
python - <<'PY'
import json
for n in ("default", "r2"):
    try:
        d = json.load(open(f"gpurun_out/r3s2/prof_{n}.json"))
        print(n, {k: (v["total_ms"], v["by_family_ms"]) for k, v in d["plans"].items()})
    except Exception as e:
        print(n, "no profile:", e)
PY
echo "=== 2. tests"
for f in tests/test_gpu_b_caption_model.py tests/test_gpu_g_device_handoff.py tests/test_gpu_d_pipeline.py; do
  n=$(basename "$f" .py)
  ( timeout 600 python -m pytest "$f" -q -m gpu -p no:cacheprovider -x > "$OUT/$n.log" 2>&1; echo "exit $?" >> "$OUT/$n.log" )
  echo "--- $n"; grep -v "Warning\|warnings.warn\|^$\|_create_method\|^tests/test_gpu\|amdgpu.ids" "$OUT/$n.log" | tail -6 | cut -c1-1600
done
echo "=== 3. bench"
( OMNI_BENCH_WATCHDOG=120 timeout 420 python bench.py --steps 5 --warmup 2 --no-extra --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "exit $?" >> "$OUT/bench.err" )
tail -3 "$OUT/bench.err" | cut -c1-300; cut -c1-400 "$OUT/bench.json"; python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r3s2/bench.json").read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["roofline"]["kernel_family_ms_per_step"], d["roofline"]["achieved"])
except Exception as e:
    print("no bench line", e)
PY
