#!/usr/bin/env bash
# Round-3 GPU session 3: polynomial GELU + LDS-transposed (coalesced) GEMM epilogue + DMA issue schedule A/B; packed window attention.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r3s3
mkdir -p "$OUT"
python tools/make_weights.py --ensure detector > /dev/null 2>&1
python tools/make_weights.py --ensure caption > /dev/null 2>&1
echo "=== 1. GEMM micro-benchmark"
( VARIANTS="dma,dma+OMNI_GEMM_SCHED=1,dma:256x128" SHAPES="s2.fc1,s2.fc2,s2.qkv,s2.proj,s0.fc1,s0.fc2,s0.qkv,s1.fc1,enc.fc1" timeout 400 python tools/gemm_bench.py > "$OUT/gemm_bench.txt" 2>&1; echo "exit $?" )
grep -v "Warn\|warn\|amdgpu.ids" "$OUT/gemm_bench.txt" | cut -c1-140
echo "=== 2. kernel + caption tests"
for f in tests/test_gpu_a_kernels.py tests/test_gpu_b_caption_model.py; do
  n=$(basename "$f" .py)
  ( timeout 600 python -m pytest "$f" -q -m gpu -p no:cacheprovider -x > "$OUT/$n.log" 2>&1; echo "exit $?" >> "$OUT/$n.log" )
  echo "--- $n"; grep -v "Warning\|warnings.warn\|^$\|_create_method\|^tests/test_gpu\|amdgpu.ids" "$OUT/$n.log" | tail -6 | cut -c1-1600
done
echo "=== 3. per-op profile"
( timeout 300 python tools/caption_profile.py 128 768 2 > "$OUT/prof_default.json" 2> "$OUT/prof_default.txt"; echo "exit $?" )
grep -v "Warn\|warn\|amdgpu.ids" "$OUT/prof_default.txt" | head -45 | cut -c1-150
echo "=== 4. bench"
( OMNI_BENCH_WATCHDOG=120 timeout 420 python bench.py --steps 5 --warmup 2 --no-extra --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "exit $?" >> "$OUT/bench.err" )
tail -2 "$OUT/bench.err" | cut -c1-300; python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r3s3/bench.json").read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["roofline"]["kernel_family_ms_per_step"], d["roofline"]["achieved"])
except Exception as e:
    print("no bench line", e)
PY
