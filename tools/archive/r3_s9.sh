#!/usr/bin/env bash
# Round-3 GPU session 9: the whole -m gpu suite at HEAD (file by file, bench-path file last), then the bench sequential vs pipelined
# (parse_stream with the device hand-off: decode of batch i on a second stream while batch i+1 encodes), then the per-op profile.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r3s9
mkdir -p "$OUT"
python tools/make_weights.py --ensure detector > /dev/null 2>&1
python tools/make_weights.py --ensure caption > /dev/null 2>&1
echo "=== 1. tests"
for f in tests/test_gpu_[a-i]_*.py tests/test_gpu_z_bench_path.py; do
  n=$(basename "$f" .py)
  t0=$(date +%s)
  ( timeout 900 python -m pytest "$f" -q -m gpu -p no:cacheprovider -x > "$OUT/$n.log" 2>&1; echo "exit $?" >> "$OUT/$n.log" )
  echo "--- $n ($(( $(date +%s) - t0 )) s)"; grep "passed\|failed\|skipped\|^exit\|Error" "$OUT/$n.log" | tail -4 | cut -c1-400
done
echo "=== 2. bench: sequential, pipelined"
for v in "" "--pipeline --lanes 1" "--pipeline"; do
  tag=${v:-sequential}; tag=${tag#--}; tag=${tag// /_}
  ( OMNI_BENCH_WATCHDOG=120 timeout 420 python bench.py --steps 6 --warmup 2 --no-extra --no-cpu-baseline $v > "$OUT/bench_$tag.json" 2> "$OUT/bench_$tag.err"; echo "exit $?" >> "$OUT/bench_$tag.err" )
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
echo "=== 3. per-op profile"
( timeout 300 python tools/caption_profile.py 128 768 2 > "$OUT/prof_default.json" 2> "$OUT/prof_default.txt"; echo "exit $?" )
grep -v "Warn\|warn\|amdgpu.ids" "$OUT/prof_default.txt" | cut -c1-150 | head -70
