#!/usr/bin/env bash
# Round-3 opening GPU session (one gpurun call, ~25 GPU-minutes): everything round 2 left unmeasured.
#   1. default bench line (the v4 stand-in has never run on a GPU: expect ~8.4 screenshots/s at 43.6 crops per screenshot)
#   2. whole GPU suite, file by file so that one failure does not hide the rest
#   3. A/B of the opt-in format-B producers (attention / channel attention / fused dwconv+LN write split output)
#   4. rocprofv3 kernel stats and PMC traffic of the SAME bench command at the final load -> copy summaries to profiles/r3_*
# usage: gpurun --timeout 1700 -- 'bash tools/r3_open.sh'
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r3open
mkdir -p "$OUT"
python tools/make_weights.py --ensure detector > /dev/null 2>&1
python tools/make_weights.py --ensure caption > /dev/null 2>&1
echo "=== 1. default bench line"
( OMNI_BENCH_WATCHDOG=120 timeout 420 python bench.py --steps 5 --warmup 2 > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "exit $?" >> "$OUT/bench.err" )
grep -v "^  File\|^Thread" "$OUT/bench.err" | tail -6 | cut -c1-300
echo "=== 2. GPU suite"
for f in tests/test_gpu_a_kernels.py tests/test_gpu_b_caption_model.py tests/test_gpu_c_detector.py tests/test_gpu_d_pipeline.py tests/test_gpu_f_overlay_png.py tests/test_gpu_g_device_handoff.py tests/test_gpu_h_service.py; do
  n=$(basename "$f" .py)
  ( timeout 600 python -m pytest "$f" -q -p no:cacheprovider --durations=5 > "$OUT/$n.log" 2>&1; echo "exit $?" >> "$OUT/$n.log" )
  echo "--- $n"; grep -v "Warning\|warnings.warn\|^$\|_create_method\|^tests/test_gpu" "$OUT/$n.log" | tail -12 | cut -c1-1200
done
echo "=== 3. opt-in kernels, the power-of-two caption plan ladder (default now has a 96-row plan), pipelined steps"
( OMNI_ATTN_SPLIT_OUT=1 OMNI_FUSE_DWLN=1 timeout 240 python -m pytest tests/test_gpu_b_caption_model.py -q -p no:cacheprovider -k "r64 or r768" 2>&1 | tail -3 | cut -c1-400 )
for v in "OMNI_ATTN_SPLIT_OUT=1" "OMNI_FUSE_DWLN=1" "OMNI_ATTN_SPLIT_OUT=1 OMNI_FUSE_DWLN=1" "OMNI_DECODE_ATTN=2" "OMNI_CAPTION_BUCKETS=8,16,32,64,128" "OMNI_BENCH_FLAGS=--pipeline"; do
  tag=$(echo "$v" | tr ' =' '__')
  extra_flags=""; case "$v" in OMNI_BENCH_FLAGS=*) extra_flags="${v#OMNI_BENCH_FLAGS=}";; esac
  ( env $v OMNI_BENCH_WATCHDOG=60 timeout 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra $extra_flags > "$OUT/ab_$tag.json" 2> "$OUT/ab_$tag.err"; echo "$v -> exit $?" )
  python - "$OUT/ab_$tag.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("   ", d["value"], "screenshots/s", d["ms_per_step"], "ms/step; gemm", r.get("gemm_ms_per_step"), "non-gemm share", r.get("non_gemm_share"))
except Exception as e:
    print("    no line:", e)
PY
done
echo "=== 4. kernel stats + PMC traffic of the bench command (same pattern as tools/r2_measure.sh)"
( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > "$OUT/stats_bench.json" 2> "$OUT/stats.err"; echo "stats exit $?" )
for ctr in FETCH_SIZE WRITE_SIZE; do
  ( timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$OUT/pmc_$ctr" -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra > /dev/null 2> "$OUT/pmc_$ctr.err"; echo "pmc $ctr exit $?" )
  python tools/pmc_summary.py "$OUT/pmc_$ctr" > "$OUT/pmc_$ctr.json" 2>> "$OUT/pmc_$ctr.err"
done
echo "=== 4b. annotate / PNG tail: host vs device (tools/annotate_bench.py)"
( timeout 120 python tools/annotate_bench.py > "$OUT/annotate_tail.json" 2> "$OUT/annotate_tail.err"; echo "annotate exit $?"; cat "$OUT/annotate_tail.json" | cut -c1-900 )
echo "=== 5. device hand-off (eager detector plan) A/B — last: the graph variant of this path stalled in round 2"
( OMNI_DEVICE_GLUE=1 OMNI_BENCH_WATCHDOG=40 timeout 90 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > "$OUT/ab_device_glue.json" 2> "$OUT/ab_device_glue.err"; echo "device glue -> exit $?" )
tail -c 600 "$OUT/ab_device_glue.json"; echo
echo "=== 6. device hand-off inside the detector graph (OMNI_DEVICE_GLUE=2: one graph, no launch between replays) — the very last GPU work of the call"
( OMNI_DEVICE_GLUE=2 OMNI_BENCH_WATCHDOG=40 timeout 90 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > "$OUT/ab_device_glue2.json" 2> "$OUT/ab_device_glue2.err"; echo "device glue 2 -> exit $?" )
tail -c 600 "$OUT/ab_device_glue2.json"; echo
echo "=== 7. the stalling arrangement with ROCm's graph packet capture off (detector graph + eager hand-off kernels) — may stall: 60 s limit"
( OMNI_DEVICE_GLUE=1 OMNI_DEVICE_GLUE_GRAPH=1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 OMNI_BENCH_WATCHDOG=30 timeout 60 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > "$OUT/ab_device_glue_nocapture.json" 2> "$OUT/ab_device_glue_nocapture.err"; echo "device glue, graph, packet capture off -> exit $?" )
tail -c 300 "$OUT/ab_device_glue_nocapture.json"; echo
find "$OUT" -name "*_kernel_stats.csv" | head -3
find "$OUT" -name "*.csv" -size +8M -delete
find "$OUT" -name "*.db" -delete
ls -la "$OUT" | head -40
