#!/usr/bin/env bash
# Round-3 GPU session 1: bisect of the red benched-path test, then the whole suite file by file, then the bench line.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r3s1
mkdir -p "$OUT"
python tools/make_weights.py --ensure detector > /dev/null 2>&1
python tools/make_weights.py --ensure caption > /dev/null 2>&1
echo "=== 1. bisect"
( timeout 600 python tools/r3_bisect.py v2 v1 > "$OUT/bisect.jsonl" 2> "$OUT/bisect.err"; echo "exit $?" >> "$OUT/bisect.err" )
cut -c1-1500 "$OUT/bisect.jsonl"; tail -3 "$OUT/bisect.err" | cut -c1-400
echo "=== 2. GPU suite, file by file"
for f in tests/test_gpu_a_kernels.py tests/test_gpu_b_caption_model.py tests/test_gpu_c_detector.py tests/test_gpu_d_pipeline.py tests/test_gpu_e_dist.py tests/test_gpu_f_overlay_png.py tests/test_gpu_g_device_handoff.py tests/test_gpu_h_service.py tests/test_gpu_z_bench_path.py; do
  n=$(basename "$f" .py)
  ( timeout 900 python -m pytest "$f" -q -m gpu -p no:cacheprovider --durations=5 -s > "$OUT/$n.log" 2>&1; echo "exit $?" >> "$OUT/$n.log" )
  echo "--- $n"; grep -v "Warning\|warnings.warn\|^$\|_create_method\|^tests/test_gpu\|amdgpu.ids" "$OUT/$n.log" | tail -8 | cut -c1-1600
done
echo "=== 3. bench"
( OMNI_BENCH_WATCHDOG=120 timeout 420 python bench.py --steps 5 --warmup 2 --no-extra --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "exit $?" >> "$OUT/bench.err" )
tail -3 "$OUT/bench.err" | cut -c1-300; cut -c1-3000 "$OUT/bench.json"
echo "=== 4. device hand-off behind graphs (bounded: the round-2 arrangement stalled)"
( OMNI_DEVICE_GLUE=2 OMNI_GRAPH_DOT=$OUT/graph_fused timeout 150 python tools/r3_glue_graph.py 100 > "$OUT/glue2.json" 2> "$OUT/glue2.err"; echo "glue=2 exit $?" )
tail -c 400 "$OUT/glue2.json"; tail -2 "$OUT/glue2.err" | cut -c1-200
( OMNI_DEVICE_GLUE=1 OMNI_DEVICE_GLUE_GRAPH=1 timeout 100 python tools/r3_glue_graph.py 100 > "$OUT/glue1g.json" 2> "$OUT/glue1g.err"; echo "glue=1 + detector graph exit $?" )
tail -c 400 "$OUT/glue1g.json"; tail -2 "$OUT/glue1g.err" | cut -c1-200
grep -c "MEMSET\|memset" $OUT/graph_fused*.dot | head -3
grep -o "label=\"[A-Za-z_]*" $OUT/graph_fused.0.dot | sort | uniq -c | sort -rn | head -8
rm -f $OUT/*.dot
