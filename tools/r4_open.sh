#!/usr/bin/env bash
# Round-4 opening GPU session (one gpurun call, ~30 GPU-minutes).  Round 3 ended with three things unmeasured:
#   A. three candidate kernels that have only run on the host emulation (window attention v2, channel-attention apply on the matrix
#      pipe, encoder attention v2: Florence2Captioner.window_attn_v2 / chan_apply_mfma / mha_v2) -> hardware parity, per-op A/B, bench A/B
#      + written later, same status: PlanBuilder.fuse_splitk (split-K without the reduce launch), Florence2Captioner.reuse_activations
#      (scratch-tensor reuse), CU-partitioned encode lanes (tools/cu_mask_probe.py, bench.py --lane-masks)
#   B. the hardware-queue cliff (one encode lane + the decode stream at GPU_MAX_HW_QUEUES=8: 1034 instead of 709 ms per step)
#      -> kernel trace of the slow and the fast case, per-queue gap summary (tools/hwq_gaps.py)
#   C. the suite and the default bench line at the round-3 final commit on a fresh box (what the driver recorded at round end)
# usage: gpurun --timeout 1700 -- 'SECTIONS="A4 A1 A2b" bash tools/r4_open.sh'   then   gpurun --timeout 1700 -- 'SECTIONS="A2 A3 B C" bash tools/r4_open.sh'
#        (A4 first: the CU-partition benches are the experiment with the largest expected effect)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r4open
mkdir -p "$OUT"
python tools/make_weights.py --ensure detector > /dev/null 2>&1
python tools/make_weights.py --ensure caption > /dev/null 2>&1
# SECTIONS="A1 A4 C" runs only those sections (default: all, ~40 GPU-minutes: split it over two calls)
want() { [[ -z "${SECTIONS:-}" || " ${SECTIONS} " == *" $1 "* ]]; }
if want A1; then
echo "=== A1. candidate kernels on hardware: kernel checks + real 768x768 crops against transformers"
( timeout 600 python tools/r4_candidates.py > "$OUT/candidates.json" 2> "$OUT/candidates.err"; echo "exit $?" )
python - "$OUT/candidates.json" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("all_passed", d.get("all_passed"))
    for k, v in d.items():
        if isinstance(v, dict):
            print("  ", k, {a: (round(b, 8) if isinstance(b, float) else b) for a, b in v.items()})
except Exception as e:
    print("no report:", e)
PY
fi
if want A2; then
echo "=== A2. per-op profile of one 128-crop plan: default, each candidate, both"
for f in "" "window_attn_v2" "chan_apply_mfma" "mha_v2" "window_attn_v2,chan_apply_mfma,mha_v2"; do
  tag=${f:-default}; tag=${tag//,/+}
  ( timeout 200 python tools/caption_profile.py 128 768 2 $f > "$OUT/per_op_$tag.json" 2> "$OUT/per_op_$tag.txt"; echo "$tag exit $?" )
  grep -E "^--- encode|attn_rows|chan_attn" "$OUT/per_op_$tag.txt" | head -12
done
fi
if want A2b; then
echo "=== A2b. batch-1 detector (configs[1], launch-bound): default vs fuse_splitk (no splitk_reduce launches)"
for f in "" "fuse_splitk"; do
  tag=det_${f:-default}
  ( timeout 120 python bench.py --mode detect --steps 200 --warmup 20 --no-cpu-baseline --no-extra ${f:+--candidates $f} > "$OUT/bench_$tag.json" 2> "$OUT/bench_$tag.err"; echo "$tag exit $?" )
  python - "$OUT/bench_$tag.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("   ", d["value"], d["unit"], d["ms_per_step"], "ms per screenshot")
except Exception as e:
    print("no result:", e)
PY
done
fi
if want A3; then
echo "=== A3. bench A/B (K = 6): default, all candidates"
for f in "" "window_attn_v2,chan_apply_mfma,mha_v2" "reuse_activations" "fuse_splitk"; do   # reuse_activations: same kernels on aliased scratch (config.hbm_peak_allocated_gb)
  tag=${f:-default}; tag=${tag//,/+}
  ( OMNI_BENCH_WATCHDOG=120 timeout 240 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra ${f:+--candidates $f} > "$OUT/bench_$tag.json" 2> "$OUT/bench_$tag.err"; echo "$tag exit $?" )
  python - "$OUT/bench_$tag.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("   ", d["value"], "screenshots/s", d["ms_per_step"], "ms/step; gemm", r.get("gemm_ms_per_step"), "non-gemm share", r.get("non_gemm_share"),
          "HBM peak", d["config"].get("hbm_peak_allocated_gb"), "GB")
    print("   ", r.get("kernel_family_ms_per_step"))
except Exception as e:
    print("    no line:", e)
PY
done
fi
if want A4; then
echo "=== A4. CU-partitioned lanes (experiment): probe (mask mapping, GEMM / LayerNorm scaling with the CU count, graph vs eager under a mask,"
echo "        GEMM queue beside a LayerNorm queue on disjoint CU sets), then the bench with the two encode lanes on disjoint halves"
( timeout 150 python tools/cu_mask_probe.py > "$OUT/cu_mask_probe.json" 2> "$OUT/cu_mask_probe.err"; echo "probe exit $?" )
python - "$OUT/cu_mask_probe.json" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    for sec in ("single", "graph", "concurrent", "two_mixed_lanes"):
        for k, v in d[sec].items():
            print("   ", sec, k, v)
except Exception as e:
    print("no report:", e)
PY
# contiguous ranges: every XCD keeps a share (profiles/r3_cu_mask_probe.md).  Third set = the decode stream: "0-191;0-191;192-255" keeps both
# encode lanes on 192 CUs (95.5 % of the GEMM rate) and gives the decode steps 64 CUs of their own instead of queueing behind 9 216-block launches
for m in "0-127;128-255" "0-159;160-255" "0-191;0-191;192-255"; do
  tag=lanes_$(echo "$m" | tr ';:-' '___')
  ( OMNI_BENCH_WATCHDOG=120 timeout 240 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra --lane-masks "$m" > "$OUT/bench_$tag.json" 2> "$OUT/bench_$tag.err"; echo "$tag exit $?" )
  python - "$OUT/bench_$tag.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("   ", d["config"].get("lane_cu_masks"), d["value"], "screenshots/s", d["ms_per_step"], "ms/step")
except Exception as e:
    print("no result:", e)
PY
done
echo "        ... and the split replay: MFMA-bound ops of every encode plan on a GEMM CU set, the rest on the other set (two lanes, eager)"
for m in "0-191;192-255" "0-175;176-255" "0-159;160-255" "0-175;176-255;176-255"; do      # 4th: the decode stream on the "other" set too
  tag=split_$(echo "$m" | tr ';:-' '___')
  ( OMNI_BENCH_WATCHDOG=120 timeout 240 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra --split-masks "$m" > "$OUT/bench_$tag.json" 2> "$OUT/bench_$tag.err"; echo "$tag exit $?" )
  python - "$OUT/bench_$tag.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("   ", d["config"].get("split_cu_masks"), d["value"], "screenshots/s", d["ms_per_step"], "ms/step")
except Exception as e:
    print("no result:", e)
PY
done
fi
if want B; then
echo "        ... more micro-batches in flight: the GEMM stream idles only while NO lane has a GEMM ready (closed two-server queue: ~0.8 busy with 2 lanes, ~0.9 with 3-4)"
for l in 3 4; do
  tag=split_lanes$l
  ( OMNI_BENCH_WATCHDOG=120 timeout 240 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra --lanes $l --split-masks "0-175;176-255" --candidates reuse_activations > "$OUT/bench_$tag.json" 2> "$OUT/bench_$tag.err"; echo "$tag exit $?" )
  python - "$OUT/bench_$tag.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("   ", d["config"].get("split_cu_masks"), d["config"].get("pipeline", "")[-60:], d["value"], "screenshots/s", d["ms_per_step"], "ms/step, HBM", d["config"].get("hbm_peak_allocated_gb"), "GB")
except Exception as e:
    print("no result:", e)
PY
done
echo "=== B. hardware-queue cliff: kernel traces of --lanes 1 at 4 and 8 hardware queues (K = 3), gap summary per queue"
for q in 4 8; do
  ( GPU_MAX_HW_QUEUES=$q OMNI_BENCH_WATCHDOG=120 timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace_hwq$q" -- \
      python bench.py --steps 3 --warmup 2 --lanes 1 --no-cpu-baseline --no-extra > "$OUT/bench_lanes1_hwq$q.json" 2> "$OUT/bench_lanes1_hwq$q.err"; echo "hwq$q exit $?" )
  f=$(find "$OUT/trace_hwq$q" -name "*kernel_trace.csv" | head -1)
  python tools/hwq_gaps.py "$f" > "$OUT/hwq_gaps_$q.json" 2> "$OUT/hwq_gaps_$q.err"; head -c 1500 "$OUT/hwq_gaps_$q.json"; echo
  find "$OUT/trace_hwq$q" -name "*.csv" -size +4M -delete; find "$OUT/trace_hwq$q" -name "*.db" -delete
done
fi
if want C; then
echo "=== C. pytest tests/ -x -q -m gpu (one process, as the driver runs it) and the default bench line"
t0=$(date +%s)
( timeout 1200 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider --durations=10 > "$OUT/pytest.log" 2>&1; echo "exit $?" >> "$OUT/pytest.log" )
echo "($(( $(date +%s) - t0 )) s)"; grep -v "Warning\|warnings.warn\|^$\|_create_method\|amdgpu.ids" "$OUT/pytest.log" | tail -16 | cut -c1-300
( OMNI_BENCH_WATCHDOG=200 timeout 900 python bench.py > "$OUT/bench_full.json" 2> "$OUT/bench_full.err"; echo "bench exit $?" )
tail -c 1200 "$OUT/bench_full.json"; echo
ls -la "$OUT" | head -40
fi
