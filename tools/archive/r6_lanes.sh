cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; OUT=gpurun_out/r6_lanes; mkdir -p $OUT
for arm in 2 1 2 1; do
tag="lanes${arm}_$(date +%s)"
( OMNI_BENCH_WATCHDOG=400 timeout 600 python3 bench.py --gpus 1 --steps 10 --warmup 5 --lanes $arm --no-cpu-baseline --no-extra > "$OUT/bench_$tag.json" 2> "$OUT/bench_$tag.err"; echo "exit $?" >> "$OUT/bench_$tag.err" )
python3 - "$OUT/bench_$tag.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
    print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], "sum", r["profiled_step_ms"], "hbm", d["config"].get("hbm_peak_allocated_gb"), d["config"].get("step_wall_ms")[:6])
except Exception as e:
    print("no line", e)
PY
done
