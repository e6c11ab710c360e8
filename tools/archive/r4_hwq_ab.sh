cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r4_hwq_ab
for q in 4 8 4 8; do
  ( GPU_MAX_HW_QUEUES=$q OMNI_BENCH_WATCHDOG=120 timeout 240 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra > gpurun_out/r4_hwq_ab/bench_hwq${q}_$RANDOM.json 2>/dev/null )
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r4_hwq_ab/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], d["config"]["hw_queues"], d["value"], d["ms_per_step"])
    except Exception as e: print(f, e)
PY
