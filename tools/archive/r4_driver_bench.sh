#!/usr/bin/env bash
# the driver's bench command as the FIRST process on a fresh lease (what BENCH_rNN.json records), then once more (warm box)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r4_driver_bench
for k in first second; do
  ( timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_driver_bench/bench_$k.json 2> gpurun_out/r4_driver_bench/bench_$k.err; echo "$k exit $?" )
  python - gpurun_out/r4_driver_bench/bench_$k.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["steps"], d["config"].get("step_wall_ms"))
except Exception as e:
    print("no line", e)
PY
done
