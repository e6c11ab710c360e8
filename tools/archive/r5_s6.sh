#!/usr/bin/env bash
# round-5 session 6: per-shape tile / split-K autotune of the detector's convolutions (tools/conv_autotune.py), then the detector parity
# tests and the detector-only bench lines with the table in place
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r5_s6
mkdir -p "$OUT"
( timeout 600 python tools/conv_autotune.py --write > "$OUT/autotune.json" 2> "$OUT/autotune.err"; echo "autotune exit $?" )
python - "$OUT/autotune.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k, v in d["plans"].items():
    print(k, {q: v[q] for q in ("ops", "tunable_convs", "shapes", "shapes_retuned", "graph_ms_heuristic", "graph_ms_tuned")})
print("choices", len(d["choices"]))
PY
grep -v Warning "$OUT/autotune.err" | tail -3 | cut -c1-300
for B in 1 8; do
  ( timeout 200 python bench.py --mode detect --batch $B --steps 200 --warmup 20 --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('tuned b$B', d['value'], d['ms_per_step'])" )
  ( OMNI_CONV_TUNING=0 timeout 200 python bench.py --mode detect --batch $B --steps 200 --warmup 20 --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('heuristic b$B', d['value'], d['ms_per_step'])" )
done
( timeout 900 python3 -m pytest tests/test_gpu_c_detector.py tests/test_gpu_a_kernels.py -x -q -m gpu -p no:cacheprovider > "$OUT/pytest_detector.log" 2>&1; echo "exit $?" >> "$OUT/pytest_detector.log" )
tail -4 "$OUT/pytest_detector.log" | cut -c1-300
