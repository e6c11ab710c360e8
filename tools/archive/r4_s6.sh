#!/usr/bin/env bash
# round-4 session 6: (1) measured (not a-priori) GPU-vs-oracle identity over seeds 0..109 (oracle finals computed on the CPU beforehand:
# tools/_scan/oracle_finals.json), (2) power / clock samples under the GEMMs and an HBM-bound kernel, (3) synchronous kernel trace
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4_s6
mkdir -p $O
( timeout 600 python tools/scan_gpu_vs_oracle.py device > $O/scan_gpu_vs_oracle.json 2> $O/scan.err; echo "scan exit $?" )
head -c 1500 $O/scan_gpu_vs_oracle.json; echo
( timeout 200 python tools/power_trace.py > $O/power_trace.json 2> $O/power.err; echo "power exit $?" )
python - $O/power_trace.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("idle", d["idle"][:1])
    for k, v in d["cases"].items():
        print(k, v["rate"], v["unit"], v["ms_per_launch"], "ms;", len(v["samples"]), "samples", v["samples"][len(v["samples"]) // 2])
except Exception as e:
    print("no power trace:", e)
PY
( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_sync" -- python bench.py --steps 3 --warmup 2 --no-pipeline --no-cpu-baseline --no-extra > "$O/stats_sync_bench.json" 2> "$O/stats_sync.err"; echo "sync trace exit $?" )
f=$(find "$O/stats_sync" -name "*kernel_stats.csv" | head -1); cp "$f" "$O/kernel_stats_sync.csv" 2>/dev/null; head -8 "$O/kernel_stats_sync.csv" | cut -c1-160
find "$O/stats_sync" -name "*.csv" -size +4M -delete; find "$O/stats_sync" -name "*.db" -delete
