#!/usr/bin/env bash
# measurement for the next round (configs[1]): per-op tables of the detector plan at batch 1 and 8, 640x640, and at 1088x1920
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r4_detprof
for cfg in "1 640" "8 640" "1 native"; do
  set -- $cfg
  ( BATCH=$1 IMGSZ=$2 timeout 200 python tools/profile_plan.py > gpurun_out/r4_detprof/detector_b$1_$2.txt 2>&1; echo "b$1 $2 exit $?" )
  head -30 gpurun_out/r4_detprof/detector_b$1_$2.txt | grep -v Warning | cut -c1-120
done
