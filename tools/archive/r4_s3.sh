#!/usr/bin/env bash
# round-4 session 3: (0) lease log: first test file + smoke at the adopted composition (window / MHA / channel-apply MFMA kernels, activation
# reuse default; losers removed); (1) persistent-block GEMM (OMNI_GEMM_GRID): parity of the persistent instantiations, then bench A/B
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4_s3
mkdir -p $O
( timeout 600 python3 -m pytest tests/test_gpu_a_kernels.py -x -q -m gpu -p no:cacheprovider > $O/pytest_a.txt 2>&1; echo "rc=$?" >> $O/pytest_a.txt )
grep -v "Warning\|warnings.warn\|^$" $O/pytest_a.txt | tail -4
( timeout 300 python3 -c 'import __graft_entry__ as e; e.smoke()' > $O/smoke.txt 2>&1; echo "rc=$?" >> $O/smoke.txt )
grep "rc=\|smoke OK" $O/smoke.txt | cut -c1-120
( OMNI_GEMM_GRID=192 timeout 300 python3 -m pytest tests/test_gpu_a_kernels.py -x -q -m gpu -p no:cacheprovider -k "gemm_dma or mlp_fused" > $O/pytest_persist192.txt 2>&1; echo "rc=$?" >> $O/pytest_persist192.txt )
grep -v "Warning\|warnings.warn\|^$" $O/pytest_persist192.txt | tail -3
( OMNI_GEMM_GRID=8 timeout 300 python3 -m pytest tests/test_gpu_a_kernels.py -x -q -m gpu -p no:cacheprovider -k "gemm_dma" > $O/pytest_persist8.txt 2>&1; echo "rc=$?" >> $O/pytest_persist8.txt )
grep -v "Warning\|warnings.warn\|^$" $O/pytest_persist8.txt | tail -3
run_bench() {  # tag, env...
  tag=$1; shift
  ( env "$@" OMNI_BENCH_WATCHDOG=120 timeout 240 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra ${EXTRA:-} > "$O/bench_$tag.json" 2> "$O/bench_$tag.err"; echo "$tag exit $?" )
  python - "$O/bench_$tag.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("   ", d["value"], "screenshots/s", d["ms_per_step"], "ms/step; gemm", r.get("gemm_ms_per_step"), "non-gemm share", r.get("non_gemm_share"),
          "HBM peak", d["config"].get("hbm_peak_allocated_gb"), "GB")
except Exception as e:
    print("    no line:", e)
PY
}
run_bench default A=1
for g in 256 224 192 160 128; do run_bench grid$g OMNI_GEMM_GRID=$g; done
EXTRA="--lanes 3" run_bench grid192_lanes3 OMNI_GEMM_GRID=192
EXTRA="--lanes 3" run_bench default_lanes3 A=1
run_bench grid192_hwq8 OMNI_GEMM_GRID=192 GPU_MAX_HW_QUEUES=8
