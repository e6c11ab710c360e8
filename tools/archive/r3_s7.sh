#!/usr/bin/env bash
# Round-3 GPU session 7: GEMM schedule 2 (barrier in front of the last unit) and the 4-wave 256x256 tile — correctness + micro-benchmark; C-API test after the crop clip.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r3s7
mkdir -p "$OUT"
python tools/make_weights.py --ensure detector > /dev/null 2>&1
python tools/make_weights.py --ensure caption > /dev/null 2>&1
echo "=== 1. kernel tests (default, SCHED=2)"
for v in "" "OMNI_GEMM_SCHED=2"; do
  ( env $v timeout 300 python -m pytest tests/test_gpu_a_kernels.py -q -m gpu -p no:cacheprovider -x -k "gemm" > "$OUT/kern_${v:-default}.log" 2>&1; echo "exit $?" >> "$OUT/kern_${v:-default}.log" )
  grep "passed\|failed\|exit" "$OUT/kern_${v:-default}.log" | tail -2
done
echo "=== 2. gemm bench"
( SHAPES=s2.fc1,s2.fc2,s2.qkv,s2.proj,enc.fc1,s3.fc1,s1.fc1,s0.fc2 VARIANTS="dma,dma+OMNI_GEMM_SCHED=2,dma:256x256w4,dma:256x256w4+OMNI_GEMM_SCHED=2,dma+OMNI_GEMM_SCHED=2,dma" timeout 400 python tools/gemm_bench.py > "$OUT/gemm_bench.txt" 2> "$OUT/gemm_bench.err"; echo "exit $?" )
cat "$OUT/gemm_bench.txt"
echo "=== 3. capi"
( timeout 300 python -m pytest tests/test_gpu_i_model_capi.py -q -m gpu -p no:cacheprovider -x > "$OUT/capi.log" 2>&1; echo "exit $?" >> "$OUT/capi.log" )
grep "passed\|failed\|exit" "$OUT/capi.log" | tail -2
