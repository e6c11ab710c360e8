set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_stress
mkdir -p "$OUT"
for i in 1 2 3 4 5 6; do
  ( timeout 600 python3 -m pytest tests/test_gpu_d_pipeline.py tests/test_gpu_c_detector.py -x -q -m gpu -p no:cacheprovider -k "tiled" > "$OUT/tiled_$i.log" 2>&1; echo "tiled run $i exit $?" ); tail -1 "$OUT/tiled_$i.log" | cut -c1-120
done
for i in 1 2 3; do
  ( timeout 600 python3 -m pytest tests/test_gpu_a_kernels.py -x -q -m gpu -p no:cacheprovider -k "schedules or gemm_dma" > "$OUT/gemm_$i.log" 2>&1; echo "gemm run $i exit $?" ); tail -1 "$OUT/gemm_$i.log" | cut -c1-120
done
