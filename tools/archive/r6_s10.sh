#!/usr/bin/env bash
# round-6 session 10 (session 9 repeated with the slim reducer: the first one took conv_split_kernel to 218 registers = 2 waves per SIMD in BOTH arms): in-launch split-K combine (OMNI_OP_CONV i24 / p6: write-through partials + arrival ticket, csrc/conv_igemm.hip) —
# kernel + detector + decode-path tests with it on, then A/B against the reduce launch (OMNI_SPLITK_COMBINE=0) within one lease:
# detector batch 1 / 8 / native (bench --mode detect), kernel trace of the batch-1 pass, then the e2e bench
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r6_s10
mkdir -p "$OUT"
t0=$(date +%s)
echo "=== 1. tests (combine on)"
( timeout 900 python3 -m pytest tests/test_gpu_a_kernels.py tests/test_gpu_c_detector.py tests/test_gpu_b_caption_model.py -x -q -m gpu -p no:cacheprovider --durations=5 > "$OUT/pytest.log" 2>&1; echo "exit $?" >> "$OUT/pytest.log" )
echo "($(( $(date +%s) - t0 )) s)"; grep -v "Warning\|warnings.warn\|^$\|_create_method\|amdgpu.ids" "$OUT/pytest.log" | tail -12 | cut -c1-400
showd() {
python3 - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], "TF/s", r["achieved"], "gemm_ms", r.get("gemm_ms_per_step"), "launches", r.get("gemm_launches_per_step"), "sum", r.get("profiled_step_ms"))
except Exception as e:
    print(sys.argv[1], "no bench line", e)
PY
}
echo "=== 2. detector A/B"
for rep in 1 2; do
for cb in 0 1; do
for cfg in "1 640" "8 640" "1 native"; do
set -- $cfg
( OMNI_SPLITK_COMBINE=$cb timeout 300 python3 bench.py --mode detect --batch $1 --imgsz $2 --steps 200 --warmup 20 --no-cpu-baseline --no-extra > "$OUT/det_c${cb}_b$1_$2_$rep.json" 2> "$OUT/det_c${cb}_b$1_$2_$rep.err"; echo "exit $?" >> "$OUT/det_c${cb}_b$1_$2_$rep.err" )
showd "$OUT/det_c${cb}_b$1_$2_$rep.json"
done
done
done
echo "=== 4. e2e A/B"
showe() {
python3 - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], "TF/s", r["achieved"], r["frac"], "gemm", r["gemm_ms_per_step"], "sum", r["profiled_step_ms"], "decode", r["parts"].get("decode_rows352"), "det", r["parts"].get("detector_batch8"), [k for k in r["per_kernel"] if k["shape_rows_n_k"][2] in (196, 224)], "enc128", r["parts"].get("caption_mb128_x2", {}).get("encode_gemm_ms"))
except Exception as e:
    print(sys.argv[1], "no bench line", e)
PY
}
for arm in "0 0" "0 1" "1 1" "0 0" "0 1" "1 1"; do
set -- $arm
tag="c$1_p$2_$(date +%s)"
( OMNI_SPLITK_COMBINE=$1 OMNI_PATCH_ROWS=$2 OMNI_BENCH_WATCHDOG=400 timeout 600 python3 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-extra > "$OUT/bench_$tag.json" 2> "$OUT/bench_$tag.err"; echo "exit $?" >> "$OUT/bench_$tag.err" )
showe "$OUT/bench_$tag.json"
done
echo "total $(( $(date +%s) - t0 )) s"
