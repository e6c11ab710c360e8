#!/usr/bin/env bash
# round-4 session 2: (0) first test file + smoke on this lease (lease log), then the round-3 candidates: hardware parity (A1),
# per-op A/B (A2, default vs all), detector batch-1 with fuse_splitk (A2b), bench A/B (A3), CU-partitioned lanes / split replay (A4, no probe)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4_s2
mkdir -p $O
( timeout 600 python3 -m pytest tests/test_gpu_a_kernels.py -x -q -m gpu -p no:cacheprovider --durations=5 > $O/pytest_a.txt 2>&1; echo "rc=$?" >> $O/pytest_a.txt )
grep -v "Warning\|warnings.warn\|^$" $O/pytest_a.txt | tail -12
( timeout 300 python3 -c 'import __graft_entry__ as e; e.smoke()' > $O/smoke.txt 2>&1; echo "rc=$?" >> $O/smoke.txt )
grep "smoke\]\|rc=\|smoke OK" $O/smoke.txt | cut -c1-200
export SKIP_PROBE=1
SECTIONS="A1 A2b A3 A4" bash tools/r4_open.sh > $O/open.txt 2>&1
cp -r gpurun_out/r4open $O/ 2>/dev/null
for f in "" "window_attn_v2,chan_apply_mfma,mha_v2"; do
  tag=${f:-default}; tag=${tag//,/+}
  ( timeout 200 python tools/caption_profile.py 128 768 2 $f > "$O/per_op_$tag.json" 2> "$O/per_op_$tag.txt"; echo "$tag exit $?" ) >> $O/open.txt
done
tail -c 7000 $O/open.txt
