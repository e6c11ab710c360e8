#!/bin/bash
# round-4 session 1: the driver's two commands on HEAD, then diagnostics if the first launch faults
mkdir -p gpurun_out/r4_s1
O=gpurun_out/r4_s1
(timeout 1500 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt)
tail -5 $O/pytest.txt
(timeout 600 python3 -c 'import __graft_entry__ as e; e.smoke()' > $O/smoke.txt 2>&1; echo "rc=$?" >> $O/smoke.txt)
tail -5 $O/smoke.txt
if ! grep -q "rc=0" $O/pytest.txt; then
  (AMD_SERIALIZE_KERNEL=3 HIP_LAUNCH_BLOCKING=1 AMD_LOG_LEVEL=3 timeout 300 python3 -m pytest tests/test_gpu_a_kernels.py::test_mfma_fragment_layout -x -q -p no:cacheprovider > $O/diag_first.txt 2>&1; echo "rc=$?" >> $O/diag_first.txt)
  tail -c 6000 $O/diag_first.txt > $O/diag_first_tail.txt
  grep -n -i "fault\|error\|ShaderName\|KernelExecution\|omni" $O/diag_first.txt | tail -60 > $O/diag_first_grep.txt
  # keep the log under the merge size limit
  head -c 20000000 $O/diag_first.txt > $O/diag_first_head.txt; rm $O/diag_first.txt
fi
