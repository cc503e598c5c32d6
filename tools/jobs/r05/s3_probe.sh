#!/bin/bash
# r05 session 3: why commit and witness kernels overlap poorly; the co-resident commit kernel; fused tree kernel tests
set -o pipefail
O=gpurun_out/r05; mkdir -p $O
for cfg in "0 0" "1 0" "1 -1" "0 -1"; do
  set -- $cfg
  echo "=== B3W_COMMIT_CO=$1 PROBE_COMMIT_PRIO=$2"
  B3W_COMMIT_CO=$1 PROBE_COMMIT_PRIO=$2 timeout -k 10 300 python tools/ubench/overlap_commit_probe.py > $O/overlap_probe_co$1_prio$2.log 2>&1; echo rc=$?
  cat $O/overlap_probe_co$1_prio$2.log | grep -v amdgpu.ids
done
echo "=== chain tests (fused tree kernel, commit overlap, 8 ranks as threads)"
timeout -k 10 1000 python -m pytest tests/test_gpu_chain.py tests/test_gpu_native_exchange.py tests/test_gpu_commit.py tests/test_node_addon.py -x -q -m gpu > $O/s3_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/s3_pytest.log
