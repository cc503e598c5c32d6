#!/bin/bash
# r05 session 17: one-wave commit workgroups and a witness kernel whose store loop runs at raised wave priority, beside each other
set -o pipefail
O=gpurun_out/r05; mkdir -p $O
for cfg in "4 0" "1 0" "4 1" "1 1"; do
  set -- $cfg
  echo "=== B3W_COMMIT_WPB=$1 B3W_WITNESS_PRIO=$2"
  B3W_COMMIT_WPB=$1 B3W_WITNESS_PRIO=$2 timeout -k 10 300 python tools/ubench/overlap_commit_probe.py > $O/overlap_probe3_wpb$1_prio$2.log 2>&1; echo rc=$?
  grep -v amdgpu.ids $O/overlap_probe3_wpb$1_prio$2.log | grep -v "persistent wave\|fills every"
done
