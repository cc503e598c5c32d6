#!/bin/bash
# r05 session 5: commitments enqueued first (as the chain does), small-footprint writers, the co-resident commit kernel, priorities
set -o pipefail
O=gpurun_out/r05; mkdir -p $O
for cfg in "0 0" "1 0" "1 -1" "0 -1"; do
  set -- $cfg
  echo "=== B3W_COMMIT_CO=$1 PROBE_COMMIT_PRIO=$2"
  B3W_COMMIT_CO=$1 PROBE_COMMIT_PRIO=$2 timeout -k 10 300 python tools/ubench/overlap_commit_probe.py > $O/overlap_probe2_co$1_prio$2.log 2>&1; echo rc=$?
  grep -v amdgpu.ids $O/overlap_probe2_co$1_prio$2.log
done
