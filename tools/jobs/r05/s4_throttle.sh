#!/bin/bash
# r05 session 4: does a writer that stays just below HBM saturation let the commit kernel's table reads through?
set -o pipefail
O=gpurun_out/r05; mkdir -p $O
PROBE_CU_MASK=1 timeout -k 10 600 python tools/ubench/overlap_commit_probe.py > $O/overlap_probe_throttle.log 2>&1; echo rc=$?
grep -v amdgpu.ids $O/overlap_probe_throttle.log
