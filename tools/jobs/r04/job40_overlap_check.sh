#!/bin/bash
# r04 job 40 (GPU box): the check of batch i on a second stream beside the witness kernel of batch i + 1, with round 4's walk kernel
# (round 2 measured the lean pair this way: 2.81 against 2.84 M steps/s)
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r04_job40
mkdir -p $out
for n in 16384 65536; do timeout -k 10 300 python3 tools/ubench/overlap_check.py nova_vesta $n 2>&1 | grep -v amdgpu; done | tee $out/overlap_check_walk.log
timeout -k 10 300 python3 tools/ubench/overlap_check.py compression 16384 2>&1 | grep -v amdgpu | tee -a $out/overlap_check_walk.log
