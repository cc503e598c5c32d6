#!/bin/bash
# r04 job 11 (GPU box): walk kernel variants — verdict tests, then ms per check against the batch size
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r04_job11
mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_r1cs.py -x -q > $out/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $out/tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python3 tools/ubench/r1cs_walk_scaling.py compression 2>&1 | grep -v amdgpu | tee $out/walk_scaling_compression.log
timeout -k 10 300 python3 tools/ubench/r1cs_walk_scaling.py nova_vesta 2>&1 | grep -v amdgpu | tee $out/walk_scaling_nova_vesta.log
