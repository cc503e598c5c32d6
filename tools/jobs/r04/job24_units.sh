#!/bin/bash
# r04 job 24 (GPU box): walk UNITS (a tile with many general rows is several units) — the circomkit nova build on the walk kernel
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r04_job24
mkdir -p $out
timeout -k 10 900 python3 -m pytest tests/test_gpu_r1cs.py -x -q -m gpu > $out/test_r1cs.log 2>&1; rc=$?; tail -3 $out/test_r1cs.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 300 python3 tools/ubench/r1cs_rate.py 2>&1 | grep -v amdgpu | tee $out/r1cs_rate_walk.log
B3W_R1CS_GATHER=4 timeout -k 10 300 python3 tools/ubench/r1cs_rate.py 2>&1 | grep -v amdgpu | tee $out/r1cs_rate_stream.log
timeout -k 10 300 python3 tools/ubench/r1cs_walk_scaling.py nova_bn254_o1 2>&1 | grep -v amdgpu | tee $out/walk_scaling_nova_bn254_o1.log
