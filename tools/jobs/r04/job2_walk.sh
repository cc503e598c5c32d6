#!/bin/bash
# r04 job 2 (GPU box): the walk kernel — every formulation's verdicts, then rates of walk / stream on the same bodies
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r04_job2
mkdir -p $out
timeout -k 10 900 python3 -m pytest tests/test_gpu_r1cs.py -x -q > $out/test_gpu_r1cs.log 2>&1; rc=$?; echo "r1cs tests rc=$rc"; tail -25 $out/test_gpu_r1cs.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python3 tools/ubench/r1cs_rate.py 2>&1 | grep -v amdgpu > $out/r1cs_rate_walk.log; echo "walk rate rc=$?"; cat $out/r1cs_rate_walk.log
B3W_R1CS_GATHER=4 timeout -k 10 300 python3 tools/ubench/r1cs_rate.py 2>&1 | grep -v amdgpu > $out/r1cs_rate_stream.log; echo "stream rate rc=$?"; cat $out/r1cs_rate_stream.log
timeout -k 10 300 python3 tools/ubench/r1cs_rate_big.py 2>&1 | grep -v amdgpu > $out/r1cs_rate_big_walk.log; cat $out/r1cs_rate_big_walk.log
