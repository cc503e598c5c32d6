#!/bin/bash
# r04 job 4 (GPU box): walk kernel with per-body deferred flags and in-kernel inverse rows: verdicts, rates, rocprofv3 profile
set -o pipefail
export TMPDIR=/tmp B3W_PROFILE_ROUND=r04
out=gpurun_out/r04_job4
mkdir -p $out
timeout -k 10 900 python3 -m pytest tests/test_gpu_r1cs.py -x -q > $out/test_gpu_r1cs.log 2>&1; rc=$?; echo "r1cs tests rc=$rc"; tail -25 $out/test_gpu_r1cs.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python3 tools/ubench/r1cs_rate.py 2>&1 | grep -v amdgpu > $out/r1cs_rate_walk.log; echo "walk rate rc=$?"; cat $out/r1cs_rate_walk.log
bash tools/profile_r1cs.sh > $out/profile_r1cs.log 2>&1; echo "profile rc=$?"; grep -E "walk|deferred|stream|check_us|frac|fetch_over" $out/profile_r1cs.log
mkdir -p $out/profiles_r04 && cp profiles/r04/r1cs_check* $out/profiles_r04/
