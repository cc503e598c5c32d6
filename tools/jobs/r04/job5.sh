#!/bin/bash
# r04 job 5 (GPU box): read ceiling of the chip in the walk kernel's pattern; walk kernel (wide rows reverted, 4-wave deferred) profile
set -o pipefail
export TMPDIR=/tmp B3W_PROFILE_ROUND=r04
out=gpurun_out/r04_job5
mkdir -p $out
timeout -k 10 200 tools/ubench/read_ceiling 12 > $out/read_ceiling.log 2>&1; echo "ceiling rc=$?"; cat $out/read_ceiling.log
timeout -k 10 600 python3 -m pytest tests/test_gpu_r1cs.py -x -q > $out/test_gpu_r1cs.log 2>&1; rc=$?; echo "r1cs tests rc=$rc"; tail -5 $out/test_gpu_r1cs.log
[ $rc -ne 0 ] && exit $rc
bash tools/profile_r1cs.sh > $out/profile_r1cs.log 2>&1; echo "profile rc=$?"; grep -E "walk|deferred|stream|check_us|frac|fetch_over" $out/profile_r1cs.log
mkdir -p $out/profiles_r04 && cp profiles/r04/r1cs_check* $out/profiles_r04/
timeout -k 10 300 python3 tools/ubench/r1cs_rate.py 2>&1 | grep -v amdgpu > $out/r1cs_rate_walk.log; cat $out/r1cs_rate_walk.log
