#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r03_job4
mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_r1cs.py -x -q -m gpu > $out/pytest_r1cs.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $out/pytest_r1cs.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 tools/ubench/r1cs_stream_dbg.py compression 4096 > $out/dbg_comp.log 2>&1; cat $out/dbg_comp.log
timeout -k 10 300 python3 tools/ubench/r1cs_stream_dbg.py nova_vesta 4096 > $out/dbg_nova.log 2>&1; cat $out/dbg_nova.log
B3W_R1CS_STAMPS=1 python3 tools/ubench/r1cs_profile_target.py compression 2>&1 | tail -10
B3W_R1CS_STAMPS=1 python3 tools/ubench/r1cs_profile_target.py nova_vesta 2>&1 | tail -10
