#!/bin/bash
# r03 job 5 (GPU box): the whole -m gpu suite with the stream check as default, then the constraint check's profiles and breakdowns
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r03_job5
mkdir -p $out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $out/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
bash tools/profile_r1cs.sh > $out/profile_r1cs.log 2>&1; echo "profile_r1cs rc=$?"; tail -3 $out/profile_r1cs.log
mkdir -p $out/profiles_r03 && cp -r profiles/r03/* $out/profiles_r03/ 2>/dev/null
python3 tools/ubench/r1cs_stream_dbg.py compression 4096 > $out/r1cs_stream_breakdown_compression.log 2>&1
python3 tools/ubench/r1cs_stream_dbg.py nova_vesta 4096 > $out/r1cs_stream_breakdown_nova_vesta.log 2>&1
python3 tools/ubench/r1cs_rate.py > $out/r1cs_rate_stream.log 2>&1; cat $out/r1cs_rate_stream.log
B3W_R1CS_STAMPS=1 python3 tools/ubench/r1cs_profile_target.py compression > $out/r1cs_stream_stamps_compression.log 2>&1
B3W_R1CS_STAMPS=1 python3 tools/ubench/r1cs_profile_target.py nova_vesta > $out/r1cs_stream_stamps_nova_vesta.log 2>&1
