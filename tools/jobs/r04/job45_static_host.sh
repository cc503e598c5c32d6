#!/bin/bash
# r04 job 45 (GPU box): the always-deferred rows' list now comes from the host builder (b3w_r1cs_host.cpp) — the check's tests and the fuzz
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r04_job45
mkdir -p $out
timeout -k 10 900 python3 -m pytest tests/test_gpu_r1cs.py -x -q -m gpu > $out/test_r1cs.log 2>&1; rc=$?; tail -3 $out/test_r1cs.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 400 python3 tools/ubench/r1cs_fuzz.py $out/walk.npz 4096 2>&1 | grep -v amdgpu | tee $out/fuzz_walk.log
for g in 4 3 1; do B3W_R1CS_GATHER=$g timeout -k 10 600 python3 tools/ubench/r1cs_fuzz.py $out/gather$g.npz 4096 2>&1 | grep -v amdgpu > $out/fuzz_gather$g.log; done
python3 tools/ubench/r1cs_fuzz_compare.py $out/walk.npz $out/gather4.npz $out/gather3.npz $out/gather1.npz | tee $out/r1cs_fuzz_compare.log; [ ${PIPESTATUS[0]} -eq 0 ] || exit 1
