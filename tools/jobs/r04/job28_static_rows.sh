#!/bin/bash
# r04 job 28 (GPU box): the always-deferred rows from their unique terms (one multiplication for a term that stands in A and in B), two
# wave sums for the price of one — parity (tests, the fuzz across formulations), then the check's profile
set -o pipefail
export TMPDIR=/tmp B3W_PROFILE_ROUND=r04
out=gpurun_out/r04_job28
mkdir -p $out
timeout -k 10 900 python3 -m pytest tests/test_gpu_r1cs.py -x -q -m gpu > $out/test_r1cs.log 2>&1; rc=$?; tail -3 $out/test_r1cs.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 400 python3 tools/ubench/r1cs_fuzz.py $out/walk.npz 4096 2>&1 | grep -v amdgpu | tee $out/fuzz_walk.log
B3W_R1CS_GATHER=1 timeout -k 10 600 python3 tools/ubench/r1cs_fuzz.py $out/gather1.npz 4096 2>&1 | grep -v amdgpu > $out/fuzz_gather1.log
python3 tools/ubench/r1cs_fuzz_compare.py $out/walk.npz $out/gather1.npz | tee $out/r1cs_fuzz_compare.log; [ ${PIPESTATUS[0]} -eq 0 ] || exit 1
bash tools/profile_r1cs.sh > $out/profile_r1cs.log 2>&1; echo "r1cs rc=$?"; python3 - <<'PY'
import json
d = json.load(open("profiles/r04/r1cs_check.json"))
for c, v in d["circuits"].items():
    print(c, v["kernel_avg_us"], round(v["roofline"]["frac"], 4), round(v["fetch_over_body_bytes"], 4))
PY
mkdir -p $out/profiles_r04 && cp profiles/r04/r1cs_check* $out/profiles_r04/
