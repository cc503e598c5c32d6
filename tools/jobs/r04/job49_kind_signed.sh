#!/bin/bash
# r04 job 49 (GPU box): the signed instantiation chosen by the circuit build — the check's tests (both instantiations for every circuit
# among the formulations), the fuzz, then the check's profile
set -o pipefail
export TMPDIR=/tmp B3W_PROFILE_ROUND=r04
out=gpurun_out/r04_job49
mkdir -p $out
timeout -k 10 900 python3 -m pytest tests/test_gpu_r1cs.py -x -q -m gpu --durations=4 > $out/test_r1cs.log 2>&1; rc=$?; tail -8 $out/test_r1cs.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 400 python3 tools/ubench/r1cs_fuzz.py $out/walk.npz 4096 2>&1 | grep -v amdgpu | tee $out/fuzz_walk.log
B3W_R1CS_SIGNED=1 timeout -k 10 400 python3 tools/ubench/r1cs_fuzz.py $out/walk_signed.npz 4096 2>&1 | grep -v amdgpu > $out/fuzz_walk_signed.log
B3W_R1CS_SIGNED=0 timeout -k 10 400 python3 tools/ubench/r1cs_fuzz.py $out/walk_unsigned.npz 4096 2>&1 | grep -v amdgpu > $out/fuzz_walk_unsigned.log
for g in 4 3 1; do B3W_R1CS_GATHER=$g timeout -k 10 600 python3 tools/ubench/r1cs_fuzz.py $out/gather$g.npz 4096 2>&1 | grep -v amdgpu > $out/fuzz_gather$g.log; done
python3 tools/ubench/r1cs_fuzz_compare.py $out/walk.npz $out/walk_signed.npz $out/walk_unsigned.npz $out/gather4.npz $out/gather3.npz $out/gather1.npz | tee $out/r1cs_fuzz_compare.log; [ ${PIPESTATUS[0]} -eq 0 ] || exit 1
bash tools/profile_r1cs.sh > $out/profile_r1cs.log 2>&1; echo "r1cs rc=$?"; cp profiles/r04/r1cs_check* $out/
python3 - <<'PY'
import json
d = json.load(open("profiles/r04/r1cs_check.json"))
for c, v in d["circuits"].items():
    print(c, v["kernel_avg_us"], round(v["roofline"]["frac"], 4), round(v["fetch_over_body_bytes"], 4))
PY
