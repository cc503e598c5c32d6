#!/bin/bash
# r04 job 15 (GPU box): the two-wave deferred kernel and the one-pass long row — parity first, then the check's profile again
set -o pipefail
export TMPDIR=/tmp B3W_PROFILE_ROUND=r04
out=gpurun_out/r04_job15
mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_r1cs.py -x -q -m gpu > $out/test_r1cs.log 2>&1; rc=$?; tail -3 $out/test_r1cs.log; [ $rc -eq 0 ] || exit 1
bash tools/profile_r1cs.sh > $out/profile_r1cs.log 2>&1; echo "r1cs rc=$?"; python3 - <<'PY'
import json
d = json.load(open("profiles/r04/r1cs_check.json"))
for c, v in d["circuits"].items():
    print(c, v["kernel_avg_us"], round(v["roofline"]["frac"], 4), round(v["fetch_over_body_bytes"], 4), v["launch_us"])
PY
timeout -k 10 300 python3 tools/ubench/r1cs_walk_scaling.py nova_vesta 2>&1 | grep -v amdgpu | tee $out/walk_scaling_nova_vesta.log
mkdir -p $out/profiles_r04 && cp profiles/r04/r1cs_check* $out/profiles_r04/
