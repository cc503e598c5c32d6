#!/bin/bash
# r04 job 33 (GPU box): the walk kernel's elements as SIGNED small numbers (p - k for -k) — parity (tests, the fuzz across formulations),
# then the check's profile and the circomkit build's kernel trace
set -o pipefail
export TMPDIR=/tmp B3W_PROFILE_ROUND=r04
out=gpurun_out/r04_job33
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
rocprofv3 --kernel-trace --stats --output-format csv -d $out/o1 -- python3 tools/ubench/r1cs_profile_target.py nova_bn254_o1 > $out/o1.log 2>&1; echo "o1 rc=$?"
python3 - <<'PY'
import csv, glob
f = max(glob.glob("gpurun_out/r04_job33/o1/**/*kernel_trace.csv", recursive=True))
per = {}
for r in sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"])):
    if "b3w_r1cs" in r["Kernel_Name"]:
        per.setdefault("deferred" if "deferred" in r["Kernel_Name"] else "walk", []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("nova_bn254_o1:", {k: round(sum(v[-10:]) / 10, 1) for k, v in per.items()})
PY
for rep in 1 2; do for c in nova_vesta compression; do timeout -k 10 200 python3 tools/ubench/r1cs_walk_scaling.py $c 2>&1 | grep "n=  8192\|n=  4096" | tr '\n' ' '; echo; done; done | tee $out/walk_scaling_signed.log
