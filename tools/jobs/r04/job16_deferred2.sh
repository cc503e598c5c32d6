#!/bin/bash
# r04 job 16 (GPU box): deferred kernel with its loads side by side (and, as an experiment, squeezed to 4 waves per SIMD)
set -o pipefail
export TMPDIR=/tmp B3W_PROFILE_ROUND=r04
out=gpurun_out/r04_job16
mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_r1cs.py -x -q -m gpu > $out/test_r1cs.log 2>&1; rc=$?; tail -3 $out/test_r1cs.log; [ $rc -eq 0 ] || exit 1
bash tools/profile_r1cs.sh > $out/profile_r1cs.log 2>&1; echo "r1cs rc=$?"; python3 - <<'PY'
import json
d = json.load(open("profiles/r04/r1cs_check.json"))
for c, v in d["circuits"].items():
    print(c, v["kernel_avg_us"], round(v["roofline"]["frac"], 4), round(v["fetch_over_body_bytes"], 4), v["launch_us"])
PY
mkdir -p $out/profiles_r04 && cp profiles/r04/r1cs_check* $out/profiles_r04/
B3W_R1CS_DEFERRED_OCC=4 rocprofv3 --kernel-trace --stats --output-format csv -d $out/occ4 -- python3 tools/ubench/r1cs_profile_target.py nova_vesta > $out/occ4.log 2>&1; echo "occ4 rc=$?"
python3 - <<'PY'
import csv, glob
f = max(glob.glob("gpurun_out/r04_job16/occ4/**/*kernel_trace.csv", recursive=True))
per = {}
for r in sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"])):
    if "b3w_r1cs" in r["Kernel_Name"]:
        per.setdefault("deferred" if "deferred" in r["Kernel_Name"] else "walk", []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("occ4:", {k: round(sum(v[-10:]) / 10, 1) for k, v in per.items()})
PY
