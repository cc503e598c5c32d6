#!/bin/bash
# r04 job 25 (GPU box): kernel trace of the check of the circomkit nova build (walk units), and nova/Vesta with its heavy tile split
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r04_job25
mkdir -p $out
summ() { python3 - "$1" <<'PY'
import csv, glob, sys
f = max(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))
per = {}
for r in sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"])):
    if "b3w_r1cs" in r["Kernel_Name"]:
        per.setdefault(r["Kernel_Name"].split("(")[0][-40:], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print({k: round(sum(v[-10:]) / len(v[-10:]), 1) for k, v in per.items()})
PY
}
rocprofv3 --kernel-trace --stats --output-format csv -d $out/o1 -- python3 tools/ubench/r1cs_profile_target.py nova_bn254_o1 > $out/o1.log 2>&1; echo "o1 rc=$?"; summ $out/o1
B3W_R1CS_GATHER=4 rocprofv3 --kernel-trace --stats --output-format csv -d $out/o1_stream -- python3 tools/ubench/r1cs_profile_target.py nova_bn254_o1 > $out/o1_stream.log 2>&1; echo "o1 stream rc=$?"; summ $out/o1_stream
for sp in 320 160 100; do
  B3W_WALK_SPLIT_GEN=$sp rocprofv3 --kernel-trace --stats --output-format csv -d $out/vesta_$sp -- python3 tools/ubench/r1cs_profile_target.py nova_vesta > $out/vesta_$sp.log 2>&1; echo "vesta split $sp rc=$?"; summ $out/vesta_$sp
done
B3W_WALK_SPLIT_GEN=100 rocprofv3 --kernel-trace --stats --output-format csv -d $out/o1_100 -- python3 tools/ubench/r1cs_profile_target.py nova_bn254_o1 > $out/o1_100.log 2>&1; echo "o1 split 100 rc=$?"; summ $out/o1_100
