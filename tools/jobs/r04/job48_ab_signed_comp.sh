#!/bin/bash
# r04 job 48 (GPU box): compression under rocprofv3 with the unsigned (B3W_R1CS_SIGNED=0) and the signed instantiation of the walk kernel, alternating
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r04_job48
mkdir -p $out
for rep in 1 2 3; do for sg in 0 1; do
  B3W_R1CS_SIGNED=$sg rocprofv3 --kernel-trace --output-format csv -d $out/t_${sg}_$rep -- python3 tools/ubench/r1cs_profile_target.py compression > $out/t.log 2>&1
  python3 - $out/t_${sg}_$rep $sg <<'PY'
import csv, glob, sys
f = max(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))
per = {}
for r in sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"])):
    if "b3w_r1cs" in r["Kernel_Name"]:
        per.setdefault("deferred" if "deferred" in r["Kernel_Name"] else "walk", []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("signed=%s" % sys.argv[2], {k: round(sum(v[-10:]) / 10, 1) for k, v in per.items()}, [round(x) for x in per["walk"]])
PY
done; done | tee $out/ab_signed_compression_rocprof.log
