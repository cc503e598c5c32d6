#!/bin/bash
# r04 job 47 (GPU box; run several times, each call a fresh box): the constraint check's rocprofv3 profile only — the boxes of this pool
# differ by +-2 % on this kernel, so the round keeps every run (profiles/r04/r1cs_check_runs.json) and quotes the median one
set -o pipefail
export TMPDIR=/tmp B3W_PROFILE_ROUND=r04
out=gpurun_out/r04_job47_$1
mkdir -p $out
bash tools/profile_r1cs.sh > $out/profile_r1cs.log 2>&1; echo "r1cs rc=$?"
cp profiles/r04/r1cs_check* $out/
python3 - <<'PY'
import json
d = json.load(open("profiles/r04/r1cs_check.json"))
for c, v in d["circuits"].items():
    print(c, v["kernel_avg_us"], round(v["roofline"]["frac"], 4), round(v["fetch_over_body_bytes"], 4))
PY
