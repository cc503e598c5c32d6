#!/bin/bash
# r04 job 10 (GPU box): SQ wave counters of the walk kernel (and round 3's stream kernel beside it)
set -o pipefail
export TMPDIR=/tmp B3W_PROFILE_ROUND=r04
out=gpurun_out/r04_job10
mkdir -p $out
bash tools/profile_sq.sh > $out/profile_sq.log 2>&1; echo "sq rc=$?"; tail -30 $out/profile_sq.log
cp profiles/r04/sq_counters.json $out/
