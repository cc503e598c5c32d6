#!/bin/bash
# tools/profile_r1cs.sh — run ON THE GPU BOX (through gpurun), from the repo root: rocprofv3 kernel statistics and the FETCH_SIZE
# counter (separate pass) of the constraint check, 4 096 bodies, for compression and the O2 nova build; the summaries go to
# gpurun_out/prof_r1cs/ and tools/profile_r1cs_collect.py distils them into profiles/rNN/.
set -o pipefail
out=gpurun_out/prof_r1cs
mkdir -p $out
export TMPDIR=/tmp
for c in compression nova_vesta; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$c -- python3 tools/ubench/r1cs_profile_target.py $c > $out/stats_$c.log 2>&1 || { echo "stats pass failed ($c)"; tail -5 $out/stats_$c.log; exit 1; }
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch_$c -- python3 tools/ubench/r1cs_profile_target.py $c > $out/fetch_$c.log 2>&1 || { echo "FETCH_SIZE pass failed ($c)"; tail -5 $out/fetch_$c.log; exit 1; }
done
python3 tools/profile_r1cs_collect.py ${B3W_PROFILE_ROUND:-r04}
