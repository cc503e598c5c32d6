#!/bin/bash
# r03 job 3 (GPU box): the stream formulation of the constraint check — parity first, then rates against the lean pair
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r03_job3
mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_r1cs.py -x -q -m gpu > $out/pytest_r1cs.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -15 $out/pytest_r1cs.log
[ $rc -eq 0 ] || exit $rc
for cfg in "0 3" "0 2" "3 0"; do set -- $cfg
  B3W_R1CS_GATHER=$1 B3W_R1CS_NBUF=$2 timeout -k 10 300 python3 tools/ubench/r1cs_rate.py > $out/r1cs_rate_mode$1_nbuf$2.log 2>&1 || { echo "rate $cfg failed"; tail -5 $out/r1cs_rate_mode$1_nbuf$2.log; exit 1; }
  echo "== mode $1 nbuf $2"; cat $out/r1cs_rate_mode$1_nbuf$2.log
done
