#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r03_job7
mkdir -p $out
timeout -k 10 1000 python -m pytest tests/test_gpu_commit.py -x -q -m gpu > $out/pytest_commit.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -12 $out/pytest_commit.log
[ $rc -eq 0 ] || exit $rc
python3 tools/ubench/commit_rate_folded.py 2>&1 | grep -v amdgpu > $out/commit_rate_folded_invtab.log; cat $out/commit_rate_folded_invtab.log
