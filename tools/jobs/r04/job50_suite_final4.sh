#!/bin/bash
# r04 job 50 (GPU box): the whole -m gpu suite with per-test durations
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r04_job50
mkdir -p $out
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q --durations=25 > $out/gpu_suite.log 2>&1; rc=$?; echo "suite rc=$rc"; tail -45 $out/gpu_suite.log
python3 -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" 2>&1 | tail -2
