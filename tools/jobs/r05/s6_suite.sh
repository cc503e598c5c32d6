#!/bin/bash
# r05 session 6: the whole -m gpu suite (durations), as the driver runs it
set -o pipefail
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 1150 python -m pytest tests/ -x -q -m gpu --durations=25 > $O/gpu_suite.log 2>&1; echo "pytest rc=$?"; tail -40 $O/gpu_suite.log
