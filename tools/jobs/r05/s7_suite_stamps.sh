#!/bin/bash
# r05 session 7: the whole -m gpu suite (durations); scaling model; then the diagnostic build's per-phase and per-tile stamps of the walk kernel
set -o pipefail
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 1000 python -m pytest tests/ -x -q -m gpu --durations=25 > $O/gpu_suite.log 2>&1; echo "pytest rc=$?"; tail -45 $O/gpu_suite.log
