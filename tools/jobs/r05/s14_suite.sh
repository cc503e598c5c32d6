#!/bin/bash
# r05 session 14: the whole -m gpu suite on the final library, as the driver runs it (durations); chained passes from a hipGraph
set -o pipefail
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu --durations=15 > $O/gpu_suite_final.log 2>&1; echo "pytest rc=$?"; tail -28 $O/gpu_suite_final.log
for w in 1 8; do timeout -k 10 200 python tools/ubench/chain_graph.py $w 1 2>&1 | grep -v amdgpu.ids | tail -2 | tee -a $O/chain_graph.log; done
