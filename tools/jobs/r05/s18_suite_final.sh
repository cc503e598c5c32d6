#!/bin/bash
# r05 session 18: the whole -m gpu suite on the final library, as the driver runs it; smoke()
set -o pipefail
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu --durations=12 > $O/gpu_suite_final.log 2>&1; echo "pytest rc=$?"; tail -20 $O/gpu_suite_final.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
