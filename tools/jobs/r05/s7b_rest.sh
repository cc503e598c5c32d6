#!/bin/bash
# r05 session 7b: the rest of the -m gpu suite behind the placement service test
set -o pipefail
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_gpu_placement.py tests/test_gpu_r1cs.py tests/test_gpu_reference_mocha_mirror.py tests/test_gpu_sweep.py tests/test_gpu_threads.py tests/test_gpu_verify.py tests/test_node_addon.py tests/test_gpu_parity.py tests/test_gpu_parity_nova.py -x -q -m gpu --durations=12 > $O/gpu_suite_rest.log 2>&1; echo "pytest rc=$?"; tail -30 $O/gpu_suite_rest.log
