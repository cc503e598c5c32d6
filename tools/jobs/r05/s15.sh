#!/bin/bash
# r05 session 15: does an L2-sized table keep a gather + VALU kernel fed beside an HBM-saturating writer?  + rest of the suite
set -o pipefail
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 300 python tools/ubench/gather_beside_writer.py 2>&1 | grep -v amdgpu.ids | tee $O/gather_beside_writer.log
timeout -k 10 800 python -m pytest tests/test_gpu_placement.py tests/test_gpu_r1cs.py tests/test_gpu_reference_mocha_mirror.py tests/test_gpu_sweep.py tests/test_gpu_threads.py tests/test_gpu_verify.py tests/test_node_addon.py -x -q -m gpu --durations=6 > $O/gpu_suite_rest.log 2>&1; echo "pytest rc=$?"; tail -14 $O/gpu_suite_rest.log
