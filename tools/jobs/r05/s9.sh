#!/bin/bash
# r05 session 9: the rest of the suite; the scaling model with a one-call stand-in, slices of 256 chunks beside the default
set -o pipefail
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_gpu_placement.py tests/test_gpu_r1cs.py tests/test_gpu_reference_mocha_mirror.py tests/test_gpu_sweep.py tests/test_gpu_threads.py tests/test_gpu_verify.py tests/test_node_addon.py -x -q -m gpu --durations=12 > $O/gpu_suite_rest.log 2>&1; echo "pytest rc=$?"; tail -22 $O/gpu_suite_rest.log
for sl in default 256 128; do
  if [ $sl = default ]; then unset B3W_CHAIN_SLICE_CHUNKS; else export B3W_CHAIN_SLICE_CHUNKS=$sl; fi
  timeout -k 10 300 python tools/ubench/chain_scaling_model.py 1 2>/dev/null | sed -n '/^{/,$p' > $O/chain_scaling_model_1mib_slice_$sl.json; echo "model slice=$sl rc=$?"
  python - $sl <<'PY'
import json,sys
d=json.load(open(f'gpurun_out/r05/chain_scaling_model_1mib_slice_{sys.argv[1]}.json'))
for r in d['rows']: print(' ', r['ranks'], r['rank0_steps'], 'queued', r['rank0_pass_ms'], 'single', r['rank0_single_pass_median_ms'], 'unsharded-of-shard', r['its_single_pass_median_ms'], r['predicted_M_steps_per_s_with_50us_per_collective'], r['standin_exchange_ms'])
PY
done
