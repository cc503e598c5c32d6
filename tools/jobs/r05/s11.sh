#!/bin/bash
# r05 session 11: LDS-resident tree kernel + single-slice copy on the caller's stream: tests, model, timeline
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_gpu_chain.py tests/test_gpu_native_exchange.py tests/test_gpu_placement.py tests/test_gpu_r1cs.py tests/test_gpu_reference_mocha_mirror.py tests/test_gpu_sweep.py tests/test_gpu_threads.py tests/test_gpu_verify.py tests/test_node_addon.py -x -q -m gpu --durations=8 > $O/gpu_suite_rest.log 2>&1; echo "pytest rc=$?"; tail -16 $O/gpu_suite_rest.log
timeout -k 10 300 python tools/ubench/chain_scaling_model.py 1 2>/dev/null | sed -n '/^{/,$p' > $O/chain_scaling_model_1mib.json; echo "model rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05/chain_scaling_model_1mib.json'))
for r in d['rows']: print(' ', r['ranks'], r['rank0_steps'], 'queued', r['rank0_pass_ms'], 'single', r['rank0_single_pass_median_ms'], 'unsharded-of-shard', r['its_single_pass_median_ms'], r['predicted_M_steps_per_s_with_50us_per_collective'], r['standin_exchange_ms'])
PY
for cfg in "8 1 none" "1 1 none"; do
  set -- $cfg
  tag=ranks$1_$2mib_$(echo $3 | tr '+' '_')
  rm -rf gpurun_out/prof_$tag
  timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/prof_$tag -- python3 tools/ubench/chain_one_pass.py $1 $2 $3 > $O/timeline_$tag.run.log 2>&1; echo "rocprof $tag rc=$?"
  tail -1 $O/timeline_$tag.run.log
  python3 tools/pass_timeline.py gpurun_out/prof_$tag 70 > $O/timeline_$tag.v2.txt 2>&1; head -20 $O/timeline_$tag.v2.txt
  rm -rf gpurun_out/prof_$tag
done
