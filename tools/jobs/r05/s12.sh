#!/bin/bash
# r05 session 12: quad-lane leaf planner: chain tests; model (default; 8-body witness kernel for the shards); timelines
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_gpu_chain.py tests/test_gpu_native_exchange.py tests/test_gpu_placement.py tests/test_gpu_commit.py -x -q -m gpu --durations=5 > $O/gpu_suite_chain.log 2>&1; echo "pytest rc=$?"; tail -12 $O/gpu_suite_chain.log
for v in default 3; do
  if [ $v = default ]; then unset B3W_VARIANT; else export B3W_VARIANT=$v; fi
  timeout -k 10 300 python tools/ubench/chain_scaling_model.py 1 2>/dev/null | sed -n '/^{/,$p' > $O/chain_scaling_model_1mib_variant_$v.json; echo "model variant=$v rc=$?"
  python - $v <<'PY'
import json,sys
d=json.load(open(f'gpurun_out/r05/chain_scaling_model_1mib_variant_{sys.argv[1]}.json'))
for r in d['rows']: print(' ', r['ranks'], r['rank0_steps'], 'queued', r['rank0_pass_ms'], 'single', r['rank0_single_pass_median_ms'], 'unsharded-of-shard', r['its_single_pass_median_ms'], r['predicted_M_steps_per_s_with_50us_per_collective'], r['standin_exchange_ms'])
PY
  for cfg in "8 1 none"; do
    set -- $cfg
    tag=ranks$1_$2mib_variant_$v
    rm -rf gpurun_out/prof_$tag
    timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/prof_$tag -- python3 tools/ubench/chain_one_pass.py $1 $2 $3 > $O/timeline_$tag.run.log 2>&1; echo "rocprof $tag rc=$?"
    python3 tools/pass_timeline.py gpurun_out/prof_$tag 70 > $O/timeline_$tag.txt 2>&1; head -16 $O/timeline_$tag.txt
    rm -rf gpurun_out/prof_$tag
  done
done
