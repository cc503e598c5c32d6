#!/bin/bash
# r05 session 10: rest of the suite; rocprofv3 timelines of ONE chained pass: 8-rank share, 1 rank (1 MiB), and the gated fold-shaped pass (8 MiB)
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_gpu_placement.py tests/test_gpu_r1cs.py tests/test_gpu_reference_mocha_mirror.py tests/test_gpu_sweep.py tests/test_gpu_threads.py tests/test_gpu_verify.py tests/test_node_addon.py -x -q -m gpu --durations=12 > $O/gpu_suite_rest.log 2>&1; echo "pytest rc=$?"; tail -22 $O/gpu_suite_rest.log
for cfg in "8 1 none" "1 1 none" "1 8 check+commit"; do
  set -- $cfg
  tag=ranks$1_$2mib_$(echo $3 | tr '+' '_')
  rm -rf gpurun_out/prof_$tag
  timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/prof_$tag -- python3 tools/ubench/chain_one_pass.py $1 $2 $3 > $O/timeline_$tag.run.log 2>&1; echo "rocprof $tag rc=$?"
  tail -1 $O/timeline_$tag.run.log
  python3 tools/pass_timeline.py gpurun_out/prof_$tag 70 > $O/timeline_$tag.txt 2>&1; head -75 $O/timeline_$tag.txt
  rm -rf gpurun_out/prof_$tag
done
