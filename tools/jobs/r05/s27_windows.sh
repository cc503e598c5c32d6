#!/bin/bash
# r05 session 27b: 18-bit commitment tables as the library's own choice: the commit tests, then the chained pass with every commit consumer
set -o pipefail
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_commit.py -x -q -m gpu 2>&1 | tail -4 || exit 1
: > $O/commit_windows_rates_auto.log
for c in commit-only commit check+commit commit-bodies; do
  for w in 16 0; do
    timeout -k 10 400 python bench.py --workload chain --preimage-mib 64 --consumer $c --commit-window $w --steps 3 --warmup 1 --cpu-seconds 0 > $O/bench_chain_64mib_consumer_${c/+/_}_w$w.json 2> $O/w.err || { tail -5 $O/w.err; exit 1; }
    python -c "
import json; d=json.load(open('$O/bench_chain_64mib_consumer_${c/+/_}_w$w.json')); print('window $w', '$c', round(d['value']/1e6,3), 'M steps/s', d['roofline'].get('point_additions_per_step'), d['roofline'].get('frac'), d['config']['consumer'][:90])" | tee -a $O/commit_windows_rates_auto.log
  done
done
