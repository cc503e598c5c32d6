#!/bin/bash
# r05 session 28: the points of a chained pass made once per run call (batched inversion) instead of once per batch: tests, then A/B
set -o pipefail
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_commit.py tests/test_gpu_graph_capture.py -x -q -m gpu 2>&1 | tail -4 || exit 1
: > $O/commit_defer_normalize.log
for c in commit-only commit check+commit; do
  for d in 0 1; do
    B3W_COMMIT_DEFER_NORMALIZE=$d timeout -k 10 400 python bench.py --workload chain --preimage-mib 64 --consumer $c --steps 3 --warmup 1 --cpu-seconds 0 > $O/b.json 2> $O/w.err || { tail -5 $O/w.err; exit 1; }
    python -c "
import json; d=json.load(open('$O/b.json')); print('defer $d', '$c', round(d['value']/1e6,3), 'M steps/s', d['roofline'].get('frac'))" | tee -a $O/commit_defer_normalize.log
    [ $d = 1 ] && cp $O/b.json $O/bench_chain_64mib_consumer_${c/+/_}_deferred.json
  done
done
