#!/bin/bash
# r05 session 29: GATED fold-shaped pass with the TRACE images written on the caller's stream in front of the witness kernel: tests, A/B, a timeline
set -o pipefail
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_commit.py -x -q -m gpu 2>&1 | tail -4 || exit 1
: > $O/commit_split_trace.log
for rnd in 1 2; do
  for d in 0 1; do
    B3W_COMMIT_SPLIT_TRACE=$d timeout -k 10 400 python bench.py --workload chain --preimage-mib 64 --consumer check+commit --steps 3 --warmup 1 --cpu-seconds 0 > $O/b.json 2> $O/w.err || { tail -5 $O/w.err; exit 1; }
    python -c "
import json; d=json.load(open('$O/b.json')); print('split $d', 'check+commit', round(d['value']/1e6,3), 'M steps/s', d['roofline'].get('frac'))" | tee -a $O/commit_split_trace.log
    [ $d = 1 ] && cp $O/b.json $O/bench_chain_64mib_consumer_check_commit_split.json
  done
done
export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/prof_split -- python3 tools/ubench/chain_one_pass.py 1 8 check+commit > $O/split_timeline.log 2>&1 && python3 tools/pass_timeline.py gpurun_out/prof_split > $O/timeline_ranks1_8mib_check_commit_split_trace.txt 2>&1; head -40 $O/timeline_ranks1_8mib_check_commit_split_trace.txt
