#!/bin/bash
# r04 job 38 (GPU box): the chained pass with the check consumer against the preimage slice size (B3W_CHAIN_SLICE_CHUNKS: 1 024 = default)
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r04_job38
mkdir -p $out
for rep in 1 2; do
for sl in 1024 2048 4096 8192; do
  for c in check none commit; do
    B3W_CHAIN_SLICE_CHUNKS=$sl timeout -k 10 600 python3 bench.py --workload chain --preimage-mib 64 --steps 3 --warmup 1 --consumer $c --cpu-seconds 0 > $out/chain_$c_$sl.json 2>/dev/null
    python3 -c "
import json
d=json.load(open('$out/chain_$c_$sl.json'))
print('slice $sl consumer $c: %.3f M steps/s' % (d['value']/1e6))
"
  done
done
done | tee $out/chain_slice_chunks.log
