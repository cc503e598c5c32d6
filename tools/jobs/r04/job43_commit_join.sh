#!/bin/bash
# r04 job 43 (GPU box): the fold-shaped pass against the slice size once more, up to ONE slice for the whole local preimage (the commit
# stream then joins the witness stream once per pass)
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r04_job43
mkdir -p $out
for rep in 1 2; do for sl in 8192 16384 65536; do
  B3W_CHAIN_SLICE_CHUNKS=$sl timeout -k 10 600 python3 bench.py --workload chain --preimage-mib 64 --steps 3 --warmup 1 --consumer commit --cpu-seconds 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('slice $sl commit: %.3f M steps/s' % (d['value']/1e6))
"; done; done | tee $out/commit_join.log
