#!/bin/bash
# r04 job 44 (GPU box): the fold-shaped pass with the witness kernel's smaller variant (fewer registers and less LDS per wave: room for
# commit waves on the same SIMD?)
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r04_job44
mkdir -p $out
for rep in 1 2; do for v in auto 0; do for c in none commit; do
  if [ $v = auto ]; then unset B3W_VARIANT; else export B3W_VARIANT=$v; fi
  timeout -k 10 600 python3 bench.py --workload chain --preimage-mib 64 --steps 3 --warmup 1 --consumer $c --cpu-seconds 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('variant $v consumer $c: %.3f M steps/s' % (d['value']/1e6))
"; done; done; done | tee $out/variant_commit.log
