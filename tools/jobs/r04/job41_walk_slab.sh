#!/bin/bash
# r04 job 41 (GPU box): the walk check's slab — 8 192 / 16 384 / 32 768 bodies per launch pair
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r04_job41
mkdir -p $out
for rep in 1 2; do for sl in 8192 16384 32768; do echo "slab $sl"; B3W_R1CS_WALK_SLAB=$sl timeout -k 10 300 python3 tools/ubench/r1cs_rate_big.py 2>&1 | grep -v amdgpu; done; done | tee $out/walk_slab.log
for sl in 8192 32768; do B3W_R1CS_WALK_SLAB=$sl timeout -k 10 600 python3 bench.py --workload chain --preimage-mib 64 --steps 3 --warmup 1 --consumer check --cpu-seconds 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('slab $sl chain check: %.3f M steps/s' % (d['value']/1e6))
"; done | tee -a $out/walk_slab.log
