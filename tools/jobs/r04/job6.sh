#!/bin/bash
# r04 job 6 (GPU box): walk kernel's fixed cost per launch; the chained pass with every consumer (new roofline fields)
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r04_job6
mkdir -p $out
timeout -k 10 300 python3 tools/ubench/r1cs_walk_scaling.py compression 2>&1 | grep -v amdgpu | tee $out/walk_scaling_compression.log
timeout -k 10 300 python3 tools/ubench/r1cs_walk_scaling.py nova_vesta 2>&1 | grep -v amdgpu | tee $out/walk_scaling_nova_vesta.log
for c in none check commit check+commit commit-only; do timeout -k 10 600 python3 bench.py --workload chain --preimage-mib 64 --steps 3 --warmup 1 --consumer $c --cpu-seconds 2 > $out/bench_chain_64mib_consumer_${c//+/_}.json 2>$out/bench_chain_$c.err; echo "chain $c rc=$?"; python3 -c "
import json,sys
d=json.load(open('$out/bench_chain_64mib_consumer_${c//+/_}.json'))
print('  value %.3f M steps/s' % (d['value']/1e6), 'roofline', {k: d['roofline'][k] for k in ('bound','achieved','peak','unit','frac')})
"; done
