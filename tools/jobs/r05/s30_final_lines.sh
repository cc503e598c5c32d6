#!/bin/bash
# r05 session 30: the bench lines of the final library (batch default, nova batch, chained pass with each consumer, config 4 / 5)
set -o pipefail
O=gpurun_out/r05/final; mkdir -p $O
run() { name=$1; shift; timeout -k 10 500 python bench.py "$@" > $O/$name.json 2> $O/$name.err || { echo "$name FAILED"; tail -3 $O/$name.err; return 1; }
  python -c "
import json; d=json.load(open('$O/$name.json')); r=d['roofline']; print('$name', round(d['value']/1e6,3), 'M/s frac', round(r['frac'],4), r.get('of_measured_ceiling'), (r.get('plain') or {}).get('frac'))"; }
run bench_default
run bench_nova_vesta_b65536_n1 --circuit nova_vesta --batch 65536 --steps 10 --warmup 2
for c in none check commit commit-only check+commit commit-bodies; do run bench_chain_64mib_consumer_${c/+/_} --workload chain --preimage-mib 64 --consumer $c --steps 3 --warmup 1 --cpu-seconds 0; done
run bench_chain_1mib_config4_n1 --workload chain --preimage-mib 1 --steps 20 --warmup 5 --cpu-seconds 0
run bench_chain_1gib_config5_n1 --workload chain --preimage-mib 1024 --steps 2 --warmup 1 --cpu-seconds 0
