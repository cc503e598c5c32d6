#!/bin/bash
# r04 job 39 (GPU box): preimage slices that grow with the preimage — the chained-pass tests, then the chain bench lines again
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r04_job39
mkdir -p $out
timeout -k 10 900 python3 -m pytest tests/test_gpu_chain.py tests/test_gpu_commit.py tests/test_gpu_native_exchange.py tests/test_node_addon.py -x -q -m gpu > $out/test_chain.log 2>&1; rc=$?; tail -3 $out/test_chain.log; [ $rc -eq 0 ] || exit 1
for c in none check commit check+commit commit-only commit-bodies; do timeout -k 10 600 python3 bench.py --workload chain --preimage-mib 64 --steps 3 --warmup 1 --consumer $c --cpu-seconds 2 > $out/bench_chain_64mib_consumer_${c//[+-]/_}.json 2>$out/bench_chain_$c.err; echo "chain $c rc=$?"; python3 -c "
import json
d=json.load(open('$out/bench_chain_64mib_consumer_${c//[+-]/_}.json'))
print('  value %.3f M steps/s' % (d['value']/1e6), 'frac %.3f' % d['roofline']['frac'], d['roofline']['bound'])
" || tail -3 $out/bench_chain_$c.err; done
python3 bench.py --workload chain --preimage-mib 1 --steps 10 --warmup 2 --cpu-seconds 2 > $out/bench_chain_1mib_config4_n1.json 2>/dev/null; echo "chain 1mib rc=$?"
python3 bench.py --workload chain --preimage-mib 1024 --steps 2 --warmup 1 --cpu-seconds 0 > $out/bench_chain_1gib_config5_n1.json 2>/dev/null; echo "chain 1gib rc=$?"
python3 -c "
import json
for f in ('bench_chain_1mib_config4_n1','bench_chain_1gib_config5_n1'):
    d=json.load(open('$out/'+f+'.json')); print(f, '%.3f M steps/s' % (d['value']/1e6), 'frac %.3f' % d['roofline']['frac'])
"
for mib in 1 64; do timeout -k 10 600 python3 tools/ubench/chain_scaling_model.py $mib 2>/dev/null | python3 -c "import sys; s=sys.stdin.read(); print(s[s.index('{'):])" > $out/chain_scaling_model_${mib}mib.json; echo "model $mib rc=$?"; done
