#!/bin/bash
# r04 job 19 (GPU box): the chunk-CV exchange on the chain's side stream — parity of the sharded pass (2 and 3 ranks, three transports),
# then the scaling model again
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r04_job19
mkdir -p $out
timeout -k 10 900 python3 -m pytest tests/test_gpu_native_exchange.py tests/test_gpu_chain.py tests/test_node_addon.py -x -q -m gpu > $out/test_exchange.log 2>&1; rc=$?; tail -3 $out/test_exchange.log; [ $rc -eq 0 ] || exit 1
for mib in 1 64; do timeout -k 10 600 python3 tools/ubench/chain_scaling_model.py $mib 2>/dev/null | python3 -c "import sys; s=sys.stdin.read(); print(s[s.index('{'):])" > $out/chain_scaling_model_${mib}mib.json; echo "model $mib rc=$?"; done
python3 -c "
import json
for mib in (1, 64):
    d = json.load(open('$out/chain_scaling_model_%dmib.json' % mib))
    for r in d['rows']: print(mib, r)
"
B3W_DIST_BACKEND=gloo python3 bench.py --gpus 2 --workload chain --preimage-mib 1 --steps 5 --warmup 2 --exchange-impl native > $out/bench_chain_1mib_config4_gloo_dryrun_2ranks_native.json 2>/dev/null; echo "chain gloo native rc=$?"; cat $out/bench_chain_1mib_config4_gloo_dryrun_2ranks_native.json | cut -c1-300
