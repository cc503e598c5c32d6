#!/bin/bash
# r04 job 22 (GPU box): the round's evidence on the final library, part B — the chained pass with every consumer, configs 4 and 5 on one GPU, the two-rank
# dry runs (torch and native exchange), the scaling model's inputs, rates
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r04_job22
mkdir -p $out
for c in none check commit check+commit commit-only commit-bodies; do timeout -k 10 600 python3 bench.py --workload chain --preimage-mib 64 --steps 3 --warmup 1 --consumer $c --cpu-seconds 2 > $out/bench_chain_64mib_consumer_${c//[+-]/_}.json 2>$out/bench_chain_$c.err; echo "chain $c rc=$?"; python3 -c "
import json
d=json.load(open('$out/bench_chain_64mib_consumer_${c//[+-]/_}.json'))
print('  value %.3f M steps/s' % (d['value']/1e6), 'frac %.3f' % d['roofline']['frac'], d['roofline']['bound'])
" || tail -3 $out/bench_chain_$c.err; done
python3 bench.py --workload chain --preimage-mib 1 --steps 10 --warmup 2 --cpu-seconds 2 > $out/bench_chain_1mib_config4_n1.json 2>/dev/null; echo "chain 1mib rc=$?"
python3 bench.py --workload chain --preimage-mib 1024 --steps 2 --warmup 1 --cpu-seconds 0 > $out/bench_chain_1gib_config5_n1.json 2>/dev/null; echo "chain 1gib rc=$?"
B3W_DIST_BACKEND=gloo python3 bench.py --gpus 2 --workload chain --preimage-mib 1 --steps 5 --warmup 2 > $out/bench_chain_1mib_config4_gloo_dryrun_2ranks_torch.json 2>/dev/null; echo "chain gloo torch rc=$?"
B3W_DIST_BACKEND=gloo python3 bench.py --gpus 2 --workload chain --preimage-mib 1 --steps 5 --warmup 2 --exchange-impl native > $out/bench_chain_1mib_config4_gloo_dryrun_2ranks_native.json 2>/dev/null; echo "chain gloo native rc=$?"
for m in every last none; do B3W_DIST_BACKEND=gloo python3 bench.py --gpus 2 --steps 3 --warmup 1 --timed-ms 600 --exchange $m > $out/bench_gloo_dryrun_2ranks_exchange_$m.json 2>/dev/null; echo "gloo $m rc=$?"; done
B3W_DIST_BACKEND=gloo python3 bench.py --gpus 2 --steps 3 --warmup 1 --timed-ms 600 --exchange every --exchange-impl native > $out/bench_gloo_dryrun_2ranks_exchange_every_native.json 2>/dev/null; echo "gloo native rc=$?"
for mib in 1 64; do timeout -k 10 600 python3 tools/ubench/chain_scaling_model.py $mib 2>/dev/null | python3 -c "import sys; s=sys.stdin.read(); print(s[s.index('{'):])" > $out/chain_scaling_model_${mib}mib.json; echo "model $mib rc=$?"; done
timeout -k 10 300 python3 tools/ubench/r1cs_rate.py 2>&1 | grep -v amdgpu > $out/r1cs_rate_walk.log; cat $out/r1cs_rate_walk.log
timeout -k 10 300 python3 tools/ubench/r1cs_rate_big.py 2>&1 | grep -v amdgpu > $out/r1cs_rate_big.log; cat $out/r1cs_rate_big.log
timeout -k 10 300 python3 tools/ubench/r1cs_walk_scaling.py compression 2>&1 | grep -v amdgpu > $out/walk_scaling_compression.log
timeout -k 10 300 python3 tools/ubench/r1cs_walk_scaling.py nova_vesta 2>&1 | grep -v amdgpu > $out/walk_scaling_nova_vesta.log
timeout -k 10 400 python3 tools/ubench/commit_rate_folded.py 2>&1 | grep -v amdgpu > $out/commit_rate_folded.log; tail -12 $out/commit_rate_folded.log
