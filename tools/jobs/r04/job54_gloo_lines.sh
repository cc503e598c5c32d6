#!/bin/bash
# r04 job 54 (GPU box): the two-rank dry runs once more on the final bench.py (cross-rank checks of what was gathered)
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r04_job54
mkdir -p $out
B3W_DIST_BACKEND=gloo python3 bench.py --gpus 2 --workload chain --preimage-mib 1 --steps 5 --warmup 2 > $out/bench_chain_1mib_config4_gloo_dryrun_2ranks_torch.json 2>/dev/null; echo "chain gloo torch rc=$?"
B3W_DIST_BACKEND=gloo python3 bench.py --gpus 2 --workload chain --preimage-mib 1 --steps 5 --warmup 2 --exchange-impl native > $out/bench_chain_1mib_config4_gloo_dryrun_2ranks_native.json 2>/dev/null; echo "chain gloo native rc=$?"
for m in every last none; do B3W_DIST_BACKEND=gloo python3 bench.py --gpus 2 --steps 3 --warmup 1 --timed-ms 600 --exchange $m > $out/bench_gloo_dryrun_2ranks_exchange_$m.json 2>/dev/null; echo "gloo $m rc=$?"; done
B3W_DIST_BACKEND=gloo python3 bench.py --gpus 2 --steps 3 --warmup 1 --timed-ms 600 --exchange every --exchange-impl native > $out/bench_gloo_dryrun_2ranks_exchange_every_native.json 2>/dev/null; echo "gloo native rc=$?"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_command_steps20_warmup5.json 2>/dev/null; echo "driver rc=$?"
wc -l $out/*.json
