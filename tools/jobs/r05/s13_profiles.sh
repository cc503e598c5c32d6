#!/bin/bash
# r05 session 13: the round's profiles with the final library: rocprofv3 kernel statistics + PMC passes of the default bench command and of
# the nova batch, the constraint check's passes, the driver's command, the chained-pass lines
set -o pipefail
export TMPDIR=/tmp B3W_PROFILE_ROUND=r05
O=gpurun_out/r05; mkdir -p $O
bash tools/profile_round.sh compression_b4096_n1 > $O/prof_comp.log 2>&1; echo "comp rc=$?"; tail -3 $O/prof_comp.log
bash tools/profile_round.sh nova_vesta_b65536_n1 --circuit nova_vesta --batch 65536 > $O/prof_nova.log 2>&1; echo "nova rc=$?"; tail -3 $O/prof_nova.log
bash tools/profile_r1cs.sh > $O/profile_r1cs.log 2>&1; echo "r1cs rc=$?"; tail -4 $O/profile_r1cs.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command_steps20_warmup5.json 2> $O/bench_driver.err; echo "driver command rc=$?"
python bench.py > $O/bench_default.json 2>/dev/null; echo "default rc=$?"
for c in none check commit commit-only check+commit; do
  python bench.py --workload chain --preimage-mib 64 --steps 5 --warmup 2 --consumer $c > $O/bench_chain_64mib_consumer_$(echo $c | tr '+-' '__').json 2>/dev/null; echo "chain 64 MiB $c rc=$?"
done
python bench.py --workload chain --preimage-mib 1 --steps 20 --warmup 5 > $O/bench_chain_1mib_config4_n1.json 2>/dev/null; echo "chain 1 MiB rc=$?"
python bench.py --workload chain --preimage-mib 1024 --steps 2 --warmup 1 --cpu-seconds 0 > $O/bench_chain_1gib_config5_n1.json 2>/dev/null; echo "chain 1 GiB rc=$?"
B3W_DIST_BACKEND=gloo python bench.py --gpus 4 --workload chain --preimage-mib 1 --steps 5 --warmup 2 --exchange-impl native > $O/bench_chain_1mib_config4_gloo_dryrun_4ranks_native.json 2>/dev/null; echo "gloo x4 rc=$?"
B3W_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 3 --warmup 1 --timed-ms 600 > $O/bench_gloo_dryrun_2ranks.json 2>/dev/null; echo "gloo x2 batch rc=$?"
mkdir -p gpurun_out/profiles_r05 && cp -r profiles/r05/* gpurun_out/profiles_r05/ && cp profiles/traffic_latest.json gpurun_out/profiles_r05/traffic_latest.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05/bench_*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']
        print(f.split('/')[-1], round(d['value']/1e6,3), 'M/s frac', round(r['frac'],3), r['bound'], d['config'].get('placement'), r.get('of_measured_ceiling'))
    except Exception as e: print(f, 'ERR', e)
PY
