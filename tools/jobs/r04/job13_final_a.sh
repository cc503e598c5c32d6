#!/bin/bash
# r04 job 13 (GPU box): the round's evidence, part A — bench lines with their rocprofv3 summaries and PMC passes, the constraint
# check's profile and SQ counters
set -o pipefail
export TMPDIR=/tmp B3W_PROFILE_ROUND=r04
out=gpurun_out/r04_job13
mkdir -p $out
bash tools/profile_round.sh compression_b4096_n1 > $out/prof_comp.log 2>&1; echo "comp rc=$?"; tail -2 $out/prof_comp.log
bash tools/profile_round.sh nova_vesta_b65536_n1 --circuit nova_vesta --batch 65536 > $out/prof_nova.log 2>&1; echo "nova rc=$?"; tail -2 $out/prof_nova.log
bash tools/profile_r1cs.sh > $out/profile_r1cs.log 2>&1; echo "r1cs rc=$?"; grep -E "walk|deferred|check_us|frac|fetch_over" $out/profile_r1cs.log
bash tools/profile_sq.sh > $out/profile_sq.log 2>&1; echo "sq rc=$?"; tail -12 $out/profile_sq.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_command_steps20_warmup5.json 2> $out/bench_driver_command.err; echo "bench rc=$?"; cat $out/bench_driver_command_steps20_warmup5.json | python3 -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['roofline']['frac'], d['config']['placement'], d['config'].get('placement_search_s'), d['cpu_baseline']['value'])"
mkdir -p $out/profiles_r04 && cp -r profiles/r04/* $out/profiles_r04/ && cp profiles/traffic_latest.json $out/
