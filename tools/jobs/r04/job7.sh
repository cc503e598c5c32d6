#!/bin/bash
# r04 job 7 (GPU box): walk with record stores behind the pack; chain consumers with commit-from-records; writer vs the filesystem alone
set -o pipefail
export TMPDIR=/tmp B3W_PROFILE_ROUND=r04
out=gpurun_out/r04_job7
mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_r1cs.py tests/test_gpu_commit.py -x -q > $out/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $out/tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python3 tools/ubench/r1cs_walk_scaling.py nova_vesta 2>&1 | grep -v amdgpu | tee $out/walk_scaling_nova_vesta.log
bash tools/profile_r1cs.sh > $out/profile_r1cs.log 2>&1; echo "profile rc=$?"; grep -E "walk|deferred|check_us|frac|fetch_over" $out/profile_r1cs.log
mkdir -p $out/profiles_r04 && cp profiles/r04/r1cs_check* $out/profiles_r04/
for c in none check commit check+commit commit-bodies; do timeout -k 10 600 python3 bench.py --workload chain --preimage-mib 64 --steps 3 --warmup 1 --consumer $c --cpu-seconds 2 > $out/bench_chain_64mib_consumer_${c//+/_}.json 2>$out/bench_chain_$c.err; echo "chain $c rc=$?"; python3 -c "
import json,sys
d=json.load(open('$out/bench_chain_64mib_consumer_${c//+/_}.json'))
print('  value %.3f M steps/s' % (d['value']/1e6), 'roofline', {k: d['roofline'][k] for k in ('bound','achieved','peak','unit','frac')})
" || tail -3 $out/bench_chain_$c.err; done
timeout -k 10 400 python3 tools/ubench/wtns_writer_rate.py 2048 2>&1 | grep -v amdgpu | tee $out/wtns_writer.log
