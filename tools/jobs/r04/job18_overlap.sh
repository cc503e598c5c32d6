#!/bin/bash
# r04 job 18 (GPU box): the deferred kernel beside the next slab's walk kernel — parity (three slabs of nova steps), then A/B on one box
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r04_job18
mkdir -p $out
timeout -k 10 900 python3 -m pytest tests/test_gpu_r1cs.py -x -q -m gpu > $out/test_r1cs.log 2>&1; rc=$?; tail -3 $out/test_r1cs.log; [ $rc -eq 0 ] || exit 1
for ov in 0 1 0 1; do
  echo "B3W_R1CS_OVERLAP=$ov"
  B3W_R1CS_OVERLAP=$ov timeout -k 10 300 python3 tools/ubench/r1cs_rate_big.py 2>&1 | grep -v amdgpu | tee -a $out/r1cs_rate_big_overlap$ov.log
done
for ov in 0 1; do
  B3W_R1CS_OVERLAP=$ov timeout -k 10 600 python3 bench.py --workload chain --preimage-mib 64 --steps 3 --warmup 1 --consumer check --cpu-seconds 0 > $out/bench_chain_64mib_consumer_check_overlap$ov.json 2>$out/bench_chain_check_$ov.err; echo "chain check overlap=$ov rc=$?"
  python3 -c "
import json
d=json.load(open('$out/bench_chain_64mib_consumer_check_overlap$ov.json'))
print('  value %.3f M steps/s' % (d['value']/1e6), 'frac %.3f' % d['roofline']['frac'])
"
done
for mib in 1 64; do timeout -k 10 600 python3 tools/ubench/chain_scaling_model.py $mib 2>/dev/null | python3 -c "import sys; s=sys.stdin.read(); print(s[s.index('{'):])" > $out/chain_scaling_model_${mib}mib.json; echo "model $mib rc=$?"; done
python3 -c "
import json
for mib in (1, 64):
    d = json.load(open('$out/chain_scaling_model_%dmib.json' % mib))
    for r in d['rows']: print(mib, r)
"
