#!/bin/bash
# r03 job 10 (GPU box): the constraint check's evidence again after the entry-list / fetch-order changes
set -o pipefail
export TMPDIR=/tmp B3W_PROFILE_ROUND=r03
out=gpurun_out/r03_job10
mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_r1cs.py tests/test_gpu_chain.py tests/test_gpu_graph_capture.py -x -q -m gpu > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $out/pytest.log
bash tools/profile_r1cs.sh > $out/profile_r1cs.log 2>&1; echo "r1cs rc=$?"
bash tools/profile_sq.sh > $out/profile_sq.log 2>&1; echo "sq rc=$?"; tail -8 $out/profile_sq.log
bash tools/r03/job8_r1cs_busy.sh > $out/r1cs_busy_counters.log 2>&1; echo "busy rc=$?"
for c in check check+commit; do python3 bench.py --workload chain --preimage-mib 64 --steps 3 --warmup 1 --consumer $c > $out/bench_chain_64mib_consumer_${c//+/_}.json 2>/dev/null; echo "chain $c rc=$?"; done
python3 tools/ubench/r1cs_stream_dbg.py compression 4096 > $out/r1cs_stream_breakdown_compression.log 2>&1
python3 tools/ubench/r1cs_stream_dbg.py nova_vesta 4096 > $out/r1cs_stream_breakdown_nova_vesta.log 2>&1
python3 tools/ubench/r1cs_rate.py 2>&1 | grep -v amdgpu > $out/r1cs_rate_stream.log
python3 tools/ubench/r1cs_rate_big.py 2>&1 | grep -v amdgpu > $out/r1cs_rate_big.log
B3W_R1CS_STAMPS=1 python3 tools/ubench/r1cs_profile_target.py compression 2>&1 | tail -10 > $out/r1cs_stream_stamps_compression.log
B3W_R1CS_STAMPS=1 python3 tools/ubench/r1cs_profile_target.py nova_vesta 2>&1 | tail -10 > $out/r1cs_stream_stamps_nova_vesta.log
mkdir -p $out/profiles_r03 && cp profiles/r03/r1cs_check* profiles/r03/sq_counters.json $out/profiles_r03/
cat $out/profiles_r03/r1cs_check.json | python3 -c "import json,sys; d=json.load(sys.stdin); [print(k, v['kernel_avg_us'], round(v['roofline']['frac'],3), round(v['fetch_over_body_bytes'],3)) for k,v in d['circuits'].items()]"
cat $out/r1cs_rate_stream.log $out/r1cs_rate_big.log; grep -h value $out/bench_chain*.json | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['config']['consumer'][:40], round(d['value']))"
