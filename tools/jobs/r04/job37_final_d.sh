#!/bin/bash
# r04 job 37 (GPU box): the check's profile, SQ counters and the driver's bench command once more on the library as it is committed
# (walk units), then the whole -m gpu suite with durations
set -o pipefail
export TMPDIR=/tmp B3W_PROFILE_ROUND=r04
out=gpurun_out/r04_job37
mkdir -p $out
bash tools/profile_r1cs.sh > $out/profile_r1cs.log 2>&1; echo "r1cs rc=$?"; python3 - <<'PY'
import json
d = json.load(open("profiles/r04/r1cs_check.json"))
for c, v in d["circuits"].items():
    print(c, v["kernel_avg_us"], round(v["roofline"]["frac"], 4), round(v["fetch_over_body_bytes"], 4))
PY
bash tools/profile_sq.sh > $out/profile_sq.log 2>&1; echo "sq rc=$?"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_command_steps20_warmup5.json 2> $out/bench_driver_command.err; echo "bench rc=$?"
timeout -k 10 600 python3 bench.py --workload chain --preimage-mib 64 --steps 3 --warmup 1 --consumer check --cpu-seconds 2 > $out/bench_chain_64mib_consumer_check.json 2>/dev/null; echo "chain check rc=$?"
timeout -k 10 300 python3 tools/ubench/r1cs_rate.py 2>&1 | grep -v amdgpu | tee $out/r1cs_rate_walk.log
timeout -k 10 300 python3 tools/ubench/r1cs_rate_big.py 2>&1 | grep -v amdgpu | tee $out/r1cs_rate_big.log
for c in compression nova_vesta nova_bn254_o1; do timeout -k 10 300 python3 tools/ubench/r1cs_walk_scaling.py $c 2>&1 | grep -v amdgpu > $out/walk_scaling_$c.log; done
mkdir -p $out/profiles_r04 && cp profiles/r04/r1cs_check* profiles/r04/sq_counters.json $out/profiles_r04/
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q --durations=15 > $out/gpu_suite.log 2>&1; rc=$?; echo "suite rc=$rc"; tail -22 $out/gpu_suite.log
