#!/bin/bash
# r04 job 9 (GPU box): the commitments on a stream of their own beside the witness kernels (with and without a CU mask); walk kernel
# with the add-counted wide terms; the predicted-scaling inputs
set -o pipefail
export TMPDIR=/tmp B3W_PROFILE_ROUND=r04
out=gpurun_out/r04_job9
mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_r1cs.py tests/test_gpu_commit.py -x -q -k "gather_kernel or chained_pass" > $out/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $out/tests.log
[ $rc -ne 0 ] && exit $rc
run_chain() { # label consumer envs...
  label=$1; c=$2; shift 2
  env "$@" timeout -k 10 600 python3 bench.py --workload chain --preimage-mib 64 --steps 3 --warmup 1 --consumer $c --cpu-seconds 0 > $out/bench_chain_$label.json 2>$out/bench_chain_$label.err; echo "chain $label rc=$?"
  python3 -c "
import json
d=json.load(open('$out/bench_chain_$label.json'))
print('  value %.3f M steps/s' % (d['value']/1e6), 'frac %.3f' % d['roofline']['frac'], d['roofline']['bound'])
" || tail -3 $out/bench_chain_$label.err
}
run_chain commit_sync commit B3W_CHAIN_COMMIT_ASYNC=0
run_chain commit_async commit B3W_CHAIN_COMMIT_ASYNC=1
run_chain commit_async_cu75 commit B3W_COMMIT_CU_PCT=75
run_chain commit_async_cu88 commit B3W_COMMIT_CU_PCT=88
run_chain check_commit_async check+commit B3W_CHAIN_COMMIT_ASYNC=1
run_chain check_commit_async_cu75 check+commit B3W_COMMIT_CU_PCT=75
timeout -k 10 300 python3 tools/ubench/r1cs_walk_scaling.py nova_vesta 2>&1 | grep -v amdgpu | tee $out/walk_scaling_nova_vesta.log
timeout -k 10 600 python3 tools/ubench/chain_scaling_model.py 1 2>&1 | grep -v amdgpu > $out/chain_scaling_model_1mib.json; echo "model rc=$?"; cat $out/chain_scaling_model_1mib.json
