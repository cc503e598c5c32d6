#!/bin/bash
# r05 session 8: config 4's scaling model with the one-launch tree; 4-rank gloo dry run; walk-kernel stamps (diagnostic build, last)
set -o pipefail
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 300 python tools/ubench/chain_scaling_model.py 1 > $O/chain_scaling_model_1mib.json 2> $O/chain_scaling_model_1mib.err; echo "model rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05/chain_scaling_model_1mib.json'))
for r in d['rows']: print(r['ranks'], r['rank0_steps'], 'pass', r['rank0_pass_ms'], 'single', r['rank0_single_pass_median_ms'], 'unsharded-of-shard', r['its_single_pass_median_ms'], r['predicted_M_steps_per_s_with_50us_per_collective'])
PY
B3W_DIST_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 4 --workload chain --preimage-mib 1 --steps 5 --warmup 2 --exchange-impl native > $O/bench_chain_1mib_config4_gloo_dryrun_4ranks_native.json 2> $O/gloo4.err; echo "gloo x4 native rc=$?"
python -c "import json;d=json.load(open('$O/bench_chain_1mib_config4_gloo_dryrun_4ranks_native.json'));print(d['value'],d['config']['exchange_impl'],d['config']['pass_ms_per_rank'],d['config']['exchange_ms_per_rank'])"
timeout -k 10 200 python bench.py --workload chain --preimage-mib 1 --steps 20 --warmup 5 --cpu-seconds 0 > $O/bench_chain_1mib_config4_n1.json 2>/dev/null; echo "chain 1 MiB rc=$?"
python -c "import json;d=json.load(open('$O/bench_chain_1mib_config4_n1.json'));print(d['value'],d['ms_per_step'])"
echo "--- diagnostic build"
B3W_BUILD_DIAG=1 python -c "import importlib; b = importlib.import_module('hot-proofs-blake3-circom_amd.build'); b.build_lib()" > $O/diag_build.log 2>&1; echo "diag build rc=$?"
for c in nova_vesta compression; do B3W_R1CS_STAMPS=1 timeout -k 10 200 python tools/ubench/r1cs_profile_target.py $c 2>&1 | grep -v amdgpu | tail -14 | tee $O/walk_stamps_$c.log; done
