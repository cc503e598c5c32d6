#!/bin/bash
# tools/profile_all.sh — run ON THE GPU BOX (through gpurun): every profile of the round with the current library
# (tools/profile_round.sh for both bench configurations, tools/profile_r1cs.sh, the default bench line); results under gpurun_out/.
set -o pipefail
export TMPDIR=/tmp
bash tools/profile_round.sh compression_b4096_n1 > gpurun_out/prof_comp.log 2>&1; echo "comp rc=$?"
bash tools/profile_round.sh nova_vesta_b65536_n1 --circuit nova_vesta --batch 65536 > gpurun_out/prof_nova.log 2>&1; echo "nova rc=$?"
bash tools/profile_r1cs.sh > gpurun_out/profile_r1cs.log 2>&1; echo "r1cs rc=$?"
python bench.py --steps 20 --warmup 5 > gpurun_out/r2_bench_final3.json 2>/dev/null; echo "bench rc=$?"
mkdir -p gpurun_out/profiles_r02 && cp -r profiles/r02/* gpurun_out/profiles_r02/ && cp profiles/traffic_latest.json gpurun_out/
