#!/bin/bash
# r04 job 3 (GPU box): the walk kernel under rocprofv3 (kernel stats + FETCH_SIZE), and its sensitivity to the number of workgroups
set -o pipefail
export TMPDIR=/tmp B3W_PROFILE_ROUND=r04
out=gpurun_out/r04_job3
mkdir -p $out
bash tools/profile_r1cs.sh > $out/profile_r1cs.log 2>&1; echo "profile rc=$?"; tail -45 $out/profile_r1cs.log
mkdir -p $out/profiles_r04 && cp profiles/r04/r1cs_check* $out/profiles_r04/
timeout -k 10 300 python3 tools/ubench/r1cs_walk_grid.py compression 2>&1 | grep -v amdgpu | tee $out/walk_grid_compression.log
timeout -k 10 300 python3 tools/ubench/r1cs_walk_grid.py nova_vesta 2>&1 | grep -v amdgpu | tee $out/walk_grid_nova_vesta.log
