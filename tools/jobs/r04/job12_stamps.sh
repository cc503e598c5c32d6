#!/bin/bash
# r04 job 12 (GPU box): per-phase cycle stamps of the walk kernel — a DIAGNOSTIC build of the library, made on the box and only there
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r04_job12
mkdir -p $out
B3W_BUILD_DIAG=1 python3 -c "import importlib; b = importlib.import_module('hot-proofs-blake3-circom_amd.build'); b.build_lib()" > $out/build.log 2>&1; echo "diag build rc=$?"; tail -2 $out/build.log
B3W_R1CS_STAMPS=1 python3 tools/ubench/r1cs_profile_target.py compression 2>&1 | grep -v amdgpu | tail -10 | tee $out/walk_stamps_compression.log
B3W_R1CS_STAMPS=1 python3 tools/ubench/r1cs_profile_target.py nova_vesta 2>&1 | grep -v amdgpu | tail -10 | tee $out/walk_stamps_nova_vesta.log
