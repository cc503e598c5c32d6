#!/bin/bash
# r04 job 26 (GPU box): tampered witnesses of all four circuits through the walk, stream, lean and gather formulations: the same verdicts?
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r04_job26
mkdir -p $out
timeout -k 10 400 python3 tools/ubench/r1cs_fuzz.py $out/walk.npz 4096 2>&1 | grep -v amdgpu | tee $out/fuzz_walk.log
for g in 4 3 1; do B3W_R1CS_GATHER=$g timeout -k 10 600 python3 tools/ubench/r1cs_fuzz.py $out/gather$g.npz 4096 2>&1 | grep -v amdgpu | tee $out/fuzz_gather$g.log; done
python3 tools/ubench/r1cs_fuzz_compare.py $out/walk.npz $out/gather4.npz $out/gather3.npz $out/gather1.npz | tee $out/r1cs_fuzz_compare.log
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
