#!/bin/bash
# tools/profile_sq.sh — run ON THE GPU BOX (through gpurun): SQ wave counters (how waves spend their cycles) of the witness kernels
# and of the constraint check (walk kernel, and round 3's stream kernel beside it); raw CSVs under gpurun_out/prof_sq/, distilled by tools/profile_sq_collect.py.
set -o pipefail
out=gpurun_out/prof_sq
mkdir -p $out
export TMPDIR=/tmp
PMC="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"
rocprofv3 --pmc $PMC --output-format csv -d $out/comp -- python3 bench.py --steps 3 --warmup 1 --inner 1 --placement plain --cpu-seconds 0 > $out/comp.log 2>&1 || { echo comp failed; tail -3 $out/comp.log; exit 1; }
rocprofv3 --pmc $PMC --output-format csv -d $out/nova -- python3 bench.py --circuit nova_vesta --batch 65536 --steps 3 --warmup 1 --inner 1 --placement plain --cpu-seconds 0 > $out/nova.log 2>&1 || { echo nova failed; tail -3 $out/nova.log; exit 1; }
rocprofv3 --pmc $PMC --output-format csv -d $out/r1cs -- python3 tools/ubench/r1cs_profile_target.py compression > $out/r1cs.log 2>&1 || { echo r1cs failed; tail -3 $out/r1cs.log; exit 1; }
rocprofv3 --pmc $PMC --output-format csv -d $out/r1cs_nova -- python3 tools/ubench/r1cs_profile_target.py nova_vesta > $out/r1cs_nova.log 2>&1 || { echo r1cs nova failed; tail -3 $out/r1cs_nova.log; exit 1; }
B3W_R1CS_GATHER=4 rocprofv3 --pmc $PMC --output-format csv -d $out/r1cs_stream -- python3 tools/ubench/r1cs_profile_target.py compression > $out/r1cs_stream.log 2>&1 || { echo r1cs stream failed; tail -3 $out/r1cs_stream.log; exit 1; }
python3 tools/profile_sq_collect.py ${B3W_PROFILE_ROUND:-r04}
