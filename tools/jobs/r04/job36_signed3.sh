#!/bin/bash
# r04 job 36 (GPU box): the walk kernel in two instantiations (SIGNED elements where the system keeps linear rows) — parity, then A/B against libb3wit_prev.so
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r04_job36
mkdir -p $out
timeout -k 10 900 python3 -m pytest tests/test_gpu_r1cs.py -x -q -m gpu > $out/test_r1cs.log 2>&1; rc=$?; tail -3 $out/test_r1cs.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 400 python3 tools/ubench/r1cs_fuzz.py $out/walk.npz 4096 2>&1 | grep -v amdgpu | tee $out/fuzz_walk.log
B3W_R1CS_GATHER=1 timeout -k 10 600 python3 tools/ubench/r1cs_fuzz.py $out/gather1.npz 4096 2>&1 | grep -v amdgpu > $out/fuzz_gather1.log
python3 tools/ubench/r1cs_fuzz_compare.py $out/walk.npz $out/gather1.npz | tee $out/r1cs_fuzz_compare.log; [ ${PIPESTATUS[0]} -eq 0 ] || exit 1
for rep in 1 2 3; do
  for lib in prev cur; do
    for c in nova_vesta compression nova_bn254_o1; do
      if [ $lib = prev ]; then export B3WIT_LIB=$PWD/hot-proofs-blake3-circom_amd/libb3wit_prev.so; else unset B3WIT_LIB; fi
      echo -n "$lib $c: "; timeout -k 10 200 python3 tools/ubench/r1cs_walk_scaling.py $c 2>&1 | grep "n=  8192\|n=  4096" | tr '\n' ' '; echo
    done
  done
done | tee $out/ab_signed.log
