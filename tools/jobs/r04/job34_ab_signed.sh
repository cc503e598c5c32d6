#!/bin/bash
# r04 job 34 (GPU box): A/B on one box — the library before the signed elements (libb3wit_prev.so, built from commit 4847ed4) against
# the current one, alternating
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r04_job34
mkdir -p $out
for rep in 1 2 3; do
  for lib in prev cur; do
    for c in nova_vesta compression; do
      if [ $lib = prev ]; then export B3WIT_LIB=$PWD/hot-proofs-blake3-circom_amd/libb3wit_prev.so; else unset B3WIT_LIB; fi
      echo -n "$lib $c: "; timeout -k 10 200 python3 tools/ubench/r1cs_walk_scaling.py $c 2>&1 | grep "n=  8192\|n=  4096" | tr '\n' ' '; echo
    done
  done
done | tee $out/ab_signed.log
