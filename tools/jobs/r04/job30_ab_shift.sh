#!/bin/bash
# r04 job 30 (GPU box): A/B on one box — the walk kernel with and without the shifting terms (B3W_WALK_SHIFT_TERMS=0: every term multiplies)
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r04_job30
mkdir -p $out
for rep in 1 2 3; do
  for sh in 0 1; do
    for c in nova_vesta compression; do
      echo -n "shift=$sh $c: "; B3W_WALK_SHIFT_TERMS=$sh timeout -k 10 200 python3 tools/ubench/r1cs_walk_scaling.py $c 2>&1 | grep "n=  8192\|n=  4096" | tr '\n' ' '; echo
    done
  done
done | tee $out/ab_shift_terms.log
