#!/bin/bash
# r05 session 26: A/B of the walk kernel, three rounds of (HEAD before | product) per circuit, medians of the last 20 launches
set -o pipefail
O=gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_ab
LOG=$O/${ABLOG:-walk_ab_stage.log}; : > $LOG
for c in ${CIRCUITS:-nova_vesta compression nova_bn254_o1}; do
  echo "#### $c" | tee -a $LOG
  for rnd in 1 2 3; do
    for m in ${MASKS:-0 product}; do
      L=hot-proofs-blake3-circom_amd/build/ablate/libb3wit_a$m.so; [ $m = product ] && L=hot-proofs-blake3-circom_amd/libb3wit.so
      D=gpurun_out/prof_ab/${c}_${m}_$rnd
      B3WIT_LIB=$L rocprofv3 --kernel-trace --output-format csv -d $D -- python3 tools/ubench/walk_ablate.py one $c > $D.log 2>&1 || { tail -5 $D.log; exit 1; }
      python3 tools/ubench/walk_trace_median.py $D "$m" | tee -a $LOG
    done
  done
done
