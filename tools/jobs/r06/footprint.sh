#!/bin/bash
# footprint.sh — does a body buffer store slower because it is BIG?  The same store-only shapes over 4 096, 16 384 and 65 536 nova bodies
# (3 / 12 / 49 GB), plain and placed (tools/ubench/store_sweep.hip, a few shapes): profiles/r06/store_footprint.log
mkdir -p gpurun_out/r06
for n in 4096 16384 65536; do
  SWEEP_ONLY=SPF tools/ubench/store_sweep $n 745312 4 quick 2>&1 | grep -E "^store_sweep|^shape|^S4 |^S8 |^S8v4|^P4x512 |^P4x512v4|^P8x512v4|^F256x256 |^best"
done
