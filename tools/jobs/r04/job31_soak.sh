#!/bin/bash
# r04 job 31 (GPU box): the native exchange tests with one more shape — 8 MiB + 77 bytes (8 193 chunks: ragged over 2 and 3 ranks, several
# preimage slices a rank, a partial last chunk) — and the chained-pass tests beside them
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r04_job31
mkdir -p $out
B3W_TEST_EXTRA_PREIMAGE_BYTES=8388685 B3W_CHAIN_SLICE_CHUNKS=1024 timeout -k 10 1100 python3 -m pytest tests/test_gpu_native_exchange.py -x -q -m gpu --durations=5 > $out/native_exchange_soak.log 2>&1; rc=$?; tail -12 $out/native_exchange_soak.log; exit $rc
