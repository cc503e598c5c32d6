#!/bin/bash
# r04 job 1 (GPU box): the whole -m gpu suite on the round's library (native multi-rank exchange over the host transport, threaded
# writer, placement limits, commit counter), then the writer's rate and the commit kernel's multiplication ceiling.
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r04_job1
mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/gpu_suite.log 2>&1; rc=$?; echo "suite rc=$rc"; tail -15 $out/gpu_suite.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python3 tools/ubench/wtns_writer_rate.py 2048 > $out/wtns_writer.log 2>&1; echo "writer rc=$?"; cat $out/wtns_writer.log | grep -v amdgpu
timeout -k 10 120 tools/ubench/fpmul29_peak > $out/fpmul29_peak.log 2>&1; echo "fpmul rc=$?"; cat $out/fpmul29_peak.log
