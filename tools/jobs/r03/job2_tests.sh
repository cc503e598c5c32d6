#!/bin/bash
# r03 job 2 (GPU box): the new N2 / exchange / watchdog tests, the ring labels under the r02 timeline's profiler flags, VALUBusy of the commit kernels
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r03_job2
mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_bench_cli.py tests/test_gpu_exchange.py -x -q -m gpu > $out/pytest_new.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest_new.log
python3 tools/ubench/placement_label.py rings > $out/rings_plain_run.json 2> $out/rings_plain_run.err; echo "rings rc=$?"
rocprofv3 --kernel-trace --memory-copy-trace --marker-trace --stats --output-format csv -d $out/rings_prof -- python3 tools/ubench/placement_label.py rings > $out/rings_under_rocprof_timeline_flags.json 2> $out/rings_under_rocprof.err; echo "rings(rocprof) rc=$?"
rocprofv3 --kernel-trace --memory-copy-trace --marker-trace --stats --output-format csv -d $out/chain1g_prof -- python3 bench.py --workload chain --preimage-mib 1024 --steps 1 --warmup 1 > $out/bench_chain_1gib_under_rocprof.json 2> $out/bench_chain_1gib_under_rocprof.err; echo "chain 1gib(rocprof) rc=$?"
rm -rf $out/rings_prof $out/chain1g_prof
rocprofv3 --pmc VALUBusy SALUBusy --output-format csv -d $out/commit_valu -- python3 tools/ubench/commit_rate_folded.py > $out/commit_rate_folded_valu.log 2>&1; echo "commit valu rc=$?"
python3 tools/pmc_distill.py $(ls $out/commit_valu/*/*counter_collection.csv | head -1) b3w_commit > $out/commit_valu_summary.json; echo "distill rc=$?"
rm -rf $out/commit_valu
ls -la $out
