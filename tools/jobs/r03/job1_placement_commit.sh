#!/bin/bash
# r03 job 1 (run ON THE GPU BOX through gpurun): placement labels and rocprof evidence for the commit kernels (#6).
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r03_job1
mkdir -p $out
python3 tools/ubench/placement_label.py rings > $out/rings_plain_run.json 2> $out/rings_plain_run.err; echo "rings rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/rings_prof -- python3 tools/ubench/placement_label.py rings > $out/rings_under_rocprof.json 2> $out/rings_under_rocprof.err; echo "rings(rocprof) rc=$?"
python3 bench.py --workload chain --preimage-mib 64 --steps 3 --warmup 1 > $out/bench_chain_64mib.json 2> $out/bench_chain_64mib.err; echo "chain rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/chain_prof -- python3 bench.py --workload chain --preimage-mib 64 --steps 3 --warmup 1 > $out/bench_chain_64mib_under_rocprof.json 2> $out/bench_chain_64mib_under_rocprof.err; echo "chain(rocprof) rc=$?"
python3 tools/ubench/placement_label.py plain50 > $out/plain50.log 2> $out/plain50.err; echo "plain50 rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/commit_stats -- python3 tools/ubench/commit_rate_folded.py > $out/commit_rate_folded_stats.log 2>&1; echo "commit stats rc=$?"
rocprofv3 --pmc VALUBusy SALUBusy --output-format csv -d $out/commit_valu -- python3 tools/ubench/commit_rate_folded.py > $out/commit_rate_folded_valu.log 2>&1; echo "commit valu rc=$?"
find $out -name "*.csv" -size +2M -delete
ls -la $out
