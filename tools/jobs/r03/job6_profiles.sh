#!/bin/bash
# r03 job 6 (GPU box): the round's evidence — bench lines, rocprofv3 summaries, PMC passes, the constraint check's profiles
set -o pipefail
export TMPDIR=/tmp B3W_PROFILE_ROUND=r03
out=gpurun_out/r03_job6
mkdir -p $out
bash tools/profile_round.sh compression_b4096_n1 > $out/prof_comp.log 2>&1; echo "comp rc=$?"; tail -2 $out/prof_comp.log
bash tools/profile_round.sh nova_vesta_b65536_n1 --circuit nova_vesta --batch 65536 > $out/prof_nova.log 2>&1; echo "nova rc=$?"; tail -2 $out/prof_nova.log
bash tools/profile_r1cs.sh > $out/profile_r1cs.log 2>&1; echo "r1cs rc=$?"
bash tools/profile_sq.sh > $out/profile_sq.log 2>&1; echo "sq rc=$?"; tail -8 $out/profile_sq.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_command_steps20_warmup5.json 2> $out/bench_driver_command.err; echo "bench rc=$?"
for c in none check commit check+commit commit-only; do python3 bench.py --workload chain --preimage-mib 64 --steps 3 --warmup 1 --consumer $c > $out/bench_chain_64mib_consumer_${c//+/_}.json 2>/dev/null; echo "chain $c rc=$?"; done
python3 bench.py --workload chain --preimage-mib 1 --steps 10 --warmup 2 > $out/bench_chain_1mib_config4_n1.json 2>/dev/null; echo "chain 1mib rc=$?"
B3W_DIST_BACKEND=gloo python3 bench.py --gpus 2 --workload chain --preimage-mib 1 --steps 5 --warmup 2 > $out/bench_chain_1mib_config4_gloo_dryrun_2ranks_one_gpu.json 2>/dev/null; echo "chain gloo rc=$?"
for m in every last none; do B3W_DIST_BACKEND=gloo python3 bench.py --gpus 2 --steps 3 --warmup 1 --timed-ms 600 --exchange $m > $out/bench_gloo_dryrun_2ranks_exchange_$m.json 2>/dev/null; echo "gloo $m rc=$?"; done
python3 tools/ubench/r1cs_stream_dbg.py compression 4096 > $out/r1cs_stream_breakdown_compression.log 2>&1
python3 tools/ubench/r1cs_stream_dbg.py nova_vesta 4096 > $out/r1cs_stream_breakdown_nova_vesta.log 2>&1
python3 tools/ubench/r1cs_rate.py 2>&1 | grep -v amdgpu > $out/r1cs_rate_stream.log
B3W_R1CS_GATHER=3 python3 tools/ubench/r1cs_rate.py 2>&1 | grep -v amdgpu > $out/r1cs_rate_lean.log
B3W_R1CS_STAMPS=1 python3 tools/ubench/r1cs_profile_target.py compression 2>&1 | tail -11 > $out/r1cs_stream_stamps_compression.log
B3W_R1CS_STAMPS=1 python3 tools/ubench/r1cs_profile_target.py nova_vesta 2>&1 | tail -11 > $out/r1cs_stream_stamps_nova_vesta.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/commit_stats -- python3 tools/ubench/commit_rate_folded.py > $out/commit_rate_folded.log 2>&1; echo "commit stats rc=$?"
cp $(ls $out/commit_stats/*/*kernel_stats.csv | head -1) $out/commit_rate_folded_kernel_stats.csv; rm -rf $out/commit_stats
rocprofv3 --pmc VALUBusy SALUBusy --output-format csv -d $out/commit_valu -- python3 tools/ubench/commit_rate_folded.py > $out/commit_valu.log 2>&1; echo "commit valu rc=$?"
python3 tools/pmc_distill.py $(ls $out/commit_valu/*/*counter_collection.csv | head -1) b3w_commit > $out/commit_rate_folded_pmc_VALUBusy.json; rm -rf $out/commit_valu
mkdir -p $out/profiles_r03 && cp -r profiles/r03/* $out/profiles_r03/ && cp profiles/traffic_latest.json $out/
ls $out
