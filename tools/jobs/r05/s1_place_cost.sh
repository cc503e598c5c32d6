#!/bin/bash
# r05 session 1: what a placement search costs, step by step, and what the bench's placement does with library defaults
set -o pipefail
O=gpurun_out/r05; mkdir -p $O
cd tools/ubench && hipcc --offload-arch=gfx950 -O2 -o place_cost place_cost.hip && cd ../.. || exit 1
timeout -k 10 300 tools/ubench/place_cost > $O/place_cost.log 2>&1; echo "place_cost rc=$?"
tail -30 $O/place_cost.log
echo "--- bench, 160 GiB search as r04's bench forced"
B3W_PLACE_DEBUG=1 timeout -k 10 300 python bench.py --steps 3 --warmup 1 --timed-ms 1500 --placement-search-gib 160 > $O/bench_search160.json 2> $O/bench_search160.err; echo rc=$?
grep -v "^$" $O/bench_search160.err | tail -20
python -c "import json;d=json.load(open('$O/bench_search160.json'));c=d['config'];print(d['value'],d['roofline']['frac'],c['placement'],{k:v for k,v in c.items() if k.startswith('placement_')})"
echo "--- bench, library defaults"
B3W_PLACE_DEBUG=1 timeout -k 10 300 python bench.py --steps 3 --warmup 1 --timed-ms 1500 > $O/bench_libdefault.json 2> $O/bench_libdefault.err; echo rc=$?
grep -v "^$" $O/bench_libdefault.err | tail -20
python -c "import json;d=json.load(open('$O/bench_libdefault.json'));c=d['config'];print(d['value'],d['roofline']['frac'],c['placement'],{k:v for k,v in c.items() if k.startswith('placement_')})"
