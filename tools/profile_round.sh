#!/bin/bash
# tools/profile_round.sh <tag> [bench.py args...] — run ON THE GPU BOX (through gpurun), from the repo root:
#   1. rocprofv3 --kernel-trace --stats   -> gpurun_out/prof_<tag>/stats    (per-kernel average duration)
#   2. rocprofv3 --pmc WRITE_SIZE         -> gpurun_out/prof_<tag>/write    (separate pass: TCC slots)
#   3. rocprofv3 --pmc FETCH_SIZE         -> gpurun_out/prof_<tag>/fetch
# and the bench JSON line of the same command.  tools/profile_collect.py then distils them into profiles/rNN/.
# The counter passes run on a plain buffer: under --pmc kernels are serialised and the placement probes time nothing
# meaningful (the bytes counted do not depend on where the buffer lies).
# The program after `--` is python3 itself (no env/bash hop: the profiler has initialised the GPU by then).
set -o pipefail
tag=$1; shift
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
python3 bench.py "$@" --cpu-seconds 0 > $out/bench.json 2> $out/bench.err || { echo "bench failed"; tail -5 $out/bench.err; exit 1; }
# (B3W_PLACE_CHECK=0: the allocator's sanity launches use another batch size and would pull the kernel's average down)
B3W_PLACE_CHECK=0 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py "$@" --cpu-seconds 0 > $out/stats.log 2>&1 || { echo "kernel-trace pass failed"; tail -5 $out/stats.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 bench.py "$@" --cpu-seconds 0 --steps 3 --warmup 1 --inner 1 --placement plain > $out/write.log 2>&1 || { echo "WRITE_SIZE pass failed"; tail -5 $out/write.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py "$@" --cpu-seconds 0 --steps 3 --warmup 1 --inner 1 --placement plain > $out/fetch.log 2>&1 || { echo "FETCH_SIZE pass failed"; tail -5 $out/fetch.log; exit 1; }
python3 tools/profile_collect.py $tag ${B3W_PROFILE_ROUND:-r02}
