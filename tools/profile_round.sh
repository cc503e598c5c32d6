#!/bin/bash
# tools/profile_round.sh <tag> [bench.py args...] — run ON THE GPU BOX (through gpurun), from the repo root:
#   1. rocprofv3 --kernel-trace --stats   -> gpurun_out/prof_<tag>/stats    (per-kernel average duration)
#   2. rocprofv3 --pmc WRITE_SIZE         -> gpurun_out/prof_<tag>/write    (separate pass: TCC slots)
#   3. rocprofv3 --pmc FETCH_SIZE         -> gpurun_out/prof_<tag>/fetch
# and the bench JSON line of the same command.  tools/profile_collect.py then distils them into profiles/rNN/.
# The counter passes run the SAME command as the timed run — placed buffer, the default launches per step — so that the bytes in
# profiles/traffic_latest.json describe the launch that was timed (r03; r01/r02 counted on a plain buffer with --inner 1).  Under
# --pmc kernels are serialised, which changes durations, not bytes; B3W_PLACE_CHECK=0 keeps the allocator's timed sanity launches
# (meaningless under the profiler) out of it, and --timed-ms bounds the number of counted dispatches.  The allocator's PROBES are timed
# stores too: under --pmc they run four times slower and noisier, a search may find one class only (r04: 165 GiB walked, all labelled
# alike) — --placement-search-s 8 ends such a search early; the bytes a launch moves do not depend on where the buffer lies.
# The program after `--` is python3 itself (no env/bash hop: the profiler has initialised the GPU by then).
set -o pipefail
tag=$1; shift
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
python3 bench.py "$@" --cpu-seconds 0 > $out/bench.json 2> $out/bench.err || { echo "bench failed"; tail -5 $out/bench.err; exit 1; }
# (B3W_PLACE_CHECK=0: the allocator's sanity launches use another batch size and would pull the kernel's average down)
B3W_PLACE_CHECK=0 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py "$@" --cpu-seconds 0 --traffic quoted > $out/stats.log 2>&1 || { echo "kernel-trace pass failed"; tail -5 $out/stats.log; exit 1; }
B3W_PLACE_CHECK=0 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 bench.py "$@" --cpu-seconds 0 --traffic quoted --timed-ms 400 --placement-search-s 8 > $out/write.log 2>&1 || { echo "WRITE_SIZE pass failed"; tail -5 $out/write.log; exit 1; }
B3W_PLACE_CHECK=0 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py "$@" --cpu-seconds 0 --traffic quoted --timed-ms 400 --placement-search-s 8 > $out/fetch.log 2>&1 || { echo "FETCH_SIZE pass failed"; tail -5 $out/fetch.log; exit 1; }
python3 tools/profile_collect.py $tag ${B3W_PROFILE_ROUND:-r06}
