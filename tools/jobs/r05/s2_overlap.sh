#!/bin/bash
# r05 session 2: the fold-shaped pass with the commit stream gated to the witness kernel; the new default bench line
set -o pipefail
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_commit.py -x -q -k "commit_only_matches or check_then_commit" > $O/s2_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/s2_pytest.log
echo "--- default bench line"
timeout -k 10 300 python bench.py > $O/bench_default_s2.json 2> $O/bench_default_s2.err; echo rc=$?
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05/bench_default_s2.json'))
r=d['roofline']; c=d['config']
print('value',d['value'],'frac',r['frac'],'plain',r.get('plain'),'of_measured',r.get('of_measured_ceiling'),r.get('plain_of_measured_ceiling'))
print('ceil',r.get('store_ceiling'))
print({k:v for k,v in c.items() if k.startswith('placement')})
PY
run() {  # tag, env..., -- args
  tag=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 300 python bench.py --workload chain --preimage-mib 64 --steps 3 --warmup 1 --cpu-seconds 0 "$@" > $O/chain_$tag.json 2> $O/chain_$tag.err
  rc=$?
  python - "$tag" "$rc" <<'PY'
import json,sys
tag,rc=sys.argv[1],sys.argv[2]
try:
    d=json.load(open(f'gpurun_out/r05/chain_{tag}.json'))
    print(f"{tag:40s} rc={rc} {d['value']/1e6:7.3f} M steps/s  frac {d['roofline']['frac']:.3f} ({d['roofline']['bound']})")
except Exception as e:
    print(tag, 'rc', rc, 'no line', e)
PY
}
run none -- --consumer none
run check -- --consumer check
run commit_auto -- --consumer commit
run commit_gated -- --consumer commit --commit-overlap gated
run commit_serial -- --consumer commit --commit-overlap serial
run cc_auto -- --consumer check+commit
run cc_free -- --consumer check+commit --commit-overlap free
run cc_serial -- --consumer check+commit --commit-overlap serial
run cc_gated_w4 B3W_VARIANT=2 -- --consumer check+commit
run cc_gated_prio_hi B3W_COMMIT_PRIORITY=-1 -- --consumer check+commit
run cc_gated_prio_lo B3W_COMMIT_PRIORITY=1 -- --consumer check+commit
run cc_gated_w4_prio_lo B3W_VARIANT=2 B3W_COMMIT_PRIORITY=1 -- --consumer check+commit
run cc_gated_w4_prio_hi B3W_VARIANT=2 B3W_COMMIT_PRIORITY=-1 -- --consumer check+commit
run commit_auto_w4 B3W_VARIANT=2 -- --consumer commit
run none_w4 B3W_VARIANT=2 -- --consumer none
