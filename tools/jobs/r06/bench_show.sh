#!/bin/bash
# bench_show.sh <tag> [bench.py args] — one bench.py run into gpurun_out/r06/<tag>.json and the fields a reader looks at first
mkdir -p gpurun_out/r06
tag=$1; shift
python bench.py "$@" > gpurun_out/r06/$tag.json 2> gpurun_out/r06/$tag.err
rc=$?
tail -3 gpurun_out/r06/$tag.err
python - "$tag" <<'PY'
import json, sys
lines = [l for l in open(f"gpurun_out/r06/{sys.argv[1]}.json") if l.startswith("{")]
if not lines:
    sys.exit("no JSON line")
d = json.loads(lines[-1]); r = d["roofline"]
print("value", round(d["value"]), d["unit"], "| frac", round(r["frac"], 4), "| kernel_ms", round(r.get("kernel_ms", 0), 4), "|", d["config"].get("kernel_variant"), d["config"].get("placement"))
for k in ("plain", "store_ceiling", "of_measured_ceiling", "plain_of_measured_ceiling", "untimed_legs_failed", "traffic"):
    if k in r:
        print(k, r[k])
print("traffic_source", (r.get("traffic_source") or "")[:90])
if "cpu_baseline" in d:
    print("cpu_baseline", {k: v for k, v in d["cpu_baseline"].items() if k in ("value", "unit", "cores", "kind")})
PY
exit $rc
