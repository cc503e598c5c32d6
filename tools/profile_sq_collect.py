#!/usr/bin/env python3
"""tools/profile_sq_collect.py [round] — distil gpurun_out/prof_sq/ (tools/profile_sq.sh) into profiles/rNN/sq_counters.json:
per kernel, the SQ wave counters averaged per launch and two fractions — parked = SQ_WAIT_ANY / SQ_WAVE_CYCLES (waves waiting on
anything: memory, barrier, dependencies) and issuing = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES."""
import csv, glob, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r04"
src = os.path.join(ROOT, "gpurun_out", "prof_sq")
doc = {"source": "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS "
                 "SQ_ACTIVE_INST_LDS (tools/profile_sq.sh), averaged per launch", "kernels": {}}
WANT = {"comp": ("b3w_compression_kernel",), "nova": ("b3w_nova_kernel",),
        "r1cs": ("b3w_r1cs_walk_kernel", "b3w_r1cs_walk_deferred_kernel", "b3w_r1cs_stream_kernel", "b3w_r1cs_lean_kernel", "b3w_r1cs_deferred_kernel"),
        "r1cs_nova": ("b3w_r1cs_walk_kernel", "b3w_r1cs_walk_deferred_kernel", "b3w_r1cs_stream_kernel", "b3w_r1cs_lean_kernel", "b3w_r1cs_deferred_kernel"),
        "r1cs_stream": ("b3w_r1cs_stream_kernel", "b3w_r1cs_deferred_kernel"),
        "r1cs_lean": ("b3w_r1cs_lean_kernel", "b3w_r1cs_deferred_kernel")}
for sub, names in WANT.items():
    hits = glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True)
    if not hits:
        continue
    rows = list(csv.DictReader(open(max(hits, key=os.path.getmtime))))
    per = {}
    for r in rows:
        kn = r["Kernel_Name"]
        if not any(n in kn for n in names):
            continue
        mm = re.search(r"::(b3w_\w+(?:<[^>]*>)?)", kn)
        short = mm.group(1) if mm else kn[:60]
        per.setdefault(short, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for short, ctr in per.items():
        launches = max(len(v) for v in ctr.values())
        if launches < 2 and "lean" not in short and "deferred" not in short and "stream" not in short and "walk" not in short:
            continue
        avg = {k: sum(v) / len(v) for k, v in ctr.items()}
        if avg.get("SQ_WAVE_CYCLES"):
            avg["parked_frac"] = avg.get("SQ_WAIT_ANY", 0.0) / avg["SQ_WAVE_CYCLES"]
            avg["issuing_frac"] = avg.get("SQ_ACTIVE_INST_ANY", 0.0) / avg["SQ_WAVE_CYCLES"]
        avg["launches"] = launches
        doc["kernels"][f"{sub}: {short}"] = avg
out = os.path.join(ROOT, "profiles", rnd, "sq_counters.json")
json.dump(doc, open(out, "w"), indent=1)
for k, v in doc["kernels"].items():
    print(f"{k}: launches {v['launches']}, parked {v.get('parked_frac', 0):.3f}, issuing {v.get('issuing_frac', 0):.3f}, "
          f"VALU insts {v.get('SQ_INSTS_VALU', 0):.3g}, LDS insts {v.get('SQ_INSTS_LDS', 0):.3g}")
