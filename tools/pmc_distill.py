"""pmc_distill.py <counter_collection.csv> [name filter] — per kernel name: dispatches, mean / min / max of every counter in a
rocprofv3 --pmc pass (the raw csv has one row per dispatch and counter and can be tens of MB; this keeps a few lines)."""
import collections, csv, json, sys
path, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
with open(path, newline="") as f:
    for row in csv.DictReader(f):
        name = row.get("Kernel_Name") or row.get("kernel_name") or ""
        if flt and flt not in name:
            continue
        short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        acc[short][row.get("Counter_Name") or row.get("counter_name")].append(float(row.get("Counter_Value") or row.get("counter_value")))
out = {}
for k, cs in acc.items():
    out[k] = {c: {"dispatches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)} for c, v in cs.items()}
print(json.dumps(out, indent=1))
