"""walk_trace_median.py <rocprofv3 output dir> [label] — the constraint check's kernels in a `rocprofv3 --kernel-trace` run: median and
minimum over the LAST 20 dispatches of each (the first launches of a process run up to 15 % longer; an average would carry them)."""
import csv, glob, statistics, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
d = {}
for r in csv.DictReader(open(f)):
    if "r1cs" in r["Kernel_Name"]:
        d.setdefault(r["Kernel_Name"], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = []
for k, v in d.items():
    v = v[-20:]
    name = "deferred" if "deferred" in k else "walk" if "walk" in k else k[:24]
    out.append(f"{name} median {statistics.median(v):7.1f} min {min(v):7.1f} us ({len(v)})")
print((sys.argv[2] if len(sys.argv) > 2 else "") + "  " + " | ".join(sorted(out, reverse=True)))
