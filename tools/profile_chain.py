#!/usr/bin/env python3
"""tools/profile_chain.py <tag> [round] — distil a rocprofv3 run of the chained pass
    rocprofv3 --kernel-trace --memory-copy-trace --marker-trace --stats --output-format csv -d gpurun_out/prof_<tag> \\
              -- python3 bench.py --workload chain --preimage-mib 1024 --steps 1 --warmup 1
into profiles/rNN/<tag>_kernel_stats.csv, <tag>_memcpy_stats.csv and <tag>_timeline.json: per-kernel share of the GPU
time (witness kernel / planner / tree), how much of the H2D copy time of the preimage slices lies under a running
kernel (the overlap BASELINE config 5 asks for), and the roctx stages the library marks (b3w:* ranges)."""
import csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r02"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles", rnd)
os.makedirs(dst, exist_ok=True)


def find(pattern):
    hits = glob.glob(os.path.join(src, "**", pattern), recursive=True)
    return max(hits, key=os.path.getmtime) if hits else None


def copy_csv(pattern, name):
    f = find(pattern)
    if not f:
        return []
    rows = list(csv.DictReader(open(f)))
    if rows:
        with open(os.path.join(dst, name), "w") as g:
            w = csv.DictWriter(g, fieldnames=rows[0].keys())
            w.writeheader()
            w.writerows(rows)
    return rows


kstats = copy_csv("*kernel_stats.csv", f"{tag}_kernel_stats.csv")
copy_csv("*memory_copy_stats.csv", f"{tag}_memcpy_stats.csv")
mstats = copy_csv("*marker_api_stats.csv", f"{tag}_marker_stats.csv")
ktrace = list(csv.DictReader(open(find("*kernel_trace.csv"))))
ctrace_f = find("*memory_copy_trace.csv")
ctrace = list(csv.DictReader(open(ctrace_f))) if ctrace_f else []

kiv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in ktrace)
t0, t1 = kiv[0][0], max(e for _, e, _ in kiv)
# union of kernel intervals
merged = []
for s, e, _ in kiv:
    if merged and s <= merged[-1][1]:
        merged[-1][1] = max(merged[-1][1], e)
    else:
        merged.append([s, e])
busy = sum(e - s for s, e in merged)


def under_kernels(s, e):
    tot = 0
    for a, b in merged:
        if b <= s:
            continue
        if a >= e:
            break
        tot += min(b, e) - max(a, s)
    return tot


import re
h2d = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Stream_Id"])) for r in ctrace if "HOST_TO_DEVICE" in r.get("Direction", "")]
# the preimage slices (1 MiB each) are the copies on the chain's own copy stream: the stream with by far the most H2D copies
per_stream = {}
for x in h2d:
    per_stream[x[2]] = per_stream.get(x[2], 0) + 1
copy_stream = max(per_stream, key=per_stream.get) if per_stream else None
big = [x for x in h2d if x[2] == copy_stream]
by_kernel = {}
for s, e, name in kiv:
    mt = re.search(r"(b3w_\w+|__amd_rocclr_\w+|at::native::\w+)", name)
    key = mt.group(1) if mt else name[:40]
    by_kernel[key] = by_kernel.get(key, 0) + (e - s)
total_k = sum(by_kernel.values())
doc = {
    "tag": tag,
    "window_ms": (t1 - t0) / 1e6,
    "gpu_busy_ms": busy / 1e6,
    "gpu_busy_frac": busy / (t1 - t0),
    "kernel_time_share": {k: v / total_k for k, v in sorted(by_kernel.items(), key=lambda kv: -kv[1])[:8]},
    "h2d_copies": len(h2d),
    "h2d_slice_copies": len(big),
    "h2d_slice_ms": sum(e - s for s, e, _ in big) / 1e6,
    "h2d_slice_ms_under_a_running_kernel": sum(under_kernels(s, e) for s, e, _ in big) / 1e6,
    "h2d_slice_bytes": "1 MiB per slice (b3w_chain_run_leaves: CHAIN_SLICE_CHUNKS = 1024 chunks)",
    "marker_ranges": {r.get("Name", r.get("Function", "?")): {"calls": int(r.get("Calls", 0)), "total_ms": float(r.get("TotalDurationNs", 0)) / 1e6}
                      for r in mstats if "b3w:" in r.get("Name", r.get("Function", ""))},
}
if big:
    doc["h2d_overlap_frac"] = doc["h2d_slice_ms_under_a_running_kernel"] / max(doc["h2d_slice_ms"], 1e-9)
bj = os.path.join(src, "bench.json")
if os.path.exists(bj):
    doc["bench_line"] = json.loads(open(bj).read().strip().splitlines()[-1])
json.dump(doc, open(os.path.join(dst, f"{tag}_timeline.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in doc.items() if k != "bench_line"}, indent=1))
