#!/usr/bin/env python3
"""pass_timeline.py <rocprofv3 output dir> [max rows] — the kernels and memory copies of the LAST burst of a trace (everything behind the
last pause of >= 10 ms), each with its start offset from the burst's first start, its duration and its queue: which kernels ran side
by side, where the device waited.
    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/prof_x -- python3 tools/ubench/chain_one_pass.py 8 1"""
import csv, glob, os, re, sys

src = sys.argv[1]
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 80


def find(pattern):
    hits = glob.glob(os.path.join(src, "**", pattern), recursive=True)
    return max(hits, key=os.path.getmtime) if hits else None


ev = []
f = find("*kernel_trace.csv")
for r in csv.DictReader(open(f)):
    mt = re.search(r"(b3w_\w+|__amd_rocclr_\w+|at::native::\w+|ncclDevKernel\w*)", r["Kernel_Name"])
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "q" + r.get("Queue_Id", "?"), mt.group(1) if mt else r["Kernel_Name"][:48]))
f = find("*memory_copy_trace.csv")
if f:
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy", r.get("Direction", "copy").replace("MEMORY_COPY_", "")))
ev.sort()
cut = 0
for i in range(1, len(ev)):
    if ev[i][0] - max(e for _, e, _, _ in ev[max(0, i - 8):i]) >= 10_000_000:
        cut = i
burst = ev[cut:]
t0 = burst[0][0]
end = max(e for _, e, _, _ in burst)
print(f"{len(burst)} events in the last burst, {(end - t0) / 1e3:.1f} us from first start to last end")
busy, cur_s, cur_e = 0, None, None
for s, e, _, _ in burst:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"device busy {busy / 1e3:.1f} us of it ({100.0 * busy / (end - t0):.1f} %)")
print(f"{'start us':>10} {'dur us':>9}  queue  what")
for s, e, q, name in burst[:limit]:
    print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:9.1f}  {q:5s}  {name}")
if len(burst) > limit:
    print(f"... {len(burst) - limit} more")
