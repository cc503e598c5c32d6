#!/usr/bin/env python3
"""tools/profile_r1cs_collect.py [round] — distil gpurun_out/prof_r1cs/ (tools/profile_r1cs.sh) into
profiles/rNN/r1cs_check_<circuit>_{kernel_stats.csv, pmc_FETCH_SIZE.csv} and profiles/rNN/r1cs_check.json: per check, the
stream (or lean) and the deferred kernel's average durations, the HBM bytes fetched (FETCH_SIZE in KiB, doubled: on gfx950 the counter
reports half of a wide coalesced read stream — /opt/skills/guides/MI355X_MICROARCH.md, HBM section) against the body bytes."""
import csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r04"
src = os.path.join(ROOT, "gpurun_out", "prof_r1cs")
dst = os.path.join(ROOT, "profiles", rnd)
os.makedirs(dst, exist_ok=True)
BODY = {"compression": 24093 * 32, "nova_vesta": 23291 * 32}
N = 4096
LAST = 10                                                   # launches averaged per kernel: the target's 3 warm-up checks stay out
doc = {"bodies": N, "checks_profiled": LAST, "warmup_checks_excluded": 3,
       "source": "rocprofv3 --kernel-trace: End - Start of the last 10 launches of each kernel (kernel_stats.csv beside it averages all 13)", "circuits": {}}


def find(sub, pattern):
    hits = glob.glob(os.path.join(src, sub, "**", pattern), recursive=True)
    return max(hits, key=os.path.getmtime) if hits else None


for c in ("compression", "nova_vesta"):
    rows = [r for r in csv.DictReader(open(find(f"stats_{c}", "*kernel_stats.csv"))) if "b3w_r1cs" in r["Name"]]
    with open(os.path.join(dst, f"r1cs_check_{c}_kernel_stats.csv"), "w") as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys(), quoting=csv.QUOTE_NONNUMERIC)
        w.writeheader(); w.writerows(rows)
    kind = lambda name: "deferred" if "deferred" in name else "walk" if "walk" in name else "stream" if "stream" in name else "lean" if "lean" in name else "init"
    trace = sorted((r for r in csv.DictReader(open(find(f"stats_{c}", "*kernel_trace.csv"))) if "b3w_r1cs" in r["Kernel_Name"]),
                   key=lambda r: int(r["Start_Timestamp"]))
    per_kind = {}
    for r in trace:
        per_kind.setdefault(kind(r["Kernel_Name"]), []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    avg = {k: sum(v[-LAST:]) / len(v[-LAST:]) for k, v in per_kind.items()}
    all_launches = {k: [round(x / 1e3, 1) for x in v] for k, v in per_kind.items()}
    pm = [r for r in csv.DictReader(open(find(f"fetch_{c}", "*counter_collection.csv")))
          if r["Counter_Name"] == "FETCH_SIZE" and "b3w_r1cs" in r["Kernel_Name"]]
    with open(os.path.join(dst, f"r1cs_check_{c}_pmc_FETCH_SIZE.csv"), "w") as f:
        w = csv.DictWriter(f, fieldnames=pm[0].keys())
        w.writeheader(); w.writerows(pm)
    per = {}
    for r in pm:
        k = kind(r["Kernel_Name"])
        per.setdefault(k, []).append(float(r["Counter_Value"]))
    fetched = {k: 2.0 * 1024.0 * sum(v) / len(v) for k, v in per.items()}
    body_bytes = N * BODY[c]
    check_ns = sum(avg.values())
    doc["circuits"][c] = {"kernel_avg_us": {k: v / 1e3 for k, v in avg.items()}, "launch_us": all_launches, "check_us": check_ns / 1e3,
                          "bodies_per_s": N / (check_ns * 1e-9), "body_bytes": body_bytes,
                          "read_rate_GBps": body_bytes / check_ns, "hbm_fetch_bytes": fetched,
                          "fetch_over_body_bytes": sum(fetched.values()) / body_bytes,
                          "roofline": {"bound": "hbm", "achieved": body_bytes / check_ns, "peak": 8000.0, "unit": "GB/s",
                                       "frac": body_bytes / check_ns / 8000.0}}
json.dump(doc, open(os.path.join(dst, "r1cs_check.json"), "w"), indent=1)
print(json.dumps(doc, indent=1))
