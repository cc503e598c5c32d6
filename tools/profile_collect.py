#!/usr/bin/env python3
"""tools/profile_collect.py <tag> [round] — distil gpurun_out/prof_<tag>/ (tools/profile_round.sh) into
profiles/rNN/<tag>_{kernel_stats.csv, pmc_WRITE_SIZE.csv, pmc_FETCH_SIZE.csv, bench.json} and update
profiles/rNN/traffic.json + profiles/traffic_latest.json.  Counter rules: /opt/skills/guides/MI355X_MICROARCH.md
(HBM): values are KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced read stream (double it);
WRITE_SIZE is exact for 16-byte-per-lane stores."""
import csv, glob, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r01"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles", rnd)
os.makedirs(dst, exist_ok=True)
WITNESS_KERNELS = ("b3w_compression_kernel", "b3w_nova_kernel", "b3w_sweep_kernel", "b3w_regionfill_kernel")
# the MODE template argument of the witness kernels: <W, NT, MODE, SL, PERSIST> / <KIND, W, NT, MODE, SL, PERSIST> (0 fused, 1 TRACE, 2 VERIFY;
# PERSIST since r06)
MODE = re.compile(r"(?:true|false), (\d), (?:true|false)(?:, (?:true|false))?>\(")


def find(sub, pattern):
    hits = glob.glob(os.path.join(src, sub, "**", pattern), recursive=True)      # merged gpurun_out keeps older runs too
    return max(hits, key=os.path.getmtime) if hits else None


bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
json.dump(bench, open(os.path.join(dst, f"bench_{tag}.json"), "w"))
stats = find("stats", "*kernel_stats.csv")
rows = list(csv.DictReader(open(stats)))
with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w") as f:
    w = csv.DictWriter(f, fieldnames=rows[0].keys(), quoting=csv.QUOTE_NONNUMERIC)
    w.writeheader()
    w.writerows(rows)
main = max((r for r in rows if any(k in r["Name"] for k in WITNESS_KERNELS)), key=lambda r: float(r["TotalDurationNs"]))
per_launch = {}
for name in ("WRITE_SIZE", "FETCH_SIZE"):
    f = find("write" if name == "WRITE_SIZE" else "fetch", "*counter_collection.csv")
    # the kernel(s) of the timed step only: the fused kernel, or TRACE (MODE 1) + SWEEP for the two-kernel path —
    # the set-up's autotune launches and the untimed VERIFY pass (MODE 2, reads every body) are other instantiations
    sweep = "sweep" in bench["config"]["kernel_variant"]
    fill = "fill-ordered" in bench["config"]["kernel_variant"]
    allrows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == name]
    # fused path: the MODE-0 instantiation this pass launched most often (its autotune may settle on another W than
    # the stats pass did — the candidates are bit-identical and write the same bytes)
    fused = {}
    for r in allrows:
        kn = r["Kernel_Name"]
        if ("b3w_compression_kernel" in kn or "b3w_nova_kernel" in kn) and MODE.search(kn) and MODE.search(kn).group(1) == "0":
            fused[kn] = fused.get(kn, 0) + 1
    chosen = max(fused, key=fused.get) if fused else None
    def timed(kn):
        if fill:                                            # (nova: the wide-slot launch behind the fill kernel, MODE 3, is part of the step)
            return "b3w_regionfill_kernel" in kn or bool("b3w_nova_kernel" in kn and MODE.search(kn) and MODE.search(kn).group(1) == "3")
        if sweep:
            return "b3w_sweep_kernel" in kn or bool(MODE.search(kn) and MODE.search(kn).group(1) == "1")
        return kn == chosen
    keep = [r for r in allrows if timed(r["Kernel_Name"])]
    with open(os.path.join(dst, f"{tag}_pmc_{name}.csv"), "w") as g:
        w = csv.DictWriter(g, fieldnames=keep[0].keys())
        w.writeheader()
        w.writerows(keep)
    # one launch = all witness kernels of one step (the sweep path has two); average over the steps in the pass
    by_kernel = {}
    for r in keep:
        by_kernel.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    per_launch[name] = sum(sum(v) / len(v) for v in by_kernel.values()) * 1024.0
traffic = per_launch["WRITE_SIZE"] + 2.0 * per_launch["FETCH_SIZE"]
cfg = bench["config"]
entry = {"tag": tag, "circuit": cfg["circuit"], "batch": cfg["batch_per_gpu"],
         "path": "sweep" if "sweep" in cfg["kernel_variant"] else "fill" if "fill-ordered" in cfg["kernel_variant"] else "fused", "placement": cfg.get("placement"),
         "hbm_bytes_per_launch": traffic, "write_bytes": per_launch["WRITE_SIZE"], "fetch_bytes_x2": 2.0 * per_launch["FETCH_SIZE"],
         "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"],
         "kernel_avg_ns_rocprof": float(main["AverageNs"]), "kernel_ms_bench": bench["roofline"]["kernel_ms"]}
tj = os.path.join(dst, "traffic.json")
doc = json.load(open(tj)) if os.path.exists(tj) else {"round": int(rnd[1:]), "entries": []}
doc["source"] = "rocprofv3 --pmc WRITE_SIZE / --pmc FETCH_SIZE (separate passes), KiB x1024, FETCH_SIZE x2 (MI355X_MICROARCH.md HBM)"
doc["entries"] = [e for e in doc["entries"] if not (e.get("circuit") == entry["circuit"] and e.get("batch") == entry["batch"]
                                                     and e.get("path") == entry["path"])] + [entry]
json.dump(doc, open(tj, "w"), indent=1)
json.dump(doc, open(os.path.join(ROOT, "profiles", "traffic_latest.json"), "w"), indent=1)
print(json.dumps(entry))
