"""batch_curve.py — witnesses/s against batch size, 1 ... 65 536, for both circuit families and every launch shape (bodies
per wave; SLICED = waves per body, B3W_VARIANT 20 + s), next to what the library's default policy (b3w_batch_run_device
without autotune: b3w_ctx.cpp) picks.
Writes profiles/r02/batch_curve.json (run on the GPU box: `python tools/ubench/batch_curve.py`).  The bodies-per-wave
setting is B3W_VARIANT: compression 1 -> 1, 2 -> 2, 0 -> 4, 3 -> 8, 8 -> 8 + occupancy limit; nova O2 1 -> 1, 0 -> 2, 3 -> 8."""
import importlib, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
m = importlib.import_module("hot-proofs-blake3-circom_amd")
dev = torch.device("cuda:0")
s = torch.cuda.current_stream().cuda_stream
OUT = os.path.join(os.getcwd(), "profiles", "r02", "batch_curve.json")
SLICED = {20 + k: f"sliced {k}" for k in (4, 8, 16, 32, 64)}
VARIANTS = {"compression": {1: "W=1", 2: "W=2", 0: "W=4", 3: "W=8", 8: "W=8 occ", **SLICED}, "nova_vesta": {1: "W=1", 0: "W=2", 3: "W=8", **SLICED}}
SIZES = [1, 8, 64, 128, 256, 512, 1024, 2048, 3072, 4096, 8192, 16384, 32768, 65536]
doc = {"unit": "witnesses/s (kernel only, HIP events over 10-40 launches, best of 3, placed body buffer)", "circuits": {}}
for circuit, variants in VARIANTS.items():
    nmax = SIZES[-1]
    recs = m.workloads.config2_compression(nmax) if circuit == "compression" else m.workloads.config3_nova(nmax)
    d_recs = torch.from_numpy(recs.view(np.int32)).to(dev)
    os.environ.pop("B3W_VARIANT", None)
    base = m.Context(circuit, 0)
    d_pub = torch.zeros((nmax, base.public_words), dtype=torch.int32, device=dev)
    d_st = torch.zeros(nmax, dtype=torch.int32, device=dev)
    buf = base.alloc_bodies(nmax * base.body_bytes)
    rows = []
    for n in SIZES:
        row = {"n": n, "rates": {}}
        iters = 40 if n <= 4096 else 10

        def rate(ctx):
            for _ in range(3):
                ctx.run_device(d_recs.data_ptr(), n, buf.ptr, 0, d_pub.data_ptr(), d_st.data_ptr(), s)
            ms = min(ctx.time_device(d_recs.data_ptr(), n, buf.ptr, 0, d_pub.data_ptr(), d_st.data_ptr(), s, iters) for _ in range(3))
            return n / ms * 1e3
        for v, name in variants.items():
            if v >= 20 and n > 4096:                          # sliced launches are for small batches
                continue
            os.environ["B3W_VARIANT"] = str(v)
            ctx = m.Context(circuit, 0)
            row["rates"][name] = rate(ctx)
            ctx.close()
        os.environ.pop("B3W_VARIANT", None)
        row["default_policy"] = rate(base)
        best = max(row["rates"], key=row["rates"].get)
        row["best"] = best
        row["default_vs_best"] = row["default_policy"] / row["rates"][best]
        row["gbps_best"] = row["rates"][best] * base.body_bytes / 1e9
        rows.append(row)
        print(circuit, n, {k: round(v / 1e6, 3) for k, v in row["rates"].items()}, "default", round(row["default_policy"] / 1e6, 3), "best", best,
              flush=True)
    doc["circuits"][circuit] = {"placement": buf.placement, "rows": rows}
    buf.free()
    base.close()
os.makedirs(os.path.dirname(OUT), exist_ok=True)
json.dump(doc, open(OUT, "w"), indent=1)
print("wrote", OUT)
