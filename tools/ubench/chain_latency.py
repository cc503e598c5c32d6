"""chain_latency.py — end-to-end time of the chained pass (plan + every step witness + parents + root) for small
preimages: the shapes of the reference's own fold tests (rust_fold/src/main.rs:478-539: 4 B ... 3 077 B) up to 1 MiB."""
import importlib, os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
m = importlib.import_module("hot-proofs-blake3-circom_amd")
chain = importlib.import_module("hot-proofs-blake3-circom_amd.chain")
ctx = m.Context("nova_vesta", 0)
for ln in (4, 68, 1024, 1028, 3077, 16384, 65536, 262144, 1048576):
    pre = m.workloads.lcg_preimage(ln)
    for _ in range(3):
        r = chain.fold_witnesses(ctx, pre)
        torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        t0 = time.perf_counter()
        r = chain.fold_witnesses(ctx, pre)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    steps = r["n_leaf_steps"] + r["n_parent_steps"]
    print(f"{ln:8d} B: {steps:6d} steps, {min(ts):.3f} ms best, {sorted(ts)[5]:.3f} ms median -> {steps / min(ts) / 1e3:.3f} M steps/s", flush=True)
