"""chain_timing2.py — where does the chained pass spend its time? (slice size of the overlapped H2D)"""
import importlib, os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
m = importlib.import_module("hot-proofs-blake3-circom_amd")
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = m.Context("nova_vesta", 0)
nbytes = mib << 20
host = torch.randint(0, 256, (nbytes,), dtype=torch.uint8).pin_memory()
for _ in range(2):
    out = m.chain.fold_witnesses(ctx, host)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2):
    out = m.chain.fold_witnesses(ctx, host)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 2
steps = out["n_leaf_steps"] + out["n_parent_steps"]
print(f"slice={os.environ.get('B3W_CHAIN_SLICE_CHUNKS', '1024')} {mib} MiB: {dt*1e3:.1f} ms, {steps/dt/1e6:.2f} M steps/s, {steps*745440/dt/1e9:.0f} GB/s placement={out['placement']}", flush=True)
