import importlib, sys, os, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
m = importlib.import_module("hot-proofs-blake3-circom_amd")
ctx = m.Context("nova_vesta", 0)
nbytes = 64 << 20
host = torch.from_numpy(np.random.default_rng(1).integers(0, 256, nbytes, dtype=np.uint8)).pin_memory()
for i in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = m.chain.fold_witnesses(ctx, host, batch_steps=16384, ring=2)
    t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"run {i}: enqueue {1e3*(t1-t0):.1f} ms, total {1e3*(t2-t0):.1f} ms", flush=True)
