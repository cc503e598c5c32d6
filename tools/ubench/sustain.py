"""sustain.py — does a variant's rate over 20 launches (what the autotuner sees) predict its rate over 3 000 (a bench step)?  Compression, 4 096
witnesses, a placed and a torch.empty buffer; HIP events."""
import importlib, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
m = importlib.import_module("hot-proofs-blake3-circom_amd")
st = torch.cuda.current_stream().cuda_stream
n = 4096
recs = m.workloads.config2_compression(n)
d_recs = torch.from_numpy(recs.view(np.int32)).cuda()
ctxs = {}
for v in (0, 3, 200, 201):
    os.environ["B3W_VARIANT"] = str(v)
    ctxs[v] = m.Context("compression", 0)
os.environ.pop("B3W_VARIANT")
placed = ctxs[0].alloc_bodies(n * ctxs[0].body_bytes)
plain = torch.empty(n * ctxs[0].body_bytes, dtype=torch.uint8, device="cuda")


def rate(ctx, ptr, launches):
    for _ in range(2):
        ctx.run_device(d_recs.data_ptr(), n, ptr, 0, 0, 0, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(launches):
        ctx.run_device(d_recs.data_ptr(), n, ptr, 0, 0, 0, st)
    e1.record(); torch.cuda.synchronize()
    return round(n * 771088 / (e0.elapsed_time(e1) / launches) / 1e9, 3)


for name, ptr in (("placed " + placed.placement, placed.ptr), ("torch.empty", plain.data_ptr())):
    for rnd in range(3):
        row = {v: (rate(c, ptr, 20), rate(c, ptr, 3000)) for v, c in ctxs.items()}
        print(f"{name:20s} round {rnd}: variant: (20 launches, 3 000 launches) TB/s {row}", flush=True)
