"""store_region_scan.py — the paced region-fill store shape (b3w_bodies_store_rate shapes 700 + p, 700 + 1000 + v) by pace, on a placed, a one-class and a
plain (torch.empty) buffer, beside the other store-only shapes and the fill-ordered witness kernel on the same buffer.
  python tools/ubench/store_region_scan.py [n=4096]"""
import importlib, os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
m = importlib.import_module("hot-proofs-blake3-circom_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
st = torch.cuda.current_stream().cuda_stream
os.environ["B3W_VARIANT"] = "200"
ctx = m.Context("compression", 0)
recs = m.workloads.config2_compression(n)
d_recs = torch.from_numpy(recs.view(np.int32)).cuda()
bufs = {"placed": ctx.alloc_bodies(n * ctx.body_bytes)}
os.environ["B3W_PLACEMENT"] = "single"
bufs["one-class"] = ctx.alloc_bodies(n * ctx.body_bytes)
os.environ.pop("B3W_PLACEMENT")
plain = torch.empty(n * ctx.body_bytes, dtype=torch.uint8, device="cuda")


def witness(ptr):
    for _ in range(3):
        ctx.run_device(d_recs.data_ptr(), n, ptr, 0, 0, 0, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ctx.run_device(d_recs.data_ptr(), n, ptr, 0, 0, 0, st)
    e1.record(); torch.cuda.synchronize()
    return n * 771088 / (e0.elapsed_time(e1) / 20) / 1e6


for name, ptr in (("placed " + bufs["placed"].placement, bufs["placed"].ptr), ("one-class", bufs["one-class"].ptr), ("torch.empty", plain.data_ptr())):
    row = {p: round(ctx.store_rate(ptr, n, 0, 700 + p, 20, st)) for p in (0, 64, 80, 84, 88, 92, 96, 1032, 1036, 1038, 1039, 1040, 1041, 1042, 1044, 1048)}
    others = {s: round(ctx.store_rate(ptr, n, 0, s, 20, st)) for s in (0, 2, 3, 6, 7)}
    print(f"{name}: region fill by pace {row} | shapes {others} | fill-ordered witness kernel {round(witness(ptr))} GB/s", flush=True)
