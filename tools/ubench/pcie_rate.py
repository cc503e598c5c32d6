"""PCIe-inclusive rate of the host-buffer entry points (never the bench `value`): records H2D, kernel,
full bodies D2H into pinned host memory, per batch."""
import importlib, sys, os, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
m = importlib.import_module("hot-proofs-blake3-circom_amd")
ctx = m.Context("compression", 0)
n = 2048
recs = torch.from_numpy(m.workloads.config2_compression(n).view(np.int32)).pin_memory()
dev = torch.device("cuda:0")
d_recs = torch.empty_like(recs, device=dev)
d_bodies = torch.empty((n, ctx.body_bytes), dtype=torch.uint8, device=dev)
h_bodies = torch.empty((n, ctx.body_bytes), dtype=torch.uint8).pin_memory()
d_pub = torch.zeros((n, 16), dtype=torch.int32, device=dev); d_st = torch.zeros(n, dtype=torch.int32, device=dev)
s = torch.cuda.current_stream().cuda_stream
def once():
    d_recs.copy_(recs, non_blocking=True)
    ctx.run_device(d_recs.data_ptr(), n, d_bodies.data_ptr(), 0, d_pub.data_ptr(), d_st.data_ptr(), s)
    h_bodies.copy_(d_bodies, non_blocking=True)
for _ in range(2): once()
torch.cuda.synchronize(); t0 = time.perf_counter()
it = 5
for _ in range(it): once()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / it
print(f"PCIe-inclusive: {n/dt:.0f} witnesses/s, D2H {n*ctx.body_bytes/dt/1e9:.1f} GB/s, {dt*1e3:.1f} ms per batch of {n}")
