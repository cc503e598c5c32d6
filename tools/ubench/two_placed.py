"""two_placed.py — several live placed buffers in one process: is each of them as fast as a lone one?"""
import importlib, os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
m = importlib.import_module("hot-proofs-blake3-circom_amd")
ctx = m.Context("nova_vesta", 0)
n = 16384
recs = m.workloads.config3_nova(n)
dev = torch.device("cuda:0")
d_recs = torch.from_numpy(recs.view(np.int32)).to(dev)
d_pub = torch.zeros((n, 15), dtype=torch.int32, device=dev); d_st = torch.zeros(n, dtype=torch.int32, device=dev)
s = torch.cuda.current_stream().cuda_stream
per = ctx.body_bytes + 128
def rate(ptr):
    for _ in range(2): ctx.run_device(d_recs.data_ptr(), n, ptr, 0, d_pub.data_ptr(), d_st.data_ptr(), s)
    ms = min(ctx.time_device(d_recs.data_ptr(), n, ptr, 0, d_pub.data_ptr(), d_st.data_ptr(), s, 5) for _ in range(3))
    return n * per / ms / 1e6
bufs = []
for k in range(4):
    t0 = time.perf_counter()
    b = ctx.alloc_bodies(n * ctx.body_bytes)
    dt = time.perf_counter() - t0
    bufs.append(b)
    print(f"buffer {k}: {b.placement}, alloc {dt*1e3:.0f} ms, kernel {rate(b.ptr):.0f} GB/s", flush=True)
print("again:", [round(rate(b.ptr)) for b in bufs], flush=True)
# alternate between two buffers like the ring does
for _ in range(3):
    for b in bufs[:2]:
        ctx.run_device(d_recs.data_ptr(), n, b.ptr, 0, d_pub.data_ptr(), d_st.data_ptr(), s)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    for b in bufs[:2]:
        ctx.run_device(d_recs.data_ptr(), n, b.ptr, 0, d_pub.data_ptr(), d_st.data_ptr(), s)
e1.record(); torch.cuda.synchronize()
print(f"alternating 2 buffers: {10 * n * per / e0.elapsed_time(e1) / 1e6:.0f} GB/s")
