"""Fused kernel rate vs output buffer placement: many allocations, print device addresses."""
import importlib, sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
m = importlib.import_module("hot-proofs-blake3-circom_amd")
ctx = m.Context("compression", 0)
n = 4096
recs = m.workloads.config2_compression(n)
dev = torch.device("cuda:0")
d_recs = torch.from_numpy(recs.view(np.int32)).to(dev)
d_pub = torch.zeros((n,16), dtype=torch.int32, device=dev); d_st = torch.zeros(n, dtype=torch.int32, device=dev)
s = torch.cuda.current_stream().cuda_stream
def t(ptr, pitch=770976, iters=30):
    for _ in range(3): ctx.run_device(d_recs.data_ptr(), n, ptr, pitch, d_pub.data_ptr(), d_st.data_ptr(), s)
    return ctx.time_device(d_recs.data_ptr(), n, ptr, pitch, d_pub.data_ptr(), d_st.data_ptr(), s, iters)
keep = []
for k in range(12):
    buf = torch.empty(n*770976 + (k % 3) * (1 << 21), dtype=torch.uint8, device=dev); keep.append(buf)
    base = buf.data_ptr()
    ms = t(base)
    print(f"buf{k:2d} addr={base:#014x}  GiB-offset={(base >> 30) & 0xFFF:4d} bits[21:30]={(base >> 21) & 0x1FF:#05x}  {ms:.4f} ms {n*771088/ms/1e6:.0f} GB/s", flush=True)
# one big arena, sub-buffers at different offsets
arena = torch.empty(40 << 30, dtype=torch.uint8, device=dev)
ab = arena.data_ptr()
for off_gib in (0, 1, 2, 3, 4, 5, 8, 9, 16, 17, 24, 32):
    ms = t(ab + (off_gib << 30))
    print(f"arena+{off_gib:2d} GiB addr={ab + (off_gib << 30):#014x}: {ms:.4f} ms {n*771088/ms/1e6:.0f} GB/s", flush=True)
for off_mib in (0, 1, 2, 3, 5, 64, 65, 512, 513, 777):
    ms = t(ab + (off_mib << 20))
    print(f"arena+{off_mib:4d} MiB: {ms:.4f} ms {n*771088/ms/1e6:.0f} GB/s", flush=True)
