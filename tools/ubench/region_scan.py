"""region_scan.py — the real compression kernel over windows of one large allocation: where are the
32 GiB region boundaries, and what does a batch buffer centred on one gain?  (python tools/ubench/region_scan.py)"""
import importlib, sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
m = importlib.import_module("hot-proofs-blake3-circom_amd")
circuit = sys.argv[1] if len(sys.argv) > 1 else "compression"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
ctx = m.Context(circuit, 0)
recs = m.workloads.config2_compression(n) if circuit == "compression" else m.workloads.config3_nova(n)
dev = torch.device("cuda:0")
d_recs = torch.from_numpy(recs.view(np.int32)).to(dev)
d_pub = torch.zeros((n, ctx.public_words), dtype=torch.int32, device=dev); d_st = torch.zeros(n, dtype=torch.int32, device=dev)
s = torch.cuda.current_stream().cuda_stream
GiB = 1 << 30
big = torch.empty(72 * GiB, dtype=torch.uint8, device=dev)
base = big.data_ptr()
pitch = ctx.body_bytes
per = pitch + 4 * recs.shape[1]
def t(off, iters=6):
    for _ in range(2): ctx.run_device(d_recs.data_ptr(), n, base + off, pitch, d_pub.data_ptr(), d_st.data_ptr(), s)
    return ctx.time_device(d_recs.data_ptr(), n, base + off, pitch, d_pub.data_ptr(), d_st.data_ptr(), s, iters)
print(f"{circuit} n={n} base={base:#x}", flush=True)
best = (0, 0)
for k in range(0, 68 * 2):
    off = k * GiB // 2
    if off + n * pitch > 72 * GiB: break
    ms = t(off)
    r = n * per / ms / 1e6
    if r > best[0]: best = (r, off)
    print(f"off={off / GiB:5.1f} GiB {ms:.4f} ms {r:6.0f} GB/s", flush=True)
print("best", best[0], best[1] / GiB)
