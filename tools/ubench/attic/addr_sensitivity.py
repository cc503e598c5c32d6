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
keep = []
def t(ptr, pitch, iters=50):
    for _ in range(5): ctx.run_device(d_recs.data_ptr(), n, ptr, pitch, d_pub.data_ptr(), d_st.data_ptr(), s)
    return ctx.time_device(d_recs.data_ptr(), n, ptr, pitch, d_pub.data_ptr(), d_st.data_ptr(), s, iters)
for k in range(4):
    buf = torch.empty(n*771072 + (1<<22), dtype=torch.uint8, device=dev); keep.append(buf)
    base = buf.data_ptr()
    for off in (0, 4096, 32, 2097152 - (base % 2097152)):
        ms = t(base+off, 770976)
        print(f"buf{k} base%2M={base % 2097152:#x} off={off:#x}: {ms:.4f} ms {n*771088/ms/1e6:.0f} GB/s", flush=True)
    print(f"buf{k} aligned pitch: {t(base, 771072):.4f} ms", flush=True)
for rep in range(3):
    print("repeat buf0:", f"{t(keep[0].data_ptr(), 770976):.4f}", "buf3:", f"{t(keep[3].data_ptr(), 770976):.4f}")
def tfill(buf, iters=20):
    for _ in range(3): buf.fill_(1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): buf.fill_(1)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for k, buf in enumerate(keep):
    ms = tfill(buf)
    print(f"buf{k} torch fill_: {ms:.4f} ms {buf.numel()/ms/1e6:.0f} GB/s   fused: {t(buf.data_ptr(), 770976):.4f} ms")
