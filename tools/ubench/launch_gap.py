"""launch_gap.py — what do per-step event records cost between back-to-back launches of the 0.44 ms kernel?"""
import importlib, os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
m = importlib.import_module("hot-proofs-blake3-circom_amd")
n = 4096
os.environ["B3W_VARIANT"] = "3"
ctx = m.Context("compression", 0)
recs = m.workloads.config2_compression(n)
dev = torch.device("cuda:0")
d_recs = torch.from_numpy(recs.view(np.int32)).to(dev)
d_pub = torch.zeros((n, 16), dtype=torch.int32, device=dev); d_st = torch.zeros(n, dtype=torch.int32, device=dev)
buf = ctx.alloc_bodies(n * ctx.body_bytes)
stream = torch.cuda.current_stream(); s = stream.cuda_stream
K = 200
def run(events):
    for _ in range(5): ctx.run_device(d_recs.data_ptr(), n, buf.ptr, 0, d_pub.data_ptr(), d_st.data_ptr(), s)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    t0 = time.perf_counter()
    for i in range(K):
        if events: ev[i][0].record(stream)
        ctx.run_device(d_recs.data_ptr(), n, buf.ptr, 0, d_pub.data_ptr(), d_st.data_ptr(), s)
        if events: ev[i][1].record(stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K * 1e3
    k = sum(a.elapsed_time(b) for a, b in ev) / K if events else float("nan")
    return dt, k
for events in (True, False, True, False):
    dt, k = run(events)
    print(f"events={events}: {dt:.4f} ms per step, kernel {k:.4f} ms ({buf.placement})", flush=True)
