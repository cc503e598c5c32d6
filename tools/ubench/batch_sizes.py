"""batch_sizes.py — kernel rate vs batch size on a placed buffer (ramp / tail share of a 0.44 ms launch)."""
import importlib, os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
m = importlib.import_module("hot-proofs-blake3-circom_amd")
dev = torch.device("cuda:0")
s = torch.cuda.current_stream().cuda_stream
nmax = 32768
recs = m.workloads.config2_compression(nmax)
d_recs = torch.from_numpy(recs.view(np.int32)).to(dev)
d_pub = torch.zeros((nmax, 16), dtype=torch.int32, device=dev); d_st = torch.zeros(nmax, dtype=torch.int32, device=dev)
base = m.Context("compression", 0)
buf = base.alloc_bodies(nmax * base.body_bytes)
print("placement", buf.placement)
for v in (0, 3, 7):
    os.environ["B3W_VARIANT"] = str(v)
    ctx = m.Context("compression", 0)
    for n in (1024, 2048, 4096, 6144, 8192, 16384, 32768):
        for _ in range(3): ctx.run_device(d_recs.data_ptr(), n, buf.ptr, 0, d_pub.data_ptr(), d_st.data_ptr(), s)
        ms = min(ctx.time_device(d_recs.data_ptr(), n, buf.ptr, 0, d_pub.data_ptr(), d_st.data_ptr(), s, 10) for _ in range(3))
        print(f"variant {v} n={n:6d}: {ms:.4f} ms {n * 771088 / ms / 1e6:6.0f} GB/s {n / ms / 1e3:.2f} M/s", flush=True)
    ctx.close()
