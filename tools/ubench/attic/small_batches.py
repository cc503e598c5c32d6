"""small_batches.py — bodies per wave vs batch size below 4096 (few workgroups per CU)."""
import importlib, os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
m = importlib.import_module("hot-proofs-blake3-circom_amd")
dev = torch.device("cuda:0"); s = torch.cuda.current_stream().cuda_stream
nmax = 4096
recs = m.workloads.config2_compression(nmax)
d_recs = torch.from_numpy(recs.view(np.int32)).to(dev)
d_pub = torch.zeros((nmax, 16), dtype=torch.int32, device=dev); d_st = torch.zeros(nmax, dtype=torch.int32, device=dev)
base = m.Context("compression", 0)
buf = base.alloc_bodies(nmax * base.body_bytes)
for n in (64, 256, 512, 1024, 2048, 3072):
    row = []
    for v in (1, 2, 0, 3):
        os.environ["B3W_VARIANT"] = str(v)
        ctx = m.Context("compression", 0)
        for _ in range(3): ctx.run_device(d_recs.data_ptr(), n, buf.ptr, 0, d_pub.data_ptr(), d_st.data_ptr(), s)
        ms = min(ctx.time_device(d_recs.data_ptr(), n, buf.ptr, 0, d_pub.data_ptr(), d_st.data_ptr(), s, 20) for _ in range(3))
        row.append(f"W={ {1:1,2:2,0:4,3:8}[v] }: {ms*1e3:6.1f} us {n/ms/1e3:5.2f} M/s")
        ctx.close()
    print(f"n={n:5d}  " + "   ".join(row), flush=True)
